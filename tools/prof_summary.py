#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results .db (kernel-trace): name, launches, total ms, mean us, share.
Usage: python tools/prof_summary.py gpurun_out/prof_x/x_results.db [--csv out.csv] [--steps N]"""
import argparse, re, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--csv")
ap.add_argument("--steps", type=int, default=0, help="divide totals by this many steps (0 = raw)")
a = ap.parse_args()
c = sqlite3.connect(a.db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if "kernel_symbol" in t][0]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
def short(n):
    n = re.sub(r"\.kd$", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) < 110 else n[:107] + "..."
div = a.steps or 1
out = ["name,launches,total_ms,mean_us,min_us,max_us,share"]
print(f"{'kernel':110s} {'n':>6s} {'ms' + ('/step' if a.steps else ''):>9s} {'mean us':>9s} {'share':>6s}")
for n, cnt, t, mn, mx in rows:
    print(f"{short(n):110s} {cnt:6d} {t / 1e6 / div:9.3f} {t / cnt / 1e3:9.1f} {t / tot:6.1%}")
    out.append(f"\"{short(n)}\",{cnt},{t / 1e6:.3f},{t / cnt / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{t / tot:.4f}")
print(f"total kernel time {tot / 1e6 / div:.3f} ms" + ("/step" if a.steps else ""))
if a.csv:
    open(a.csv, "w").write("\n".join(out) + "\n")
