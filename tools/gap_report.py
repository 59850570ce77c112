#!/usr/bin/env python3
"""Idle time of the GPU inside one optimisation step, from a rocprofv3 --kernel-trace results .db: the union of all kernel
intervals against the step's span (steps are delimited by adamw_kernel), the gaps attributed to the kernel that FOLLOWS them.
Usage: python tools/gap_report.py x_results.db [--skip N]"""
import collections, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,stream_id,queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
tot = collections.Counter(); cnt = collections.Counter(); spans = []; idles = []
for a, b in zip(idx[1:-1], idx[2:]):
    step = sorted(rows[a + 1:b + 1], key=lambda r: r[1])
    t0, e = step[0][1], step[0][2]
    idle = 0
    for r in step[1:]:
        if r[1] > e:
            g = r[1] - e
            idle += g
            n = r[0].replace("void ", "").replace("vit::", "").split("(")[0][:48]
            tot[n] += g; cnt[n] += 1
        e = max(e, r[2])
    spans.append((e - t0) / 1e6); idles.append(idle / 1e6)
n = len(spans)
print(f"{n} steps: span {statistics.mean(spans):.3f} ms, idle {statistics.mean(idles):.3f} ms ({100 * sum(idles) / sum(spans):.1f} %), "
      f"{sum(cnt.values()) / n:.0f} gaps per step, streams/queues {sorted(set((r[3], r[4]) for r in rows))}")
for k, v in tot.most_common(14):
    print(f"  {v / n / 1e3:8.1f} us/step in {cnt[k] / n:5.1f} gaps (mean {v / cnt[k] / 1e3:5.1f} us) before {k}")
