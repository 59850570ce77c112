#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC passes (one directory per pass, each a separate run with --kernel-trace --pmc ...
--output-format csv, never combined with other trace domains), merged into one JSON with derived figures:

  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) x CUs x 4 SIMDs)   (north_star's "MFMA utilisation";
                   rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: 7.8 M "cycles" for a 450 us kernel at
                   2.2 GHz -- checked against duration x clock and against FLOP / time: dW GEMM 0.34 vs 817 TFLOP/s / 2.5 P)
  wave-cycle split: active = SQ_ACTIVE_INST_ANY, issue-stall = SQ_WAIT_INST_ANY, parked = SQ_WAIT_ANY, each / SQ_WAVE_CYCLES
  hbm_bytes      = 2 x FETCH_SIZE KiB (gfx950: wide reads are half-counted) + WRITE_SIZE KiB     (MI355X_MICROARCH.md)
  l2_hit         = TCC_HIT / (TCC_HIT + TCC_MISS)

usage: python tools/pmc_kernels.py OUT.json DIR [DIR ...] [--match substring] [--cus 256]"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.build import source_stamp
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(zip(sys.argv[1:], sys.argv[2:]))
    match, cus = opts.get("--match"), int(opts.get("--cus", 256))
    out, dirs = args[0], [a for a in args[1:] if os.path.isdir(a)]
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("vit::", "")
                    if match and match not in name:
                        continue
                    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[name][row["Counter_Name"]] += 1
    doc = {"source": "rocprofv3 --kernel-trace --pmc <counters> (one pass per directory: " + ", ".join(dirs) + ")",
           "corrections": "FETCH_SIZE doubled (gfx950 half-counts wide coalesced reads), WRITE_SIZE exact, both KiB; SQ_* "
                          "wave-cycle counters are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)",
           "lib_stamp": source_stamp(), "kernels": {}}
    for name in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0))):
        c = {k: acc[name][k] / cnt[name][k] for k in acc[name]}
        e = {"launches": max(cnt[name].values()), "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * cus * 4), 4)
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for key, ctr in (("active", "SQ_ACTIVE_INST_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("parked", "SQ_WAIT_ANY"),
                             ("valu_active", "SQ_ACTIVE_INST_VALU"), ("lds_active", "SQ_ACTIVE_INST_LDS"),
                             ("lds_issue_stall", "SQ_WAIT_INST_LDS")):
                if ctr in c:
                    e["frac_" + key] = round(c[ctr] / wc, 4)
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            e["hbm_bytes"] = int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["l2_hit"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        doc["kernels"][name] = e
    json.dump(doc, open(out, "w"), indent=1)
    for name, e in list(doc["kernels"].items())[:14]:
        print(name[:64].ljust(64), {k: v for k, v in e.items() if k not in ("counters",)})


if __name__ == "__main__":
    main()
