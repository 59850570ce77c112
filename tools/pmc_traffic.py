#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (separate runs, as MI355X_MICROARCH.md prescribes):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w --output-format csv -- python3 bench.py ...   (same command)
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/rNN_pmc_traffic.json

gfx950 corrections (guide, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
exact; both counters are in KiB."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.build import source_stamp
from collections import defaultdict

def load(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = re.sub(r"^void ", "", row["Kernel_Name"])
                name = re.sub(r"\(.*$", "", name).replace("vit::", "")
                acc[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return acc, cnt

def main():
    fdir, wdir, out = sys.argv[1:4]
    fa, fc = load(fdir, "FETCH_SIZE")
    wa, wc = load(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fa) | set(wa), key=lambda k: -(2 * fa.get(k, 0) + wa.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0))
        if n == 0:
            continue
        f_kb, w_kb = fa.get(k, 0.0) / max(fc.get(k, 1), 1), wa.get(k, 0.0) / max(wc.get(k, 1), 1)
        kernels[k] = {"fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                      "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024), "launches": n}
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                     "--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing",
           "correction": "gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact "
                         "(MI355X_MICROARCH.md, HBM section); KB -> bytes x1024",
           "lib_stamp": source_stamp(),  # sha256 of vit_amd/csrc + headers at collection time: bench.py refuses a stale file
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:12]:
        print(f"{k[:70]:70s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch x{v['launches']}")

if __name__ == "__main__":
    main()
