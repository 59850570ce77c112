#!/bin/bash
# Clock and package power while bench.py's timed steps run, under two settings of the half-tile tail launch (one box):
#   tools/power_ab.sh OUTDIR
out=$1; mkdir -p "$out"
for v in ht1 ht0 ht1 ht0; do
  o=""; [ $v = ht1 ] && o="gemm_half_tail=1"
  VIT_OPTIONS=$o bash tools/smi_sample.sh "$out/smi_$v.txt" -- python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing \
    --no-secondary --no-input-probe --no-overlap > "$out/bench_$v.json" 2>/dev/null
  python - "$out/bench_$v.json" "$out/smi_$v.txt" $v <<'PY'
import json, re, sys
d = json.loads(open(sys.argv[1]).read())
t = open(sys.argv[2]).read()
sclk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", t)]
print(f"{sys.argv[3]}: {d['ms_per_step']:.3f} ms/step; sclk MHz last samples {sclk[-6:]} mean {sum(sclk[-6:]) / max(1, len(sclk[-6:])):.0f}; "
      f"power W {pw[-6:]} mean {sum(pw[-6:]) / max(1, len(pw[-6:])):.0f}")
PY
done
