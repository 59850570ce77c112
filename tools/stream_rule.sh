#!/bin/bash
# one stream vs two streams (half-tail off) at several batch sizes: which launch mode does the tile count call for?
export VIT_OPTIONS=gemm_half_tail=0
for r in 1 2; do
for b in 64 128 192 256 384 512; do
  for f in "" "--no-overlap"; do
    python bench.py --batch $b --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-input-probe --no-secondary $f 2>/dev/null | \
      python -c "import sys,json; t=sys.stdin.read(); d=json.loads(t[t.index(chr(123)):]); print('B', $b, 'one' if '$f' else 'two', d['ms_per_step'], round(d['value']))"
  done
done
done
