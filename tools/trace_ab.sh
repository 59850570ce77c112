#!/bin/bash
# per-kernel means of one-stream steps under two option sets: tools/trace_ab.sh OUTDIR "NAME|VIT_OPTIONS value" ...
out=$1; shift
mkdir -p "$out"; R=$(pwd)
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; opts=${spec#*|}
  export VIT_OPTIONS="$opts"
  (cd /tmp && rocprofv3 --kernel-trace -d "$R/$out/$name" -o t -- python3 "$R/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --no-input-probe --no-overlap > "$R/$out/$name.log" 2>&1)
  python tools/prof_summary.py "$out/$name/t_results.db" --steps 9 > "$out/$name.summary.txt"
  echo "== $name ($opts)"; head -16 "$out/$name.summary.txt" | cut -c1-58,108-150; tail -1 "$out/$name.summary.txt"
done
