#!/bin/bash
# A/B of bench.py variants on one box, interleaved: tools/ab_bench.sh OUTDIR REPS "NAME|ENV=.. ENV=..|extra bench args" ...
out=$1; reps=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$reps"); do
  for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
    env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-input-probe $args \
      > "$out/$name.$r.json" 2> "$out/$name.$r.err" || { echo "$name run $r FAILED"; tail -3 "$out/$name.$r.err"; }
    python - "$out/$name.$r.json" "$name" "$r" <<'PY'
import json, sys
t = open(sys.argv[1]).read()
try:
    d = json.loads(t[t.index('{"metric"'):])
    print(f"{sys.argv[2]:>16} run {sys.argv[3]}: median {d['ms_per_step']:.3f} ms  min {d['timing']['min_ms']:.3f}  {d['value']:.0f} img/s")
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  done
done
