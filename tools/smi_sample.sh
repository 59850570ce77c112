#!/bin/bash
# Sample clocks / power while a command runs: tools/smi_sample.sh OUT.txt -- cmd ...
OUT="$1"; shift; [[ "$1" == "--" ]] && shift
"$@" &
PID=$!
sleep 8
for i in 1 2 3 4 5 6; do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  { date +%s.%N; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|mclk\|power\|busy" ; } >> "$OUT"
  sleep 1.5
done
wait $PID
