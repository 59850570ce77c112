#!/bin/bash
# tools/micro/mfma_power beside a rocm-smi sampler: sustained MFMA rate, clock, package power per block shape
out=${1:-gpurun_out/mfma_power.txt}
( for i in $(seq 1 28); do sleep 1.2; echo "t=$i $(rocm-smi --showclocks --showpower 2>&1 | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')"; done ) > "$out.smi" &
tools/micro/mfma_power 6 | tee "$out"
wait
cat "$out.smi" >> "$out"; rm -f "$out.smi"
