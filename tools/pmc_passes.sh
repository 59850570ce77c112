#!/bin/bash
# rocprofv3 PMC passes of one command, one counter group per run (PMC is never combined with other trace domains).
#   tools/pmc_passes.sh OUTDIR -- python3 tools/attn_prof.py
# The program must follow `--` directly (no env / bash -c wrappers): the profiler initialises the GPU before it starts.
set -euo pipefail
OUT="$1"; shift; [[ "$1" == "--" ]] && shift
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  [[ -n "${PMC_ONLY:-}" && " $PMC_ONLY " != *" $i "* ]] && continue
  rocprofv3 --kernel-trace --pmc $P -d "$OUT/p$i" --output-format csv -- "$@" > "$OUT.p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT.p$i.log"; }
done
