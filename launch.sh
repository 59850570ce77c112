#!/bin/bash
# launch.sh -- the reference's user entry (launch.sh:16-308) for the MI355X path.  Implemented modes: run, test.
#   ./launch.sh run  -c CONFIG [-g NUM_GPUS] [-w 0|1] [--save] [--ckpt PATH] [--debug N] [--seed S] [--synthetic N]
#   ./launch.sh test -c CONFIG [--ckpt PATH|best|last|none]      (evaluation only)
# The lr / sweep modes (W&B, task-parallel LR sweeps) are outside the hot path.
set -euo pipefail
MODE="${1:-run}"; shift || true
CONFIG="configs/baseline.yaml"; GPUS=""; EXTRA=()
while [[ $# -gt 0 ]]; do
  case "$1" in
    -c|--config) CONFIG="$2"; shift 2 ;;
    -g|--gpu|--gpus) GPUS="$2"; shift 2 ;;
    *) EXTRA+=("$1"); shift ;;
  esac
done
cd "$(dirname "$0")"
[[ -f .env ]] && set -a && source .env && set +a
case "$MODE" in
  run)   # scripts/run.py starts its own rank processes when -g N > 1 (no torchrun needed)
    exec python scripts/run.py -f "$CONFIG" ${GPUS:+-g "$GPUS"} "${EXTRA[@]}" ;;
  test)  # evaluation only: nothing is trained (reference launch.sh test -> scripts/test.py)
    exec python scripts/test.py -f "$CONFIG" ${GPUS:+-g "$GPUS"} "${EXTRA[@]}" ;;
  *) echo "mode '$MODE' is outside the MI355X hot path (implemented: run, test)"; exit 2 ;;
esac
