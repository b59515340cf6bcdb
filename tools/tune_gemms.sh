#!/bin/bash
# Regenerate dehaze_hip/tunableop_gfx950.csv: PyTorch TunableOp over the GEMM shapes of the config-2 training step
# (about 2 minutes on one MI355X).  The rotating buffer makes every candidate read its operands cold, as in the step.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/tunableop_new.csv}"
export DHZ_NO_TUNED_GEMMS=1 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$OUT" \
       PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10 PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=512
python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline
echo "wrote ${OUT%.csv}0.csv - copy it to research-and-..._amd/dehaze_hip/tunableop_gfx950.csv"
