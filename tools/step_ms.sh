# prints the config-2 (default) or config-4 (C4=1) step time of the library named by DHZ_LIB_PATH (CMD for tools/variants.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
ARGS=""; [ -n "$C4" ] && ARGS="--dtype bf16 --embed_dim 64 --ps 256 --batch 8"
python $R/bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 $ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'], 'ms/step')"
