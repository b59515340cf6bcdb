# A/B of environment settings on the training step, same box, alternating passes:
#   VARIANTS="DHZ_S6_ROUTE= DHZ_S6_ROUTE=old" bash tools/ab_step.sh        (an empty value = the default)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in $(seq 1 ${PASSES:-2}); do
  for v in $VARIANTS; do
    ms=$(env $v python $R/bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$v : $ms ms/step (pass $rep)"
  done
done
