#!/bin/bash
# single-GPU cost of leaving k CUs to the collective kernels (dhz_set_reserved_cus; DHZ_COMM_RESERVE_CUS when world > 1):
#   bash tools/reserve_cus.sh > gpurun_out/r05_reserve_cus.txt       (config 2: E = 32, 128 x 128, bs 32, Charbonnier + CR, AdamW)
R=$(cd "$(dirname "$0")/.." && pwd)
echo "# bench.py --reserve-cus k --steps 40 --warmup 10 (config 2, one MI355X); grid_cus = CUs the persistent grids are sized for"
for k in 0 4 8 16 32 0; do
  python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 --reserve-cus $k 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('reserve_cus %2d  grid_cus %3s  %.3f ms/step  %.1f patches/s' % ($k, d['config'].get('grid_cus', 256), d['ms_per_step'], d['value']))"
done
