#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root:  bash tools/profile_round.sh r01o
# Three separate passes of the same bench command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE - PMC passes never share a
# run with a trace), summarised into gpurun_out/<name>_*; copy those into profiles/ afterwards.
N=${1:?name}
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-fp32-pipe --no-config4 $BENCH_ARGS > /dev/null 2>&1
if [ "$2" = "trace" ]; then
  cd $R
  T=$(ls $O/p_trace/*/*kernel_trace.csv | head -1)
  python tools/prof_summary.py $T --skip 5 --top 90 > $O/${N}_steady_state.txt
  cp $(ls $O/p_trace/*/*kernel_stats.csv | head -1) $O/${N}_kernel_stats.csv
  rm -rf $O/p_*
  exit 0
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 $BENCH_ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 $BENCH_ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p_mfma -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fp32-pipe --no-config4 $BENCH_ARGS > /dev/null 2>&1
cd $R
python tools/pmc_kernel.py $O/p_mfma --mfma "fused_window|winograd|linear_|gemm_bf16|wgrad_bf16|leff_fused|ps_attn|dense_attn|thin_conv|split6|gemm_split|wgrad_split|conv_gemm|conv_wgrad" > $O/${N}_pmc_mfma.txt
T=$(ls $O/p_trace/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $T --skip 5 --top 70 > $O/${N}_steady_state.txt
cp $(ls $O/p_trace/*/*kernel_stats.csv | head -1) $O/${N}_kernel_stats.csv
python tools/pmc_summary.py $(ls $O/p_fetch/*/*counter_collection.csv | head -1) $(ls $O/p_write/*/*counter_collection.csv | head -1) \
    --json $O/${N}_pmc_traffic.json --stamp \
    --filter "ps_attn|fused_window|winograd|linear_wgrad|linear_gemm|split6|gemm_split|wgrad_split|conv_gemm|conv_wgrad|leff|ln_partition|reverse_residual|charbonnier|adamw|bias_|maxpool|l1_pair|blocked|crop|thin_conv|conv3x3_in3" \
    > $O/${N}_pmc_traffic.txt
rm -rf $O/p_*
python bench.py $BENCH_ARGS > $O/${N}_bench_line.json
cut -c1-200 $O/${N}_bench_line.json
