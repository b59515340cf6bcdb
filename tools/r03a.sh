#!/bin/bash
R=$(pwd); O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_trace
rocprofv3 --kernel-trace --output-format csv -d $O/p_trace -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cd $R
T=$(ls $O/p_trace/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $T --skip 5 --top 90 > $O/r03a_steady_state.txt
cp $T $O/r03a_trace.csv
rm -rf $O/p_trace
python tools/bench_gemm.py nolib > $O/r03a_gemm.txt 2>&1
python tools/bench_wgrad.py nolib > $O/r03a_wgrad.txt 2>&1
python bench.py --no-cpu-baseline > $O/r03a_bench.json 2>$O/r03a_bench.err
tail -3 $O/r03a_gemm.txt; tail -2 $O/r03a_wgrad.txt; cut -c1-300 $O/r03a_bench.json
