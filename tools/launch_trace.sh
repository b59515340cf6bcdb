#!/bin/bash
# per-launch durations of selected kernels within one steady step:  bash tools/launch_trace.sh out_name 'regex' [bench args]
N=${1:?name}; RX=${2:?regex}; shift 2
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_trace
rocprofv3 --kernel-trace --output-format csv -d $O/p_trace -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2>&1
cd $R
python tools/prof_summary.py $(ls $O/p_trace/*/*kernel_trace.csv | head -1) --skip 5 --top 10 --launches "$RX" > $O/${N}_launches.txt
rm -rf $O/p_trace
