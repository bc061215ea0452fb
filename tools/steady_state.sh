#!/bin/bash
# Steady-state kernel table of the bench step (rocprofv3 --kernel-trace): gpurun_out/ss/steady_state.{txt,csv}
set -o pipefail
R=$PWD; O=$R/gpurun_out/ss; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/bench_under_rocprof.json 2> $O/trace.err || exit 1
cd $R
python3 tools/steady_state_stats.py $(ls $O/trace/*/*kernel_trace.csv | head -1) --steps 16 --top 200 --csv $O/steady_state.csv > $O/steady_state.txt
rm -rf $O/trace
head -3 $O/steady_state.txt
