#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root (gpurun): everything lands under gpurun_out/r03/ and the summaries
# are then copied into profiles/.  Counter passes are their own runs (never combined with tracing), as the guide prescribes.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. headline bench line + the same command under rocprofv3 --kernel-trace --stats
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err || exit 1
# 2. HBM traffic of the scan kernels: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || exit 1
# 3. issue / LDS / MFMA counters of the scan and GEMM kernels
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_mfma.err || exit 1
cd $R
python3 tools/steady_state_stats.py $(ls $O/trace/*/*kernel_trace.csv | head -1) --steps 16 --top 60 --csv $O/steady_state.csv > $O/steady_state.txt
python3 tools/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/scan_traffic.json T-224-bs64 > /dev/null
python3 tools/pmc_summary.py $O/pmc_sq ss2d_fwd_kernel ss2d_bwd_kernel gemm_bf16_kernel conv3x3_nhwc_kernel > $O/scan_pmc_summary.json
python3 tools/pmc_summary.py $O/pmc_mfma gemm_bf16_kernel ss2d_bwd_kernel ss2d_fwd_kernel conv3x3_nhwc_kernel conv3x3_wgrad_kernel > $O/mfma_pmc_summary.json
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma      # gpurun copies back at most 64 MiB
echo collected
