#!/bin/bash
# issue / LDS counters of ms_linear_bwd_bf16 at the stage-0 shapes (own --pmc passes, no tracing): gpurun_out/lb_pmc.json
R=$PWD; O=$R/gpurun_out/lbpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $O/a --output-format csv -- python3 $R/tools/bench_linear_bwd.py T > $O/a.out 2> $O/a.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/b --output-format csv -- python3 $R/tools/bench_linear_bwd.py T > /dev/null 2> $O/b.err
cd $R
python3 tools/pmc_summary.py $O/a "linear_bwd_kernel<10" "linear_bwd_kernel<12" "linear_bwd_kernel<4" > gpurun_out/lb_pmc_a.json
python3 tools/pmc_summary.py $O/b "linear_bwd_kernel<10" "linear_bwd_kernel<12" "linear_bwd_kernel<4" > gpurun_out/lb_pmc_b.json
cp $O/a.err gpurun_out/lb_a.err; cp $O/a.out gpurun_out/lb_a.out; grep -c linear_bwd $O/a/*/*counter_collection.csv; cp $O/b.err gpurun_out/lb_b.err; ls -R $O | head -20; rm -rf $O
cat gpurun_out/lb_pmc_a.json gpurun_out/lb_pmc_b.json
