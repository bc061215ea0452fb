#!/bin/bash
# issue / LDS / MFMA counters of ms_conv3x3_nhwc_bf16 per stage shape (own --pmc passes, no tracing): gpurun_out/conv_pmc_s<stage>_{a,b}.json
R=$PWD; O=$R/gpurun_out/convpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in ${STAGES:-0 2}; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $O/a$S --output-format csv -- python3 $R/tools/bench_conv3x3.py 64 T $S > $O/a$S.out 2> $O/a$S.err || exit 1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/b$S --output-format csv -- python3 $R/tools/bench_conv3x3.py 64 T $S > /dev/null 2> $O/b$S.err || exit 1
python3 $R/tools/pmc_summary.py $O/a$S "conv3x3_nhwc_kernel" "conv3x3_wgrad_kernel" > $R/gpurun_out/conv_pmc_s${S}_a.json
python3 $R/tools/pmc_summary.py $O/b$S "conv3x3_nhwc_kernel" "conv3x3_wgrad_kernel" > $R/gpurun_out/conv_pmc_s${S}_b.json
cp $O/a$S.out $R/gpurun_out/conv_pmc_s${S}.out
done
rm -rf $O
cat $R/gpurun_out/conv_pmc_s*.json
