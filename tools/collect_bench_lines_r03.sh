#!/bin/bash
# The other bench lines of round 3 (run on the GPU box from the repo root): fp32 dense precision (the reference's own),
# MedMamba-B 512^2 bs 32 (BASELINE.json configs[2]) with its own PMC traffic, per-stage scan kernel table, projection GEMMs.
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
python3 bench.py --steps 8 --warmup 3 --variant B --res 512 --batch-size 32 --no-cpu-baseline > $O/bench_B512.json 2> $O/bench_B512.err
python3 tools/scan_kernel_bench.py 64 10 > $O/scan_stage_table.txt 2>&1
python3 tools/scan_kernel_bench.py 32 5 0,1,2,3 B > $O/scan_stage_table_B512.txt 2>&1
python3 tools/bench_gemm.py > $O/gemm_bench.txt 2>&1
python3 tools/bench_gemm_f32.py > $O/gemm_f32_bench.txt 2>&1
MEDSCAN_F32_GEMM=0 python3 bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline > $O/bench_fp32_blas.json 2> /dev/null
python3 bench.py --steps 10 --warmup 3 --variant SSD --batch-size 32 --no-cpu-baseline > $O/bench_ssd.json 2> $O/bench_ssd.err
python3 tools/bench_fusion.py > $O/vfefm_bench.json 2> $O/vfefm_bench.err
MEDSCAN_BWD_FAST=0 python3 tools/scan_kernel_bench.py 64 10 > $O/scan_stage_table_general_kernel.txt 2>&1
MEDSCAN_MFMA_GEMM=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_blas_projections.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmcB_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --variant B --res 512 --batch-size 32 --no-cpu-baseline > /dev/null 2> $O/pmcB_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmcB_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --variant B --res 512 --batch-size 32 --no-cpu-baseline > /dev/null 2> $O/pmcB_write.err
cd $R
python3 tools/collect_traffic.py $O/pmcB_fetch $O/pmcB_write $O/scan_traffic_B512.json B-512-bs32 > /dev/null
rm -rf $O/pmcB_fetch $O/pmcB_write
echo lines collected
