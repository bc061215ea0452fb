#!/bin/bash
# per-kernel time of the SSD chunk kernels (rocprofv3 --kernel-trace --stats over tools/bench_ssd_chunk.py bwd): gpurun_out/ssd_kernel_stats.txt
R=$PWD; O=$R/gpurun_out/ssdprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/tools/bench_ssd_chunk.py bwd > $O/bench.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ssdprof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
out = open("gpurun_out/ssd_kernel_stats.txt", "w")
for r in rows:
    n = r["Name"]
    if "ssd_" in n or "carry" in n:
        out.write(f'{n.split("(")[0][-40:]:42s} calls {r["Calls"]:>5s} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:9.1f} us\n')
out.close()
print(open("gpurun_out/ssd_kernel_stats.txt").read())
PY
grep "^b32" $O/bench.txt
rm -rf $O
