R=$PWD; O=$R/gpurun_out/vf; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/tools/bench_fusion.py > $O/bench.json 2> $O/err.txt
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/vf/trace/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time ms', tot/1e6)
for r in rows[:32]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
rm -rf $O/trace
