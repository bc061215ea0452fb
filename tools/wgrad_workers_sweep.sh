R=$PWD; cd /tmp; export TMPDIR=/tmp
for n in 256 384 512 768 1024; do
  rm -rf /tmp/cabl; MEDSCAN_WGRAD_WGS=$n rocprofv3 --kernel-trace -d /tmp/cabl --output-format csv -- python3 $R/tools/bench_conv3x3.py 64 T > /dev/null 2>&1
  python3 - $n <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/cabl/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    for k in ('conv3x3_wgrad_kernel', 'conv3x3_wgrad_finalize'):
        if k in r['Kernel_Name']: d[(k, int(r['Grid_Size_Y']) if 'finalize' not in k else 0, )].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
main = {k[1]: sorted(v)[len(v)//2] for k, v in d.items() if 'finalize' not in k[0]}
fin = sorted(sum([v for k, v in d.items() if 'finalize' in k[0]], []))
print('WGS', sys.argv[1], 'main by nblk*4:', {k: round(v, 1) for k, v in sorted(main.items())}, 'finalize min/med/max', round(fin[0],1), round(fin[len(fin)//2],1), round(fin[-1],1))
PY
done
