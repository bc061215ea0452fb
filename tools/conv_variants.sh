R=$PWD; cd /tmp; export TMPDIR=/tmp
for a in ${VARS:-0 pf2}; do
  L=$R/build/variants/libmedscan_conv$a.so; [ $a == 0 ] && L=$R/medical_image_classification_amd/libmedscan.so
  rm -rf /tmp/cabl; MEDSCAN_LIBRARY=$L rocprofv3 --kernel-trace -d /tmp/cabl --output-format csv -- python3 $R/tools/bench_conv3x3.py 64 T > /dev/null 2>&1
  python3 - $a <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/cabl/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'conv3x3_nhwc_kernel' in r['Kernel_Name']:
        d[(int(r['Grid_Size_X']), int(r['Grid_Size_Y']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('VAR', sys.argv[1], ' '.join(f"grid{k}: med {sorted(v)[len(v)//2]:.1f} min {min(v):.1f} us" for k, v in sorted(d.items(), reverse=True)))
PY
done
