#!/bin/bash
# LDS bank-conflict share of every kernel of the training step (one --pmc pass, no tracing): gpurun_out/lds_conflicts.txt
R=$PWD; O=$R/gpurun_out/ldspmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $O/bench.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - $O > $R/gpurun_out/lds_conflicts.txt <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])[:90]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]['SQ_LDS_BANK_CONFLICT'])
print(f"{'conflict cyc':>14} {'LDS active':>14} {'share':>6} {'GUI active':>14} {'calls':>6}  kernel")
for k, c in rows[:40]:
    a = c['SQ_LDS_IDX_ACTIVE']
    print(f"{c['SQ_LDS_BANK_CONFLICT']:14.0f} {a:14.0f} {c['SQ_LDS_BANK_CONFLICT'] / a if a else 0:6.2f} {c['GRBM_GUI_ACTIVE']:14.0f} {n[k]:6d}  {k}")
PY
rm -rf $O; head -30 $R/gpurun_out/lds_conflicts.txt
