"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel (substring filter).
usage: python tools/pmc_summary.py <dir> [kernel-substring ...] ; prints JSON."""
import collections, csv, glob, json, sys
d, pats = sys.argv[1], sys.argv[2:] or ["scan_fwd_kernel", "scan_bwd_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = next((p for p in pats if p in r["Kernel_Name"]), None)
        if key is None:
            continue
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({k: {c: v[0] / v[1] for c, v in sorted(cs.items())} | {"dispatches": max(v[1] for v in cs.values())}
                  for k, cs in acc.items()}, indent=1))
