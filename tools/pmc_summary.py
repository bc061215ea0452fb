"""Average the counters of a rocprofv3 --pmc run per kernel family (scan_fwd / scan_bwd)."""
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "scan_bwd" if "scan_bwd_kernel" in n else "scan_fwd" if "scan_fwd_kernel" in n else None
            if k:
                a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({k: {c: round(v[0] / v[1], 1) for c, v in cs.items()} for k, cs in acc.items()}, indent=1))
