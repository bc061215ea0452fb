"""Category totals and the small-kernel tail of a steady-state table (tools/steady_state_stats.py --csv): python tools/step_categories.py file.csv [max_us]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))[1:]
max_us = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0


def cat(k):
    for key, name in (("scan_bwd", "scan_bwd"), ("ss2d_fwd", "scan_fwd"), ("scan_fwd", "scan_fwd"), ("gemm_bf16", "ms gemm"), ("ms::bn_", "ms bn"),
                      ("ms::dwconv", "ms dwconv"), ("ms::ln", "ms ln/ln_gate"), ("ms::dtproj", "ms dtproj"), ("ms::", "ms other"),
                      ("igemm", "miopen igemm"), ("SubTensor", "miopen helpers"), ("batched_transpose", "miopen helpers"),
                      ("Cijk", "hipblaslt"), ("multi_tensor", "optimizer"), ("FusedAdam", "optimizer"), ("FillFunctor", "fill"),
                      ("fillBuffer", "fill"), ("reduce_kernel", "aten reduce"), ("CatArray", "aten cat")):
        if key in k:
            return name
    return "aten copy/cast" if "copy" in k.lower() else "aten other"


tot = collections.defaultdict(lambda: [0.0, 0.0])
for r in rows:
    c = cat(r["kernel"]); tot[c][0] += float(r["ms_per_step"]); tot[c][1] += float(r["calls_per_step"])
print(f"total {sum(v[0] for v in tot.values()):.3f} ms, {sum(v[1] for v in tot.values()):.0f} launches")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{v[0]:7.3f} ms {v[1]:6.0f}  {k}")
small = [r for r in rows if float(r["avg_us"]) <= max_us]
print(f"<= {max_us} us: {sum(float(r['calls_per_step']) for r in small):.0f} launches, {sum(float(r['ms_per_step']) for r in small):.3f} ms")
for r in sorted(small, key=lambda r: -float(r["calls_per_step"])):
    print(f"{float(r['ms_per_step']):7.3f} {float(r['calls_per_step']):6.1f} {float(r['avg_us']):6.1f}  {r['kernel'][:140]}")
