"""Steady-state per-kernel summary of a rocprofv3 --kernel-trace CSV of `bench.py`: keeps only the last `--steps`
training steps (warm-up, MIOpen/hipBLASLt first-call work and the cpu_baseline leg drop out), prints per-kernel time per
step and the total GPU-busy time per step.  A step boundary = one launch of the loss kernel (nll_loss_forward).

    rocprofv3 --kernel-trace -d gpurun_out/prof -o tr --output-format csv -- python3 bench.py --steps 20 --warmup 8
    python tools/steady_state_stats.py gpurun_out/prof/.../tr_kernel_trace.csv --steps 16 [--csv out.csv]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        return "hipBLASLt " + name[:40]
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--marker", default="nll_loss_forward")
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + 1:
        raise SystemExit(f"only {len(marks)} step markers in the trace")
    lo, hi = marks[-a.steps - 1], marks[-1]              # marker k .. marker k+steps: exactly `steps` steps
    win = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / a.steps
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in win:
        agg[short(n)][0] += e - s
        agg[short(n)][1] += 1
    busy = sum(v[0] for v in agg.values()) / a.steps
    out = sorted(agg.items(), key=lambda kv: -kv[1][0])
    print(f"steps {a.steps}: wall {wall / 1e6:.3f} ms/step, GPU busy {busy / 1e6:.3f} ms/step, {len(win) / a.steps:.0f} launches/step")
    print(f"{'ms/step':>8} {'calls/step':>10} {'avg us':>9}  kernel")
    for n, (t, c) in out[:a.top]:
        print(f"{t / a.steps / 1e6:8.3f} {c / a.steps:10.1f} {t / c / 1e3:9.1f}  {n}")
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "ms_per_step", "calls_per_step", "avg_us"])
            w.writerow([f"TOTAL (wall {wall / 1e6:.3f} ms/step)", f"{busy / 1e6:.4f}", f"{len(win) / a.steps:.1f}", ""])
            for n, (t, c) in out:
                w.writerow([n, f"{t / a.steps / 1e6:.4f}", f"{c / a.steps:.2f}", f"{t / c / 1e3:.2f}"])


if __name__ == "__main__":
    main()
