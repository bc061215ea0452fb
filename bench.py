#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, which sets
   RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, or bare -- then bench.py starts its own N rank processes, self_launch())

A "step" = one full training step (forward + backward + Adam) of MedMamba-T (depths [2,2,4,2], dims
[96,192,384,768], d_state 16) on one synthetic batch of 64 3x224x224 images per GPU (BASELINE.json
configs[1]; configs[3] for N > 1 = the same per-rank work under DDP over RCCL: weak scaling).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line:
  value     = images/s of the whole job (N * 64 * K / max-over-ranks time)
  roofline  = dominant hand-written kernel (the selective-scan backward): algorithmic bytes
              (SURVEY.md section 8d formulas) / its HIP-event-measured duration inside the timed region,
              against the 8 TB/s HBM3E peak; `traffic` = measured HBM bytes per launch from rocprofv3 PMC
              passes when profiles/scan_traffic.json is present, else null
  cpu_baseline = the CPU restatement of the same training step (oracle scan in C/OpenMP + torch CPU ops),
              timed on this host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn as nn

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
# The scan kernels' second ceiling: vector-instruction issue.  Measured on MI355X (tools/microbench/valu_rate.hip,
# profiles/r02_valu_rate.txt): one wave64 VALU instruction occupies a SIMD for 4.3 cycles (v_fma_f32, v_mul_f32, v_pk_*_f32
# alike), v_exp_f32 for 8.2; 1024 SIMDs; 2.4 GHz is the chip's maximum clock (in-kernel measurement: 2.2-2.35 GHz).
# Wave-level VALU instructions per (channel, state, position) element of the shipped kernels, counted in their ISA
# (tools/kernel_mix.py): per 32-position chunk of an 8-channel x 16-state wave (64 lanes x 64 elements).
VALU_CYCLES_PER_INSTR, N_SIMD, MAX_CLOCK_HZ = 4.3, 1024, 2.4e9
# scan_bwd: VALU wave-instructions per state element of the software-pipelined SS2D backward (round 3, scan_ss2d_bwd.hip), from the
# rocprofv3 counters of the bench step itself (profiles/r03_scan_pmc_summary.json: SQ_INSTS_VALU 161.3 M per launch x 64 lanes over
# 524 M state elements per launch -- 154.4 M before the 16-byte LDS operand layout, which traded 7 M register moves for half the LDS
# cycles and is 2.5 % faster; 175.2 M = 21.4 before this round's instruction diet); round 2's kernel: 24.3 (1555 per 32-position
# chunk of an 8-channel wave).  scan_fwd: 43.6 M x 64 / 524 M.  NOTE: the kernels are bound by the issue of ALL
# instruction kinds (~5.5 cycles each across a SIMD's waves), so this vector-only floor understates what the shipped code needs.
VALU_INSTR_PER_STATE_ELEM = {"scan_bwd": 19.70, "scan_fwd": 5.33}

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=64, help="per GPU (BASELINE.json: bs=64)")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--num-classes", type=int, default=8)
    ap.add_argument("--variant", default="T", choices=["T", "B", "SSD"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="autocast dtype of the dense GEMM/conv ops; the scan is fp32 in both (MedMamba.py:403-409)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--miopen-find", dest="miopen_find", action="store_true",
                    help="torch.backends.cudnn.benchmark (MIOpen find mode for the dense-conv branch).  Off by default: with "
                         "the BatchNorm layers on our own kernels it no longer wins on average (27.0 vs 27.0 ms per step) and its "
                         "per-process algorithm picks spread the step time from 26.9 to 33.5 ms")
    ap.add_argument("--no-miopen-find", dest="miopen_find", action="store_false", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal: rendezvous + one all-reduce + the JSON line, no model and no GPU needed "
                         "(MEDSCAN_DIST_BACKEND=gloo on a CPU-only host)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def host_cores():
    """CPU cores this process may really use: min(affinity mask, cgroup CPU quota).  The GPU box shows 256
    logical CPUs but grants a 16-CPU quota; spinning 256 OpenMP threads on that quota stalls for minutes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


WORKLOADS = {"T": "depths/dims/d_state of BASELINE.json configs[1]",
             "B": "depths [2,2,12,2], dims [128..1024]: BASELINE.json configs[2]",
             "SSD": "CNN_Mamba.VSSM defaults, the SSD/Mamba-2 variant the reference's train.py imports"}


def valu_ceiling(kind, k):
    """Issue-bound floor of the kernel: state elements / 64 lanes x instructions per element x cycles per instruction over
    all SIMDs at the maximum clock, next to the measured time -- the HBM fraction cannot exceed frac / busy_at_max_clock."""
    elems, ipe = k.get("state_elems", 0), VALU_INSTR_PER_STATE_ELEM.get(kind)
    if not elems or not ipe:
        return None
    floor_ms = elems / 64.0 * ipe * VALU_CYCLES_PER_INSTR / (N_SIMD * MAX_CLOCK_HZ) * 1e3
    return {"instr_per_state_element": round(ipe, 2), "cycles_per_instr": VALU_CYCLES_PER_INSTR,
            "issue_floor_ms_per_launch": round(floor_ms / k["launches"], 4),
            "busy_at_max_clock": round(floor_ms / k["ms"], 3)}


def cpu_baseline_worker(args):
    """The same train step on the host: module surface + oracle/ss2d_oracle.py (torch CPU ops + C/OpenMP scan).
    Bounded sample: `cpu_batch` images x (1 warm-up + cpu_steps timed) steps.  Runs in its own process."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    from medical_image_classification_amd.train import build_model
    from oracle import ss2d_oracle
    torch.manual_seed(0)
    n_cls = 2                               # BASELINE.json configs[0]: "CPU-only PyTorch train.py on 2 synthetic classes"
    net = build_model(num_classes=n_cls, variant=args.variant)
    ss2d_oracle.install(net)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    lossf = nn.CrossEntropyLoss()
    x = torch.randn(args.cpu_batch, 3, args.res, args.res)
    y = torch.randint(0, n_cls, (args.cpu_batch,))

    def step():
        opt.zero_grad(set_to_none=True)
        loss = lossf(net(x), y)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()                                   # warm-up (allocator, oneDNN primitives)
    print(f"[cpu_baseline] warm-up step {time.perf_counter() - t0:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(args.cpu_batch * args.cpu_steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"MedMamba-{args.variant} fp32 full train step (fwd+bwd+Adam), 2 classes (BASELINE.json configs[0]), bs={args.cpu_batch}, "
                      f"{args.res}x{args.res}, 1 warm-up + {args.cpu_steps} timed step(s), {dt:.1f} s timed; "
                      "scan = oracle/scan_oracle.c (OpenMP), other ops = torch CPU. The reference's own CPU path "
                      "(Python-loop selective_scan_ref) measured 0.0046 images/s on 8 vCPU (BASELINE.md section 2)."}


def cpu_baseline(args):
    """Run the CPU leg in a fresh interpreter (no HIP runtime in it, OMP settings applied before any OpenMP
    runtime starts) under a hard timeout, so it can never hang the bench."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-batch", str(args.cpu_batch),
           "--cpu-steps", str(args.cpu_steps), "--res", str(args.res), "--num-classes", str(args.num_classes),
           "--variant", args.variant]
    env = dict(os.environ, OMP_NUM_THREADS=str(host_cores()), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=300, check=True)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 -- report, never hang or crash the GPU result
        return {"value": None, "unit": "images/s", "cores": host_cores(), "kind": "port",
                "sample": f"cpu baseline leg failed: {type(e).__name__}: {e}"}


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without an external launcher: start N FRESH interpreters of this script, one rank per GPU, with
    the env:// rendezvous the reference's ddp_train.py reads (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT,
    ddp_train.py:64-81), wait for all of them, relay rank 0's single JSON line, and fail if any rank fails.  This process
    has not touched the GPU (no HIP call before this point) and never execs: the ranks are ordinary child processes."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                LOCAL_WORLD_SIZE=str(args.gpus))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between the ranks on this host driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    pending = set(range(args.gpus))
    while pending and failed is None:
        for r in list(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
        if pending and failed is None:
            time.sleep(0.05)          # (rank 0's stdout carries one short line at its very end: the pipe cannot fill up meanwhile)
    if failed is not None:                  # one rank died: the others would wait in the rendezvous / a collective for ever
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print(f"[bench] rank {failed[0]} exited with code {failed[1]}", file=sys.stderr, flush=True)
        sys.exit(failed[1] if 0 < failed[1] < 256 else 1)
    out = procs[0].stdout.read().decode()
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr, flush=True)
        sys.exit(1)
    print(lines[-1], flush=True)


def dry_launch(args):
    """Launcher rehearsal (tests/test_ddp_cpu.py): everything bench.py does around the model -- rendezvous, barrier, one
    all-reduce (max over ranks of a per-rank time), rank 0's one JSON line -- and nothing else."""
    from medical_image_classification_amd.ddp_train import setup_distributed
    distributed, rank, world, local_rank = setup_distributed(None)
    if args.gpus > 1 and (not distributed or world != args.gpus):
        raise RuntimeError(f"--gpus {args.gpus}: WORLD_SIZE is {world}")
    on_gpu = distributed and dist.get_backend() == "nccl"
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=f"cuda:{local_rank}" if on_gpu else "cpu")
    if distributed:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry-launch", "value": t.item(), "unit": "ranks", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "backend": dist.get_backend() if distributed else None}), flush=True)
    if distributed:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker(args)), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:          # before the first torch.cuda.* call of this process
        self_launch(args)
        return
    if args.dry_launch:
        dry_launch(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON result: anything a library writes to fd 1 meanwhile (RCCL prints a version
    # banner there when its first communicator is created) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback in the product path)")
    if world != args.gpus and args.gpus > 1:
        raise RuntimeError(f"--gpus {args.gpus}: launched with WORLD_SIZE {world} (torch.distributed.run --nproc-per-node must "
                           f"equal --gpus; without a launcher bench.py starts its own ranks)")
    from medical_image_classification_amd import selective_scan_interface as ssi
    from medical_image_classification_amd.ddp_train import setup_distributed, wrap_ddp
    from medical_image_classification_amd.train import build_model, make_adam, synthetic_batch, train_step

    distributed, rank, world, local_rank = setup_distributed("nccl")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)   # MIOpen find mode for the conv branch (first warm-up step)

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    torch.manual_seed(0)
    net = build_model(num_classes=args.num_classes, variant=args.variant).to(device)
    net.train()
    model = wrap_ddp(net, distributed, local_rank)
    opt = make_adam(net.parameters(), lr=1e-4)
    lossf = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    images, labels = synthetic_batch(args.batch_size, args.num_classes, args.res, device, gen)
    ac = torch.bfloat16 if args.dtype == "bf16" else None

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model on {device}, world {world}; warm-up {args.warmup} steps")
    from medical_image_classification_amd import medmamba as _mm
    for i in range(args.warmup):
        train_step(model, opt, lossf, images, labels, ac)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
        if i == 0:
            # two-stream blocks (opt-in, MEDSCAN_BRANCH_STREAMS=auto): decided per process by measurement -- 15+ extra untimed
            # steps, every rank runs the same number of them (they contain the gradient all-reduce); a no-op by default
            _mm.autotune_branch_streams(lambda: train_step(model, opt, lossf, images, labels, ac), device, log=log)
    barrier()
    ssi.TIMER.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, opt, lossf, images, labels, ac)
    barrier()
    elapsed = time.perf_counter() - t0
    ssi.TIMER.enabled = False
    kern = ssi.TIMER.summary()
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()
    log(f"timed {args.steps} steps in {elapsed:.3f} s")

    if rank == 0:
        value = world * args.batch_size * args.steps / elapsed
        dom = "scan_bwd" if kern.get("scan_bwd", {}).get("ms", 0) >= kern.get("scan_fwd", {}).get("ms", 0) else "scan_fwd"
        if dom not in kern:          # the chunked-GEMM SSD form ran (MEDSCAN_SSD_CHUNKED_MIN_STATE): no selective-scan launch to price
            kern = {}
            k = {"bytes": 0, "ms": 1.0, "launches": 1}
        else:
            k = kern[dom]
        achieved = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tp):            # PMC passes are tied to the workload they were taken on
            tj = json.load(open(tp))
            if tj.get("workload") == f"{args.variant}-{args.res}-bs{args.batch_size}":
                traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 4),
                    "algorithmic_bytes_per_launch": k["bytes"] // k["launches"],
                    "valu": valu_ceiling(dom, k),
                    "other": {n: {"GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                  "ms_per_step": round(v["ms"] / args.steps, 3)} for n, v in kern.items()}}
        out = {"metric": f"images/sec MedMamba-{args.variant} 3x{args.res}x{args.res} bs={args.batch_size} train step",
               "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype + "+f32scan" if args.dtype == "bf16" else "f32",
               "data": "synthetic",
               "config": {"workload": f"MedMamba-{args.variant} ({WORKLOADS[args.variant]}) full "
                                      f"training step fwd+bwd+Adam, {args.batch_size} x 3x{args.res}x{args.res} per GPU, "
                                      f"{args.num_classes} classes, random-init weights; ONE resident synthetic batch re-used "
                                      "every step, no per-step loss.item() host sync (the reference loop has one, train.py:80)",
                          "global_batch": world * args.batch_size, "parallelism": f"dp{world}",
                          "loss": round(final_loss, 4)},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline and args.variant == "SSD":
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": host_cores(), "kind": "port",
                                   "sample": "not timed for the non-headline SSD variant (its restatement is a float64 "
                                             "Python loop over L, sized for parity tests only)"}
        elif world == 1 and not args.no_cpu_baseline:
            log("cpu_baseline leg ...")
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
