#!/usr/bin/env python3
"""bench.py -- Ginteractions/s of the all-pairs hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one small step (one pass of the O(N^2) force accumulation + fused Euler
update) over the whole synthetic system.  Workload: Plummer sphere, N = 1,048,576,
fp64, seed 42 (BASELINE.json configs[3]; the configuration the metric is quoted on);
with --gpus N the SAME system is sharded over N ranks (strong scaling), positions
all-gathered with RCCL every step.  Inputs are resident in HBM before the timed
region.  One JSON line on stdout (rank 0).

Extra objects in the line:
  roofline      FP64 vector-ALU roofline of the step kernel: 20 FLOP per interaction
                 / average launch duration measured with HIP
                events on the launch stream, against the 78.6 TFLOP/s vector peak.
  cpu_baseline  the reference's own SimulateAVX2() (compiled from /root/reference in the
                build container into oracle/_ref/libnbody_ref.so; kind "reference"), or
                the oracle's restatement of it (kind "port") when that file is absent,
                on this host's cores, on a bounded sample.  N=1 only, rank 0 only.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))
sys.path.insert(0, str(REPO / "tests"))

FLOP_PER_INTERACTION = 20.0      # BASELINE.md section 3 / SURVEY 8d convention
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CU x 64 FMA lanes x 2 x 2.4 GHz (MI355X_MICROARCH.md)


def cpu_baseline(budget_s=20.0):
    """Time the reference's multithreaded AVX2 path on this host (bounded sample)."""
    import numpy as np  # noqa: F401
    import nbody_mi355x as nb
    import oracle_lib as ol
    threads = max(1, min(16, os.cpu_count() or 1))   # the box's CPU share for one GPU is 16
    n = 32768
    p, m = nb.ic_plummer(n, 42)
    kind = "reference" if ol.have_ref() else "port"

    def run(sub):
        if kind == "reference":
            _, sec = ol.ref_simulate(p, m, 1, sub, 1e-3, threads=threads, mode="avx2")
            return sec
        t0 = time.perf_counter()
        ol.simulate(p, m, 1, sub, 1e-3, threads=threads, kind="avx2")
        return time.perf_counter() - t0

    t_probe = run(2) / 2
    sub = int(max(3, min(600, budget_s / max(t_probe, 1e-6))))
    sec = run(sub)
    return {
        "value": round(n * n * sub / sec / 1e9, 3), "unit": "Ginteractions/s", "cores": threads,
        "kind": kind,
        "sample": f"SimulateAVX2(), Plummer N={n} fp64 seed 42, {sub} small steps, {threads} threads, {sec:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=1048576)
    ap.add_argument("--ic", choices=["plummer", "cube"], default="plummer",
                    help="plummer: BASELINE configs 3/4 (seed 42, dt 1e-3); cube: config 5 (uniform cube, seed 7, dt 1e-4)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--precision", choices=["fp64", "fp32"], default="fp64")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-devices", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs (device = local_rank %% device_count)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal only: with one rank, still take the multi-rank code path (RCCL "
                         "all-gather + reduce-scatter over a world of 1, split steps)")
    args = ap.parse_args()

    # stdout carries ONE JSON line: libraries that print banners there (RCCL prints its
    # version on the first communicator) are sent to stderr until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import nbody_mi355x as nb
    from nbody_mi355x.sharded import HipBackend, ShardedNBody

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run "
                     f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev_index = local_rank % torch.cuda.device_count() if args.share_devices else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n, K, W = args.n, args.steps, args.warmup
    prec = nb.FP64 if args.precision == "fp64" else nb.FP32
    if args.ic == "plummer":                          # same ICs on every rank
        (p, m), dt, scene = nb.ic_plummer(n, 42), 1e-3, f"plummer_sphere_n{n}_{args.precision}_seed42_dt1e-3"
    else:
        (p, m), dt, scene = nb.ic_uniform_cube(n, 7), 1e-4, f"uniform_cube_n{n}_{args.precision}_seed7_dt1e-4"
    sim = ShardedNBody(p, m, dt, rank=rank, world=world, device=device,
                       backend=HipBackend(precision=prec, variant=args.variant),
                       overlap=not args.no_overlap, force_collectives=args.force_collectives)
    del p, m

    def barrier():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    sim.warm_collectives()                            # first-call set-up of RCCL's collectives
    for _ in range(W):
        sim.step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                       # torch's current stream == the launch stream
    for _ in range(K):
        sim.step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_s = ev0.elapsed_time(ev1) * 1e-3
    if world > 1:
        t = torch.tensor([wall, dev_s], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_s = float(t[0]), float(t[1])

    if rank == 0:
        inter_per_step = float(n) * float(n)
        value = inter_per_step * K / wall / 1e9
        # dominant kernel: the step kernel; one launch per step and rank at N=1 (two at N>1,
        # own-slab half + remote half, timed together), each rank doing n * n/world pairs
        ach = FLOP_PER_INTERACTION * (inter_per_step / world) * K / dev_s / 1e12
        peak = FP64_VECTOR_PEAK_TFLOPS * (2.0 if prec == nb.FP32 else 1.0)
        traffic = None
        tf = REPO / "profiles" / "traffic.json"
        if tf.exists() and world == 1:
            try:
                traffic = json.loads(tf.read_text()).get(f"n{n}_{args.precision}")
            except Exception:
                traffic = None
        uses_sym = sim.workspace is not None and args.variant in (0, 10, 11, 12, 14, 18) and not (
            args.variant == 0 and world == 1 and prec == nb.FP64 and n <= nb.WAVE_MAX_BODIES)
        kernel_label = ("pair-symmetric tiles + finish" if uses_sym else
                        "one-sided step_wave_kernel" if (world == 1 and prec == nb.FP64 and n <= nb.WAVE_MAX_BODIES
                                                         and args.variant in (0, 7)) else "one-sided step_kernel")
        line = {
            "metric": "Ginteractions/s at N=1M fp64, 1/2/4/8 MI355X; % of FP64 vector peak",
            "value": round(value, 2), "unit": "Ginteractions/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall / K * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if prec == nb.FP64 else "f32", "data": "synthetic",
            "config": {"workload": scene,
                       "bodies": n, "interactions_per_step": inter_per_step,
                       "parallelism": (f"i-shard x{world}, RCCL all-gather of positions"
                                       + (" + reduce-scatter of pair-symmetric partial sums" if sim.acc_all is not None else "")
                                       + " per step"
                                       + ("" if args.no_overlap or world == 1 else ", all-gather overlapped with the own-slab part")),
                       "kernel": kernel_label,
                       "kernel_variant": args.variant},
            "roofline": {"bound": "valu_fp64" if prec == nb.FP64 else "valu_fp32", "achieved": round(ach, 3),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                         "kernel": ("sym_tiles_kernel+sym_finish_kernel" if uses_sym else kernel_label.split()[-1]),
                         "avg_launch_ms": round(dev_s / K * 1e3, 3),
                         "flop_per_interaction": FLOP_PER_INTERACTION},
            "pct_of_fp64_vector_peak": round(100 * ach / FP64_VECTOR_PEAK_TFLOPS, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
