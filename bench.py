#!/usr/bin/env python3
"""bench.py -- Ginteractions/s of the all-pairs hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts its own N
rank processes -- one per GPU, before this process has touched the GPU or imported torch --
and relays rank 0's line (spawn_ranks below).

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
THREAD_CAP = 16                  # CPU share of one GPU on the GPU box (OMP_NUM_THREADS there)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CU x 64 FMA lanes x 2 x 2.4 GHz (MI355X_MICROARCH.md)


def cpu_baseline(budget_s=20.0):
    """Time the reference's multithreaded AVX2 path on this host (bounded sample)."""
    import numpy as np  # noqa: F401
    import nbody_mi355x as nb
    import oracle_lib as ol
    threads = max(1, min(THREAD_CAP, os.cpu_count() or 1))
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
        # the machine: what the OS reports, and why fewer threads were used (a one-GPU share
        # of a GPU box is 16 CPUs; os.cpu_count() there shows the whole machine)
        "host_cpu_count": os.cpu_count(), "host_cpu_model": _cpu_model(), "thread_cap": THREAD_CAP,
    }


def _gint(n, steps, seconds):
    return float(n) * float(n) * steps / seconds / 1e9


def _timed_runs(sim, substeps, reps):
    """`reps` big steps of `substeps` small steps through the handle API; device seconds of
    each (hipEvents around the kernels, nbody_mi355x_last_run_seconds), sorted."""
    times = []
    for _ in range(reps):
        sim.run(substeps, snapshot=False)
        times.append(sim.last_run_seconds())
    return sorted(times)


def other_configs(nb):
    """The other BASELINE.json configs on ONE GPU, beside the headline (config 4): bounded to
    well under a minute of GPU time.  All through the handle API (create / upload / run), the
    path SimulateMI355X() takes; times are device times of the kernels."""
    import numpy as np
    out = {}
    frac = lambda gint, prec="fp64": round(gint * FLOP_PER_INTERACTION / 1e3 / (FP64_VECTOR_PEAK_TFLOPS * (2 if prec == "fp32" else 1)), 4)
    # config 2: the saturn ring scene of saturn_ring.cc (N = 500, dt = 0.01 s; ICs = the golden
    # fixture dumped from the reference's own RNG streams), one big step of the reference's
    # 21 600 small steps, launch-bound: microseconds per small step, no roofline
    with np.load(REPO / "tests" / "golden" / "g3_saturn_ring_500.npz") as z:
        p, m, dt = z["particles"], z["mass"], float(z["dt"])
    with nb.Simulation(p.shape[0], dt) as sim:
        sim.upload(p, m)
        sim.run(2048, snapshot=False)
        t = _timed_runs(sim, 21600, 3)
    out["config2_saturn_ring_n500_fp64"] = {"us_per_small_step": round(t[1] / 21600 * 1e6, 3), "small_steps": 21600,
                                            "repetitions": 3, "statistic": "median",
                                            "kernel": "step_wave_kernel, hipGraph replay of 64 small steps"}
    # config 3: Plummer N = 65 536 fp64 ("single-GPU roofline")
    n = 65536
    p, m = nb.ic_plummer(n, 42)
    with nb.Simulation(n, 1e-3) as sim:
        sim.upload(p, m)
        sim.run(50, snapshot=False)
        t = _timed_runs(sim, 50, 5)
    g = _gint(n, 50, t[2])
    out["config3_plummer_n65536_fp64"] = {"value": round(g, 1), "unit": "Ginteractions/s", "frac": frac(g),
                                          "ms_per_step": round(t[2] / 50 * 1e3, 4), "small_steps": 50,
                                          "repetitions": 5, "statistic": "median"}
    # config 4 again with the equal-mass forms of the tile loop switched off: what a scene
    # with mixed masses gets (the three benchmark scenes are equal-mass systems)
    n = 1048576
    p, m = nb.ic_plummer(n, 42)
    os.environ["NBODY_MI355X_SYM_NOEQ"] = "1"         # read by the library at every step
    try:
        with nb.Simulation(n, 1e-3) as sim:
            sim.upload(p, m)
            sim.run(1, snapshot=False)
            t = _timed_runs(sim, 1, 5)
    finally:
        del os.environ["NBODY_MI355X_SYM_NOEQ"]
    g = _gint(n, 1, t[2])
    out["config4_plummer_n1048576_fp64_general_masses"] = {
        "value": round(g, 1), "unit": "Ginteractions/s", "frac_general_masses": frac(g),
        "ms_per_step": round(t[2] * 1e3, 3), "repetitions": 5, "statistic": "median"}
    # config 5 on one GPU: uniform cube N = 4 194 304, fp64 and FP32 forces
    n = 4194304
    p, m = nb.ic_uniform_cube(n, 7)
    for prec, code in (("fp64", nb.FP64), ("fp32", nb.FP32)):
        with nb.Simulation(n, 1e-4, precision=code) as sim:
            sim.upload(p, m)
            sim.run(1, snapshot=False)
            t = _timed_runs(sim, 1, 2)
        g = _gint(n, 1, 0.5 * (t[0] + t[1]))
        out[f"config5_uniform_cube_n4194304_{prec}_1gpu"] = {
            "value": round(g, 1), "unit": "Ginteractions/s", "frac": frac(g, prec),
            "ms_per_step": round(0.5 * (t[0] + t[1]) * 1e3, 2), "repetitions": 2, "statistic": "mean"}
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(cmd, n, env=None, timeout=None):
    """Start `cmd` n times as rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, as torch.distributed.run would set them) and wait.
    Returns (exit code, rank 0's stdout).  Rank 0's stdout is captured; the other ranks' goes
    to this process's stderr.  The first rank to fail ends the others; the exit code is the
    first failure's (or 0).  The caller must not have initialised the GPU: children are fresh
    processes, nothing is re-executed in place.
    Stands where the reference has one process and one GPU: SimulateGPU(),
    particle_simulation_GPU.cu:277-303."""
    import subprocess
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(_free_port()))
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=True))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc, out0 = 0, ""
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124
                print("bench.py: rank processes timed out", file=sys.stderr)
            for r in live:
                procs[r].terminate()          # the exact processes started above
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.05)
    reader.join(timeout=10)
    out0 = "".join(chunks)
    return rc, out0


def self_launch(args):
    """--gpus N > 1 without a launcher: N children of this script, rank 0's JSON line relayed
    as this process's only stdout line."""
    rc, out = spawn_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus)
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc != 0:
        sys.stderr.write(out)
        sys.exit(rc if 0 < rc < 256 else 1)
    print(lines[-1], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=1048576)
    ap.add_argument("--ic", choices=["plummer", "cube"], default="plummer",
                    help="plummer: BASELINE configs 3/4 (seed 42, dt 1e-3); cube: config 5 (uniform cube, seed 7, dt 1e-4)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--precision", choices=["fp64", "fp32"], default="fp64")
    ap.add_argument("--integrator", choices=["euler", "kick_drift", "kdk"], default="euler",
                    help="euler = the reference's (the headline); kdk = leapfrog, one more O(N) kernel per step")
    ap.add_argument("--device-ics", action="store_true",
                    help="make the scene on the device(s), one random stream per body (statistically the default scene, not bit for bit)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE configs that the default N=1 run measures beside the headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-devices", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs (device = local_rank %% device_count)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal only: with one rank, still take the multi-rank code path (RCCL "
                         "all-gather + reduce-scatter over a world of 1, split steps)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)              # (nothing above has imported torch or touched the GPU)

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
    if args.device_ics:                               # every rank makes the bodies it holds, on its GPU
        kind, seed, dt = (nb.SCENE_PLUMMER, 42, 1e-3) if args.ic == "plummer" else (nb.SCENE_UNIFORM_CUBE, 7, 1e-4)
        p, m = nb.make_scene(kind, n, seed=seed), None
        scene = (f"{'plummer_sphere' if args.ic == 'plummer' else 'uniform_cube'}_n{n}_{args.precision}_seed{seed}_dt{dt:g}"
                 "_one_stream_per_body_made_on_device")
    elif args.ic == "plummer":                        # same ICs on every rank
        (p, m), dt, scene = nb.ic_plummer(n, 42), 1e-3, f"plummer_sphere_n{n}_{args.precision}_seed42_dt1e-3"
    else:
        (p, m), dt, scene = nb.ic_uniform_cube(n, 7), 1e-4, f"uniform_cube_n{n}_{args.precision}_seed7_dt1e-4"
    sim = ShardedNBody(p, m, dt, rank=rank, world=world, device=device,
                       backend=HipBackend(precision=prec, variant=args.variant),
                       overlap=not args.no_overlap, force_collectives=args.force_collectives,
                       integrator={"euler": nb.EULER, "kick_drift": nb.KICK_DRIFT, "kdk": nb.KDK}[args.integrator])
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
    # one HIP event between every two steps, on torch's current stream == the launch stream:
    # the K steps are timed as a whole (the contract's number) AND one by one (their median)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(K):
        sim.step()
        evs[k + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    dev_s = evs[0].elapsed_time(evs[K]) * 1e-3
    per_step = sorted(evs[k].elapsed_time(evs[k + 1]) * 1e-3 for k in range(K))
    med_s = per_step[K // 2] if K % 2 else 0.5 * (per_step[K // 2 - 1] + per_step[K // 2])
    if world > 1:
        t = torch.tensor([wall, dev_s, med_s], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_s, med_s = float(t[0]), float(t[1]), float(t[2])

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
        uses_sym = sim.workspace is not None and args.variant in (0,) + nb.SYM_VARIANTS and not (
            args.variant == 0 and world == 1 and prec == nb.FP64 and n <= nb.WAVE_MAX_BODIES)
        kernel_label = ("pair-symmetric tiles + finish" if uses_sym else
                        "one-sided step_wave_kernel" if (world == 1 and prec == nb.FP64 and n <= nb.WAVE_MAX_BODIES
                                                         and args.variant in (0, 7)) else "one-sided step_kernel")
        # a step may walk the block distances in batches (one launch of the tile kernel each)
        launches = 1
        if uses_sym:
            launches = sim.backend.tile_launches(sim.S, world, sim.count, n, sim.workspace,
                                                 variant=sim.sym_variant if world > 1 or args.force_collectives else None)
            if world > 1 or args.force_collectives:
                launches += 1          # + the diagonal tiles' launch (phase 1)
        line = {
            "metric": "Ginteractions/s at N=1M fp64, 1/2/4/8 MI355X; % of FP64 vector peak",
            "value": round(value, 2), "unit": "Ginteractions/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall / K * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if prec == nb.FP64 else "f32", "data": "synthetic",
            "config": {"workload": scene,
                       "bodies": n, "interactions_per_step": inter_per_step,
                       "parallelism": (f"i-shard x{world}, {'RCCL' if args.backend == 'nccl' else args.backend + ' (rehearsal, host-staged)'} all-gather of positions"
                                       + (" + reduce-scatter of pair-symmetric partial sums" if sim.acc_all is not None else "")
                                       + " per step"
                                       + ("" if args.no_overlap or world == 1 else ", all-gather overlapped with the own-slab part")),
                       "kernel": kernel_label,
                       "kernel_variant": args.variant, "integrator": args.integrator},
            "roofline": {"bound": "valu_fp64" if prec == nb.FP64 else "valu_fp32", "achieved": round(ach, 3),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         # PMC bytes (profiles/traffic.json holds them per STEP), per launch like `achieved`
                         "traffic": (round(traffic / launches) if traffic else None), "traffic_per_step": traffic,
                         "kernel": ("sym_tiles_kernel+sym_finish_kernel" if uses_sym else kernel_label.split()[-1]),
                         # `achieved` = the step's algorithmic FLOP / the step's device time; a step is
                         # `launches_per_step` launches of the dominant kernel, each covering that share
                         "launches_per_step": launches,
                         "avg_step_ms": round(dev_s / K * 1e3, 3),
                         "avg_launch_ms": round(dev_s / K / launches * 1e3, 3),
                         "interactions_per_launch": inter_per_step / world / launches,
                         "flop_per_interaction": FLOP_PER_INTERACTION},
            "pct_of_fp64_vector_peak": round(100 * ach / FP64_VECTOR_PEAK_TFLOPS, 2),
            # the median of the K individually timed steps (HIP events), max over ranks
            "median_of_steps": {"timed_steps": K, "ms_per_step": round(med_s * 1e3, 3),
                                "value": round(inter_per_step / med_s / 1e9, 2),
                                "min_ms": round(per_step[0] * 1e3, 3), "max_ms": round(per_step[-1] * 1e3, 3)},
        }
        headline = (world == 1 and n == 1048576 and args.ic == "plummer" and prec == nb.FP64 and args.variant == 0
                    and not args.force_collectives)
        if headline and not args.no_other_configs:
            del sim                                    # (its 7 GB workspace)
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(nb)
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
