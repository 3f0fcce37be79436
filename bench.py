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
With more than one rank (or --force-collectives) the line also proves what ran:
  collectives   what the communicator itself reports: backend, world size as
                torch.distributed sees it, RCCL version, every rank's device (name, ordinal,
                PCI bus id) -- gathered THROUGH the communicator
  per_rank_ms_per_step   every rank's own wall clock over the K timed steps
  collective_ms the two collectives of a step (all-gather of positions, reduce-scatter of
                partial sums), timed on their own after the timed region, max over ranks
  verify        (default at N > 1, --verify at N = 1; after the timed region, never inside)
                one more step whose accelerations are checked on 1024 sampled bodies against
                the oracle's exact sums (the checker, tests/oracle_lib.py), total momentum
                change ~ 0, and the positions of all bodies against ONE GPU running the same
                steps through the handle API (max relative difference <= 1e-12, SURVEY 8d
                config 4) or against a --reference-checksum file a one-GPU run wrote.
                A failed check makes the exit code non-zero.
  other_configs the other BASELINE configs: on one GPU (N = 1: configs 2, 3, 5, mixed masses, the
                FP64_FAST tier) or, with more than one rank beside the N = 1M headline, config 5
                (uniform cube N = 4 194 304, fp64 and FP32) over the SAME ranks -- north_star's second
                multi-GPU figure; never part of `value`.
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


def config2_cpu_side(p, m, dt, small_steps=4000):
    """BASELINE config 2 is "SimulateMI355X() on 1 GPU vs SimulateAVX2()" (ref: saturn_ring.cc:70):
    the CPU side of it on this host, same scene (the golden fixture's ICs), all threads of the
    GPU's CPU share -- the REAL reference's SimulateAVX2() (oracle/_ref, a thread spawned per
    small step as the reference does, particle_simulation.h:355-366) and this repository's host
    mirror of it (persistent worker pool), through nbody_host_cli.  Part of the cpu_baseline leg."""
    import subprocess
    import tempfile
    import numpy as np
    import oracle_lib as ol
    threads = max(1, min(THREAD_CAP, os.cpu_count() or 1))
    out = {"cpu_threads": threads, "cpu_small_steps": small_steps}
    if ol.have_ref():
        _, sec = ol.ref_simulate(p, m, 1, small_steps, dt, threads=threads, mode="avx2")
        out["reference_avx2_us_per_small_step"] = round(sec / small_steps * 1e6, 3)
    cli = REPO / "n-body-sim_amd" / "host" / "nbody_host_cli"
    if cli.exists():
        try:
            with tempfile.TemporaryDirectory() as tmp:
                ics = Path(tmp) / "ics.bin"
                with open(ics, "wb") as f:
                    f.write(np.int64(p.shape[0]).tobytes())
                    f.write(np.ascontiguousarray(p, dtype=np.float64).tobytes())
                    f.write(np.ascontiguousarray(m, dtype=np.float64).tobytes())
                r = subprocess.run([str(cli), "run", "avx2", str(ics), "2", str(small_steps), repr(dt), str(threads),
                                    str(Path(tmp) / "out.bin")], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                sec = json.loads(line[-1])["big_step_seconds"][1]     # (the first big step holds the pool's start)
                out["host_mirror_avx2_us_per_small_step"] = round(sec / small_steps * 1e6, 3)
            else:
                out["host_mirror_avx2_error"] = (r.stderr or "no result line").strip()[-200:]
        except Exception as e:  # noqa: BLE001
            out["host_mirror_avx2_error"] = f"{type(e).__name__}: {e}"
    return out


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
    out["config2_saturn_ring_n500_fp64"].update(config2_cpu_side(p, m, dt))
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
    # config 4 with the opt-in FP64_FAST tier (the correction series of r^-3 cut after its linear
    # term: 17 instead of 18 operations per pair; NOT the parity path, include/nbody_mi355x.h)
    with nb.Simulation(n, 1e-3, precision=nb.FP64_FAST) as sim:
        sim.upload(p, m)
        sim.run(1, snapshot=False)
        t = _timed_runs(sim, 1, 5)
    g = _gint(n, 1, t[2])
    out["config4_plummer_n1048576_fp64_fast_tier"] = {
        "value": round(g, 1), "unit": "Ginteractions/s", "frac": frac(g), "ms_per_step": round(t[2] * 1e3, 3),
        "repetitions": 5, "statistic": "median", "note": "opt-in precision tier, <= 2.7e-14 per pair; not the headline"}
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


def config5_multi_rank(nb, make_sim, barrier, reduce_max, n, world, rank):
    """BASELINE config 5 as north_star states it -- uniform cube N = 4 194 304, fp64 and FP32 forces,
    strong scaling over the ranks of this run -- beside the multi-rank headline: every rank takes
    part (same sharding, same collectives as the headline), one warm-up step, two timed steps
    between barriers, the slowest rank's time.  Rank 0 returns the entries; the one-GPU figures of
    the same scene are in the N = 1 line's `other_configs`."""
    import time as _time
    out = {}
    p, m = nb.ic_uniform_cube(n, 7)
    for name, code in (("fp64", nb.FP64), ("fp32", nb.FP32)):
        sim = make_sim(p, m, 1e-4, code)
        sim.warm_collectives()
        sim.step()
        barrier()
        t0 = _time.perf_counter()
        for _ in range(2):
            sim.step()
        barrier()
        sec = reduce_max(_time.perf_counter() - t0) / 2
        del sim
        g = float(n) * float(n) / sec / 1e9
        peak = FP64_VECTOR_PEAK_TFLOPS * (2 if name == "fp32" else 1) * world
        out[f"config5_uniform_cube_n{n}_{name}_{world}gpu"] = {
            "value": round(g, 1), "unit": "Ginteractions/s", "n_gpus": world, "scaling": "strong",
            "frac": round(g * FLOP_PER_INTERACTION / 1e3 / peak, 4), "ms_per_step": round(sec * 1e3, 2),
            "repetitions": 2, "statistic": "mean of two steps, slowest rank"}
    return out if rank == 0 else None


VERIFY_ROWS = 1024
VERIFY_TOL = {"f64": {"rows": 1e-12, "momentum": 1e-12, "positions": 1e-12},
              "f32": {"rows": 1e-4, "momentum": 1e-5, "positions": 1e-4}}


def sampled_rows(n, count=VERIFY_ROWS, seed=0):
    """The bodies whose accelerations are checked: the same on every rank."""
    import numpy as np
    return np.sort(np.random.default_rng(seed).choice(n, min(count, n), replace=False))


def rows_of_rank(rows, begin, count):
    """Which of the sampled bodies a rank owns: (their positions in `rows`, their local indices)."""
    import numpy as np
    sel = np.nonzero((rows >= begin) & (rows < begin + count))[0]
    return sel, rows[sel] - begin


def check_rows(a_gpu, a_ref, tol):
    """Sampled-row accelerations against the checker's: the largest vector difference over the
    largest acceleration of the sample (the metric of tests/test_gpu_parity.py)."""
    import numpy as np
    err = float(np.linalg.norm(a_gpu - a_ref, axis=1).max() / np.linalg.norm(a_ref, axis=1).max())
    worst = float((np.linalg.norm(a_gpu - a_ref, axis=1) / np.linalg.norm(a_ref, axis=1)).max())
    return {"max_err_over_max_acc": err, "max_rel_err": worst, "tol": tol, "ok": bool(err <= tol)}


def check_momentum(f_parts, scale_parts, tol):
    """sum_i m_i a_i over ALL bodies (added rank by rank) against sum_i m_i |a_i|: Newton's third
    law -- a size-independent property of a correct force sum, whatever the number of ranks."""
    import numpy as np
    f = np.sum(np.asarray(f_parts, dtype=np.float64), axis=0)
    scale = float(np.sum(scale_parts))
    rel = float(np.linalg.norm(f) / scale) if scale > 0 else float("inf")
    return {"rel_net_force": rel, "tol": tol, "ok": bool(rel <= tol)}


def check_positions(pos, pos_ref, tol):
    """max_i |dr_i| / |r_i| (vector norms) between two runs of the same steps."""
    import numpy as np
    if pos.shape != pos_ref.shape:
        return {"ok": False, "error": f"shape {pos.shape} against {pos_ref.shape}"}
    den = np.linalg.norm(pos_ref, axis=1)
    den = np.where(den == 0, 1.0, den)
    err = float((np.linalg.norm(pos - pos_ref, axis=1) / den).max())
    return {"max_rel_position_diff": err, "tol": tol, "ok": bool(err <= tol)}


def verify_run(args, sim, nb, dist, world, rank, steps_done, ics, dt, prec, integrator):
    """After the timed region: see `verify` in the module docstring.  Collective (every rank
    calls it); the result dict is returned on rank 0, None elsewhere."""
    import numpy as np
    n = sim.n
    tol = VERIFY_TOL["f32" if prec == nb.FP32 else "f64"]
    use_dist = dist is not None and dist.is_initialized()
    masses = sim.all_masses()                    # (n,) host copy of what the device holds
    pos_before = sim.all_positions()             # every slab made current first (collective)
    sim.step(snapshot=True)
    snap = sim.own_snapshot()                    # own bodies: {r', v', a G} at pos_before
    begin, count = sim.begins[rank], sim.count
    rows = sampled_rows(n)
    sel, local = rows_of_rank(rows, begin, count)
    mine = {"sel": sel, "acc": snap[local, 6:9],
            "f": (masses[begin:begin + count, None] * snap[:, 6:9]).sum(0),
            "scale": float((masses[begin:begin + count] * np.linalg.norm(snap[:, 6:9], axis=1)).sum())}
    parts = [None] * world
    if use_dist:
        dist.all_gather_object(parts, mine)
    else:
        parts = [mine]
    pos_after = sim.all_positions()              # collective
    if rank != 0:
        return None
    import oracle_lib as ol                      # the CHECKER (tests/): never on the timed path
    out = {"steps_before": steps_done, "sampled_rows": int(len(rows))}
    a_gpu = np.zeros((len(rows), 3))
    for part in parts:
        a_gpu[part["sel"]] = part["acc"]
    state = np.zeros((n, 6))
    state[:, :3] = pos_before
    threads = max(1, min(THREAD_CAP, os.cpu_count() or 1))
    out["rows_vs_oracle"] = check_rows(a_gpu, ol.accel_rows(state, masses, rows, threads=threads), tol["rows"])
    out["momentum"] = check_momentum([q["f"] for q in parts], [q["scale"] for q in parts], tol["momentum"])
    total = steps_done + 1
    if args.write_checksum:
        np.save(args.write_checksum, pos_after)
        out["checksum_written"] = {"file": str(args.write_checksum), "steps": total}
    if args.reference_checksum:
        out["positions_vs_reference_file"] = dict(check_positions(pos_after, np.load(args.reference_checksum), tol["positions"]),
                                                  file=str(args.reference_checksum))
    # the same steps on ONE GPU through the handle API (what SimulateMI355X() runs), when the
    # ICs are host arrays and the re-run fits the budget
    est = total * float(n) * float(n) / 2.5e12
    if ics is not None and (world > 1 or args.force_collectives) and est <= args.verify_budget_s:
        with nb.Simulation(n, dt, precision=prec, integrator=integrator, device=sim.device.index or 0) as one:
            one.upload(*ics)
            if integrator == nb.KDK:              # (the sharded driver's leapfrog steps are whole steps)
                for _ in range(total):
                    one.run(1, snapshot=False)
            else:
                one.run(total, snapshot=False)
            ref = one.download()[:, :3]
        out["positions_vs_one_gpu"] = dict(check_positions(pos_after, ref, tol["positions"]), steps=total)
    elif ics is not None and (world > 1 or args.force_collectives):
        out["positions_vs_one_gpu"] = {"skipped": f"estimated {est:.0f} s > --verify-budget-s {args.verify_budget_s:g}"}
    out["ok"] = all(v.get("ok", True) for v in out.values() if isinstance(v, dict))
    return out


def time_collectives(sim, dist, torch, world, reps=5):
    """The two collectives of a small step on their own, after the timed region: HIP events on
    the launch stream around each blocking call (the stream waits for RCCL's), max over ranks."""
    t = sim.time_collectives(reps)
    if t is None:
        return None
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, t)
        for key in ("all_gather", "reduce_scatter"):
            vals = [q[key] for q in parts if q.get(key) is not None]
            t[key] = max(vals) if vals else None
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}


def describe_collectives(torch, dist, world, backend, dev_index):
    """What the communicator reports, gathered through it."""
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": dist.get_rank(), "device_index": dev_index, "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None,
          "pid": os.getpid(), "host": os.uname().nodename}
    devices = [None] * dist.get_world_size()
    dist.all_gather_object(devices, me)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            ver = f"unavailable ({e})"
    return {"backend": dist.get_backend(), "is_rccl": backend == "nccl", "world_size_seen": dist.get_world_size(),
            "world_size_env": world, "rccl_version": ver, "hip": getattr(torch.version, "hip", None),
            "distinct_devices": len({(d["host"], d["pci_bus_id"], d["device_index"]) for d in devices}),
            "devices": devices}



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
    ap.add_argument("--precision", choices=["fp64", "fp32", "fp64fast"], default="fp64",
                    help="fp64 = the parity path (the headline); fp32 / fp64fast = the optional tiers of include/nbody_mi355x.h")
    ap.add_argument("--integrator", choices=["euler", "kick_drift", "kdk"], default="euler",
                    help="euler = the reference's (the headline); kdk = leapfrog, one more O(N) kernel per step")
    ap.add_argument("--device-ics", action="store_true",
                    help="make the scene on the device(s), one random stream per body (statistically the default scene, not bit for bit)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE configs that the default N=1 run measures beside the headline")
    ap.add_argument("--config5-bodies", type=int, default=None,
                    help="rehearsal only: size of the multi-rank config-5 leg (default 4 194 304, run beside the "
                         "N=1M headline with more than one rank); giving it runs the leg beside any headline size")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-devices", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs (device = local_rank %% device_count)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal only: with one rank, still take the multi-rank code path (RCCL "
                         "all-gather + reduce-scatter over a world of 1, split steps)")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                    help="after the timed region: sampled rows against the oracle, momentum, positions against "
                         "one GPU (default with more than one rank; see the module docstring)")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--write-checksum", type=Path, default=None,
                    help="with verification: save the positions after the verified step (.npy) for --reference-checksum")
    ap.add_argument("--reference-checksum", type=Path, default=None,
                    help="with verification: positions a ONE-GPU run of the same command wrote with --write-checksum")
    ap.add_argument("--verify-budget-s", type=float, default=90.0,
                    help="skip the one-GPU re-run of the verification when it would take longer than this")
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
    prec = {"fp64": nb.FP64, "fp32": nb.FP32, "fp64fast": nb.FP64_FAST}[args.precision]
    if args.device_ics:                               # every rank makes the bodies it holds, on its GPU
        kind, seed, dt = (nb.SCENE_PLUMMER, 42, 1e-3) if args.ic == "plummer" else (nb.SCENE_UNIFORM_CUBE, 7, 1e-4)
        p, m = nb.make_scene(kind, n, seed=seed), None
        scene = (f"{'plummer_sphere' if args.ic == 'plummer' else 'uniform_cube'}_n{n}_{args.precision}_seed{seed}_dt{dt:g}"
                 "_one_stream_per_body_made_on_device")
    elif args.ic == "plummer":                        # same ICs on every rank
        (p, m), dt, scene = nb.ic_plummer(n, 42), 1e-3, f"plummer_sphere_n{n}_{args.precision}_seed42_dt1e-3"
    else:
        (p, m), dt, scene = nb.ic_uniform_cube(n, 7), 1e-4, f"uniform_cube_n{n}_{args.precision}_seed7_dt1e-4"
    integrator = {"euler": nb.EULER, "kick_drift": nb.KICK_DRIFT, "kdk": nb.KDK}[args.integrator]
    sim = ShardedNBody(p, m, dt, rank=rank, world=world, device=device,
                       backend=HipBackend(precision=prec, variant=args.variant),
                       overlap=not args.no_overlap, force_collectives=args.force_collectives,
                       integrator=integrator)
    explicit_verify = args.verify is True
    do_verify = (world > 1) if args.verify is None else args.verify
    # the one-GPU re-run of the verification starts from the same host arrays (rank 0 only)
    ics = (p, m) if (do_verify and rank == 0 and not args.device_ics) else None
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
    own_wall = wall
    if world > 1:
        t = torch.tensor([wall, dev_s, med_s], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_s, med_s = float(t[0]), float(t[1]), float(t[2])

    # ---- after the timed region: what proves a multi-rank run (nothing here is timed into `value`)
    proof = {}
    if use_dist:
        proof["collectives"] = describe_collectives(torch, dist, world, args.backend, dev_index)
        walls = [None] * world
        dist.all_gather_object(walls, own_wall)
        proof["per_rank_ms_per_step"] = [round(w / K * 1e3, 3) for w in walls]
        proof["collective_ms"] = time_collectives(sim, dist, torch, world)
    verify_failed = False
    if do_verify:
        try:
            v = verify_run(args, sim, nb, dist if use_dist else None, world, rank, W + K, ics, dt, prec, integrator)
        except Exception as e:  # noqa: BLE001  (an unusable checker is reported, a failed check is fatal)
            if explicit_verify:
                raise
            v = {"ok": None, "error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            proof["verify"] = v
            verify_failed = v.get("ok") is False
    del ics

    # north_star's other multi-GPU figure (config 5: N = 4M, fp64 and FP32) beside the multi-rank headline
    multi_other = None
    if (world > 1 and not args.no_other_configs and args.ic == "plummer" and prec == nb.FP64 and args.variant == 0
            and integrator == nb.EULER and (n == 1048576 or args.config5_bodies is not None)):
        # (the headline's handle stays: a few GB per rank of 288)
        def make_sim(p5, m5, dt5, code):
            return ShardedNBody(p5, m5, dt5, rank=rank, world=world, device=device,
                                backend=HipBackend(precision=code, variant=0), overlap=not args.no_overlap)

        def reduce_max(x):
            t5 = torch.tensor([x], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t5, op=dist.ReduceOp.MAX)
            return float(t5[0])

        try:
            multi_other = config5_multi_rank(nb, make_sim, barrier, reduce_max, args.config5_bodies or 4194304, world, rank)
        except Exception as e:  # noqa: BLE001  (the headline line is not lost to its companion)
            multi_other = {"config5_error": f"{type(e).__name__}: {e}"} if rank == 0 else None

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
            args.variant == 0 and world == 1 and prec != nb.FP32 and n <= nb.WAVE_MAX_BODIES)
        kernel_label = ("pair-symmetric tiles + finish" if uses_sym else
                        "one-sided step_wave_kernel" if (world == 1 and prec != nb.FP32 and n <= nb.WAVE_MAX_BODIES
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
            "vs_baseline": None, "dtype": "f32" if prec == nb.FP32 else "f64", "data": "synthetic",
            "config": {"workload": scene,
                       "bodies": n, "interactions_per_step": inter_per_step,
                       "parallelism": ("one GPU, no collective" if not use_dist else
                                       f"i-shard x{world}, {'RCCL' if args.backend == 'nccl' else args.backend + ' (rehearsal, host-staged)'} all-gather of positions"
                                       + (" + reduce-scatter of pair-symmetric partial sums" if sim.acc_all is not None else "")
                                       + " per step"
                                       + ("" if args.no_overlap else ", all-gather overlapped with the own-slab part")
                                       + (" (one rank: --force-collectives rehearsal)" if world == 1 else "")),
                       "kernel": kernel_label,
                       "kernel_variant": args.variant, "integrator": args.integrator},
            "roofline": {"bound": "valu_fp32" if prec == nb.FP32 else "valu_fp64", "achieved": round(ach, 3),
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
        line.update(proof)
        headline = (world == 1 and n == 1048576 and args.ic == "plummer" and prec == nb.FP64 and args.variant == 0
                    and not args.force_collectives)
        if headline and not args.no_other_configs:
            del sim                                    # (its 7 GB workspace)
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(nb)
        if multi_other:
            line["other_configs"] = multi_other
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    if use_dist:
        failed = torch.tensor([1 if verify_failed else 0], dtype=torch.int32,
                              device=device if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.broadcast(failed, src=0)            # every rank leaves with rank 0's verdict
        verify_failed = bool(int(failed[0]))
        dist.barrier()
        dist.destroy_process_group()
    if verify_failed:
        sys.exit("bench.py: verification FAILED (see `verify` in the line)" if rank == 0 else 1)


if __name__ == "__main__":
    main()
