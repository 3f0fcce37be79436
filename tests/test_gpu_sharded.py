"""Level-1 (device pointer) entry points on torch-owned memory: split steps and the
multi-slab layout, on ONE GPU (the 8-GPU run is the driver's; here the slabs of the other
"ranks" live on the same device and the exchange is a device copy)."""
import numpy as np
import pytest

import torch  # noqa: F401  (before nbody_mi355x: one shared HIP runtime)
import nbody_mi355x as nb
import oracle_lib as ol
from conftest import rel_pos_err
from nbody_mi355x.sharded import ShardedNBody

pytestmark = pytest.mark.gpu


class LoopbackGroup:
    """Stands in for torch.distributed on one device: the 'all-gather' copies each fake
    rank's own slab into every other fake rank's buffer."""

    def __init__(self, sims):
        self.sims = sims

    def exchange(self):
        for a in self.sims:
            src = a.pos[a.cur][a.rank]
            for b in self.sims:
                if b is not a:
                    b.pos[b.cur][a.rank].copy_(src)
        for s in self.sims:
            s._fresh = True


@pytest.mark.parametrize("world,n", [(2, 1000), (4, 4096), (3, 777)])
def test_multi_slab_layout_equals_oracle(world, n):
    p, m = nb.ic_plummer(n, 9)
    sims = [ShardedNBody(p, m, 1e-3, rank=r, world=world, device="cuda", use_workspace=False) for r in range(world)]
    grp = LoopbackGroup(sims)
    steps = 10
    for k in range(steps):
        for s in sims:
            s.step(snapshot=(k == steps - 1))
        grp.exchange()
    torch.cuda.synchronize()
    fin, out = ol.simulate(p, m, 1, steps, 1e-3, threads=8)
    state = np.concatenate([s.own_state() for s in sims])
    snap = np.concatenate([s.own_snapshot() for s in sims])
    assert rel_pos_err(state[:, :3], fin[:, :3]) <= 1e-13
    assert rel_pos_err(snap[:, :3], out[0, :, :3]) <= 1e-13


def test_split_step_equals_full_step():
    """acc_out/acc_in halves (own slab first, then the rest) == one full step, up to the
    changed summation order."""
    n, world, rank = 3000, 3, 1
    p, m = nb.ic_plummer(n, 11)
    full = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cuda", use_workspace=False)
    split = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cuda", use_workspace=False)
    full.step(snapshot=True)
    # drive the split path by hand (no process group on one device)
    S, P, r = split.S, world, rank
    cur, nxt = split.pos[0], split.pos[1]
    common = dict(pos_cur=cur, pos_next=nxt, mass=split.mass, vel=split.vel, S=S, world=P, rank=r,
                  count=split.count, dt=split.dt, G=split.G, eps2=0.0)
    split.backend.step(acc_in=None, acc_out=split.acc, snapshot=None, ranges=((r * S, (r + 1) * S), (0, 0)), **common)
    split.backend.step(acc_in=split.acc, acc_out=None, snapshot=split.snap,
                       ranges=((0, r * S), ((r + 1) * S, P * S)), **common)
    split.cur = 1
    torch.cuda.synchronize()
    a, b = full.own_snapshot(), split.own_snapshot()
    assert np.array_equal(a[:, :3], b[:, :3])  # r' = r + dt v does not depend on the forces
    num = np.linalg.norm(a[:, 6:9] - b[:, 6:9], axis=1)
    assert (num / np.linalg.norm(a[:, 6:9], axis=1)).max() <= 1e-13


def test_pack_unpack_roundtrip():
    import ctypes as C
    n, S = 1000, 1024
    aos = torch.randn(n, 6, dtype=torch.float64, device="cuda")
    pos = torch.zeros(3, S, dtype=torch.float64, device="cuda")
    vel = torch.zeros(3, S, dtype=torch.float64, device="cuda")
    back = torch.zeros_like(aos)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb.check(nb.lib().nbody_mi355x_pack_f64(aos.data_ptr(), n, pos.data_ptr(), vel.data_ptr(), S, st))
    nb.check(nb.lib().nbody_mi355x_unpack_f64(pos.data_ptr(), vel.data_ptr(), n, S, back.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(aos, back)
    assert torch.equal(pos[:, :n], aos[:, :3].T) and torch.all(pos[:, n:] == 0)


@pytest.mark.parametrize("world,n,variant", [(2, 20000, 0), (4, 16384, 0), (3, 30000, 0), (8, 65536, 0),
                                             (2, 20000, nb.VARIANT_SYMW4), (3, 30000, nb.VARIANT_SYMW2),
                                             (4, 65536, nb.VARIANT_SYMW8), (5, 1500, nb.VARIANT_SYMW1)])
def test_symmetric_multi_slab_equals_oracle(world, n, variant):
    """Pair-symmetric kernels over several slabs: every fake rank computes partial sums
    for ALL slots, the loop-back 'reduce-scatter' adds them, each rank integrates its slab."""
    p, m = nb.ic_plummer(n, 13)
    from nbody_mi355x.sharded import HipBackend
    sims = [ShardedNBody(p, m, 1e-3, rank=r, world=world, device="cuda", backend=HipBackend(variant=variant))
            for r in range(world)]
    assert all(s.workspace is not None and s.acc_all is not None for s in sims)
    grp = LoopbackGroup(sims)
    steps = 4
    for k in range(steps):
        for s in sims:
            common = dict(pos_cur=s.pos[s.cur], pos_next=s.pos[s.cur ^ 1], mass=s.mass, vel=s.vel, S=s.S,
                          world=world, rank=s.rank, count=s.count, dt=s.dt, G=s.G, eps2=s.eps2)
            s._common = common
            s._sym_forces(s.pos[s.cur], common)          # _fresh is True: no collective inside
        total = sum(s.acc_all for s in sims)             # what reduce_scatter_tensor would deliver
        for s in sims:
            s.acc.copy_(total[s.rank])
            s._integrate(s._common, s.snap if k == steps - 1 else None)
            s.cur ^= 1
        grp.exchange()
    torch.cuda.synchronize()
    fin, out = ol.simulate(p, m, 1, steps, 1e-3, threads=16)
    state = np.concatenate([s.own_state() for s in sims])
    snap = np.concatenate([s.own_snapshot() for s in sims])
    assert rel_pos_err(state[:, :3], fin[:, :3]) <= 1e-13
    num = np.linalg.norm(snap[:, 6:9] - out[0, :, 6:9], axis=1)
    assert (num / np.linalg.norm(out[0, :, 6:9], axis=1)).max() <= 1e-12


def test_config4_one_gpu_vs_eight_ranks_agree():
    """BASELINE config 4 (Plummer N = 1 048 576 over 8 ranks), SURVEY 8d: 'parity by ... 1-GPU vs
    P-GPU position agreement <= 1e-12'.  The eight ranks' kernels run one after the other on the
    one GPU of the test box (same slab layout and launches as under bench.py --gpus 8, the two
    collectives done by hand); the single-GPU run is the handle API.  Two small steps: the second
    one works on positions that went through the exchange."""
    n, world, steps = 1048576, 8, 2
    p, m = nb.ic_plummer(n, 42)
    with nb.Simulation(n, 1e-3) as one:
        one.upload(p, m)
        ref = one.run(steps, snapshot=True)
    sims = [ShardedNBody(p, m, 1e-3, rank=r, world=world, device="cuda") for r in range(world)]
    assert all(s.workspace is not None and s.acc_all is not None for s in sims)
    grp = LoopbackGroup(sims)
    for k in range(steps):
        for s in sims:
            s._common = dict(pos_cur=s.pos[s.cur], pos_next=s.pos[s.cur ^ 1], mass=s.mass, vel=s.vel, S=s.S,
                             world=world, rank=s.rank, count=s.count, dt=s.dt, G=s.G, eps2=s.eps2)
            s._sym_forces(s.pos[s.cur], s._common)
        total = sum(s.acc_all for s in sims)
        for s in sims:
            s.acc.copy_(total[s.rank])
            s._integrate(s._common, s.snap if k == steps - 1 else None)
            s.cur ^= 1
        del total
        grp.exchange()
    torch.cuda.synchronize()
    snap = np.concatenate([s.own_snapshot() for s in sims])
    assert rel_pos_err(snap[:, :3], ref[:, :3]) <= 1e-12
    num = np.linalg.norm(snap[:, 6:9] - ref[:, 6:9], axis=1)
    assert (num / np.linalg.norm(ref[:, 6:9], axis=1)).max() <= 1e-11   # sums of 1e6 terms in another order
    num = np.linalg.norm(snap[:, 3:6] - ref[:, 3:6], axis=1)
    assert (num / np.linalg.norm(ref[:, 3:6], axis=1)).max() <= 1e-12


@pytest.mark.parametrize("variant", [nb.VARIANT_SYM4, nb.VARIANT_SYM8, nb.VARIANT_SYMW2, nb.VARIANT_SYMW8])
def test_symmetric_phases_equal_whole(variant):
    """phase 1 (diagonal tiles, own slab only) + phase 2 (the rest) == phase 0."""
    n, world, rank = 40000, 4, 2
    p, m = nb.ic_plummer(n, 21)
    from nbody_mi355x.sharded import HipBackend
    s = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cuda", backend=HipBackend(variant=variant))
    common = dict(pos_cur=s.pos[0], pos_next=s.pos[1], mass=s.mass, vel=s.vel, S=s.S, world=world, rank=rank,
                  count=s.count, dt=s.dt, G=s.G, eps2=0.0)
    kw = dict(acc_in=None, snapshot=None, ranges=((0, 0), (0, 0)), workspace=s.workspace, variant=variant, **common)
    s.backend.step(acc_out=s.acc_all, phase=0, **kw)
    whole = s.acc_all.clone()
    s.acc_all.zero_()
    s.backend.step(acc_out=None, phase=1, **kw)
    s.backend.step(acc_out=s.acc_all, phase=2, **kw)
    torch.cuda.synchronize()
    scale = whole.abs().max()
    assert (whole - s.acc_all).abs().max() <= 1e-13 * scale
    assert whole.abs().max() > 0


def _rank_main(rank, world, port, n, steps, q, integrator=0, scene_kind=None):
    """A real rank of the sharded driver (own process, own process group); the two ranks
    share the one GPU of the test box and talk over gloo, staged through the host."""
    import os
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(repo / "n-body-sim_amd"))
    sys.path.insert(0, str(repo / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_mi355x as nb
        from nbody_mi355x.sharded import ShardedNBody
        if scene_kind is None:
            p, m = nb.ic_plummer(n, 31)
        else:                              # made on the device by every rank, no host arrays
            p, m = nb.make_scene(scene_kind, n, seed=31), None
        sim = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cuda:0", overlap=True, integrator=integrator)
        for k in range(steps):
            sim.step(snapshot=(k == steps - 1))
        torch.cuda.synchronize()
        q.put((rank, sim.own_state(), sim.own_snapshot(), sim.workspace is not None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,expect_sym,world", [(1500, False, 2), (20000, True, 2), (70000, True, 4)])
def test_two_real_ranks_end_to_end(n, expect_sym, world):
    """ShardedNBody.step() as bench.py drives it -- collectives, buffer rotation, overlap
    split (one-sided) and partial-sum exchange (pair-symmetric) -- against the oracle."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n, steps, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=100) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(r[3] == expect_sym for r in res)
    p, m = nb.ic_plummer(n, 31)
    fin, out = ol.simulate(p, m, 1, steps, 1e-3, threads=16)
    state = np.concatenate([r[1] for r in res])
    snap = np.concatenate([r[2] for r in res])
    assert rel_pos_err(state[:, :3], fin[:, :3]) <= 1e-13
    assert rel_pos_err(snap[:, :3], out[0, :, :3]) <= 1e-13
    num = np.linalg.norm(snap[:, 6:9] - out[0, :, 6:9], axis=1)
    assert (num / np.linalg.norm(out[0, :, 6:9], axis=1)).max() <= 1e-12


@pytest.mark.parametrize("integrator", ["KICK_DRIFT", "KDK"])
@pytest.mark.parametrize("n,expect_sym,world", [(1500, False, 2), (20000, True, 2), (20000, True, 3)])
def test_real_ranks_other_integrators(n, expect_sym, world, integrator):
    """The sharded driver with the symplectic Euler step and the kick-drift-kick leapfrog (a
    drift of the own slab, the exchange, one force evaluation whose epilogue is the second
    kick): real ranks sharing the box's GPU, one-sided split steps and pair-symmetric partial
    sums, against the C handle's own sequencing of the same integrator on one device (which
    fuses {kick, kick, drift} between small steps: the same operations in another grouping)."""
    import socket
    import torch.multiprocessing as mp
    which = getattr(nb, integrator)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n, steps, q, which)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=100) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(r[3] == expect_sym for r in res)
    p, m = nb.ic_plummer(n, 31)
    with nb.Simulation(n, 1e-3, integrator=which) as sim:
        sim.upload(p, m)
        out = sim.run(steps, snapshot=True)
        fin = sim.download()
    state = np.concatenate([r[1] for r in res])
    snap = np.concatenate([r[2] for r in res])
    assert rel_pos_err(state[:, :3], fin[:, :3]) <= 1e-13
    assert rel_pos_err(snap[:, :3], out[:, :3]) <= 1e-13
    assert np.abs(snap[:, 3:6] - out[:, 3:6]).max() <= 1e-12 * np.abs(out[:, 3:6]).max()
    num = np.linalg.norm(snap[:, 6:9] - out[:, 6:9], axis=1)
    assert (num / np.linalg.norm(out[:, 6:9], axis=1)).max() <= 1e-11
    # ... and it is not the Euler step: the integrators differ at O(dt^2) per step
    euler, _ = ol.simulate(p, m, 1, steps, 1e-3, threads=16)
    assert rel_pos_err(state[:, :3], euler[:, :3]) > 1e-9


@pytest.mark.parametrize("kind,n,world", [(0, 20000, 2), (0, 1500, 3), (2, 20000, 2)])
def test_real_ranks_make_their_scene_on_the_device(kind, n, world):
    """ShardedNBody given a Scene record instead of arrays: every rank fills the slabs it
    holds on its own device (one random stream per body), the Plummer sphere's centre of mass
    goes through one all-reduce of 6 doubles.  Against the C handle's generate() + run()."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n, steps, q, 0, kind)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=100) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    with nb.Simulation(n, 1e-3) as sim:
        sim.generate(nb.make_scene(kind, n, seed=31))
        first = sim.download()
        out = sim.run(steps, snapshot=True)
        fin = sim.download()
    state = np.concatenate([r[1] for r in res])
    snap = np.concatenate([r[2] for r in res])
    # (the centre of mass is summed in another order: 1e-16 of the sphere's size)
    assert np.abs(state[:, :3] - fin[:, :3]).max() <= 1e-12 * np.abs(fin[:, :3]).max()
    assert np.abs(snap[:, 3:6] - out[:, 3:6]).max() <= 1e-11 * np.abs(out[:, 3:6]).max()
    assert np.abs(fin[:, :3] - first[:, :3]).max() > 0


def _rccl_single_rank_main(port, n, steps, q):
    """One rank, backend nccl (= RCCL), forced through the multi-rank code path."""
    import os
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(repo / "n-body-sim_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import nbody_mi355x as nb
        from nbody_mi355x.sharded import ShardedNBody
        p, m = nb.ic_plummer(n, 31)
        sim = ShardedNBody(p, m, 1e-3, rank=0, world=1, device="cuda:0", overlap=True, force_collectives=True)
        sim.warm_collectives()   # as bench.py does: must not change the state
        for k in range(steps):
            sim.step(snapshot=(k == steps - 1))
        torch.cuda.synchronize()
        pos = sim.all_positions()
        q.put((sim.own_state(), sim.own_snapshot(), sim.workspace is not None, pos))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,expect_sym", [(1500, False), (20000, True)])
def test_rccl_collectives_single_rank(n, expect_sym):
    """The collectives bench.py issues on an 8-GPU node -- all_gather_into_tensor of the own slab
    (asynchronous, overlapped with the own-slab part, work.wait() before the rest) and
    reduce_scatter_tensor of the pair-symmetric partial sums -- through RCCL itself, with the
    one rank a one-GPU box allows: same calls, same buffers and stream ordering, result
    against the oracle."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_single_rank_main, args=(port, n, steps, q))
    pr.start()
    state, snap, used_sym, pos = q.get(timeout=150)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    assert used_sym == expect_sym
    p, m = nb.ic_plummer(n, 31)
    fin, out = ol.simulate(p, m, 1, steps, 1e-3, threads=16)
    assert rel_pos_err(state[:, :3], fin[:, :3]) <= 1e-13
    assert rel_pos_err(pos, fin[:, :3]) <= 1e-13
    assert rel_pos_err(snap[:, :3], out[0, :, :3]) <= 1e-13
    num = np.linalg.norm(snap[:, 6:9] - out[0, :, 6:9], axis=1)
    assert (num / np.linalg.norm(out[0, :, 6:9], axis=1)).max() <= 1e-12


@pytest.mark.parametrize("world,n,drop", [(1, 20000, 1), (1, 20000, 7), (2, 30000, 5), (1, 5000, 3)])
def test_symmetric_distance_batching(world, n, drop):
    """A workspace too small for every distance at once: the symmetric kernels work through
    the distances in batches that reuse it.  Same answer as with the full workspace."""
    p, m = nb.ic_plummer(n, 17)
    sims = [ShardedNBody(p, m, 1e-3, rank=r, world=world, device="cuda") for r in range(world)]
    for s in sims:
        assert s.workspace is not None
        row = 3 * s.S
        rows = s.workspace.numel() // row - drop         # the preferred size holds every distance
        s.workspace = torch.empty(rows * row, dtype=torch.float64, device="cuda")
    for s in sims:
        common = dict(pos_cur=s.pos[0], pos_next=s.pos[1], mass=s.mass, vel=s.vel, S=s.S, world=world,
                      rank=s.rank, count=s.count, dt=s.dt, G=s.G, eps2=0.0)
        if world == 1:
            s.backend.step(acc_in=None, acc_out=None, snapshot=s.snap, ranges=((0, n), (0, 0)),
                           workspace=s.workspace, **common)
        else:
            s._common = common
            s._sym_forces(s.pos[0], common)
    if world > 1:
        total = sum(s.acc_all for s in sims)
        for s in sims:
            s.acc.copy_(total[s.rank])
            s._integrate(s._common, s.snap)
    torch.cuda.synchronize()
    _, out = ol.simulate(p, m, 1, 1, 1e-3, threads=16)
    snap = np.concatenate([s.own_snapshot() for s in sims])
    num = np.linalg.norm(snap[:, 6:9] - out[0, :, 6:9], axis=1)
    assert (num / np.linalg.norm(out[0, :, 6:9], axis=1)).max() <= 1e-12
    assert rel_pos_err(snap[:, :3], out[0, :, :3]) <= 1e-15


def test_symmetric_workspace_too_small_is_an_error():
    p, m = nb.ic_plummer(20000, 17)
    s = ShardedNBody(p, m, 1e-3, device="cuda")
    s.workspace = torch.empty(3 * s.S * 2, dtype=torch.float64, device="cuda")
    with pytest.raises(nb.NBodyError) as e:
        s.backend.step(pos_cur=s.pos[0], pos_next=s.pos[1], mass=s.mass, vel=s.vel, S=s.S, world=1, rank=0,
                       count=s.count, dt=s.dt, G=s.G, eps2=0.0, acc_in=None, acc_out=None, snapshot=None,
                       ranges=((0, 20000), (0, 0)), workspace=s.workspace, variant=nb.VARIANT_SYM)
    assert e.value.code == nb.ERR_INVALID and "workspace" in str(e.value)


def test_bench_self_launch_two_ranks_share_the_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU)
    starts the two rank processes itself and relays rank 0's line -- here as a rehearsal, both
    ranks on the one GPU of the box, gloo staged through the host instead of RCCL."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--share-devices", "--backend", "gloo",
                        "--bodies", "131072", "--steps", "3", "--warmup", "1", "--config5-bodies", "262144"],
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["bodies"] == 131072 and line["value"] > 0
    assert "cpu_baseline" not in line                                      # N = 1 only
    # north_star's other multi-GPU figure (config 5: the uniform cube, fp64 and FP32) is run by the SAME ranks
    # beside the headline -- at N = 4M beside the N = 1M headline; a rehearsal size here
    oc = line["other_configs"]
    assert set(oc) == {"config5_uniform_cube_n262144_fp64_2gpu", "config5_uniform_cube_n262144_fp32_2gpu"}
    for e in oc.values():
        assert e["n_gpus"] == 2 and e["scaling"] == "strong" and e["value"] > 0 and 0 < e["frac"] < 1
    assert oc["config5_uniform_cube_n262144_fp32_2gpu"]["value"] > oc["config5_uniform_cube_n262144_fp64_2gpu"]["value"]
    # what makes a multi-rank line self-proving: the communicator's own view ...
    c = line["collectives"]
    assert c["world_size_seen"] == 2 and c["backend"] == "gloo" and not c["is_rccl"]
    assert [d["rank"] for d in c["devices"]] == [0, 1] and len({d["pid"] for d in c["devices"]}) == 2
    assert c["distinct_devices"] == 1, "--share-devices: both ranks on the one GPU, and the line says so"
    assert "MI355X" in c["devices"][0]["name"] or "AMD" in c["devices"][0]["name"]
    # ... every rank's own clock, the two collectives on their own ...
    assert len(line["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert max(line["per_rank_ms_per_step"]) == pytest.approx(line["ms_per_step"], rel=1e-3)
    cm = line["collective_ms"]
    assert cm["all_gather"] > 0 and cm["reduce_scatter"] > 0 and cm["all_gather_bytes_per_rank"] == 3 * 8 * 65536
    # ... and the verification leg, on by default with more than one rank
    v = line["verify"]
    assert v["ok"] is True and v["steps_before"] == 4
    assert v["rows_vs_oracle"]["ok"] and v["rows_vs_oracle"]["max_err_over_max_acc"] <= 1e-12
    assert v["momentum"]["ok"] and v["positions_vs_one_gpu"]["ok"] and v["positions_vs_one_gpu"]["steps"] == 5
    assert "RCCL" not in line["config"]["parallelism"] and "gloo" in line["config"]["parallelism"]


def _bench(*flags, timeout=400):
    import json
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / "bench.py"), *flags], capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, (json.loads(lines[0]) if len(lines) == 1 else None)


def test_bench_one_rank_through_rccl_verifies_itself(tmp_path):
    """`--gpus 1 --force-collectives --verify`: the multi-rank code path through REAL RCCL (a
    communicator of one rank is all a one-GPU box allows) -- the line reports what the
    communicator saw and the verification passes; a one-GPU run of the same command writes the
    checksum file the RCCL run is then also compared with."""
    common = ["--bodies", "65536", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    ck = tmp_path / "pos.npy"
    r, one = _bench(*common, "--verify", "--write-checksum", str(ck))
    assert r.returncode == 0 and one is not None, r.stderr[-2000:]
    assert one["config"]["parallelism"] == "one GPU, no collective" and "collectives" not in one
    assert one["verify"]["ok"] and one["verify"]["checksum_written"]["steps"] == 5 and ck.exists()
    assert "positions_vs_one_gpu" not in one["verify"]
    r, line = _bench(*common, "--force-collectives", "--verify", "--reference-checksum", str(ck))
    assert r.returncode == 0 and line is not None, r.stderr[-2000:]
    c = line["collectives"]
    assert c["is_rccl"] and c["backend"] == "nccl" and c["world_size_seen"] == 1 and c["rccl_version"][0].isdigit()
    assert line["collective_ms"]["timer"] == "hip_events_on_launch_stream"
    assert line["collective_ms"]["all_gather"] > 0 and line["collective_ms"]["reduce_scatter"] > 0
    v = line["verify"]
    assert v["ok"] and v["rows_vs_oracle"]["ok"] and v["momentum"]["ok"]
    assert v["positions_vs_one_gpu"]["ok"] and v["positions_vs_reference_file"]["ok"]
    assert "RCCL" in line["config"]["parallelism"] and "--force-collectives rehearsal" in line["config"]["parallelism"]


def test_bench_verification_failure_is_a_non_zero_exit(tmp_path):
    """A reference file that does not match fails the run: non-zero exit, the line still printed
    with verify.ok false (a wrong multi-rank result must not pass as a number)."""
    import numpy as np
    bad = tmp_path / "bad.npy"
    np.save(bad, np.zeros((8192, 3)))
    r, line = _bench("--bodies", "8192", "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--verify",
                     "--reference-checksum", str(bad))
    assert r.returncode != 0 and "verification FAILED" in r.stderr
    assert line is not None and line["verify"]["ok"] is False
    assert line["verify"]["rows_vs_oracle"]["ok"] and not line["verify"]["positions_vs_reference_file"]["ok"]
