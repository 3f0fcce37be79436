"""nbody_mi355x_step_f64_static -- the one entry point that allocates nothing (its workspace is
a device array inside the library, include/nbody_mi355x.h) -- against nbody_mi355x_step_f64
on the same inputs, through the C ABI.  Same kernels, same launch plan, same number of
workspace rows: the outputs must agree bit for bit; and against the oracle to the FP64 bar.
Two calls in a row, the second on a smaller system (the array keeps the first call's
contents: nothing of them may leak into the second result).  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

import nbody_mi355x as nb
import oracle_lib as ol
from conftest import rel_pos_err

pytestmark = pytest.mark.gpu


def _slab(p, m, S, dev):
    import torch
    n = p.shape[0]
    pos = np.zeros((3, S))
    pos[:, :n] = p[:, :3].T
    vel = np.zeros((3, S))
    vel[:, :n] = p[:, 3:].T
    mass = np.zeros(S)
    mass[:n] = m
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(pos), t(vel), t(mass)


def _one_step(entry, p, m, dt, *, workspace, precision=nb.FP64):
    """One small step with snapshot through `entry` ("static" or "workspace")."""
    import torch
    dev = torch.device("cuda", 0)
    n = p.shape[0]
    from nbody_mi355x.sharded import slab_slots
    S = slab_slots(n, 0)   # csrc/nbody_internal.h and its mirror
    pos, vel, mass = _slab(p, m, S, dev)
    nxt = torch.zeros_like(pos)
    snap = torch.full((n, 9), np.nan, dtype=torch.float64, device=dev)
    a = nb.StepArgs()
    a.pos_cur, a.pos_next, a.mass, a.vel = pos.data_ptr(), nxt.data_ptr(), mass.data_ptr(), vel.data_ptr()
    a.snapshot = snap.data_ptr()
    a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, 1, 0, n
    a.j0_begin, a.j0_end = 0, n
    a.dt, a.G, a.eps2 = dt, nb.G_REFERENCE, 0.0
    a.precision, a.variant = precision, nb.VARIANT_SYM
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L = nb.lib()
    if entry == "static":
        a.workspace, a.workspace_bytes = None, 0
        nb.check(L.nbody_mi355x_step_f64_static(C.byref(a), stream))
    else:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * 8
        nb.check(L.nbody_mi355x_step_f64(C.byref(a), stream))
    torch.cuda.synchronize(dev)
    return nxt.cpu().numpy(), vel.cpu().numpy(), snap.cpu().numpy()


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_static_entry_point_matches_the_allocating_one(precision):
    import torch
    dev = torch.device("cuda", 0)
    prec = nb.FP64 if precision == "fp64" else nb.FP32
    # the allocating entry point gets a workspace of exactly the static array's size: the same
    # number of rows -> the same batching of the distances -> the same order of additions
    ws = torch.empty(nb.STATIC_SCRATCH_BYTES // 8, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(3)
    for n in (30000, 9000):                       # the second call is the smaller one
        p, m = nb.ic_plummer(n, 500 + n)
        m = m * rng.uniform(0.5, 2.0, n)
        ws.fill_(float("nan"))                    # (what the other entry point starts from must not matter either)
        ref = _one_step("workspace", p, m, 1e-3, workspace=ws, precision=prec)
        got = _one_step("static", p, m, 1e-3, workspace=None, precision=prec)
        for r, g in zip(ref, got):
            assert np.array_equal(r, g)
        _, orc = ol.simulate(p, m, 1, 1, 1e-3, threads=16)
        snap = got[2]
        assert np.all(np.isfinite(snap))
        tol_r, tol_a = (1e-14, 1e-12) if precision == "fp64" else (1e-9, 1e-4)
        assert rel_pos_err(snap[:, :3], orc[0, :, :3]) <= tol_r
        scale = np.linalg.norm(orc[0, :, 6:9], axis=1).max()
        assert np.abs(snap[:, 6:9] - orc[0, :, 6:9]).max() <= tol_a * scale


def _call(entry, a, ws, stream):
    L = nb.lib()
    if entry == "static":
        a.workspace, a.workspace_bytes = None, 0
        nb.check(L.nbody_mi355x_step_f64_static(C.byref(a), stream))
    else:
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
        nb.check(L.nbody_mi355x_step_f64(C.byref(a), stream))


def test_static_entry_point_workgroup_tiles_in_several_batches():
    """The other kernel family through the shared array: workgroup tiles with 256-body
    blocks (variant SYM1) on a system whose block distances do not fit the 128 MiB at once -- the
    distances are walked in batches that reuse the C rows, the accumulator lives in the array
    too.  Bit for bit against the allocating entry point with a workspace of the same size."""
    import torch
    dev = torch.device("cuda", 0)
    n = 100000
    p, m = nb.ic_plummer(n, 9)
    S = nb.round_up(n, 2048)
    ws = torch.full((nb.STATIC_SCRATCH_BYTES // 8,), float("nan"), dtype=torch.float64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for entry in ("workspace", "static"):
        pos, vel, mass = _slab(p, m, S, dev)
        nxt = torch.zeros_like(pos)
        snap = torch.full((n, 9), np.nan, dtype=torch.float64, device=dev)
        a = nb.StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel, a.snapshot = pos.data_ptr(), nxt.data_ptr(), mass.data_ptr(), vel.data_ptr(), snap.data_ptr()
        a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, 1, 0, n
        a.j0_begin, a.j0_end = 0, n
        a.dt, a.G, a.eps2, a.precision, a.variant = 1e-3, nb.G_REFERENCE, 0.0, nb.FP64, nb.VARIANT_SYM1
        if entry == "workspace":
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
            launches = C.c_int32(0)
            nb.check(nb.lib().nbody_mi355x_step_tile_launches(C.byref(a), C.byref(launches)))
            assert launches.value >= 3, "the case is meant to need several distance batches"
        _call(entry, a, ws, stream)
        torch.cuda.synchronize(dev)
        out[entry] = (nxt.cpu().numpy(), vel.cpu().numpy(), snap.cpu().numpy())
    for r, g in zip(out["workspace"], out["static"]):
        assert np.array_equal(r, g)
    rows = np.random.default_rng(0).choice(n, 512, replace=False)
    acc = ol.accel_rows(p, m, rows, threads=16)
    got = out["static"][2][rows, 6:9]
    assert (np.linalg.norm(got - acc, axis=1) / np.linalg.norm(acc, axis=1)).max() <= 1e-12


@pytest.mark.parametrize("variant", [nb.VARIANT_SYM, nb.VARIANT_SYM2])
def test_static_entry_point_two_slabs_phase_1_then_2(variant):
    """Several slabs through the shared array: phase 1 leaves the diagonal tiles' sums
    in the array's R rows for the phase 2 call that follows (with acc_out: this rank's partial
    sums for every slot).  Both "ranks" of a two-slab layout in turn, on one GPU, bit for bit
    against the allocating entry point; the two ranks' partial sums added = the oracle's sums."""
    import torch
    dev = torch.device("cuda", 0)
    n, P = 24000, 2
    p, m = nb.ic_plummer(n, 77)
    half = n // P
    S = nb.round_up(half, 1024)
    pos = np.zeros((P, 3, S)); mass = np.zeros((P, S))
    for r in range(P):
        pos[r, :, :half] = p[r * half:(r + 1) * half, :3].T
        mass[r, :half] = m[r * half:(r + 1) * half]
    pos_d, mass_d = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev)
    nxt = torch.zeros_like(pos_d)
    vel = torch.zeros((3, S), dtype=torch.float64, device=dev)
    ws = torch.full((nb.STATIC_SCRATCH_BYTES // 8,), float("nan"), dtype=torch.float64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    sums = {}
    for entry in ("workspace", "static"):
        total = torch.zeros((P, 3, S), dtype=torch.float64, device=dev)
        for r in range(P):
            acc_all = torch.full((P, 3, S), float("nan"), dtype=torch.float64, device=dev)
            for phase in (1, 2):
                a = nb.StepArgs()
                a.pos_cur, a.pos_next, a.mass, a.vel = pos_d.data_ptr(), nxt.data_ptr(), mass_d.data_ptr(), vel.data_ptr()
                a.acc_out = acc_all.data_ptr() if phase == 2 else None
                a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, P, r, half
                a.dt, a.G, a.eps2, a.precision, a.variant, a.phase = 1e-3, nb.G_REFERENCE, 0.0, nb.FP64, variant, phase
                _call(entry, a, ws, stream)
            torch.cuda.synchronize(dev)
            sums[(entry, r)] = acc_all.cpu().numpy()
            total += acc_all
        sums[entry] = total.cpu().numpy()
    for r in range(P):
        assert np.array_equal(sums[("workspace", r)], sums[("static", r)])
    got = np.concatenate([sums["static"][r, :, :half].T for r in range(P)]) * nb.G_REFERENCE
    acc = ol.accel_rows(p, m, np.arange(n), threads=16)
    assert (np.linalg.norm(got - acc, axis=1) / np.linalg.norm(acc, axis=1)).max() <= 1e-12


def test_static_entry_point_under_hip_graph_capture():
    """What the entry point is for: a caller under HIP graph capture, where hipMalloc is illegal.
    Two small steps (ping-pong buffers) are captured into one graph through torch's stream
    capture and replayed three times; the state must equal six direct calls, bit for bit.
    (The first, un-captured call resolves the array's address on the device.)"""
    import torch
    dev = torch.device("cuda", 0)
    n = 20000
    p, m = nb.ic_plummer(n, 21)
    S = nb.round_up(n, 1024)

    def fresh():
        pos, vel, mass = _slab(p, m, S, dev)
        return [pos, pos.clone()], vel, mass

    def args(bufs, vel, mass, cur):
        a = nb.StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel = bufs[cur].data_ptr(), bufs[cur ^ 1].data_ptr(), mass.data_ptr(), vel.data_ptr()
        a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, 1, 0, n
        a.j0_begin, a.j0_end = 0, n
        a.dt, a.G, a.eps2, a.precision, a.variant = 1e-3, nb.G_REFERENCE, 0.0, nb.FP64, nb.VARIANT_SYM
        return a

    L = nb.lib()
    # direct: six steps on the current stream
    bufs, vel, mass = fresh()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for k in range(6):
        nb.check(L.nbody_mi355x_step_f64_static(C.byref(args(bufs, vel, mass, k & 1)), stream))
    torch.cuda.synchronize(dev)
    want = (bufs[0].cpu().numpy(), vel.cpu().numpy())
    # captured: two steps per graph, three replays
    bufs, vel, mass = fresh()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for k in range(2):
            nb.check(L.nbody_mi355x_step_f64_static(C.byref(args(bufs, vel, mass, k & 1)), cap))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    assert np.array_equal(bufs[0].cpu().numpy(), want[0]) and np.array_equal(vel.cpu().numpy(), want[1])
    assert not np.array_equal(want[0][:, :n], p[:, :3].T)      # the bodies did move


def test_static_entry_point_rejects_what_does_not_fit():
    import torch
    dev = torch.device("cuda", 0)
    L = nb.lib()
    a = nb.StepArgs()
    S = 1 << 20                                   # minimum workspace far beyond 128 MiB
    buf = torch.zeros(8, dtype=torch.float64, device=dev)   # never dereferenced: rejected before any launch
    a.pos_cur, a.pos_next, a.mass, a.vel = buf.data_ptr(), buf.data_ptr() + 8, buf.data_ptr(), buf.data_ptr()
    a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, 1, 0, S
    a.j0_end = S
    a.dt, a.G = 1e-3, nb.G_REFERENCE
    assert L.nbody_mi355x_step_f64_static(C.byref(a), None) == nb.ERR_INVALID
    assert "static" in nb.last_error() and "bytes" in nb.last_error()
    a.shard_slots = a.own_count = a.j0_end = 4096
    a.variant = 2                                 # a one-sided kernel: nothing to do with the scratch
    assert L.nbody_mi355x_step_f64_static(C.byref(a), None) == nb.ERR_INVALID
    assert "one-sided" in nb.last_error()
