"""The N>1 host logic (partition, slab layout, in-place all-gather, split step with
overlap) on CPU: world_size 2 and 3 over gloo, with the ORACLE standing in for the kernel.
Each rank plays one reference worker thread (particle_simulation.h:250-253); the result
must be bit-identical to the single-process oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "n-body-sim_amd"))


class OracleBackend:
    """Checker backend: reassembles Particle rows from the slab layout and runs the
    oracle's per-thread small step.  A split step (acc_out, then acc_in) is emulated by
    doing all the work in the second half."""

    def __init__(self, n, world):
        from nbody_mi355x.sharded import shard_counts
        self.counts, self.begins = shard_counts(n, world)
        self.n = n

    def step(self, *, pos_cur, pos_next, mass, vel, acc_in, acc_out, snapshot, S, world, rank, count,
             ranges, dt, G, eps2):
        import oracle_lib as ol
        if acc_out is not None:
            acc_out.zero_()
            return
        cur = np.zeros((self.n, 6))
        m = np.zeros(self.n)
        for r in range(world):
            b, c = self.begins[r], self.counts[r]
            cur[b:b + c, :3] = pos_cur[r, :, :c].T.numpy()
            m[b:b + c] = mass[r, :c].numpy()
        b, c = self.begins[rank], count
        cur[b:b + c, 3:] = vel[:, :c].T.numpy()
        nxt, out = ol.small_step(cur, m, dt, b, c, want_out=snapshot is not None)
        pos_next[rank, :, :c] = torch.from_numpy(nxt[b:b + c, :3].T.copy())
        vel[:, :c] = torch.from_numpy(nxt[b:b + c, 3:].T.copy())
        if snapshot is not None:
            snapshot[:c] = torch.from_numpy(out[b:b + c])


def _worker(rank, world, port, n, steps, overlap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_mi355x as nb
        from nbody_mi355x.sharded import ShardedNBody
        p, m = nb.ic_plummer(n, 42)
        sim = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cpu",
                           backend=OracleBackend(n, world), overlap=overlap)
        for k in range(steps):
            sim.step(snapshot=(k == steps - 1))
        q.put((rank, sim.own_state(), sim.own_snapshot(), sim.all_positions()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,overlap", [(2, 300, True), (2, 301, False), (3, 257, True)])
def test_sharded_driver_matches_single_process_oracle(world, n, overlap):
    import nbody_mi355x as nb
    import oracle_lib as ol
    steps = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, overlap, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p, m = nb.ic_plummer(n, 42)
    fin, out = ol.simulate(p, m, 1, steps, 1e-3, threads=1)
    state = np.concatenate([r[1] for r in res], axis=0)
    snap = np.concatenate([r[2] for r in res], axis=0)
    assert np.array_equal(state, fin)
    assert np.array_equal(snap, out[0])
    for r in res:
        assert np.array_equal(r[3], fin[:, :3])  # every rank sees every position


class LeapfrogBackend:
    """Checker backend for the driver's integrator sequencing: direct sums in numpy (the oracle
    is the reference's Euler step and has no other integrator).  Split steps add their j range
    to acc_out / start from acc_in, as the HIP step does."""

    def __init__(self, n, world):
        from nbody_mi355x.sharded import shard_counts
        self.counts, self.begins = shard_counts(n, world)
        self.n = n

    @staticmethod
    def _sums(pos_all, mass_all, xi, ranges):
        a = np.zeros_like(xi)
        for lo, hi in ranges:
            if hi <= lo:
                continue
            d = pos_all[lo:hi][None, :, :] - xi[:, None, :]
            r2 = (d * d).sum(-1)
            w = np.where(r2 > 0, mass_all[lo:hi][None, :] / np.where(r2 > 0, r2, 1.0) ** 1.5, 0.0)
            a += (w[..., None] * d).sum(1)
        return a

    def step(self, *, pos_cur, pos_next, mass, vel, acc_in, acc_out, snapshot, S, world, rank, count,
             ranges, dt, G, eps2, integrator=0, kdk_acc=None):
        import nbody_mi355x as nb
        flat = pos_cur.numpy().transpose(0, 2, 1).reshape(world * S, 3)      # slot-major positions
        xi = flat[rank * S:rank * S + count]
        a = self._sums(flat, mass.numpy().reshape(-1), xi, ranges)
        if acc_in is not None:
            a += acc_in.numpy()[:, :count].T
        if acc_out is not None:
            acc_out.zero_()
            acc_out[:, :count] = torch.from_numpy(a.T.copy())
            return
        v = vel.numpy()[:, :count].T.copy()
        if integrator == nb.KDK:                       # second kick; nothing moves
            v += 0.5 * dt * a * G
            kdk_acc[:, :count] = torch.from_numpy(a.T.copy())
            x = xi
        elif integrator == nb.KICK_DRIFT:
            v += dt * a * G
            x = xi + dt * v
        else:
            x = xi + dt * v
            v += dt * a * G
        if integrator != nb.KDK:
            pos_next[rank, :, :count] = torch.from_numpy(x.T.copy())
        vel[:, :count] = torch.from_numpy(v.T.copy())
        if snapshot is not None:
            snapshot[:count] = torch.from_numpy(np.concatenate([xi, v, a * G], axis=1))

    def kdk_drift(self, *, pos_cur, pos_next, vel, kdk_acc, S, world, rank, count, dt, G):
        v = vel.numpy()[:, :count] + 0.5 * dt * G * kdk_acc.numpy()[:, :count]
        vel[:, :count] = torch.from_numpy(v)
        pos_next[rank, :, :count] = pos_cur[rank, :, :count] + dt * torch.from_numpy(v)


def _leapfrog_worker(rank, world, port, n, steps, overlap, integrator, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_mi355x as nb
        from nbody_mi355x.sharded import ShardedNBody
        p, m = nb.ic_plummer(n, 42)
        sim = ShardedNBody(p, m, 1e-2, rank=rank, world=world, device="cpu", G=1.0,
                           backend=LeapfrogBackend(n, world), overlap=overlap, integrator=integrator)
        for k in range(steps):
            sim.step(snapshot=(k == steps - 1))
        q.put((rank, sim.own_state(), sim.own_snapshot(), sim.all_positions()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,overlap,integrator", [(2, 120, True, "KDK"), (3, 97, False, "KDK"),
                                                        (2, 120, True, "KICK_DRIFT"), (2, 120, True, "EULER")])
def test_sharded_driver_integrator_sequencing(world, n, overlap, integrator):
    """The driver's leapfrog: prime (a KDK evaluation with dt = 0), then per small step a drift
    of the own slab, the exchange, one evaluation with the second kick -- several ranks over
    gloo against a single-process numpy leapfrog; symplectic Euler and Euler through the same
    backend for contrast."""
    import nbody_mi355x as nb
    which = getattr(nb, integrator)
    steps, dt = 7, 1e-2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_leapfrog_worker, args=(r, world, port, n, steps, overlap, which, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p, m = nb.ic_plummer(n, 42)
    x, v = p[:, :3].copy(), p[:, 3:].copy()
    acc = lambda x: LeapfrogBackend._sums(x, m, x, [(0, n)])   # noqa: E731
    a = acc(x)
    for _ in range(steps):
        if which == nb.KDK:
            v += 0.5 * dt * a
            x = x + dt * v
            a = acc(x)
            v += 0.5 * dt * a
        elif which == nb.KICK_DRIFT:
            a = acc(x)
            v += dt * a
            x = x + dt * v
        else:
            a = acc(x)
            x = x + dt * v
            v += dt * a
    state = np.concatenate([r[1] for r in res], axis=0)
    snap = np.concatenate([r[2] for r in res], axis=0)
    np.testing.assert_allclose(state[:, :3], x, rtol=0, atol=1e-13 * np.abs(x).max())
    np.testing.assert_allclose(state[:, 3:], v, rtol=0, atol=1e-13 * np.abs(v).max())
    if which == nb.KDK:          # the snapshot is the synchronised state {r, v, a(r)} after the step
        np.testing.assert_allclose(snap[:, :3], x, rtol=0, atol=1e-13 * np.abs(x).max())
        np.testing.assert_allclose(snap[:, 6:9], a, rtol=0, atol=1e-12 * np.abs(a).max())
    for r in res:
        np.testing.assert_allclose(r[3], x, rtol=0, atol=1e-13 * np.abs(x).max())
