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
