"""bench.py's verification leg (what makes a multi-rank line self-proving) on CPU: the check
functions on known inputs, and verify_run end to end over gloo with world_size 2, the ORACLE
standing in for the kernel (tests/test_sharded_gloo.py's backend) -- sampled rows, momentum,
checksum file written by one run and read by the next.  The device-side fields
(`collectives`, `collective_ms`, the one-GPU re-run) are covered on the GPU box by
tests/test_gpu_sharded.py::test_bench_self_launch_two_ranks_share_the_gpu."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "n-body-sim_amd"))

import bench  # noqa: E402
from test_sharded_gloo import OracleBackend, _free_port  # noqa: E402


def test_sampled_rows_are_the_same_everywhere_and_partition_over_ranks():
    from nbody_mi355x.sharded import shard_counts
    n = 5000
    rows = bench.sampled_rows(n)
    assert np.array_equal(rows, bench.sampled_rows(n)) and len(rows) == bench.VERIFY_ROWS
    assert len(np.unique(rows)) == len(rows) and rows.min() >= 0 and rows.max() < n
    counts, begins = shard_counts(n, 3)
    seen = np.zeros(len(rows), dtype=int)
    for b, c in zip(begins, counts):
        sel, local = bench.rows_of_rank(rows, b, c)
        assert np.array_equal(rows[sel], local + b) and (local >= 0).all() and (local < c).all()
        seen[sel] += 1
    assert (seen == 1).all(), "every sampled body belongs to exactly one rank"
    assert len(bench.sampled_rows(100)) == 100      # small systems: every body


def test_check_functions_pass_and_fail_where_they_should():
    rng = np.random.default_rng(3)
    a = rng.normal(size=(64, 3))
    assert bench.check_rows(a, a, 1e-12)["ok"]
    assert bench.check_rows(a * (1 + 1e-14), a, 1e-12)["ok"]
    bad = a.copy()
    bad[5] *= 1.001
    r = bench.check_rows(bad, a, 1e-12)
    assert not r["ok"] and r["max_err_over_max_acc"] > 1e-6
    # momentum: equal and opposite forces cancel; a missing partner does not
    f = rng.normal(size=(10, 3))
    assert bench.check_momentum([f.sum(0), -f.sum(0)], [1.0, 1.0], 1e-12)["ok"]
    assert not bench.check_momentum([f.sum(0), np.zeros(3)], [1.0, 1.0], 1e-12)["ok"]
    assert not bench.check_momentum([np.zeros(3)], [0.0], 1e-12)["ok"], "no force at all is not a pass"
    x = rng.normal(size=(100, 3))
    assert bench.check_positions(x, x, 1e-12)["ok"]
    y = x.copy()
    y[7, 1] += 1e-9
    assert not bench.check_positions(y, x, 1e-12)["ok"]
    assert not bench.check_positions(x[:50], x, 1e-12)["ok"]


def _args(**kw):
    base = dict(write_checksum=None, reference_checksum=None, force_collectives=False, verify_budget_s=0.0)
    base.update(kw)
    return argparse.Namespace(**base)


def _worker(rank, world, port, n, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_mi355x as nb
        from nbody_mi355x.sharded import ShardedNBody
        p, m = nb.ic_plummer(n, 42)
        out = []
        for reference in (None, Path(tmp) / "pos.npy", Path(tmp) / "bad.npy"):
            sim = ShardedNBody(p, m, 1e-3, rank=rank, world=world, device="cpu", backend=OracleBackend(n, world))
            for _ in range(3):
                sim.step()
            if reference is not None and reference.name == "bad.npy" and rank == 0:
                good = np.load(Path(tmp) / "pos.npy")
                good[n // 2, 0] *= 1 + 1e-9
                np.save(reference, good)
            dist.barrier()
            args = _args(write_checksum=Path(tmp) / "pos.npy" if reference is None else None,
                         reference_checksum=reference)
            out.append(bench.verify_run(args, sim, nb, dist, world, rank, 3, None, 1e-3, nb.FP64, nb.EULER))
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1500), (3, 700)])
def test_verify_run_over_gloo_with_the_oracle_as_the_kernel(world, n, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, str(tmp_path), q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for r in range(1, world):
        assert res[r] == [None, None, None], "only rank 0 holds the verdict"
    first, second, third = res[0]
    assert first["ok"] and first["steps_before"] == 3 and first["sampled_rows"] == min(n, bench.VERIFY_ROWS)
    assert first["rows_vs_oracle"]["ok"] and first["rows_vs_oracle"]["max_err_over_max_acc"] == 0.0   # the oracle against itself
    assert first["momentum"]["ok"] and 0 <= first["momentum"]["rel_net_force"] <= 1e-12
    assert first["checksum_written"]["steps"] == 4 and (tmp_path / "pos.npy").exists()
    assert "positions_vs_one_gpu" not in first, "no host arrays given: no one-GPU re-run"
    assert second["ok"] and second["positions_vs_reference_file"]["max_rel_position_diff"] == 0.0
    assert not third["ok"] and not third["positions_vs_reference_file"]["ok"]
    assert third["rows_vs_oracle"]["ok"], "only the position check failed"
