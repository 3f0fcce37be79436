"""The handle API over several shards driven by one host thread (nbody_mi355x_create_multi,
csrc/nbody_multi.hip).  A one-GPU box has one device, so the shards are placed on device 0
several times over: the partition, the slab layout, both exchanges (partial sums, new
positions), their event ordering and the snapshot / download / energy paths are the real
ones; only the peer copies are same-device copies instead of xGMI transfers.
Comparator: the oracle (tests/oracle_lib.py), as everywhere."""
import numpy as np
import pytest

import nbody_mi355x as nb
import oracle_lib as ol
from conftest import load_golden, rel_pos_err

pytestmark = pytest.mark.gpu


def multi_simulate(p, m, shards, big, sub, dt, **kw):
    outs = []
    with nb.Simulation(p.shape[0], dt, devices=[0] * shards, **kw) as sim:
        sim.upload(p, m)
        for _ in range(big):
            outs.append(sim.run(sub, snapshot=True))
        fin = sim.download()
        e = sim.energy()
    return fin, np.stack(outs), e


@pytest.mark.parametrize("shards,n", [(2, 1000), (3, 777), (4, 4096), (8, 300)])
def test_one_sided_shards_against_oracle(shards, n):
    """Small systems: every shard runs the one-sided kernel over all slabs; ragged shard sizes."""
    p, m = nb.ic_plummer(n, 100 + n)
    fin_ref, ref = ol.simulate(p, m, 2, 5, 1e-3, threads=8)
    fin, out, _ = multi_simulate(p, m, shards, 2, 5, 1e-3, variant=1)
    assert rel_pos_err(out[1, :, :3], ref[1, :, :3]) <= 1e-13
    assert rel_pos_err(fin[:, :3], fin_ref[:, :3]) <= 1e-13
    assert np.abs(out[1, :, 3:6] - ref[1, :, 3:6]).max() <= 1e-13 * np.abs(ref[1, :, 3:6]).max()
    num = np.linalg.norm(out[1, :, 6:9] - ref[1, :, 6:9], axis=1)
    assert (num / np.linalg.norm(ref[1, :, 6:9], axis=1)).max() <= 1e-12


@pytest.mark.parametrize("shards,n,variant", [(2, 20000, 0), (4, 16384, 0), (3, 30000, 0), (8, 65536, 0),
                                              (2, 9000, nb.VARIANT_SYM2), (4, 40000, nb.VARIANT_SYM8)])
def test_symmetric_shards_against_oracle(shards, n, variant):
    """Large systems: pair-symmetric tiles per shard, partial sums pulled and added in shard
    order, then integrated; several big steps so that both position buffers, the event
    ordering across steps and the snapshot of a later big step are exercised."""
    p, m = nb.ic_plummer(n, 13)
    fin_ref, ref = ol.simulate(p, m, 2, 3, 1e-3, threads=16)
    fin, out, _ = multi_simulate(p, m, shards, 2, 3, 1e-3, variant=variant)
    assert rel_pos_err(out[0, :, :3], ref[0, :, :3]) <= 1e-13
    assert rel_pos_err(out[1, :, :3], ref[1, :, :3]) <= 1e-13
    assert rel_pos_err(fin[:, :3], fin_ref[:, :3]) <= 1e-13
    num = np.linalg.norm(out[1, :, 6:9] - ref[1, :, 6:9], axis=1)
    assert (num / np.linalg.norm(ref[1, :, 6:9], axis=1)).max() <= 1e-12


def test_multi_golden_100_steps_reproducible_and_energy():
    """The reference's golden vector through 4 shards: 100 steps inside the 1e-10 bar, the
    same bits run to run, and the energy diagnostic equal to the single-handle one."""
    g = load_golden("g4_plummer_4096")
    p, m, dt = g["particles"], g["mass"], float(g["dt"])
    fin_a, a, e_multi = multi_simulate(p, m, 4, 1, 100, dt)
    fin_b, b, _ = multi_simulate(p, m, 4, 1, 100, dt)
    assert rel_pos_err(a[0, :, :3], g["out_cpu"][0, :, :3]) <= 1e-12
    assert np.array_equal(a, b) and np.array_equal(fin_a, fin_b)
    with nb.Simulation(p.shape[0], dt) as sim:
        sim.upload(p, m)
        sim.run(100, snapshot=False)
        e_one = sim.energy()
    assert e_multi[0] == pytest.approx(e_one[0], rel=1e-12)
    assert e_multi[1] == pytest.approx(e_one[1], rel=1e-12)


def test_multi_float_host_records_and_fp32():
    g = load_golden("g6_plummer_1024_f32")
    p, m = g["particles"], g["mass"]
    with nb.Simulation(p.shape[0], float(g["dt"]), devices=[0, 0], host_is_float=True) as sim:
        sim.upload(p, m)
        out = sim.run(100, snapshot=True)
        fin = sim.download()
    assert out.dtype == np.float32 and fin.dtype == np.float32
    _, ref = ol.simulate(p.astype(np.float64), m.astype(np.float64), 1, 100, float(g["dt"]), threads=8)
    assert rel_pos_err(out[:, :3].astype(np.float64), ref[0, :, :3]) <= 2e-7
    p2, m2 = nb.ic_plummer(20000, 3)
    _, ref2 = ol.simulate(p2, m2, 1, 5, 1e-3, threads=16)
    _, out2, _ = multi_simulate(p2, m2, 2, 1, 5, 1e-3, precision=nb.FP32)
    err = rel_pos_err(out2[0, :, :3], ref2[0, :, :3])
    assert 1e-13 < err <= 1e-4


def test_multi_errors():
    with pytest.raises(nb.NBodyError):
        nb.Simulation(100, 0.1, devices=[0, 99])
    with pytest.raises(nb.NBodyError):
        nb.Simulation(3, 0.1, devices=[0] * 4)   # more shards than bodies
    with nb.Simulation(100, 0.1, devices=[0]) as sim:   # one shard = the plain handle
        p, m = nb.ic_plummer(100, 1)
        sim.upload(p, m)
        assert sim.run(2).shape == (100, 9)


def _rccl_fallback_main(q):
    """Own process: the communicator is built, its self-check is made to fail (test hook), and the
    handle must fall back to the peer copies, say why, and still compute the right thing."""
    import os
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(repo / "n-body-sim_amd"))
    os.environ["NBODY_MI355X_MULTI_FORCE"] = "1"
    os.environ["NBODY_MI355X_MULTI_RCCL"] = "1"
    os.environ["NBODY_MI355X_MULTI_SELFCHECK_FAIL"] = "1"
    import torch  # noqa: F401
    import nbody_mi355x as nb
    n = 20000
    p, m = nb.ic_plummer(n, 13)
    with nb.Simulation(n, 1e-3, devices=[0]) as sim:
        text = nb.last_error()
        backend = sim.exchange_backend()
        sim.upload(p, m)
        out = sim.run(3, snapshot=True)
    q.put((text, backend, out))


def test_rccl_self_check_failure_falls_back_to_peer_copies():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_fallback_main, args=(q,))
    pr.start()
    text, backend, out = q.get(timeout=200)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    assert backend == nb.EXCHANGE_PEER_COPIES
    assert "self-check" in text and "using peer copies" in text, text
    p, m = nb.ic_plummer(20000, 13)
    _, ref = ol.simulate(p, m, 1, 3, 1e-3, threads=16)
    assert rel_pos_err(out[:, :3], ref[0, :, :3]) <= 1e-13


def _rccl_one_rank_main(n, q):
    """Own process (the two environment switches are read at create): one shard through the
    multi-shard code with its exchanges done by RCCL -- ncclCommInitAll over one device,
    in-place ncclAllGather of the positions, ncclReduceScatter of the partial sums."""
    import os
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(repo / "n-body-sim_amd"))
    os.environ["NBODY_MI355X_MULTI_FORCE"] = "1"
    os.environ["NBODY_MI355X_MULTI_RCCL"] = "1"
    import torch  # noqa: F401  (maps torch's librccl: the library must pick that copy up)
    import nbody_mi355x as nb
    p, m = nb.ic_plummer(n, 13)
    with nb.Simulation(n, 1e-3, devices=[0]) as sim:
        text = nb.last_error()              # create leaves a note here if it fell back to peer copies
        backend = sim.exchange_backend()
        sim.upload(p, m)
        outs = [sim.run(3, snapshot=True) for _ in range(2)]
        seconds = sim.last_exchange_seconds()
        fin = sim.download()
    q.put((text, outs, fin, backend, seconds))


@pytest.mark.parametrize("n", [1500, 20000])
def test_rccl_exchanges_with_a_one_rank_communicator(n):
    """NBODY_MI355X_MULTI_RCCL=1: the C library's own RCCL path (resolved with dlopen, no link
    dependency) with the one rank a one-GPU box allows, one-sided (1500) and pair-symmetric
    (20 000: phases 1 and 2 around the exchange, reduce-scatter of the partial sums)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_one_rank_main, args=(n, q))
    pr.start()
    text, outs, fin, backend, seconds = q.get(timeout=200)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    # RCCL was loaded, the communicator built, and its SELF-CHECK at create passed (all-gather
    # and reduce-scatter of a patterned buffer, bit for bit against the peer copies)
    assert "peer copies" not in text, text
    assert backend == nb.EXCHANGE_RCCL
    assert seconds[0] > 0 and (seconds[1] > 0) == (n >= 2048), seconds   # the sums are exchanged on the pair-symmetric path only
    p, m = nb.ic_plummer(n, 13)
    fin_ref, ref = ol.simulate(p, m, 2, 3, 1e-3, threads=16)
    for k in range(2):
        assert rel_pos_err(outs[k][:, :3], ref[k, :, :3]) <= 1e-13
        num = np.linalg.norm(outs[k][:, 6:9] - ref[k, :, 6:9], axis=1)
        assert (num / np.linalg.norm(ref[k, :, 6:9], axis=1)).max() <= 1e-12
    assert rel_pos_err(fin[:, :3], fin_ref[:, :3]) <= 1e-13


def test_multi_overlap_many_steps_bit_reproducible():
    """The split step (diagonal tiles under the exchange on the comm stream, the rest after it)
    over 40 small steps on 8 shards: same bits run to run -- any event missing between the
    compute and comm streams would show as a difference or as an error against the oracle."""
    n = 24000
    p, m = nb.ic_plummer(n, 5)
    fin_a, a, _ = multi_simulate(p, m, 8, 2, 20, 1e-3)
    fin_b, b, _ = multi_simulate(p, m, 8, 2, 20, 1e-3)
    assert np.array_equal(a, b) and np.array_equal(fin_a, fin_b)
    _, ref = ol.simulate(p, m, 2, 20, 1e-3, threads=16)
    assert rel_pos_err(a[1, :, :3], ref[1, :, :3]) <= 1e-12


def test_exchange_backend_and_measured_exchange_cost():
    """Which exchange a handle uses and what the two exchanges of a small step cost on the device
    (nbody_mi355x_exchange_backend / _last_exchange_seconds): shards that share a GPU take the
    event-ordered peer copies (RCCL needs a GPU per rank -- on a multi-GPU node it is the default
    behind its self-check); a one-device handle exchanges nothing."""
    n = 24000
    p, m = nb.ic_plummer(n, 5)
    with nb.Simulation(n, 1e-3, devices=[0, 0, 0, 0]) as sim:
        assert sim.exchange_backend() == nb.EXCHANGE_PEER_COPIES
        assert nb.last_error() == ""        # nothing to report: RCCL was not tried for shards sharing a GPU
        sim.upload(p, m)
        sim.run(4, snapshot=False)
        pos_s, sums_s = sim.last_exchange_seconds()
        step_s = sim.last_run_seconds() / 4
    assert 0 < pos_s < step_s and 0 < sums_s < step_s
    with nb.Simulation(1000, 1e-3, devices=[0, 0], variant=1) as sim:   # one-sided: positions only
        sim.upload(*nb.ic_plummer(1000, 1))
        sim.run(2, snapshot=False)
        pos_s, sums_s = sim.last_exchange_seconds()
    assert pos_s > 0 and sums_s == 0
    with nb.Simulation(n, 1e-3) as sim:
        sim.upload(p, m)
        sim.run(1, snapshot=False)
        assert sim.exchange_backend() == nb.EXCHANGE_NONE and sim.last_exchange_seconds() == (0.0, 0.0)
