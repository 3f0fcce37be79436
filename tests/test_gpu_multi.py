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
