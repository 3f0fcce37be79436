"""The oracle (oracle/nbody_oracle.c) against the golden vectors produced by the REAL
reference (tests/golden/make_golden.py).  CPU only.  Bar: bit-for-bit -- the oracle is a
restatement of the same expressions compiled with the same flags."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_golden


def _run(g, kind, threads):
    fin, out = ol.simulate(g["particles"], g["mass"], int(g["big_steps"]), int(g["substeps"]),
                           float(g["dt"]), threads=threads, kind=kind)
    return fin, out


@pytest.mark.parametrize("name,threads", [("g1_earth_moon", 1), ("g2_earth_7moons", 1),
                                          ("g3_saturn_ring_500", 4), ("g7_edge_cases", 3)])
def test_scalar_f64_full_records(name, threads):
    g = load_golden(name)
    _, out = _run(g, "f64", threads)
    ref = g["out_cpu"]  # (big+1, n, 11) = {r,v,a,m,t}; row 0 is the ICs with a = 0
    assert np.array_equal(ref[0, :, :6], g["particles"])
    assert np.all(ref[0, :, 6:9] == 0)
    assert np.array_equal(out, ref[1:, :, :9])
    # m and t bookkeeping (particle_simulation.h:317-319): t = dt * big * substeps
    big = np.arange(ref.shape[0])
    assert np.array_equal(ref[:, 0, 10], float(g["dt"]) * big * int(g["substeps"]))
    assert np.array_equal(ref[1, :, 9], g["mass"])


@pytest.mark.parametrize("name", ["g4_plummer_1024", "g4_plummer_4096", "g8_si_cloud_512"])
def test_scalar_f64_plummer_and_si(name):
    g = load_golden(name)
    fin, out = _run(g, "f64", 8)
    assert np.array_equal(out, g["out_cpu"])
    assert np.array_equal(fin, g["out_cpu"][-1, :, :6])  # final state == last snapshot's r,v


def test_thread_count_does_not_change_results():
    g = load_golden("g4_plummer_1024")
    _, o1 = _run(g, "f64", 1)
    _, o5 = _run(g, "f64", 5)
    assert np.array_equal(o1, o5)


def test_single_force_evaluation():
    ics = load_golden("g4_plummer_4096")
    g = load_golden("g5_plummer_4096_force")
    _, out = ol.simulate(ics["particles"], ics["mass"], 1, 1, float(g["dt"]), threads=8)
    assert np.array_equal(out, g["out_cpu"])
    rows = np.array([0, 1, 77, 4095])
    acc = ol.accel_rows(ics["particles"], ics["mass"], rows)
    assert np.array_equal(acc, g["out_cpu"][0, rows, 6:9])


@pytest.mark.parametrize("name,threads", [("g2_earth_7moons", 2), ("g3_saturn_ring_500", 4),
                                          ("g4_plummer_1024", 8), ("g8_si_cloud_512", 8)])
def test_avx2_restatement(name, threads):
    g = load_golden(name)
    _, out = _run(g, "avx2", threads)
    ref = g["out_avx2"]
    if ref.shape[-1] == 11:
        ref = ref[1:, :, :9]
    assert np.array_equal(out, ref)


def test_avx2_ragged_n_is_safe_and_close():
    """N % 4 != 0 corrupts the reference's heap (particle_simulation.h:514-540); the
    restatement must handle it and agree with the scalar path to rounding."""
    g = load_golden("g7_edge_cases")
    _, o_avx = _run(g, "avx2", 2)
    _, o_cpu = _run(g, "f64", 1)
    np.testing.assert_allclose(o_avx, o_cpu, rtol=1e-12, atol=1e-18)


def test_float_instantiation():
    g = load_golden("g6_plummer_1024_f32")
    _, out = _run(g, "f32", 8)
    assert out.dtype == np.float32
    assert np.array_equal(out, g["out_cpu"][1:, :, :9])


def test_energy_diagnostic_two_body():
    """KE/PE of the earth-moon ICs against the closed form."""
    g = load_golden("g1_earth_moon")
    p, m = g["particles"], g["mass"]
    ke, pe = ol.energy(p, m, threads=2)
    G = 6.7430e-11
    assert ke == pytest.approx(0.5 * m[1] * np.sum(p[1, 3:] ** 2), rel=1e-15)
    assert pe == pytest.approx(-G * m[0] * m[1] / 3.844e8, rel=1e-15)


def test_earth_moon_physics_known_answers():
    """SURVEY 4 known-answer row: momentum conserved to rounding, separation within a few %."""
    g = load_golden("g1_earth_moon")
    p, m = g["particles"], g["mass"]
    _, out = ol.simulate(p, m, 4, 25000, 0.1)  # 10 000 s
    mom0 = (m[:, None] * p[:, 3:]).sum(0)
    mom = (m[:, None] * out[-1, :, 3:6]).sum(0)
    assert np.linalg.norm(mom - mom0) <= 1e-12 * np.linalg.norm(mom0)
    sep = np.linalg.norm(out[-1, 1, :3] - out[-1, 0, :3])
    assert abs(sep / 3.844e8 - 1) < 0.03
