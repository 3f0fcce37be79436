"""Scenes with one random stream per body (csrc/nbody_scene.hip; SURVEY 8f-3: initial conditions
at scale): the host side here, the device side in test_gpu_scene.py.  The recipes are those of
the serial generators (csrc/nbody_ics.cpp), which follow the reference's saturn_ring.cc:42-63 /
particle_initializer.h:120-175 for the ring."""
import ctypes as C

import numpy as np
import pytest

import nbody_mi355x as nb


@pytest.mark.parametrize("kind", [nb.SCENE_PLUMMER, nb.SCENE_UNIFORM_CUBE, nb.SCENE_RING])
def test_a_body_is_a_function_of_scene_and_index(kind):
    """Any range of bodies, made on its own, is that range of the whole scene -- whatever the
    thread count (ranges below and above the threading threshold); other seeds, other bodies."""
    n = 70000
    sc = nb.make_scene(kind, n, seed=9)
    p, m = nb.scene_host(sc)
    for first, count in [(0, 1), (123, 4567), (n - 20000, 20000), (n - 1, 1), (5, 0)]:
        q, mq = nb.scene_host(sc, first, count)
        assert np.array_equal(q, p[first:first + count]) and np.array_equal(mq, m[first:first + count])
    p2, _ = nb.scene_host(sc)
    assert np.array_equal(p, p2)
    other, _ = nb.scene_host(nb.make_scene(kind, n, seed=10))
    assert not np.array_equal(other[1:, :3], p[1:, :3])
    assert np.all(np.isfinite(p)) and np.all(np.isfinite(m)) and np.all(m >= 0)


def test_plummer_scene_is_a_plummer_sphere():
    """Half-mass radius 1.305 a, <v^2> = 3 pi / 32 (G M = a = 1), virial ratio 2K/|W| = 1,
    truncated at 10 a, equal masses 1/(G n) -- and statistically the serial generator's sphere."""
    n = 200000
    sc = nb.make_scene(nb.SCENE_PLUMMER, n, seed=3)
    p, m = nb.scene_host(sc)
    r = np.linalg.norm(p[:, :3], axis=1)
    assert abs(np.median(r) - 1.305) < 0.02 and r.max() <= 10.0      # (truncation moves the median by ~1 %)
    assert abs((p[:, 3:] ** 2).sum(1).mean() - 3 * np.pi / 32) < 0.01
    assert np.allclose(m, 1.0 / (nb.G_REFERENCE * n), rtol=1e-15)
    assert np.abs(p.mean(0)).max() < 0.02                            # no systematic offset (not yet in the c.o.m. frame)
    ps, _ = nb.ic_plummer(n, 3)
    rs = np.linalg.norm(ps[:, :3], axis=1)
    for q in (0.1, 0.5, 0.9):
        assert abs(np.quantile(r, q) / np.quantile(rs, q) - 1) < 0.02
    k = 4000                                                         # virial ratio on a subsample, O(k^2)
    x, v = p[:k, :3], p[:k, 3:]
    d = np.linalg.norm(x[:, None] - x[None], axis=2)
    w = -0.5 * (1.0 / k) ** 2 * (1.0 / d[d > 0]).sum()                # G M = 1, masses 1/k
    kin = 0.5 * (1.0 / k) * (v ** 2).sum()
    assert abs(2 * kin / abs(w) - 1) < 0.1


def test_cube_and_ring_scenes():
    n = 100000
    p, m = nb.scene_host(nb.make_scene(nb.SCENE_UNIFORM_CUBE, n, seed=7))
    assert np.abs(p[:, :3]).max() < 1 and np.abs(p[:, :3].mean(0)).max() < 0.01 and not p[:, 3:].any()
    assert abs(p[:, :3].var() - 1 / 3) < 0.01
    sc = nb.make_scene(nb.SCENE_RING, n, seed=42)
    p, m = nb.scene_host(sc)
    assert not p[0].any() and m[0] == 5.683e26                       # the planet, at rest at the origin
    rad = np.linalg.norm(p[1:, :2], axis=1)
    assert abs(rad.mean() / 1.12e8 - 1) < 0.02 and abs(rad.std() / 3.0e7 - 1) < 0.05
    assert abs(p[1:, 2].std() / 50 - 1) < 0.05 and not p[1:, 5].any()
    # circular speed about the planet plus 50 m/s of scatter, prograde (ref: particle_initializer.h:120-175)
    speed = np.linalg.norm(p[1:, 3:5], axis=1)
    assert np.abs(speed / np.sqrt(6.67430e-11 * 5.683e26 / rad) - 1).max() < 0.05
    lz = p[1:, 0] * p[1:, 4] - p[1:, 1] * p[1:, 3]
    assert (lz > 0).all()
    assert abs(m[1:].mean() / 0.1 ** 2 - 1) < 0.05                   # masses = N(0, 0.1)^2


def test_scene_argument_checks():
    L = nb.lib()
    p = np.empty((4, 6))
    m = np.empty(4)
    args = (p.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
    ok = nb.make_scene(nb.SCENE_PLUMMER, 4)
    assert L.nbody_mi355x_scene_host(C.byref(ok), 0, 4, *args) == 0
    for bad in (nb.make_scene(7, 4), nb.make_scene(nb.SCENE_PLUMMER, 0), nb.make_scene(nb.SCENE_PLUMMER, 4, G=0.0),
                nb.make_scene(nb.SCENE_RING, 4, central_mass=-1.0), nb.make_scene(nb.SCENE_RING, 4, width_sigma=-1.0)):
        assert L.nbody_mi355x_scene_host(C.byref(bad), 0, 4, *args) == nb.ERR_INVALID
        assert nb.last_error()
    for first, count in [(-1, 2), (3, 2), (0, 5), (0, -1)]:
        assert L.nbody_mi355x_scene_host(C.byref(ok), first, count, *args) == nb.ERR_INVALID
    assert L.nbody_mi355x_scene_host(C.byref(ok), 0, 4, None, args[1]) == nb.ERR_INVALID
    assert L.nbody_mi355x_scene_host(None, 0, 4, *args) == nb.ERR_INVALID
