"""Scenes made on the device (nbody_mi355x_generate / nbody_mi355x_scene_fill_f64,
csrc/nbody_scene.hip) against the same code on the host's threads: one random stream per
body, so device and host produce the same bodies up to the last bits of pow / log / cos."""
import numpy as np
import pytest

import torch  # noqa: F401  (before nbody_mi355x: one shared HIP runtime)
import nbody_mi355x as nb
from conftest import rel_pos_err

pytestmark = pytest.mark.gpu


def host_scene(sc):
    p, m = nb.scene_host(sc)
    if sc.scene == nb.SCENE_PLUMMER:          # generate() moves the sphere into its centre-of-mass frame
        p = p - p.mean(0)
    return p, m


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("kind,n", [(nb.SCENE_PLUMMER, 30001), (nb.SCENE_UNIFORM_CUBE, 5000), (nb.SCENE_RING, 20000),
                                    (nb.SCENE_PLUMMER, 300)])
def test_device_scene_equals_host_scene(kind, n, devices):
    kw = {} if devices is None else {"devices": devices}
    sc = nb.make_scene(kind, n, seed=17)
    p, m = host_scene(sc)
    with nb.Simulation(n, 1e-3, **kw) as sim:
        sim.generate(sc)
        got = sim.download()
        out = sim.run(1, snapshot=True)          # the masses are only visible through what they do
        sim.upload(p, m)
        ref = sim.run(1, snapshot=True)
    scale = np.abs(p[:, :3]).max()
    assert np.abs(got[:, :3] - p[:, :3]).max() <= 1e-13 * scale
    vmax = np.abs(p[:, 3:]).max()
    assert np.abs(got[:, 3:] - p[:, 3:]).max() <= 1e-13 * vmax + 1e-300
    amax = np.linalg.norm(ref[:, 6:9], axis=1).max()
    assert np.abs(out[:, 6:9] - ref[:, 6:9]).max() <= 1e-10 * amax
    if kind == nb.SCENE_PLUMMER:
        assert np.abs(got.mean(0)).max() <= 1e-15 * n ** 0.5
    if kind == nb.SCENE_RING:
        assert not got[0].any()


def test_generate_at_scale_and_argument_checks():
    """1M bodies made in place on the device, then stepped: a virialised sphere stays one; a
    scene of another size, or an unknown one, is refused."""
    n = 1 << 20
    with nb.Simulation(n, 1e-3) as sim:
        sim.generate(nb.make_scene(nb.SCENE_PLUMMER, n, seed=42))
        ke, pe = sim.energy()
        assert abs(2 * ke / abs(pe) - 1) < 0.02
        snap = sim.run(1, snapshot=True)
        assert np.all(np.isfinite(snap))
        before = sim.download()
        with pytest.raises(nb.NBodyError):
            sim.generate(nb.make_scene(nb.SCENE_PLUMMER, n - 1))
        with pytest.raises(nb.NBodyError):
            sim.generate(nb.make_scene(9, n))
        assert np.array_equal(sim.download(), before)      # a refused call leaves the state alone


def test_scene_fill_level_1_into_torch_slabs():
    """Level 1 as the one-process-per-GPU driver would use it: a rank fills its own slab of
    bodies [first, first + count) of a scene into torch-owned memory, sums it and shifts it."""
    import ctypes as C
    n, first, count, S = 50000, 12345, 20000, 20480
    sc = nb.make_scene(nb.SCENE_PLUMMER, n, seed=5)
    dev = torch.device("cuda", 0)
    pos = torch.zeros((3, S), dtype=torch.float64, device=dev)
    vel = torch.zeros((3, S), dtype=torch.float64, device=dev)
    mass = torch.zeros(S, dtype=torch.float64, device=dev)
    sums = torch.zeros(6, dtype=torch.float64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L = nb.lib()
    nb.check(L.nbody_mi355x_scene_fill_f64(C.byref(sc), first, count, C.c_void_p(pos.data_ptr()), C.c_void_p(vel.data_ptr()),
                                           C.c_void_p(mass.data_ptr()), S, stream))
    nb.check(L.nbody_mi355x_slab_sums_f64(C.c_void_p(pos.data_ptr()), C.c_void_p(vel.data_ptr()), count, S,
                                          C.c_void_p(sums.data_ptr()), stream))
    p, m = nb.scene_host(sc, first, count)
    torch.cuda.synchronize()
    assert np.abs(pos[:, :count].cpu().numpy().T - p[:, :3]).max() <= 1e-13 * np.abs(p[:, :3]).max()
    assert np.abs(vel[:, :count].cpu().numpy().T - p[:, 3:]).max() <= 1e-13
    assert np.array_equal(mass[:count].cpu().numpy(), m) and not pos[:, count:].any() and not mass[count:].any()
    s = sums.cpu().numpy()
    assert np.abs(s - p.sum(0)).max() <= 1e-12 * np.abs(p).sum(0).max()
    shift = (C.c_double * 6)(*(s / count))
    nb.check(L.nbody_mi355x_slab_shift_f64(C.c_void_p(pos.data_ptr()), C.c_void_p(vel.data_ptr()), count, S, shift, stream))
    torch.cuda.synchronize()
    assert np.abs(pos[:, :count].cpu().numpy().mean(1)).max() <= 1e-15 and not pos[:, count:].any()
    assert L.nbody_mi355x_scene_fill_f64(C.byref(sc), n - 5, 6, C.c_void_p(pos.data_ptr()), None, C.c_void_p(mass.data_ptr()), S, stream) == nb.ERR_INVALID
    assert L.nbody_mi355x_scene_fill_f64(C.byref(sc), 0, S + 1, C.c_void_p(pos.data_ptr()), None, C.c_void_p(mass.data_ptr()), S, stream) == nb.ERR_INVALID
