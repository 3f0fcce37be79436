"""ctypes binding of libnbody_mi355x.so (C ABI: include/nbody_mi355x.h).

This package is plumbing for tests and bench.py: it loads the in-tree shared
library and mirrors its C structs.  There is NO fallback: if the library is not
built, or a call fails, an exception is raised -- nothing here computes forces
on the CPU.

If torch is going to be used in the same process, import torch BEFORE this
module: both link libamdhip64.so.7 and must share one HIP runtime (the first
one loaded wins).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent  # n-body-sim_amd/
# (NBODY_MI355X_LIB: another build of the same ABI, for A/B measurements on one GPU lease)
LIB_PATH = Path(os.environ["NBODY_MI355X_LIB"]) if os.environ.get("NBODY_MI355X_LIB") else ROOT / "lib" / "libnbody_mi355x.so"

TILE = 256
G_REFERENCE = 6.7430e-11  # reference: particle_simulation.h:713
FP64, FP32, FP64_FAST = 0, 1, 2   # FP64_FAST: opt-in tier, include/nbody_mi355x.h
EULER, KICK_DRIFT, KDK, KDK_CONTINUE = 0, 1, 2, 3   # KDK_CONTINUE: level 1 (StepArgs) only

OK = 0
ERR_INVALID, ERR_ALLOC, ERR_H2D, ERR_D2H, ERR_FREE, ERR_LAUNCH, ERR_NO_DEVICE = range(-1, -8, -1)


class NBodyError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"nbody_mi355x error {code}: {msg}")
        self.code = code


class StepArgs(C.Structure):
    """nbody_mi355x_step_args"""
    _fields_ = [
        ("pos_cur", C.c_void_p), ("pos_next", C.c_void_p), ("mass", C.c_void_p),
        ("vel", C.c_void_p), ("acc_in", C.c_void_p), ("acc_out", C.c_void_p),
        ("snapshot", C.c_void_p),
        ("shard_slots", C.c_int64), ("num_shards", C.c_int32), ("own_shard", C.c_int32),
        ("own_count", C.c_int64),
        ("j0_begin", C.c_int64), ("j0_end", C.c_int64), ("j1_begin", C.c_int64), ("j1_end", C.c_int64),
        ("dt", C.c_double), ("G", C.c_double), ("eps2", C.c_double),
        ("precision", C.c_int32), ("variant", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("phase", C.c_int32), ("integrator", C.c_int32),
        ("kdk_acc", C.c_void_p),
    ]


EXCHANGE_NONE, EXCHANGE_PEER_COPIES, EXCHANGE_RCCL = 0, 1, 2
VARIANT_WAVE, VARIANT_TINY = 7, 8
STATIC_SCRATCH_BYTES = 134217728   # NBODY_MI355X_STATIC_SCRATCH_BYTES
PLAN_FIELDS = 12          # NBODY_MI355X_PLAN_FIELDS
WAVE_MAX_BODIES = 2048   # NBODY_MI355X_WAVE_MAX_BODIES: below, whole FP64 steps take the wave kernels
VARIANT_SYM, VARIANT_SYM1, VARIANT_SYM2, VARIANT_SYM4, VARIANT_SYM8 = 10, 11, 12, 14, 18
VARIANT_SYMW1, VARIANT_SYMW2, VARIANT_SYMW4, VARIANT_SYMW8 = 21, 22, 24, 28   # pair-symmetric, wave tiles
SYM_VARIANTS = (VARIANT_SYM, VARIANT_SYM1, VARIANT_SYM2, VARIANT_SYM4, VARIANT_SYM8,
                VARIANT_SYMW1, VARIANT_SYMW2, VARIANT_SYMW4, VARIANT_SYMW8)


class Config(C.Structure):
    """nbody_mi355x_config"""
    _fields_ = [
        ("num_particles", C.c_int64), ("stepsize", C.c_double), ("G", C.c_double),
        ("eps2", C.c_double), ("precision", C.c_int32), ("device", C.c_int32),
        ("host_is_float", C.c_int32), ("variant", C.c_int32),
        ("integrator", C.c_int32), ("reserved", C.c_int32),
    ]


# every symbol include/nbody_mi355x.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "nbody_mi355x_last_error": (C.c_char_p, []),
    "nbody_mi355x_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nbody_mi355x_step_f64": (C.c_int, [C.POINTER(StepArgs), _P]),
    "nbody_mi355x_step_f64_static": (C.c_int, [C.POINTER(StepArgs), _P]),
    "nbody_mi355x_kdk_drift_f64": (C.c_int, [C.POINTER(StepArgs), _P]),
    "nbody_mi355x_step_tile_launches": (C.c_int, [C.POINTER(StepArgs), C.POINTER(C.c_int32)]),
    "nbody_mi355x_step_plan": (C.c_int, [C.POINTER(StepArgs), C.POINTER(C.c_int32)]),
    "nbody_mi355x_step_variant": (C.c_int, [C.POINTER(StepArgs), C.POINTER(C.c_int32)]),
    "nbody_mi355x_step_workspace_bytes": (C.c_int, [C.POINTER(StepArgs), C.POINTER(C.c_int64)]),
    "nbody_mi355x_step_workspace_min_bytes": (C.c_int, [C.POINTER(StepArgs), C.POINTER(C.c_int64)]),
    "nbody_mi355x_pack_f64": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P]),
    "nbody_mi355x_unpack_f64": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    "nbody_mi355x_energy_f64": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                          C.c_double, _P, _P, _P]),
    "nbody_mi355x_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "nbody_mi355x_create_multi": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_P)]),
    "nbody_mi355x_destroy": (C.c_int, [_P]),
    "nbody_mi355x_upload": (C.c_int, [_P, _P, _P]),
    "nbody_mi355x_run": (C.c_int, [_P, C.c_int64, _P]),
    "nbody_mi355x_run_async": (C.c_int, [_P, C.c_int64, _P]),
    "nbody_mi355x_wait": (C.c_int, [_P]),
    "nbody_mi355x_host_alloc": (C.c_int, [C.c_int64, C.POINTER(_P)]),
    "nbody_mi355x_host_free": (C.c_int, [_P]),
    "nbody_mi355x_download": (C.c_int, [_P, _P]),
    "nbody_mi355x_energy": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nbody_mi355x_kernel_in_use": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "nbody_mi355x_last_run_seconds": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "nbody_mi355x_last_exchange_seconds": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nbody_mi355x_exchange_backend": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "nbody_mi355x_ic_plummer": (C.c_int, [C.c_int64, C.c_uint64, C.c_double, _P, _P]),
    "nbody_mi355x_ic_uniform_cube": (C.c_int, [C.c_int64, C.c_uint64, C.c_double, _P, _P]),
    "nbody_mi355x_ic_ring": (C.c_int, [C.c_int64, C.c_uint64] + [C.c_double] * 6 + [_P, _P]),
    "nbody_mi355x_scene_fill_f64": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, _P]),
    "nbody_mi355x_slab_sums_f64": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    "nbody_mi355x_slab_shift_f64": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    "nbody_mi355x_scene_host": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "nbody_mi355x_generate": (C.c_int, [_P, _P]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C {ROOT}` (or __graft_entry__.build()); "
                "there is no CPU fallback")
        try:
            # torch bundles its own libamdhip64.so.7; load it first so that this process
            # has ONE HIP runtime whichever of the two the caller touches first
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().nbody_mi355x_last_error() or b"").decode()


def check(rc: int) -> None:
    if rc != OK:
        raise NBodyError(rc, last_error())


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().nbody_mi355x_device_count(C.byref(n))
    return n.value if rc == OK else 0


def round_up(v: int, m: int = TILE) -> int:
    return (v + m - 1) // m * m


def _np(a, dtype=np.float64):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- initial conditions
def ic_plummer(n: int, seed: int = 42, G: float = G_REFERENCE):
    """Particle<double>[n] as an (n,6) array {r,v} and masses (n,)."""
    p = np.empty((n, 6), dtype=np.float64)
    m = np.empty(n, dtype=np.float64)
    check(lib().nbody_mi355x_ic_plummer(n, seed, G, p.ctypes.data_as(_P), m.ctypes.data_as(_P)))
    return p, m


def ic_uniform_cube(n: int, seed: int = 7, G: float = G_REFERENCE):
    p = np.empty((n, 6), dtype=np.float64)
    m = np.empty(n, dtype=np.float64)
    check(lib().nbody_mi355x_ic_uniform_cube(n, seed, G, p.ctypes.data_as(_P), m.ctypes.data_as(_P)))
    return p, m


class PinnedArray:
    """A page-locked (n, cols) host array from nbody_mi355x_host_alloc; .array is the numpy view.
    Free it with close() (or use it as a context manager) once no copy into it is pending."""

    def __init__(self, shape, dtype=np.float64):
        self._p = C.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        check(lib().nbody_mi355x_host_alloc(nbytes, C.byref(self._p)))
        raw = (C.c_char * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(raw, dtype=dtype).reshape(shape)

    def close(self) -> None:
        if self._p:
            p, self._p = self._p, None
            self.array = None
            check(lib().nbody_mi355x_host_free(p))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def ic_ring(n: int, seed: int = 42, central_mass: float = 5.683e26, ring_radius: float = 1.12e8,
            width_sigma: float = 3.0e7, thickness_sigma: float = 50.0, v_sigma: float = 50.0,
            mass_sigma: float = 0.1):
    """saturn_ring.cc's scene for any n (defaults: the reference's parameters)."""
    p = np.empty((n, 6), dtype=np.float64)
    m = np.empty(n, dtype=np.float64)
    check(lib().nbody_mi355x_ic_ring(n, seed, central_mass, ring_radius, width_sigma, thickness_sigma, v_sigma,
                                     mass_sigma, p.ctypes.data_as(_P), m.ctypes.data_as(_P)))
    return p, m


# scenes with one random stream per body: made on the device (Simulation.generate) or by the
# same code on the host's threads (scene_host)
SCENE_PLUMMER, SCENE_UNIFORM_CUBE, SCENE_RING = 0, 1, 2


class Scene(C.Structure):
    _fields_ = [("scene", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64), ("num_particles", C.c_int64),
                ("G", C.c_double), ("central_mass", C.c_double), ("ring_radius", C.c_double),
                ("width_sigma", C.c_double), ("thickness_sigma", C.c_double), ("v_sigma", C.c_double),
                ("mass_sigma", C.c_double)]


def make_scene(scene: int, n: int, seed: int = 42, G: float = G_REFERENCE, central_mass: float = 5.683e26,
               ring_radius: float = 1.12e8, width_sigma: float = 3.0e7, thickness_sigma: float = 50.0,
               v_sigma: float = 50.0, mass_sigma: float = 0.1) -> Scene:
    """A scene record (ring defaults: the reference's saturn_ring.cc parameters)."""
    return Scene(scene, 0, seed, n, G, central_mass, ring_radius, width_sigma, thickness_sigma, v_sigma, mass_sigma)


def scene_host(scene: Scene, first: int = 0, count: int | None = None):
    """Bodies [first, first + count) of the scene on the host's threads: (count,6) {r,v}, masses.
    (A Plummer sphere is NOT moved into its centre-of-mass frame here: that needs every body.)"""
    count = scene.num_particles - first if count is None else count
    p = np.empty((count, 6), dtype=np.float64)
    m = np.empty(count, dtype=np.float64)
    check(lib().nbody_mi355x_scene_host(C.byref(scene), first, count, p.ctypes.data_as(_P), m.ctypes.data_as(_P)))
    return p, m


# ---------------------------------------------------------------- handle API
class Simulation:
    """The handle API, as ParticleSimulation::SimulateMI355X() uses it:
    create -> upload -> run (one call per big step) -> destroy.
    devices = [d0, d1, ...]: one shard of bodies per entry, driven by this one host thread
    (nbody_mi355x_create_multi; an ordinal may repeat: several shards on one GPU)."""

    def __init__(self, num_particles: int, stepsize: float, *, G: float = G_REFERENCE,
                 eps2: float = 0.0, precision: int = FP64, device: int = 0,
                 host_is_float: bool = False, variant: int = 0, integrator: int = EULER, devices=None):
        self.n = int(num_particles)
        self.host_dtype = np.float32 if host_is_float else np.float64
        cfg = Config(self.n, stepsize, G, eps2, precision, device, int(host_is_float), variant, integrator, 0)
        h = C.c_void_p()
        if devices is None:
            check(lib().nbody_mi355x_create(C.byref(cfg), C.byref(h)))
        else:
            devs = (C.c_int32 * len(devices))(*devices)
            check(lib().nbody_mi355x_create_multi(C.byref(cfg), devs, len(devices), C.byref(h)))
        self._h = h

    def upload(self, particles, mass) -> None:
        p, pp = _np(np.asarray(particles).reshape(self.n, 6), self.host_dtype)
        m, mp = _np(np.asarray(mass).reshape(self.n), self.host_dtype)
        check(lib().nbody_mi355x_upload(self._h, pp, mp))

    def generate(self, scene: Scene) -> None:
        """The scene made on the device(s), straight into the handle's state, instead of upload()."""
        check(lib().nbody_mi355x_generate(self._h, C.byref(scene)))

    def run(self, substeps: int, snapshot: bool = True):
        """One big step.  Returns DeviceOutputParticle[n] as (n,9) {r,v,a*G} or None."""
        out = np.empty((self.n, 9), dtype=self.host_dtype) if snapshot else None
        check(lib().nbody_mi355x_run(self._h, substeps, out.ctypes.data_as(_P) if snapshot else None))
        return out

    def run_async(self, substeps: int, out=None):
        """Enqueue one big step; `out` ((n,9) array of the host dtype, ideally a PinnedArray's) receives the snapshot.  wait() completes it."""
        if out is not None:
            assert out.shape == (self.n, 9) and out.dtype == self.host_dtype and out.flags.c_contiguous
        check(lib().nbody_mi355x_run_async(self._h, substeps, out.ctypes.data_as(_P) if out is not None else None))

    def wait(self) -> None:
        check(lib().nbody_mi355x_wait(self._h))

    def download(self):
        p = np.empty((self.n, 6), dtype=self.host_dtype)
        check(lib().nbody_mi355x_download(self._h, p.ctypes.data_as(_P)))
        return p

    def energy(self):
        ke, pe = C.c_double(), C.c_double()
        check(lib().nbody_mi355x_energy(self._h, C.byref(ke), C.byref(pe)))
        return ke.value, pe.value

    def kernel_in_use(self):
        """(variant, workspace_bytes): the kernel build a small step runs, the pair-symmetric workspace held."""
        v, b = C.c_int32(), C.c_int64()
        check(lib().nbody_mi355x_kernel_in_use(self._h, C.byref(v), C.byref(b)))
        return v.value, b.value

    def last_run_seconds(self) -> float:
        s = C.c_double()
        check(lib().nbody_mi355x_last_run_seconds(self._h, C.byref(s)))
        return s.value

    def last_exchange_seconds(self):
        """(positions, partial_sums): device seconds of the two exchanges of the last small step of the last run."""
        a, b = C.c_double(), C.c_double()
        check(lib().nbody_mi355x_last_exchange_seconds(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exchange_backend(self) -> int:
        """EXCHANGE_NONE (one device), EXCHANGE_PEER_COPIES or EXCHANGE_RCCL."""
        v = C.c_int32()
        check(lib().nbody_mi355x_exchange_backend(self._h, C.byref(v)))
        return v.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            check(lib().nbody_mi355x_destroy(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
