"""ctypes access to the CHECKERS: oracle/libnbody_oracle.so (the CPU restatement) and,
where it has been built, oracle/_ref/libnbody_ref.so (the real reference header).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
ORACLE_SO = REPO / "oracle" / "libnbody_oracle.so"
REF_SO = REPO / "oracle" / "_ref" / "libnbody_ref.so"

_P = C.c_void_p
_oracle = None
_ref = None


def oracle() -> C.CDLL:
    global _oracle
    if _oracle is None:
        if not ORACLE_SO.exists():
            raise FileNotFoundError(f"{ORACLE_SO} missing: run `make -C oracle`")
        L = C.CDLL(str(ORACLE_SO))
        sim = [C.c_long, C.c_long, C.c_long, C.c_double, C.c_int, _P, _P, _P]
        L.nbody_oracle_simulate_f64.argtypes = sim
        L.nbody_oracle_simulate_avx2_f64.argtypes = sim
        L.nbody_oracle_simulate_f32.argtypes = [C.c_long, C.c_long, C.c_long, C.c_float, C.c_int, _P, _P, _P]
        L.nbody_oracle_small_step_f64.argtypes = [C.c_long, _P, _P, _P, C.c_double, C.c_long, C.c_long, _P]
        L.nbody_oracle_accel_rows_f64.argtypes = [C.c_long, _P, _P, _P, C.c_long, C.c_int, _P]
        L.nbody_oracle_energy_f64.argtypes = [C.c_long, _P, C.c_long, _P, C.c_int,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _oracle = L
    return _oracle


def have_ref() -> bool:
    return REF_SO.exists()


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        L = C.CDLL(str(REF_SO))
        L.nbody_ref_simulate_f64.argtypes = [C.c_int, C.c_int, C.c_long, C.c_long, C.c_double, C.c_int,
                                             _P, _P, _P, C.POINTER(C.c_double)]
        L.nbody_ref_simulate_f32.argtypes = [C.c_int, C.c_long, C.c_long, C.c_float, C.c_int,
                                             _P, _P, _P, C.POINTER(C.c_double)]
        L.nbody_ref_scene_earth_moon.argtypes = [_P, _P]
        L.nbody_ref_scene_earth_7moons.argtypes = [_P, _P]
        L.nbody_ref_scene_saturn_ring.argtypes = [C.c_int, _P, _P]
        _ref = L
    return _ref


def _ptr(a):
    return a.ctypes.data_as(_P)


# ----------------------------------------------------------------------- oracle
def simulate(particles, mass, big_steps, substeps, dt, threads=1, kind="f64"):
    """SimulateCPU()/SimulateAVX2() restated.  Returns (final (n,6), out (big,n,9))."""
    dtype = np.float32 if kind == "f32" else np.float64
    p = np.array(particles, dtype=dtype, order="C").reshape(-1, 6).copy()
    m = np.ascontiguousarray(mass, dtype=dtype)
    n = p.shape[0]
    out = np.zeros((big_steps, n, 9), dtype=dtype)
    fn = {"f64": oracle().nbody_oracle_simulate_f64, "avx2": oracle().nbody_oracle_simulate_avx2_f64,
          "f32": oracle().nbody_oracle_simulate_f32}[kind]
    rc = fn(n, big_steps, substeps, dt, threads, _ptr(p), _ptr(m), _ptr(out))
    if rc != 0:
        raise RuntimeError(f"oracle simulate rc={rc}")
    return p, out


def small_step(cur, mass, dt, i_begin, i_count, want_out=False):
    """One reference worker thread's share of one small step.  Returns next (n,6)
    with only rows [i_begin, i_begin+i_count) written (others zero) and out9 or None."""
    cur = np.ascontiguousarray(cur, dtype=np.float64).reshape(-1, 6)
    m = np.ascontiguousarray(mass, dtype=np.float64)
    n = cur.shape[0]
    nxt = np.zeros_like(cur)
    out = np.zeros((n, 9)) if want_out else None
    rc = oracle().nbody_oracle_small_step_f64(n, _ptr(cur), _ptr(nxt), _ptr(m), dt, i_begin, i_count,
                                              _ptr(out) if want_out else None)
    if rc != 0:
        raise RuntimeError(f"oracle small_step rc={rc}")
    return nxt, out


def accel_rows(particles, mass, rows, threads=8):
    p = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 6)
    m = np.ascontiguousarray(mass, dtype=np.float64)
    r = np.ascontiguousarray(rows, dtype=np.int64)
    acc = np.zeros((len(r), 3))
    oracle().nbody_oracle_accel_rows_f64(p.shape[0], _ptr(p), _ptr(m), _ptr(r), len(r), threads, _ptr(acc))
    return acc


def energy(state, mass, threads=8):
    """state: (n,6) Particle rows or (n,9) DeviceOutputParticle rows."""
    s = np.ascontiguousarray(state, dtype=np.float64)
    m = np.ascontiguousarray(mass, dtype=np.float64)
    ke, pe = C.c_double(), C.c_double()
    oracle().nbody_oracle_energy_f64(s.shape[0], _ptr(s), s.shape[1], _ptr(m), threads, C.byref(ke), C.byref(pe))
    return ke.value, pe.value


# -------------------------------------------------------------------- reference
def ref_simulate(particles, mass, big_steps, substeps, dt, threads=1, mode="cpu", f32=False):
    """The REAL reference.  Returns (out11 (big+1,n,11), seconds)."""
    dtype = np.float32 if f32 else np.float64
    p = np.ascontiguousarray(particles, dtype=dtype).reshape(-1, 6)
    m = np.ascontiguousarray(mass, dtype=dtype)
    n = p.shape[0]
    out = np.zeros((big_steps + 1, n, 11), dtype=dtype)
    sec = C.c_double()
    if f32:
        rc = ref().nbody_ref_simulate_f32(n, big_steps, substeps, dt, threads, _ptr(p), _ptr(m), _ptr(out), C.byref(sec))
    else:
        rc = ref().nbody_ref_simulate_f64(1 if mode == "avx2" else 0, n, big_steps, substeps, dt, threads,
                                          _ptr(p), _ptr(m), _ptr(out), C.byref(sec))
    if rc != 0:
        raise RuntimeError(f"reference simulate rc={rc}")
    return out, sec.value


def ref_scene(name, n=500):
    size = {"earth_moon": 2, "earth_7moons": 8, "saturn_ring": n}[name]
    p = np.zeros((size, 6))
    m = np.zeros(size)
    if name == "earth_moon":
        ref().nbody_ref_scene_earth_moon(_ptr(p), _ptr(m))
    elif name == "earth_7moons":
        ref().nbody_ref_scene_earth_7moons(_ptr(p), _ptr(m))
    else:
        ref().nbody_ref_scene_saturn_ring(size, _ptr(p), _ptr(m))
    return p, m
