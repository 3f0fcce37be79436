#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs oracle/_ref/libnbody_ref.so, i.e.
/root/reference compiled by `make -C oracle ref`).  Each .npz holds inputs
(Particle rows (n,6), masses, dt, step counts) and the reference's own
out_put_data_ (rows {r,v,a,m,t}) -- data only, no reference source.

    python tests/golden/make_golden.py

The reference has no tests or golden files of its own, so
these files are the pin for oracle/nbody_oracle.c.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "n-body-sim_amd"))

import oracle_lib as ol  # noqa: E402
import nbody_mi355x as nb  # noqa: E402


def save(name, p, m, big, sub, dt, **outs):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, particles=p, mass=m, big_steps=np.int64(big), substeps=np.int64(sub),
                        dt=np.float64(dt), **outs)
    print(f"{path.name}: n={max(p.shape[0], next(iter(outs.values())).shape[1])} big={big} sub={sub} dt={dt} -> {path.stat().st_size/1024:.0f} KiB")


def main():
    if not ol.have_ref():
        sys.exit("oracle/_ref/libnbody_ref.so missing: run `make -C oracle ref` in the build container")

    # G1 earth-moon (earth_moon.cc:13-38): dt = 0.1 s; 2 big steps x 500 substeps
    p, m = ol.ref_scene("earth_moon")
    out, _ = ol.ref_simulate(p, m, 2, 500, 0.1, threads=1, mode="cpu")
    save("g1_earth_moon", p, m, 2, 500, 0.1, out_cpu=out)

    # G2 earth + 7 moons (earth_7moons.cc:15-50): dt = 0.05 s, 1000 substeps; CPU and AVX2
    p, m = ol.ref_scene("earth_7moons")
    out, _ = ol.ref_simulate(p, m, 1, 1000, 0.05, threads=1, mode="cpu")
    out2, _ = ol.ref_simulate(p, m, 1, 1000, 0.05, threads=2, mode="avx2")
    save("g2_earth_7moons", p, m, 1, 1000, 0.05, out_cpu=out, out_avx2=out2)

    # G3 saturn ring (saturn_ring.cc:15-63): N=500, seed 42, dt = 0.01 s, 100 substeps
    p, m = ol.ref_scene("saturn_ring", 500)
    out, _ = ol.ref_simulate(p, m, 1, 100, 0.01, threads=4, mode="cpu")
    out2, _ = ol.ref_simulate(p, m, 1, 100, 0.01, threads=4, mode="avx2")
    save("g3_saturn_ring_500", p, m, 1, 100, 0.01, out_cpu=out, out_avx2=out2)

    # G4 Plummer (library generator, seed 42), dt = 1e-3 N-body time units, 100 substeps
    for n in (1024, 4096):
        p, m = nb.ic_plummer(n, 42)
        out, _ = ol.ref_simulate(p, m, 1, 100, 1e-3, threads=8, mode="cpu")
        out2, _ = ol.ref_simulate(p, m, 1, 100, 1e-3, threads=8, mode="avx2")
        # positions/velocities/accelerations only: m and t are trivial here
        # (snapshot 0 is the ICs again and is dropped for the large scenes)
        extra = {"out_avx2": out2[1:, :, :9]} if n == 1024 else {}
        save(f"g4_plummer_{n}", p, m, 1, 100, 1e-3, out_cpu=out[1:, :, :9], **extra)

    # G5 one force evaluation, N=4096 (pure force check: out a = G * sum); the ICs are
    # those of g4_plummer_4096.npz and are not stored twice
    p, m = nb.ic_plummer(4096, 42)
    out, _ = ol.ref_simulate(p, m, 1, 1, 1e-3, threads=8, mode="cpu")
    save("g5_plummer_4096_force", p[:0], m[:0], 1, 1, 1e-3, out_cpu=out[1:, :, :9])

    # G6 float instantiation ParticleSimulation<float,8>::SimulateCPU on the Plummer 1024 scene
    p, m = nb.ic_plummer(1024, 42)
    pf, mf = p.astype(np.float32), m.astype(np.float32)
    out, _ = ol.ref_simulate(pf, mf, 1, 100, 1e-3, threads=8, f32=True)
    save("g6_plummer_1024_f32", pf, mf, 1, 100, 1e-3, out_cpu=out)

    # G7 edge cases, CPU path (N % 4 != 0 would corrupt the reference's AVX2 path):
    # two coincident bodies (r^2 == 0 skip, particle_simulation.h:269), a massless body,
    # a body at the origin, N = 7.
    p = np.array([[0, 0, 0, 0, 0, 0],
                  [1, 0, 0, 0, 1, 0],
                  [1, 0, 0, 0, -1, 0],      # coincident with body 1 at t = 0
                  [0, 2, 0, -0.5, 0, 0],
                  [0, 0, -3, 0, 0.3, 0],
                  [-1.5, -1.5, 0.5, 0.1, 0.1, 0.1],
                  [4, 4, 4, 0, 0, 0]], dtype=np.float64)
    m = np.array([1.0, 1e-3, 2e-3, 0.0, 5e-4, 1e-3, 1e-6]) / nb.G_REFERENCE
    out, _ = ol.ref_simulate(p, m, 3, 50, 1e-3, threads=3, mode="cpu")
    save("g7_edge_cases", p, m, 3, 50, 1e-3, out_cpu=out)

    # G8 strongly coupled SI-unit cloud (SURVEY Appendix A note 1): masses 1e24..1e26 kg
    # in a 1e8 m box, dt = 1 s, 100 substeps; exercises large magnitudes.
    rng = np.random.default_rng(1234)
    n = 512
    p = np.concatenate([rng.uniform(-1e8, 1e8, (n, 3)), rng.normal(0, 1e3, (n, 3))], axis=1)
    m = 10.0 ** rng.uniform(24, 26, n)
    out, _ = ol.ref_simulate(p, m, 1, 100, 1.0, threads=8, mode="cpu")
    out2, _ = ol.ref_simulate(p, m, 1, 100, 1.0, threads=8, mode="avx2")
    save("g8_si_cloud_512", p, m, 1, 100, 1.0, out_cpu=out[1:, :, :9], out_avx2=out2[1:, :, :9])


if __name__ == "__main__":
    main()
