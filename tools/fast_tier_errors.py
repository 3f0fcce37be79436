#!/usr/bin/env python3
"""How far the opt-in FP64_FAST tier is from the oracle, beside the default FP64 path: sampled-row
accelerations at N = 1M and 262 144 (one step), positions and energy drift at 65 536 after 100 steps.
    python tools/fast_tier_errors.py > profiles/r03_fast_tier_errors.txt"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))
sys.path.insert(0, str(REPO / "tests"))
import nbody_mi355x as nb  # noqa: E402
import oracle_lib as ol  # noqa: E402  (the checker)


def vec_rel(a, b):
    return float((np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max())


for n in (262144, 1048576):
    p, m = nb.ic_plummer(n, 42)
    rows = np.random.default_rng(0).choice(n, 1024, replace=False)
    acc = ol.accel_rows(p, m, rows, threads=16)
    for name, prec in (("FP64", nb.FP64), ("FP64_FAST", nb.FP64_FAST)):
        with nb.Simulation(n, 1e-3, precision=prec) as sim:
            sim.upload(p, m)
            out = sim.run(1, snapshot=True)
        d = out[rows, 6:9] - acc
        rel = np.linalg.norm(d, axis=1) / np.linalg.norm(acc, axis=1)
        # signed error along the acceleration: the truncated term is one-signed (r^-3 too small)
        along = (d * acc).sum(1) / (acc * acc).sum(1)
        print(f"N={n:8d} {name:9s} sampled rows vs oracle: max rel {rel.max():.3e}  median {np.median(rel):.3e}  "
              f"mean signed error along a {along.mean():+.3e}", flush=True)
n = 65536
p, m = nb.ic_plummer(n, 42)
_, ref = ol.simulate(p, m, 1, 100, 1e-3, threads=16, kind="avx2")
e0 = sum(ol.energy(p, m, threads=16))
dref = sum(ol.energy(ref[0], m, threads=16)) - e0
for name, prec in (("FP64", nb.FP64), ("FP64_FAST", nb.FP64_FAST)):
    with nb.Simulation(n, 1e-3, precision=prec) as sim:
        sim.upload(p, m)
        out = sim.run(100, snapshot=True)
    perr = float((np.linalg.norm(out[:, :3] - ref[0, :, :3], axis=1) / np.linalg.norm(ref[0, :, :3], axis=1)).max())
    drift = sum(ol.energy(out, m, threads=16)) - e0
    print(f"N={n:8d} {name:9s} 100 steps vs oracle (AVX2 path): max rel position error {perr:.3e}  accelerations {vec_rel(out[:, 6:9], ref[0, :, 6:9]):.3e}  "
          f"energy drift / reference's {drift / dref:.9f}", flush=True)
