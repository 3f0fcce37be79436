#!/usr/bin/env python3
"""Ginteractions/s of the default path over the mid sizes (handle API, device time of the kernels).
    python tools/sweep_mid.py [--sizes 8192,16384,...] [--precision fp64|fp32] [--general] [--variant V]"""
import argparse
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))
import numpy as np  # noqa: E402
import nbody_mi355x as nb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="8192,16384,32768,65536,131072,262144")
ap.add_argument("--precision", default="fp64")
ap.add_argument("--general", action="store_true", help="mixed masses (the general form of the tile loop)")
ap.add_argument("--variants", default="0", help="comma-separated kernel variants (0 = the library's choice)")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
prec = nb.FP64 if args.precision == "fp64" else nb.FP32
for n, variant in [(int(x), int(v)) for x in args.sizes.split(",") for v in args.variants.split(",")]:
    p, m = nb.ic_plummer(n, 42)
    if args.general:
        m = m * np.random.default_rng(1).uniform(0.5, 2.0, n)
    steps = max(2, min(200, int(3e12 / (float(n) * n))))
    try:
        with nb.Simulation(n, 1e-3, precision=prec, variant=variant) as sim:
            sim.upload(p, m)
            used, ws = sim.kernel_in_use()
            sim.run(steps, snapshot=False)
            t = []
            for _ in range(args.reps):
                sim.run(steps, snapshot=False)
                t.append(sim.last_run_seconds())
    except nb.NBodyError as e:
        print(f"n={n:8d} variant={variant:3d}  not applicable: {e}", flush=True)
        continue
    t.sort()
    med = t[len(t) // 2]
    g = float(n) * n * steps / med / 1e9
    peak = 78.6 * (2 if prec == nb.FP32 else 1)
    print(f"n={n:8d} variant={variant:3d}->{used:2d} ws={ws / 1e6:7.1f}MB steps={steps:4d} median {med / steps * 1e6:10.1f} us/step  {g:8.1f} Gint/s  "
          f"{100 * g * 20 / 1e3 / peak:5.1f} % of peak   (min {t[0] / steps * 1e6:.1f} max {t[-1] / steps * 1e6:.1f})", flush=True)
