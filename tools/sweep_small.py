#!/usr/bin/env python3
"""tools/sweep_small.py -- microseconds per small step of the handle API (graph-replayed where it
applies) for small and mid-size systems, per kernel variant: where do the one-sided wave-per-body
kernel (7; its shapes JPL x IPW through NBODY_MI355X_WAVE_SHAPE), the 16-lanes-per-body kernel (6),
and the pair-symmetric kernels (10) cross?"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "n-body-sim_amd"))
import nbody_mi355x as nb  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [500, 1024, 1500, 2048, 3000, 4096, 6000, 8192, 10000, 12288, 16384]
cols = [(0, None), (7, 81), (7, 41), (7, 42), (7, 44), (7, 82), (7, 84), (6, None), (10, None)]
print("bodies   " + "".join(f"{('v' + str(v) + ('/' + str(sh) if sh else '')):>10s}" for v, sh in cols))
for n in sizes:
    p, m = nb.ic_plummer(n, 42)
    row = []
    for v, shape in cols:
        if (v == 6 and n > 4096) or (v == 10 and n < 512):
            row.append(float("nan"))
            continue
        if shape:
            os.environ["NBODY_MI355X_WAVE_SHAPE"] = str(shape)
        else:
            os.environ.pop("NBODY_MI355X_WAVE_SHAPE", None)
        sub = max(200, min(4000, int(4e9 / (n * n))))
        try:
            with nb.Simulation(n, 1e-3, variant=v) as sim:
                sim.upload(p, m)
                sim.run(sub, snapshot=False)
                sim.run(sub, snapshot=False)
                row.append(sim.last_run_seconds() / sub * 1e6)
        except Exception:
            row.append(float("nan"))
    print(f"{n:7d}  " + "".join(f"{t:10.2f}" for t in row), flush=True)
