#!/usr/bin/env python3
"""The data behind the launch model of the wave tiles (csrc/nbody_sym.hip: plan_wave): microseconds per
small step for every (bodies, block = 64 * ipt, workgroups per row) of a grid, equal masses, FP64.
(Round 2 swept entries per workgroup, NBODY_MI355X_SYMW_PER: profiles/r02_wave_tile_sweep.csv; since
round 3 a row's list is dealt in octets and the knob is the number of workgroups per row.)
    python tools/sweep_wave_plan.py [sizes] > profiles/r03_wave_tile_sweep.csv      (SWEEP_FP32=1: the FP32 tier)"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))
import nbody_mi355x as nb  # noqa: E402

SIZES = [5120, 6144, 7168, 8192, 9216, 10240, 12288, 14336, 16384, 18432, 20480, 24576, 28672, 32768, 36864,
         40960, 45056, 49152, 57344, 65536, 81920, 98304, 131072]
CHUNKS = [1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64]
if len(sys.argv) > 1:
    SIZES = [int(x) for x in sys.argv[1].split(",")]
print("bodies,ipt,chunks,us_per_step")
for n in SIZES:
    p, m = nb.ic_plummer(n, 42)
    steps = max(3, min(300, int(1.5e12 / (float(n) * n))))
    for ipt in (2, 4, 8):
        for per in CHUNKS:
            os.environ["NBODY_MI355X_SYMW_CHUNKS"] = str(per)
            try:
                with nb.Simulation(n, 1e-3, variant=20 + ipt, precision=nb.FP32 if os.environ.get("SWEEP_FP32") else nb.FP64) as sim:
                    sim.upload(p, m)
                    sim.run(steps, snapshot=False)
                    t = []
                    for _ in range(3):
                        sim.run(steps, snapshot=False)
                        t.append(sim.last_run_seconds())
                print(f"{n},{ipt},{per},{sorted(t)[1] / steps * 1e6:.2f}", flush=True)
            except nb.NBodyError as e:
                print(f"# {n},{ipt},{per}: {e}", flush=True)
