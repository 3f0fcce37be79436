#!/usr/bin/env python3
"""One mid-size case for the profiler: N bodies, a pinned kernel variant, STEPS small steps through the
handle API (tuning knobs come from the environment: NBODY_MI355X_SYMW_CHUNKS, NBODY_MI355X_SYM_PARTS).
    python tools/mid_case.py N VARIANT [STEPS]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "n-body-sim_amd"))
import nbody_mi355x as nb  # noqa: E402

n, variant = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
p, m = nb.ic_plummer(n, 42)
with nb.Simulation(n, 1e-3, variant=variant) as sim:
    sim.upload(p, m)
    sim.run(steps, snapshot=False)
    sim.run(steps, snapshot=False)
    print(f"n={n} variant={variant}->{sim.kernel_in_use()[0]} {sim.last_run_seconds() / steps * 1e6:.2f} us/step")
