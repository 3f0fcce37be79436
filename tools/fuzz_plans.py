#!/usr/bin/env python3
"""tools/fuzz_plans.py [cases] [seed] -- random systems through the library's own choice of kernel and
launch plan against the oracle's exact sums on sampled rows: sizes between the wave-per-body kernel's
limit and 150 000 bodies (every block size, even and uneven dealing, padded last blocks, entries cut more
than once), mass patterns (one mass / mixed / two species / a few massless test bodies / one heavy body),
FP64 and the FP32 tier, one to three small steps, one GPU or two to four shards of the multi-device
handle on it.  Prints one line per case
and exits non-zero on the first failure."""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))
sys.path.insert(0, str(REPO / "tests"))
import nbody_mi355x as nb   # noqa: E402
import oracle_lib as ol     # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for c in range(cases):
    n = int(np.exp(rng.uniform(np.log(2049), np.log(150000))))
    p, m = nb.ic_plummer(n, 1000 + c)
    kind = int(rng.integers(0, 5))
    if kind == 1:
        m = m * rng.uniform(0.5, 2.0, n)
    elif kind == 2:
        m = m * np.where(rng.random(n) < 0.5, 1.0, 3.0)
    elif kind == 3:
        m = m.copy()
        m[rng.choice(n, max(1, n // 50), replace=False)] = 0.0
    elif kind == 4:
        m = m.copy()
        m[int(rng.integers(0, n))] *= 1.0e4
    fp32 = bool(rng.random() < 0.25)
    steps = int(rng.integers(1, 4))
    _, ref = ol.simulate(p, m, 1, steps, 1e-3, threads=16) if n <= 20000 else (None, None)
    shards = int(rng.integers(2, 5)) if rng.random() < 0.3 else 1      # several shards of the C handle on the one GPU
    kw = {"devices": [0] * shards} if shards > 1 else {}
    with nb.Simulation(n, 1e-3, precision=nb.FP32 if fp32 else nb.FP64, **kw) as sim:
        sim.upload(p, m)
        variant, _ = sim.kernel_in_use()
        out = sim.run(steps if ref is not None else 1, snapshot=True)
    if ref is not None:     # the whole trajectory
        acc, got = ref[0, :, 6:9], out[:, 6:9]
    else:                   # exact sums for sampled rows of the first step
        rows = rng.choice(n, 768, replace=False)
        acc, got = ol.accel_rows(p, m, rows, threads=16), out[rows, 6:9]
    err = float(np.abs(got - acc).max() / np.linalg.norm(acc, axis=1).max())
    tol = 1e-4 if fp32 else 1e-12
    worst = max(worst, err / tol)
    print(f"case {c:3d}: n={n:6d} masses={('one', 'mixed', 'two species', 'some massless', 'one heavy')[kind]:13s} "
          f"{'fp32' if fp32 else 'fp64'} steps={steps} shards={shards} variant={variant:2d} err={err:.2e} (tol {tol:g})", flush=True)
    if not err <= tol:
        sys.exit(f"FAILED: case {c}")
print(f"{cases} cases, worst error {worst:.3f} of its tolerance")
