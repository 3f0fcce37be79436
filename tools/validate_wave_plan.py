#!/usr/bin/env python3
"""The library's own choice of block size and chunking for the wave tiles (csrc/nbody_sym.hip:
plan_wave) against six pinned shapes (256- and 512-body blocks x 4 / 8 / 16 entries per workgroup)
over 39 sizes: microseconds per small step, and how far the choice is from the best of them.
    python -u tools/validate_wave_plan.py > profiles/r02_wave_tile_validation.txt   (from the repo root)"""
import os, sys
sys.path.insert(0, 'n-body-sim_amd')
import nbody_mi355x as nb
SIZES = [4352, 5120, 6144, 7168, 8192, 9216, 10240, 11264, 12288, 13312, 14336, 16384, 18432, 20480, 22528, 24576, 26624, 28672,
         30720, 32768, 34816, 36864, 40960, 45056, 49152, 53248, 57344, 61440, 65536, 73728, 81920, 90112, 98304, 114688,
         131072, 163840, 196608, 229376, 260096]
def run(n, p, m, variant, per):
    if per: os.environ["NBODY_MI355X_SYMW_PER"] = str(per)
    else: os.environ.pop("NBODY_MI355X_SYMW_PER", None)
    steps = max(3, min(300, int(1.5e12 / (float(n) * n))))
    with nb.Simulation(n, 1e-3, variant=variant) as sim:
        sim.upload(p, m)
        used, ws = sim.kernel_in_use()
        sim.run(steps, snapshot=False)
        t = []
        for _ in range(3):
            sim.run(steps, snapshot=False); t.append(sim.last_run_seconds())
    return sorted(t)[1] / steps * 1e6, used, ws
for n in SIZES:
    p, m = nb.ic_plummer(n, 42)
    auto, used, ws = run(n, p, m, 0, 0)
    refs = {}
    for v in (24, 28):
        for per in (4, 8, 16):
            try: refs[(v, per)] = run(n, p, m, v, per)[0]
            except nb.NBodyError: pass
    b = min(refs.values())
    print(f"n={n:7d} auto->{used} ws={ws/1e6:7.1f}MB {auto:9.1f} us {n*n/auto/1e3:7.0f} Gint/s | best ref {b:9.1f} {min(refs, key=refs.get)} | auto vs best {100*(auto/b-1):+5.1f}%", flush=True)
