import sys, time
sys.path.insert(0, "n-body-sim_amd")
import torch
import nbody_mi355x as nb
for n in (1 << 20, 1 << 22):
    with nb.Simulation(n, 1e-3) as sim:
        for kind in (0, 1, 2):
            sc = nb.make_scene(kind, n, seed=1)
            sim.generate(sc)
            t = time.time(); sim.generate(sc); dt = time.time() - t
            t = time.time(); nb.scene_host(sc); th = time.time() - t
            print(f"n={n} scene={kind} device {dt*1e3:.2f} ms  host threads {th*1e3:.1f} ms", flush=True)
    t = time.time(); nb.ic_plummer(n, 1); print(f"n={n} serial ic_plummer {time.time()-t:.2f} s", flush=True)
