#!/bin/bash
# tools/diag_mid.sh [sizes] -- where the workgroups of the wave-tile launches ran and where a wave's
# time goes (NBODY_MI355X_SYMW_TRACE; tuning only: allocates, blocks, a few steps per size)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZES=${1:-8192,16384,32768,65536}
cd $R
rm -f gpurun_out/wg_trace.txt
NBODY_MI355X_SYMW_TRACE=$R/gpurun_out/wg_trace.txt SIZES=$SIZES timeout -k 10 200 python3 - <<'PY'
import os, sys
sys.path.insert(0, 'n-body-sim_amd')
import nbody_mi355x as nb
for n in [int(x) for x in os.environ["SIZES"].split(",")]:
    p, m = nb.ic_plummer(n, 42)
    with nb.Simulation(n, 1e-3) as sim:
        sim.upload(p, m)
        sim.run(30, snapshot=False)
    print("traced", n, flush=True)
PY
python3 tools/analyze_wg_trace.py gpurun_out/wg_trace.txt | awk '/^launch/ {n++} { if ((n % 30) == 0 || (n % 30) == 29) print }'
