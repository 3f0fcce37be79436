#!/bin/bash
# tools/ab_mid.sh [sizes] -- the default path at the mid sizes with two builds of the library on ONE
# GPU lease: n-body-sim_amd/lib/libnbody_mi355x_prev.so (an earlier commit, built by hand) and the
# current one, alternating.  Box-to-box spread is 3-5 %: A/B numbers must come from one lease.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZES=${1:-8192,12288,16384,20480,24576,28672,32768,40960,49152,65536,98304,131072}
for rep in 1 2; do
  echo "== previous build (pass $rep)"
  NBODY_MI355X_LIB=$R/n-body-sim_amd/lib/libnbody_mi355x_prev.so timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 || exit 1
  echo "== current build (pass $rep)"
  timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 || exit 1
done
