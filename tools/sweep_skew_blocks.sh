#!/bin/bash
# tools/sweep_skew_blocks.sh -- 256- against 512-body wave tiles (variants 24 / 28) at the small mid sizes by
# the least share of an older wave from which the dealing is uneven (NBODY_MI355X_SYMW_SKEW_MIN), one lease.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZES=${1:-8192,9216,10240,11264,12288,13312,14336,15360,16384,18432}
for min in ${2:-12 16 20 24 1000}; do
  echo "== min $min"
  NBODY_MI355X_SYMW_SKEW_MIN=$min timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 --variants 24,28 || exit 1
done
