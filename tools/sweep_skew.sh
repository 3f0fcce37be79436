#!/bin/bash
# tools/sweep_skew.sh [sizes] [percents] -- wave tiles, uneven dealing (set_octets, nbody_sym.hip): the
# step time by the later waves' share (NBODY_MI355X_SYMW_SKEW, percent of an older wave's; 0 = even
# dealing by the fitted chunk rule) on ONE GPU lease.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZES=${1:-10240,12288,16384,20480,24576,32768,49152,65536,98304,131072}
for pct in ${2:-0 15 20 25 30 40}; do
  echo "== skew $pct"
  NBODY_MI355X_SYMW_SKEW=$pct timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 $SWEEP_ARGS || exit 1
done
