#!/bin/bash
# tools/ab_env.sh VAR SIZES [sweep_mid args] -- tools/sweep_mid.py with VAR unset and VAR=1, alternating, on one lease
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
VAR=$1; SIZES=$2; shift 2
for rep in 1 2; do
  echo "== $VAR unset (pass $rep)"; env -u $VAR timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 "$@" || exit 1
  echo "== $VAR=1 (pass $rep)"; env $VAR=1 timeout -k 10 300 python3 $R/tools/sweep_mid.py --sizes $SIZES --reps 3 "$@" || exit 1
done
