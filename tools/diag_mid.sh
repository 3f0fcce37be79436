#!/bin/bash
# tools/diag_mid.sh -- per-kernel times (rocprofv3 kernel trace) of a few mid-size launch shapes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r3_g
rm -rf $OUT; mkdir -p $OUT
for case in "8192 24 8" "8192 24 16" "8192 28 16" "16384 24 8" "16384 24 12" "16384 24 33" "16384 28 16" "16384 28 8" "32768 28 16"; do
  set -- $case
  export NBODY_MI355X_SYMW_CHUNKS=$3
  tag=n$1_v$2_c$3
  timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/tools/mid_case.py $1 $2 > $OUT/$tag.log 2>&1 || exit 1
  echo "== $tag: $(grep us/step $OUT/$tag.log)"
  python3 $R/tools/trace_gaps.py $OUT/$tag 100
done
