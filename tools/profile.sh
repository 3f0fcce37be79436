#!/bin/bash
# tools/profile.sh TAG [bench args...] -- rocprofv3 passes for bench.py on the GPU box.
# One pass per counter group (rocprofv3 does not split counters over passes); the
# kernel-trace/stats pass is separate from every --pmc pass.  Writes under gpurun_out/TAG/.
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || exit 4
echo "profile passes done: $OUT"
