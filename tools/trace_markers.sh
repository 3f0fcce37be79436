#!/bin/bash
# tools/trace_markers.sh -- rocprofv3 kernel + marker (roctx) traces of the two multi-GPU drivers on
# one GPU: bench.py --force-collectives (one rank through RCCL) and the C library's multi-shard
# handle (4 shards on the one GPU).  The named ranges show what is enqueued around which kernels.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03_markers
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $OUT/bench -- python3 $R/bench.py --gpus 1 --force-collectives --no-verify --bodies 131072 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $OUT/multi -- $R/n-body-sim_amd/host/nbody_host_cli bench plummer 131072 mi355x-shards4 3 2 > $OUT/multi.log 2>&1 || exit 2
find $OUT -name "*marker*" -o -name "*kernel_trace*" | head
