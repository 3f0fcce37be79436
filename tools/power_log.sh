#!/bin/bash
# tools/power_log.sh [bench args] -- what the chip draws and clocks while the headline step runs: rocm-smi sampled
# twice a second beside `bench.py --steps 20` (no other configs, no CPU baseline).  Evidence for DESIGN.md's
# statement that the step kernels run at the power limit.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/power_log.txt
: > $OUT
rocm-smi --showmaxpower --showsclkrange >> $OUT 2>&1 || echo "(rocm-smi limits query failed)" >> $OUT
( for i in $(seq 1 60); do echo "== sample $i $(date +%s.%N)"; rocm-smi --showmetrics 2>&1 | grep -E "current_socket_power|current_gfxclks|temperature_hotspot|XCP\[0\].*below_host_limit" ; sleep 0.5; done ) >> $OUT 2>&1 &
SAMPLER=$!
sleep 2
timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/power_bench.json 2> $R/gpurun_out/power_bench.err
RC=$?
sleep 2
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
exit $RC
