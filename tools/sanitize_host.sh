#!/bin/bash
# tools/sanitize_host.sh -- AddressSanitizer + UBSan build of the HOST side (g++ only; the
# reference has no sanitizer runs) and a pass over every CPU path of the
# host mirror: ragged N, float objects, AVX2 / AVX-512, scene set-up, text writers.
# GPU code is not sanitized (not available on this pool).  Exits non-zero on any report.
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
g++ -std=c++17 -Wall -Wextra -O1 -g -mavx2 -mfma -pthread -fsanitize=address,undefined -fno-sanitize-recover=all \
    -fno-omit-frame-pointer -I"$R/include" -I"$R/n-body-sim_amd/host" "$R/n-body-sim_amd/host/nbody_host_cli.cc" \
    -o "$T/cli" -L"$R/n-body-sim_amd/lib" -lnbody_mi355x -Wl,-rpath,"$R/n-body-sim_amd/lib" -Wl,-rpath,/opt/rocm/lib
export ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0
"$T/cli" scene saturn_ring "$T/ring.bin"
"$T/cli" scene earth_7moons "$T/e7.bin"
for path in cpu avx2 avx512 cpu-f32 avx512-f32; do
  "$T/cli" run $path "$T/ring.bin" 2 7 0.01 5 "$T/out.bin" 2>&1 | tail -1
  "$T/cli" run $path "$T/e7.bin" 2 50 0.1 3 "$T/out.bin" 2>&1 | tail -1
done
"$T/cli" bench plummer 1023 avx2 3 2 3 > /dev/null
"$T/cli" bench ring 1001 avx512 3 2 3 > /dev/null
"$T/cli" bench cube 777 cpu 3 2 2 > /dev/null
"$T/cli" demo earth_moon cpu 2 1000 "$T/em_" > /dev/null
"$T/cli" demo saturn_ring avx2 2 1000 "$T/sr_" > /dev/null
rm -rf "$T"
echo "host paths clean under ASan + UBSan"
