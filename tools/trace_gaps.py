#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive kernels from a rocprofv3 kernel trace.
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [last N rows]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70],
                         r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
if tail:
    rows = rows[-tail:]
dur = defaultdict(list)
gap = defaultdict(list)
for k, (s, e, name, gx, gy, gz) in enumerate(rows):
    key = f"{name} grid=({gx},{gy},{gz})"
    dur[key].append(e - s)
    if k:
        gap[key].append(s - rows[k - 1][1])
for key in dur:
    v = sorted(dur[key])
    g = sorted(gap[key]) if gap[key] else [0]
    print(f"{len(v):6d} x {key}\n         median {v[len(v) // 2] / 1e3:9.2f} us  min {v[0] / 1e3:9.2f}  max {v[-1] / 1e3:9.2f}   "
          f"gap before: median {g[len(g) // 2] / 1e3:7.2f} us")
