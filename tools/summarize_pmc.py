#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel averages of every counter,
kernel-trace stats, and derived figures for the step kernel.  Usage:
    python tools/summarize_pmc.py gpurun_out/TAG N_BODIES [KERNEL_SUBSTRING [STEPS]] > profiles/NAME.txt
STEPS = small steps the profiled command ran (bench.py: --steps + --warmup): a step may take
several launches of one kernel (distance batches), so the per-interaction figures are taken per
STEP (sum over the step's launches), not per launch."""
import collections
import csv
import glob
import sys

d, n = sys.argv[1], float(sys.argv[2])
sel = sys.argv[3] if len(sys.argv) > 3 else "step"
steps = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
for f in glob.glob(f"{d}/trace/vm/*_kernel_stats.csv"):
    print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
    print(open(f).read())
c = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/pmc_*/vm/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        c[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
print("== PMC counters, average per launch (separate --pmc passes) ==")
for k, cs in c.items():
    if sel not in k:
        continue
    print(k)
    for name, v in sorted(cs.items()):
        print(f"  {name:24s} {sum(v)/len(v):.6e}   (n={len(v)})")
    t = sum(dur[k]) / len(dur[k])
    g = {a: sum(v) / len(v) for a, v in cs.items()}
    pairs = n * n / 64.0
    print(f"  avg duration under PMC      {t:.4f} s per launch")
    if steps > 0:
        calls = max(len(v) for v in cs.values())
        per_step = calls / steps
        pairs = pairs / per_step     # (the figures below: per 64 interactions of a whole STEP)
        print(f"  launches per step           {per_step:g}   ({t * per_step:.4f} s per step)")
    if "SQ_INSTS_VALU" in g:
        print(f"  VALU wave-instr per 64 interactions   {g['SQ_INSTS_VALU']/pairs:.2f}")
    if "SQ_ACTIVE_INST_VALU" in g:
        cyc = 4 * g["SQ_ACTIVE_INST_VALU"]
        print(f"  VALU-active cycles per 64 interactions {cyc/pairs:.1f}  (SQ_ACTIVE_INST_VALU counts quad-cycles)")
        if "GRBM_GUI_ACTIVE" in g:
            clk = g["GRBM_GUI_ACTIVE"] / 8
            print(f"  shader cycles per launch (GRBM_GUI_ACTIVE/8 XCDs) {clk:.4e}  -> effective clock {clk/t/1e9:.2f} GHz")
            print(f"  VALU busy fraction per SIMD  {cyc/1024/clk:.3f}")
    if "FETCH_SIZE" in g:
        print(f"  FETCH_SIZE {g['FETCH_SIZE']*1024:.4e} B raw, x2 gfx950 read correction = {2*g['FETCH_SIZE']*1024:.4e} B")
    if "WRITE_SIZE" in g:
        print(f"  WRITE_SIZE {g['WRITE_SIZE']*1024:.4e} B")
