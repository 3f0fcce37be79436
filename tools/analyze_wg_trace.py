#!/usr/bin/env python3
"""Where and when the workgroups of a wave-tile launch ran (NBODY_MI355X_SYMW_TRACE=FILE, written by
csrc/nbody_symw.hip in tuning runs): per launch, how many workgroups each CU received, their
durations, the launch's span and how level it was.
    python tools/analyze_wg_trace.py FILE [launch]"""
import collections
import sys

rows = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) not in (7, 13, 14):
        continue
    launch, row, chunk, t0, t1, hw, xcc = (int(x) for x in f[:7])
    extra = [int(x) for x in f[7:]]
    # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    cu = (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    rows[launch].append((row, chunk, t0, t1, cu, extra))
only = int(sys.argv[2]) if len(sys.argv) > 2 else None
for launch, ws in sorted(rows.items()):
    if only is not None and launch != only:
        continue
    t_first = min(w[2] for w in ws)
    t_last = max(w[3] for w in ws)
    dur = sorted((w[3] - w[2]) / 100.0 for w in ws)          # 100 MHz ticks -> microseconds
    per_cu = collections.Counter(w[4] for w in ws)
    busy = collections.defaultdict(float)
    for w in ws:
        busy[w[4]] += (w[3] - w[2]) / 100.0
    hist = collections.Counter(per_cu.values())
    ends = sorted((w[3] - t_first) / 100.0 for w in ws)
    starts = sorted((w[2] - t_first) / 100.0 for w in ws)
    print(f"launch {launch}: {len(ws)} workgroups on {len(per_cu)} CUs; span {(t_last - t_first) / 100.0:.1f} us; "
          f"workgroup duration min {dur[0]:.1f} median {dur[len(dur) // 2]:.1f} max {dur[-1]:.1f} us")
    print(f"   workgroups per CU: " + ", ".join(f"{k} on {v} CUs" for k, v in sorted(hist.items())))
    print(f"   starts: first 0.0, median {starts[len(starts) // 2]:.1f}, last {starts[-1]:.1f} us;  "
          f"ends: first {ends[0]:.1f}, median {ends[len(ends) // 2]:.1f}, last {ends[-1]:.1f} us")
    xcd = collections.Counter(w[4][0] for w in ws)
    print("   workgroups per XCD: " + " ".join(f"{k}:{v}" for k, v in sorted(xcd.items())))
    ck = sorted(w[5][6] / ((w[3] - w[2]) / 100.0) / 1e3 for w in ws if len(w[5]) > 6 and w[3] > w[2])
    if ck:   # s_memtime ticks per microsecond of the 100 MHz clock
        print(f"   shader clock while the workgroups ran: min {ck[0]:.2f} median {ck[len(ck) // 2]:.2f} max {ck[-1]:.2f} GHz")
    ex = [w for w in ws if w[5] and w[5][5] > 0]
    if ex:   # where wave 0's time goes (microseconds, medians over the workgroups)
        med = lambda v: sorted(v)[len(v) // 2] / 100.0
        print(f"   wave 0: prologue (row in registers) {med([w[5][0] - w[2] for w in ex]):.2f} us; per entry: staged after "
              f"{med([w[5][1] / w[5][5] for w in ex]):.2f}, walk {med([w[5][2] / w[5][5] for w in ex]):.2f}, j sums stored "
              f"{med([w[5][3] / w[5][5] for w in ex]):.2f} us ({med([100.0 * w[5][5] for w in ex]):.0f} entries or parts); "
              f"epilogue (i sums through LDS, R row) {med([w[3] - w[5][4] for w in ex]):.2f} us")
