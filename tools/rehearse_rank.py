#!/usr/bin/env python3
"""tools/rehearse_rank.py -- what ONE rank of an N-rank run computes, timed on one GPU.

    python tools/rehearse_rank.py --world 8 --rank 3 [--bodies 1048576] [--steps 5]

The rank's kernels run exactly as under bench.py --gpus N (same slab layout, same launches,
phases 1 + 2, gather, integrate); the two collectives are replaced by what they would leave
in this rank's buffers if every other rank stood still (remote slabs keep their initial
positions; the reduce-scatter keeps only this rank's own partial sums).  The trajectory is
therefore NOT the real one -- this measures the per-rank compute time, i.e. the part of the
strong-scaling efficiency that does not depend on xGMI: efficiency_bound = T(1) / (world * T).
"""
import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "n-body-sim_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--bodies", type=int, default=1048576)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", choices=["fp64", "fp32"], default="fp64")
    ap.add_argument("--ic", choices=["plummer", "cube"], default="plummer")
    ap.add_argument("--no-overlap", action="store_true")
    args = ap.parse_args()
    import torch
    import nbody_mi355x as nb
    from nbody_mi355x.sharded import HipBackend, ShardedNBody

    prec = nb.FP64 if args.precision == "fp64" else nb.FP32
    p, m = nb.ic_plummer(args.bodies, 42) if args.ic == "plummer" else nb.ic_uniform_cube(args.bodies, 7)
    sim = ShardedNBody(p, m, 1e-3 if args.ic == "plummer" else 1e-4, rank=args.rank, world=args.world, device="cuda:0",
                       backend=HipBackend(precision=prec), overlap=not args.no_overlap,
                       force_collectives=True)

    class _Done:
        def wait(self):
            return True

    def fake_all_gather(buf, async_op):
        # remote slabs of `buf` <- their (stale) content in the other buffer: same bytes moved
        other = sim.pos[0] if buf is sim.pos[1] else sim.pos[1]
        for r in range(sim.world):
            if r != sim.rank:
                buf[r].copy_(other[r])
        return _Done()

    def fake_reduce_scatter():
        sim.acc.copy_(sim.acc_all[sim.rank])

    sim._all_gather = fake_all_gather
    sim._reduce_scatter = fake_reduce_scatter
    for _ in range(args.warmup):
        sim.step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        sim.step()
    ev1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    dev = ev0.elapsed_time(ev1) * 1e-3 / args.steps
    n = float(args.bodies)
    print(json.dumps({"world": args.world, "rank": args.rank, "bodies": args.bodies, "precision": args.precision,
                      "ms_per_step_rank": round(dev * 1e3, 3), "wall_ms": round(wall * 1e3, 3),
                      "Ginteractions_per_s_if_all_ranks_like_this": round(n * n / dev / 1e9, 1),
                      "sym": sim.workspace is not None}))


if __name__ == "__main__":
    main()
