"""One-rank-per-GPU driver of the hot path: i-bodies sharded, positions all-gathered.

The reference parallelises MakeSmallStep over contiguous i ranges, remainder to
the low ids (particle_simulation.h:98-110), with all positions in shared
memory.  Across GPUs the same split is kept -- rank r owns the bodies thread r
would own -- and the shared memory becomes one exchange per small step: every
rank all-gathers the positions it has just updated (velocities stay private;
no reduction is needed because each rank produces complete a_i).

Layout: the slab layout of include/nbody_mi355x.h, [world][3][S] with
S = max shard size rounded up to the 256-body tile (1024 for large systems), so that
the exchange is ONE in-place all_gather_into_tensor of 3*S doubles per rank.

Large systems use the pair-symmetric kernels: a rank evaluates each unordered pair of
(own row, block within half the ring ahead) once, which leaves it with partial sums for
bodies of OTHER ranks; those go home in one reduce_scatter_tensor (3*N doubles in, 3*S
out) before the Euler update.  Two collectives per step, both tiny next to the tiles.

Overlap: while the collective for small step k is in flight (RCCL runs it on
its own stream), the rank already accumulates the forces from its OWN slab,
which it holds (step_args.acc_out, first half of a split step); after
work.wait() a second launch adds the remote slabs and integrates.

Integrators: the reference's forward Euler (default), the symplectic Euler step, and the
kick-drift-kick leapfrog of include/nbody_mi355x.h (KDK): a small step is then a drift of
the own slab (v += dt/2 a, r += dt v), the exchange, and the force evaluation at the new
positions whose epilogue is the second kick -- still one force evaluation and the same two
collectives per small step.

torch is plumbing here (device memory, streams, torch.distributed); the
arithmetic is in libnbody_mi355x.so.  `backend` exists so that the CPU test of
this host logic (gloo, world_size 2) can substitute a checker for the kernel;
the default is the HIP library and there is no fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import (EULER, FP64, G_REFERENCE, KDK, KICK_DRIFT, SCENE_PLUMMER, SYM_VARIANTS, TILE, VARIANT_SYM, Scene, StepArgs,
               check, lib, round_up)

SYM_MIN_BODIES = 2048  # as kSymMinBodies in csrc/nbody_kernels.hip


class _Range:
    """A named range in a rocprofv3 trace (--marker-trace): torch.cuda.nvtx is roctx on ROCm
    builds.  The reference's only timing hooks are wall-clock timers around a big step
    (particle_simulation.h:203-207); these make a multi-GPU trace readable -- does the
    all-gather really sit under phase 1?  No-ops off the GPU."""
    enabled = True

    def __init__(self, name, on):
        self.name, self.on = name, on and _Range.enabled

    def __enter__(self):
        if self.on:
            try:
                torch.cuda.nvtx.range_push(self.name)
            except Exception:  # noqa: BLE001  (a torch build without the marker library)
                self.on = _Range.enabled = False
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.nvtx.range_pop()
        return False


# slots per block of a variant that pins one (block_of_variant, csrc/nbody_internal.h)
BLOCK_OF_VARIANT = {11: 256, 12: 512, 14: 1024, 18: 2048, 21: 64, 22: 128, 24: 256, 28: 512}


def slab_slots(bodies: int, variant, shards: int = 1) -> int:
    """Slots of a slab holding `bodies`, one of `shards`: slab_slots() of csrc/nbody_internal.h -- whole
    tiles and whole blocks of the pair-symmetric kernels the library would choose for the WHOLE system
    or the variant pins."""
    shards = max(1, shards)
    unit = 512 if bodies * shards > 12800 else TILE
    if round_up(bodies, unit) * shards > 262143:   # beyond the wave tiles: whole 2048-slot blocks
        unit = 2048
    unit = max(unit, BLOCK_OF_VARIANT.get(variant, 0))
    return round_up(bodies, unit)


def shard_counts(n: int, world: int):
    """particle_simulation.h:98-110: n // world each, remainder to the low ranks."""
    base, rem = divmod(n, world)
    counts = [base + (1 if r < rem else 0) for r in range(world)]
    begins = [base * r + min(r, rem) for r in range(world)]
    return counts, begins


class HipBackend:
    """Calls nbody_mi355x_step_f64 on torch-owned device memory."""

    def __init__(self, precision=FP64, variant=0):
        self.precision, self.variant = precision, variant
        self._lib = lib()

    def workspace_bytes(self, S, world):
        """Scratch the pair-symmetric kernels want for this layout (0: not applicable)."""
        a = StepArgs()
        a.shard_slots, a.num_shards, a.variant, a.precision = S, world, self.variant, self.precision
        need = C.c_int64(0)
        check(self._lib.nbody_mi355x_step_workspace_bytes(C.byref(a), C.byref(need)))
        return need.value

    def workspace_min_bytes(self, S, world):
        """The smallest workspace they accept (the distances are then walked in more batches)."""
        a = StepArgs()
        a.shard_slots, a.num_shards, a.variant, a.precision = S, world, self.variant, self.precision
        least = C.c_int64(0)
        check(self._lib.nbody_mi355x_step_workspace_min_bytes(C.byref(a), C.byref(least)))
        return least.value

    def tile_launches(self, S, world, count, n, workspace, variant=None):
        """Launches of the force kernel per whole step for this layout and workspace (distance batches)."""
        a = StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel = 8, 16, 8, 8     # (validated, never dereferenced)
        a.acc_out = 8 if world > 1 else None
        a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, world, 0, count
        a.j0_end = n if world == 1 else 0
        a.precision, a.variant = self.precision, self.variant if variant is None else variant
        if workspace is not None:
            a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        out = C.c_int32(0)
        check(self._lib.nbody_mi355x_step_tile_launches(C.byref(a), C.byref(out)))
        return out.value

    def step(self, *, pos_cur, pos_next, mass, vel, acc_in, acc_out, snapshot, S, world, rank, count,
             ranges, dt, G, eps2, workspace=None, phase=0, variant=None, integrator=EULER, kdk_acc=None):
        a = StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel = pos_cur.data_ptr(), pos_next.data_ptr(), mass.data_ptr(), vel.data_ptr()
        a.acc_in = acc_in.data_ptr() if acc_in is not None else None
        a.acc_out = acc_out.data_ptr() if acc_out is not None else None
        a.snapshot = snapshot.data_ptr() if snapshot is not None else None
        a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, world, rank, count
        (a.j0_begin, a.j0_end), (a.j1_begin, a.j1_end) = ranges
        a.dt, a.G, a.eps2 = dt, G, eps2
        a.precision, a.variant = self.precision, self.variant if variant is None else variant
        a.phase = phase
        a.integrator = integrator
        a.kdk_acc = kdk_acc.data_ptr() if kdk_acc is not None else None
        if workspace is not None:
            a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        stream = torch.cuda.current_stream(pos_cur.device).cuda_stream
        check(self._lib.nbody_mi355x_step_f64(C.byref(a), C.c_void_p(stream)))

    def scene_fill(self, scene, first, count, pos_slab, vel, mass_slab, S):
        """Bodies [first, first + count) of the scene into one slab (nbody_mi355x_scene_fill_f64)."""
        stream = torch.cuda.current_stream(pos_slab.device).cuda_stream
        check(self._lib.nbody_mi355x_scene_fill_f64(C.byref(scene), first, count, C.c_void_p(pos_slab.data_ptr()),
                                                    C.c_void_p(vel.data_ptr()) if vel is not None else None,
                                                    C.c_void_p(mass_slab.data_ptr()), S, C.c_void_p(stream)))

    def slab_sums(self, pos_slab, vel, count, S, sums6):
        stream = torch.cuda.current_stream(pos_slab.device).cuda_stream
        check(self._lib.nbody_mi355x_slab_sums_f64(C.c_void_p(pos_slab.data_ptr()), C.c_void_p(vel.data_ptr()), count, S,
                                                   C.c_void_p(sums6.data_ptr()), C.c_void_p(stream)))

    def slab_shift(self, pos_slab, vel, count, S, shift6):
        stream = torch.cuda.current_stream(pos_slab.device).cuda_stream
        check(self._lib.nbody_mi355x_slab_shift_f64(C.c_void_p(pos_slab.data_ptr()),
                                                    C.c_void_p(vel.data_ptr()) if vel is not None else None, count, S,
                                                    (C.c_double * 6)(*shift6), C.c_void_p(stream)))

    def kdk_drift(self, *, pos_cur, pos_next, vel, kdk_acc, S, world, rank, count, dt, G):
        """First half of a leapfrog small step for the own slab (nbody_mi355x_kdk_drift_f64)."""
        a = StepArgs()
        a.pos_cur, a.pos_next, a.vel, a.kdk_acc = pos_cur.data_ptr(), pos_next.data_ptr(), vel.data_ptr(), kdk_acc.data_ptr()
        a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, world, rank, count
        a.dt, a.G = dt, G
        stream = torch.cuda.current_stream(pos_cur.device).cuda_stream
        check(self._lib.nbody_mi355x_kdk_drift_f64(C.byref(a), C.c_void_p(stream)))


class ShardedNBody:
    def __init__(self, particles, mass, dt, *, rank=0, world=1, device="cuda", G=G_REFERENCE, eps2=0.0,
                 backend=None, overlap=True, group=None, use_workspace=True, force_collectives=False,
                 integrator=EULER):
        # `particles`: an (n, 6) array {r, v} with `mass` (n,), uploaded; or a Scene record
        # (mass = None): every rank then makes the bodies it holds on its own device
        scene = particles if isinstance(particles, Scene) else None
        if scene is None:
            particles = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 6)
            mass = np.ascontiguousarray(mass, dtype=np.float64)
        self.n, self.rank, self.world = (scene.num_particles if scene is not None else particles.shape[0]), rank, world
        self.dt, self.G, self.eps2 = float(dt), float(G), float(eps2)
        self.overlap, self.group = overlap, group
        if integrator not in (EULER, KICK_DRIFT, KDK):
            raise ValueError(f"integrator {integrator}: EULER, KICK_DRIFT or KDK")
        self.integrator = integrator
        # rehearsal: take the multi-rank code path (collectives, split steps, partial-sum
        # exchange) even with one rank, so that RCCL itself can be exercised on a one-GPU box
        self.multi = world > 1 or force_collectives
        self.device = torch.device(device)
        self.backend = backend if backend is not None else HipBackend()
        self.counts, self.begins = shard_counts(self.n, world)
        self.count = self.counts[rank]
        # large systems: whole 1024- or 2048-body blocks, so that the pair-symmetric kernels apply
        # with their largest block
        big = max(self.counts)
        variant = getattr(self.backend, "variant", None)
        self.S = S = slab_slots(big, variant, world)

        dev = self.device
        if scene is None:
            # host image of the slab layout; padding slots: mass 0 at the origin
            pos = np.zeros((world, 3, S))
            m = np.zeros((world, S))
            for r in range(world):
                b, c = self.begins[r], self.counts[r]
                pos[r, :, :c] = particles[b:b + c, :3].T
                m[r, :c] = mass[b:b + c]
            vel = np.zeros((3, S))
            b, c = self.begins[rank], self.count
            vel[:, :c] = particles[b:b + c, 3:].T
            self.pos = [torch.from_numpy(pos).to(dev), torch.from_numpy(pos.copy()).to(dev)]
            self.mass = torch.from_numpy(m).to(dev)
            self.vel = torch.from_numpy(vel).to(dev)
        else:
            self._generate(scene)
        self.acc = torch.zeros((3, S), dtype=torch.float64, device=dev)
        self.snap = torch.zeros((max(self.count, 1), 9), dtype=torch.float64, device=dev)
        # pair-symmetric kernels: used when the backend offers them, the
        # system is large enough and their workspace fits; otherwise the one-sided kernel
        self.workspace = None
        self.acc_all = None
        if (use_workspace and hasattr(self.backend, "workspace_bytes") and self.device.type == "cuda"
                and (variant in SYM_VARIANTS or (variant == 0 and self.n >= SYM_MIN_BODIES))):
            # the preferred size if it is under half of the free memory, else what is under
            # half of it down to the kernels' minimum (as nbody_mi355x_create does)
            need = self.backend.workspace_bytes(S, world)
            free_b, _ = torch.cuda.mem_get_info(self.device)
            budget = free_b // 2 - (3 * S * world * 8 if self.multi else 0)
            if need > budget and need > 0:
                least = self.backend.workspace_min_bytes(S, world)
                need = budget if 0 < least <= budget else 0
            if need > 0:
                self.workspace = torch.empty(need // 8, dtype=torch.float64, device=dev)
                if self.multi:
                    self.acc_all = torch.zeros((world, 3, S), dtype=torch.float64, device=dev)
        self.sym_variant = variant if variant in SYM_VARIANTS else VARIANT_SYM
        # leapfrog: the raw sums of the last force evaluation, own slab (the next drift's kick)
        self.kacc = torch.zeros((3, S), dtype=torch.float64, device=dev) if integrator == KDK else None
        self._kdk_primed = False
        self._send = self._recv = None
        self.cur = 0
        self._fresh = True  # every slab of pos[cur] is current (true right after upload)

    def _generate(self, scene):
        """The scene made on this rank's device: every slab's positions and masses (O(N) per
        rank, no exchange), the own slab's velocities.  A Plummer sphere goes into its
        centre-of-mass frame: own-slab sums, one all-reduce of 6 doubles, a shift."""
        dev, S, P = self.device, self.S, self.world
        self.pos = [torch.zeros((P, 3, S), dtype=torch.float64, device=dev)]
        self.mass = torch.zeros((P, S), dtype=torch.float64, device=dev)
        self.vel = torch.zeros((3, S), dtype=torch.float64, device=dev)
        for r in range(P):
            self.backend.scene_fill(scene, self.begins[r], self.counts[r], self.pos[0][r],
                                    self.vel if r == self.rank else None, self.mass[r], S)
        if scene.scene == SCENE_PLUMMER:
            sums = torch.zeros(6, dtype=torch.float64, device=dev)
            self.backend.slab_sums(self.pos[0][self.rank], self.vel, self.count, S, sums)
            if P > 1:
                import torch.distributed as dist
                if dist.get_backend(self.group) == "nccl":
                    dist.all_reduce(sums, group=self.group)
                else:                      # gloo rehearsal: staged through the host
                    host = sums.cpu()
                    dist.all_reduce(host, group=self.group)
                    sums = host
            shift = (sums.cpu().numpy() / self.n).tolist()
            for r in range(P):
                self.backend.slab_shift(self.pos[0][r], self.vel if r == self.rank else None, self.counts[r], S, shift)
        self.pos.append(self.pos[0].clone())

    def _range(self, name):
        return _Range(name, self.device.type == "cuda")

    # ------------------------------------------------------------------ one small step
    def step(self, snapshot=False):
        with self._range("nbody small step"):
            self._step(snapshot)

    def _step(self, snapshot):
        snap = self.snap if snapshot else None
        if self.integrator != KDK:
            self._evaluate(snap, self.dt)       # forces at pos[cur]; the update writes pos[cur ^ 1]
            self.cur ^= 1
            self._fresh = False
            return
        if not self._kdk_primed:                # a(r0): a KDK evaluation with dt = 0 moves nothing
            self._evaluate(None, 0.0)
            self._kdk_primed = True
        cur, nxt = self.pos[self.cur], self.pos[self.cur ^ 1]
        self.backend.kdk_drift(pos_cur=cur, pos_next=nxt, vel=self.vel, kdk_acc=self.kacc, S=self.S, world=self.world,
                               rank=self.rank, count=self.count, dt=self.dt, G=self.G)
        self.cur ^= 1
        self._fresh = False
        self._evaluate(snap, self.dt)           # forces at the new positions + the second kick; nothing moves

    def _evaluate(self, snap, dt):
        """One force evaluation at pos[cur] and the integrator's epilogue (exchange included)."""
        cur, nxt = self.pos[self.cur], self.pos[self.cur ^ 1]
        S, P, r = self.S, self.world, self.rank
        common = dict(pos_cur=cur, pos_next=nxt, mass=self.mass, vel=self.vel, S=S, world=P, rank=r,
                      count=self.count, dt=dt, G=self.G, eps2=self.eps2)
        if self.integrator != EULER:            # (a backend without integrators keeps its signature)
            common.update(integrator=self.integrator, kdk_acc=self.kacc)
        if not self.multi:
            extra = {"workspace": self.workspace} if self.workspace is not None else {}
            self.backend.step(acc_in=None, acc_out=None, snapshot=snap, ranges=((0, self.n), (0, 0)), **common, **extra)
        elif self.workspace is not None:
            self._sym_forces(cur, common)
            with self._range("reduce-scatter of partial sums"):
                self._reduce_scatter()
            with self._range("integrate"):
                self._integrate(common, snap)
        elif not self.overlap or self._fresh:
            if not self._fresh:
                self._all_gather(cur, async_op=False)
            self.backend.step(acc_in=None, acc_out=None, snapshot=snap, ranges=((0, P * S), (0, 0)), **common)
        else:
            with self._range("all-gather of positions (issue)"):
                work = self._all_gather(cur, async_op=True)       # remote slabs, in flight
            with self._range("phase 1: own slab"):
                self.backend.step(acc_in=None, acc_out=self.acc, snapshot=None,
                                  ranges=((r * S, (r + 1) * S), (0, 0)), **common)   # own slab meanwhile
            with self._range("all-gather of positions (wait)"):
                work.wait()                                       # stream-ordered, not a host block
            with self._range("phase 2: remote slabs"):
                self.backend.step(acc_in=self.acc, acc_out=None, snapshot=snap,
                                  ranges=((0, r * S), ((r + 1) * S, P * S)), **common)

    # -- pair-symmetric step over several ranks, in its three parts (the one-GPU loop-back
    # test drives them separately): partial sums for EVERY slot -> reduce-scatter -> integrate
    def _sym_forces(self, cur, common):
        kw = dict(acc_in=None, snapshot=None, ranges=((0, 0), (0, 0)), workspace=self.workspace,
                  variant=self.sym_variant, **common)
        if self.overlap and not self._fresh:
            with self._range("all-gather of positions (issue)"):
                work = self._all_gather(cur, async_op=True)             # remote slabs, in flight
            with self._range("phase 1: diagonal tiles"):
                self.backend.step(acc_out=None, phase=1, **kw)          # diagonal tiles: own slab only
            with self._range("all-gather of positions (wait)"):
                work.wait()
            with self._range("phase 2: all other tiles"):
                self.backend.step(acc_out=self.acc_all, phase=2, **kw)  # all other tiles, then gather
        else:
            if not self._fresh:
                with self._range("all-gather of positions (blocking)"):
                    self._all_gather(cur, async_op=False)
            with self._range("phase 0: all tiles"):
                self.backend.step(acc_out=self.acc_all, phase=0, **kw)

    def _reduce_scatter(self):
        import torch.distributed as dist
        if dist.get_backend(self.group) == "nccl":   # RCCL
            dist.reduce_scatter_tensor(self.acc.view(-1), self.acc_all.view(-1), group=self.group)
        else:  # gloo (rehearsals without one GPU per rank): no reduce_scatter; staged via the host
            host = self.acc_all.cpu()
            dist.all_reduce(host, group=self.group)
            self.acc.copy_(host[self.rank])

    def _integrate(self, common, snap):
        # a one-sided step with no j range: acc_in -> the integrator's update (+ snapshot)
        self.backend.step(acc_in=self.acc, acc_out=None, snapshot=snap, ranges=((0, 0), (0, 0)), variant=1, **common)

    def _all_gather(self, buf, async_op):
        import torch.distributed as dist
        flat = buf.view(-1)
        n3 = 3 * self.S
        own = flat[self.rank * n3:(self.rank + 1) * n3]
        if dist.get_backend(self.group) == "nccl" or buf.device.type == "cpu":
            # ONE collective: every rank's slab (3*S doubles).  Neither side of it aliases
            # memory a kernel is using: the send side is a private copy of the own slab and the
            # receive side a staging tensor -- the own-slab kernels (phase 1) read `buf` while
            # the collective is in flight on RCCL's stream; the REMOTE slabs are copied into
            # their places (two device copies, microseconds) after the wait, in stream order.
            if self._send is None:
                self._send = torch.empty_like(own)
                self._recv = torch.empty_like(flat)
            self._send.copy_(own)
            work = dist.all_gather_into_tensor(self._recv, self._send, group=self.group, async_op=async_op)
            lo, hi, recv = self.rank * n3, (self.rank + 1) * n3, self._recv

            class _Placed:
                def wait(self_inner):
                    if work is not None:
                        work.wait()              # stream-ordered, not a host block
                    if lo > 0:
                        flat[:lo].copy_(recv[:lo])
                    if hi < flat.numel():
                        flat[hi:].copy_(recv[hi:])
                    return True
            placed = _Placed()
            if not async_op:
                placed.wait()
            return placed
        # gloo with device tensors (rehearsals): staged via the host, out of place
        mine = own.cpu()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        for r, part in enumerate(parts):
            if r != self.rank:
                flat[r * n3:(r + 1) * n3].copy_(part)

        class _Done:
            def wait(self):
                return True
        return _Done()

    def warm_collectives(self):
        """Run each collective of a small step once, on the real buffers, without changing the
        state: the all-gather re-sends the own slab (already in place everywhere), the
        reduce-scatter lands in the scratch the next step overwrites.  RCCL sets a collective
        up on its first call (channels, kernels: milliseconds); a driver that times steps calls
        this before its warm-up steps so that no timed step holds a first call."""
        if not self.multi or self.device.type != "cuda":
            return
        self._all_gather(self.pos[self.cur], async_op=False)
        if self.acc_all is not None:
            self._reduce_scatter()
        torch.cuda.synchronize(self.device)

    def time_collectives(self, reps=5):
        """Milliseconds of each collective of a small step on its own (mean of `reps` blocking
        calls after one untimed call; the state is not changed, as in warm_collectives).  HIP
        events on the launch stream, which waits for the collective's own stream; a gloo
        rehearsal is staged through the host and is timed with the host's clock instead.
        None when this object runs no collective."""
        if not self.multi or self.device.type != "cuda":
            return None
        import time
        import torch.distributed as dist
        events = dist.get_backend(self.group) == "nccl"

        def timed(call):
            call()
            torch.cuda.synchronize(self.device)
            if events:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize(self.device)
                return e0.elapsed_time(e1) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / reps * 1e3

        out = {"all_gather": timed(lambda: self._all_gather(self.pos[self.cur], async_op=False)),
               "reduce_scatter": timed(self._reduce_scatter) if self.acc_all is not None else None,
               "timer": "hip_events_on_launch_stream" if events else "host_clock_host_staged",
               "reps": reps, "all_gather_bytes_per_rank": 3 * self.S * 8,
               "reduce_scatter_bytes_in_per_rank": 3 * self.S * 8 * self.world if self.acc_all is not None else 0}
        return out

    def all_masses(self):
        """(n,) masses of every body, from the device's copy."""
        m = self.mass.cpu().numpy()
        return np.concatenate([m[r, :self.counts[r]] for r in range(self.world)])

    def sync_positions(self):
        """Make every slab of the current buffer current (needed before reading remote slabs)."""
        if self.multi and not self._fresh:
            self._all_gather(self.pos[self.cur], async_op=False)
            self._fresh = True

    # ------------------------------------------------------------------ results
    def own_state(self):
        """Particle rows (count, 6) of this rank's bodies."""
        c = self.count
        pos = self.pos[self.cur][self.rank, :, :c].T.cpu().numpy()
        vel = self.vel[:, :c].T.cpu().numpy()
        return np.concatenate([pos, vel], axis=1)

    def own_snapshot(self):
        """DeviceOutputParticle rows (count, 9) written by the last step(snapshot=True)."""
        return self.snap[:self.count].cpu().numpy()

    def all_positions(self):
        """(n, 3) positions of every body (collective)."""
        self.sync_positions()
        p = self.pos[self.cur].cpu().numpy()
        return np.concatenate([p[r, :, :self.counts[r]].T for r in range(self.world)], axis=0)
