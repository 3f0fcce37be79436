"""CPU-only checks of the boundary and the host logic: the C-ABI library loads and exports
every symbol include/nbody_mi355x.h declares (no compute calls without a GPU), argument
validation fails loudly, the IC generators are deterministic, and the shard partition
follows the reference's thread partition."""
import re
from pathlib import Path

import numpy as np
import pytest

import nbody_mi355x as nb
from nbody_mi355x.sharded import shard_counts, slab_slots

REPO = Path(__file__).resolve().parent.parent


def test_header_symbols_all_exported_and_bound():
    header = (REPO / "include" / "nbody_mi355x.h").read_text()
    declared = set(re.findall(r"\b(nbody_mi355x_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    L = nb.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert declared == set(nb.SYMBOLS), "python binding and header disagree"


def test_struct_sizes_match_header_layout():
    import ctypes as C
    # 7 pointers + int64 + 2 int32 + 5 int64 + 3 double + 2 int32 + pointer + int64 + 2 int32 + pointer
    assert C.sizeof(nb.StepArgs) == 7 * 8 + 8 + 8 + 5 * 8 + 3 * 8 + 8 + 16 + 8 + 8
    assert C.sizeof(nb.Config) == 8 + 3 * 8 + 6 * 4


def test_invalid_arguments_fail_loudly_without_gpu():
    import ctypes as C
    L = nb.lib()
    a = nb.StepArgs()
    assert L.nbody_mi355x_step_f64(C.byref(a), None) == nb.ERR_INVALID
    assert "null" in nb.last_error()
    a.pos_cur, a.mass = 8, 8  # never dereferenced: validation rejects before any launch
    a.shard_slots = 100  # not a multiple of the tile
    assert L.nbody_mi355x_step_f64(C.byref(a), None) == nb.ERR_INVALID
    assert "multiple of 256" in nb.last_error()
    cfg = nb.Config(0, 0.1, 0, 0, 0, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert L.nbody_mi355x_create(C.byref(cfg), C.byref(h)) == nb.ERR_INVALID
    with pytest.raises(nb.NBodyError):
        nb.ic_plummer(0)


def test_product_package_never_touches_the_oracle():
    for f in (REPO / "n-body-sim_amd").rglob("*"):
        if f.suffix in {".py", ".hip", ".cpp", ".h", ".hpp"} or f.name == "Makefile":
            assert "oracle" not in f.read_text().replace("test_oracle", ""), f"{f} mentions the oracle"


def test_ic_plummer_deterministic_and_sane():
    p, m = nb.ic_plummer(4096, 42)
    p2, m2 = nb.ic_plummer(4096, 42)
    assert np.array_equal(p, p2) and np.array_equal(m, m2)
    p3, _ = nb.ic_plummer(4096, 43)
    assert not np.array_equal(p, p3)
    assert np.allclose(m * nb.G_REFERENCE * 4096, 1.0)
    assert np.abs(p[:, :3].mean(0)).max() < 1e-12 and np.abs(p[:, 3:].mean(0)).max() < 1e-12
    r = np.linalg.norm(p[:, :3], axis=1)
    assert r.max() < 10.5
    assert 1.2 < np.median(r) < 1.4  # Plummer half-mass radius = 1.305 a (slightly less when truncated)
    v2 = (p[:, 3:] ** 2).sum(1).mean()
    assert 0.2 < v2 < 0.4  # <v^2> = 3 pi / 32 = 0.295 in G M = a = 1


def test_ic_uniform_cube():
    p, m = nb.ic_uniform_cube(1000, 7)
    assert np.all(np.abs(p[:, :3]) <= 1) and np.all(p[:, 3:] == 0)
    assert np.allclose(m.sum() * nb.G_REFERENCE, 1.0)


def test_ic_ring_follows_the_reference_recipe():
    """saturn_ring.cc:42-63 at any n from the portable generator: central body at rest at
    the origin, ring bodies near the ring plane on (perturbed) circular orbits about it with
    the initializer's G = 6.67430e-11, radii ~ N(ring_radius, width), masses = N(0, .1)^2."""
    n = 20000
    p, m = nb.ic_ring(n, 42)
    p2, m2 = nb.ic_ring(n, 42)
    assert np.array_equal(p, p2) and np.array_equal(m, m2)
    assert not np.array_equal(p, nb.ic_ring(n, 43)[0])
    assert np.all(p[0] == 0) and m[0] == 5.683e26
    r = np.hypot(p[1:, 0], p[1:, 1])
    assert abs(r.mean() - 1.12e8) < 3 * 3.0e7 / np.sqrt(n) and abs(r.std() / 3.0e7 - 1) < 0.05
    assert abs(p[1:, 2].std() / 50 - 1) < 0.05 and np.all(p[1:, 5] == 0)
    phase = np.arctan2(p[1:, 1], p[1:, 0])
    assert abs(phase.mean()) < 0.1 and abs(phase.std() - np.pi / np.sqrt(3)) < 0.05   # uniform in angle
    # tangential speed = circular speed (+- 50 m/s), radial speed = +- 50 m/s
    vt = (p[1:, 0] * p[1:, 4] - p[1:, 1] * p[1:, 3]) / r
    vr = (p[1:, 0] * p[1:, 3] + p[1:, 1] * p[1:, 4]) / r
    vc = np.sqrt(6.67430e-11 * 5.683e26 / r)
    assert abs((vt - vc).std() / 50 - 1) < 0.05 and abs(vr.std() / 50 - 1) < 0.05
    assert np.all(m[1:] >= 0) and abs(m[1:].mean() / 0.01 - 1) < 0.05   # E[N(0,.1)^2] = .01
    with pytest.raises(nb.NBodyError):
        nb.ic_ring(10, 1, central_mass=-1.0)



@pytest.mark.parametrize("n,world", [(5, 3), (500, 8), (1048576, 8), (7, 8), (4194304, 8)])
def test_shard_partition_matches_reference_threads(n, world):
    counts, begins = shard_counts(n, world)
    # particle_simulation.h:98-110, e.g. 5 particles, 3 threads -> 2,2,1
    assert sum(counts) == n and max(counts) - min(counts) <= 1
    assert counts == sorted(counts, reverse=True)
    assert begins == [sum(counts[:r]) for r in range(world)]
    if (n, world) == (5, 3):
        assert counts == [2, 2, 1]


def test_python_constants_mirror_the_header():
    import re
    text = (REPO / "include" / "nbody_mi355x.h").read_text()
    defs = {k: v for k, v in re.findall(r"#define\s+NBODY_MI355X_(\w+)\s+([-\w.+]+)", text)}
    assert int(defs["WAVE_MAX_BODIES"]) == nb.WAVE_MAX_BODIES
    assert int(defs["VARIANT_WAVE"]) == nb.VARIANT_WAVE and int(defs["VARIANT_TINY"]) == nb.VARIANT_TINY
    assert int(defs["VARIANT_SYM"]) == nb.VARIANT_SYM and int(defs["VARIANT_SYM8"]) == nb.VARIANT_SYM8
    assert int(defs["TILE"]) == nb.TILE and float(defs["G_REFERENCE"]) == nb.G_REFERENCE
    assert [int(defs[k]) for k in ("EULER", "KICK_DRIFT", "KDK", "KDK_CONTINUE")] == [nb.EULER, nb.KICK_DRIFT, nb.KDK, nb.KDK_CONTINUE]
    assert [int(defs[f"VARIANT_SYMW{k}"]) for k in (1, 2, 4, 8)] == [nb.VARIANT_SYMW1, nb.VARIANT_SYMW2, nb.VARIANT_SYMW4, nb.VARIANT_SYMW8]
    assert int(defs["STATIC_SCRATCH_BYTES"]) == nb.STATIC_SCRATCH_BYTES
    assert [int(defs[k]) for k in ("FP64", "FP32", "FP64_FAST")] == [nb.FP64, nb.FP32, nb.FP64_FAST]
    assert [int(defs[f"EXCHANGE_{k}"]) for k in ("NONE", "PEER_COPIES", "RCCL")] == [nb.EXCHANGE_NONE, nb.EXCHANGE_PEER_COPIES, nb.EXCHANGE_RCCL]


def test_step_variant_reports_the_kernel_choice_without_a_gpu():
    """nbody_mi355x_step_variant is the read-only view of the one place the kernel is chosen
    (csrc/nbody_kernels.hip: resolve_variant): no launch, so it runs here.  Without a workspace
    a large system gets a ONE-SIDED build -- that is how a caller sees the fallback."""
    import ctypes as C
    L = nb.lib()

    def variant_of(n, workspace_bytes=0, variant=0, precision=nb.FP64, shards=1):
        a = nb.StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel = 8, 16, 8, 8    # never dereferenced
        S = slab_slots(n, 0)   # csrc/nbody_internal.h and its mirror
        a.shard_slots, a.num_shards, a.own_count, a.j0_end = S, shards, n, n
        a.precision, a.variant = precision, variant
        if workspace_bytes:
            a.workspace, a.workspace_bytes = 8, workspace_bytes
        v = C.c_int32(-1)
        nb.check(L.nbody_mi355x_step_variant(C.byref(a), C.byref(v)))
        return v.value

    big = 1 << 40
    assert variant_of(500) == nb.VARIANT_WAVE and variant_of(nb.WAVE_MAX_BODIES, big) == nb.VARIANT_WAVE
    assert variant_of(100000) in (1, 2) and variant_of(1048576) == 2          # no workspace: one-sided
    assert variant_of(8192, big) == nb.VARIANT_SYMW4 and variant_of(12288, big) == nb.VARIANT_SYMW4
    assert variant_of(16384, big) == nb.VARIANT_SYMW8          # 512-body blocks from 13 312 slots (round 3's sweeps)
    assert variant_of(1048576, big, precision=nb.FP64_FAST) == nb.VARIANT_SYM8   # the tier changes the arithmetic, not the plan
    assert variant_of(65536, big) == nb.VARIANT_SYMW8 and variant_of(1048576, big) == nb.VARIANT_SYM8
    assert variant_of(65536, big, precision=nb.FP32) == nb.VARIANT_SYMW8 and variant_of(1048576, big, precision=nb.FP32) == nb.VARIANT_SYM8
    assert variant_of(65536, 4096) in (1, 2)                                  # workspace below the minimum
    assert variant_of(20000, big, variant=nb.VARIANT_SYM4) == nb.VARIANT_SYM4
    assert variant_of(20000, 0, variant=3) == 3
    a = nb.StepArgs()
    assert L.nbody_mi355x_step_variant(C.byref(a), C.byref(C.c_int32())) == nb.ERR_INVALID


def test_launch_plans_exist_and_are_consistent_for_every_layout():
    """The plan of the pair-symmetric kernels is a pure function of (slots, slabs, variant,
    precision): for every slab size from 2 blocks up, one slab or several, the workspace query
    and the variant query must agree, the preferred workspace must respect the 4 GiB cap (or
    four times the minimum), and no layout may be left without a kernel.  No launch: runs here."""
    import ctypes as C
    L = nb.lib()
    sizes = list(range(256, 16384, 256)) + list(range(16384, 300000, 7 * 1024)) + [1 << 20, 3 << 20, 1 << 22]
    for shards in (1, 2, 3, 8):
        for prec in (nb.FP64, nb.FP32):
            for S in sizes:
                a = nb.StepArgs()
                a.pos_cur, a.pos_next, a.mass, a.vel, a.acc_out = 8, 16, 8, 8, (8 if shards > 1 else None)
                a.shard_slots, a.num_shards, a.own_count, a.precision = S, shards, S, prec
                a.variant = nb.VARIANT_SYM
                need, v = C.c_int64(-1), C.c_int32(-1)
                nb.check(L.nbody_mi355x_step_workspace_bytes(C.byref(a), C.byref(need)))
                total = S * shards
                if total < 128:
                    continue
                assert need.value > 0, (S, shards, prec)
                assert need.value % (24 * S) == 0                     # whole rows of 3*S doubles
                assert need.value <= max(4 << 30, 4 * 24 * S * 80)    # capped (80 rows: a generous bound on the minimum)
                least = C.c_int64(-1)
                nb.check(L.nbody_mi355x_step_workspace_min_bytes(C.byref(a), C.byref(least)))
                assert 0 < least.value <= need.value and least.value % (24 * S) == 0
                vmin = C.c_int32(-1)
                a.workspace, a.workspace_bytes = 8, least.value       # the minimum is accepted ...
                nb.check(L.nbody_mi355x_step_variant(C.byref(a), C.byref(vmin)))
                assert vmin.value in nb.SYM_VARIANTS, (S, shards, prec, vmin.value)
                a.workspace, a.workspace_bytes = 8, need.value
                nb.check(L.nbody_mi355x_step_variant(C.byref(a), C.byref(v)))
                assert vmin.value == v.value                          # ... by the same kernel build
                assert v.value in nb.SYM_VARIANTS and v.value != nb.VARIANT_SYM, (S, shards, prec, v.value)
                block = {11: 256, 12: 512, 14: 1024, 18: 2048, 21: 64, 22: 128, 24: 256, 28: 512}[v.value]
                assert S % block == 0 and total // block >= 2
                assert (v.value > 20) == (total <= 262143), (S, shards, v.value)   # wave tiles up to 262 143 slots


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own (no C++, no HIP)."""
    import subprocess
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                    str(REPO / "include" / "nbody_mi355x.h")], check=True)


def test_step_tile_launches_counts_the_distance_batches():
    """nbody_mi355x_step_tile_launches: 1 for one-sided / unbatched steps, the number of batches
    when the workspace holds only part of the block distances (N = 1M with the default 4 GiB
    workspace: 2 -- bench.py divides the step's work by it for its per-launch figures)."""
    import ctypes as C
    L = nb.lib()

    def launches(S, ws_bytes, variant=0):
        a = nb.StepArgs()
        a.pos_cur, a.pos_next, a.mass, a.vel = 8, 16, 8, 8
        a.shard_slots, a.num_shards, a.own_count, a.j0_end, a.variant = S, 1, S, S, variant
        if ws_bytes:
            a.workspace, a.workspace_bytes = 8, ws_bytes
        out = C.c_int32(-1)
        nb.check(L.nbody_mi355x_step_tile_launches(C.byref(a), C.byref(out)))
        return out.value

    def preferred(S):
        a = nb.StepArgs()
        a.shard_slots, a.num_shards = S, 1
        need = C.c_int64()
        nb.check(L.nbody_mi355x_step_workspace_bytes(C.byref(a), C.byref(need)))
        return need.value

    assert launches(1 << 20, 0) == 1                                  # one-sided
    assert launches(1 << 20, preferred(1 << 20)) == 2                 # 4 GiB of the 7.3 GB
    assert launches(1 << 20, 8 << 30) == 1
    assert launches(65536, preferred(65536)) == 1
    assert launches(65536, preferred(65536) // 2) >= 2
    assert launches(1 << 22, preferred(1 << 22)) >= 8


def _plan(n, shards=1, variant=0, precision=None, own_shard=0):
    import ctypes as C
    from nbody_mi355x.sharded import shard_counts, slab_slots
    counts, _ = shard_counts(n, shards)
    a = nb.StepArgs()
    a.pos_cur, a.pos_next, a.mass, a.vel = 8, 16, 8, 8    # never dereferenced
    S = slab_slots(max(counts), variant, shards)
    a.shard_slots, a.num_shards, a.own_shard, a.own_count = S, shards, own_shard, counts[own_shard]
    a.j0_end = S * shards
    a.variant = variant
    a.precision = nb.FP64 if precision is None else precision
    a.workspace, a.workspace_bytes = 8, 1 << 40
    out = (C.c_int32 * nb.PLAN_FIELDS)()
    nb.check(nb.lib().nbody_mi355x_step_plan(C.byref(a), out))
    return list(out), S


def test_wave_tile_plans_deal_every_octet_once_and_every_cut_part_is_found():
    """nbody_mi355x_step_plan (host logic, no GPU): over sizes from the wave tiles' first to their
    last, one to eight slabs, both precisions -- the shares the plan deals (csrc/nbody_sym.hip:
    plan_wave, set_octets) cover a row's list exactly once, respect their minima (an older wave >= 8
    octets, a later one >= 2), and the rule by which the finishing kernels find the parts of a cut
    entry (sum_partials: first boundary behind the entry's start by one division, then boundary by
    boundary) names exactly the waves that begin inside the entry."""
    sizes = [2049, 2305, 2560, 3000, 4096, 4609, 5000, 8192, 9000, 9216, 10000, 12288, 12800, 12801, 13312, 15000,
             16384, 18432, 20000, 24576, 30000, 32768, 40960, 49152, 50000, 65536, 70000, 98304, 100000, 124928,
             131072, 200000, 261632]
    seen_uneven = seen_multicut = 0
    for n in sizes:
        for shards, prec in ((1, nb.FP64), (1, nb.FP32), (2, nb.FP64), (3, nb.FP64), (8, nb.FP64)):
            if n // shards < 300:
                continue
            # (several slabs: the drivers ask for the pair-symmetric family by name, as the handles do)
            p, S = _plan(n, shards, variant=0 if shards == 1 else nb.VARIANT_SYM, precision=prec)
            family, block, rows, D, chunks, ws, q, qb, octets, r_rows, cb_rows = p[:11]
            if S * shards > 262143:       # several slabs, each rounded up: past the wave tiles' last size
                assert family == 1 and S % 2048 == 0
                continue
            assert family == 2, (n, shards, p)
            assert S % block == 0 and rows == S // block and D == rows * shards // 2
            assert octets == (D + 1) * (block // 64) * 8
            waves = 4 * chunks
            assert 0 < ws <= waves and q >= 8 and 2 <= qb <= q
            if ws == waves:
                assert qb == q                                    # even dealing
            else:
                seen_uneven += 1
                assert ws * 2 == waves and rows * (ws // 4) <= 256 and 16 * rows * (ws // 4) >= 15 * 256
            begins = [k * q if k <= ws else ws * q + (k - ws) * qb for k in range(waves + 1)]
            assert begins[-1] >= octets and begins[-1] - octets < 4 * q + 4 * qb    # the list is covered; at most a workgroup's worth to spare
            assert r_rows >= chunks and cb_rows * 3 * S >= rows * waves * 192    # every wave has its Cb part's room
            # the finishing kernels' rule against brute force, for the low rows' list and the high rows' (one tile shorter)
            for n_oct in (octets, octets - (block // 64) * 8 if (rows * shards) % 2 == 0 else octets):
                for o in range(0, n_oct, 8):
                    brute = [k for k in range(1, waves) if o < begins[k] < min(o + 8, n_oct)]
                    split = ws * q
                    if o < split:
                        k = o // q + 1
                        b = k * q
                    else:
                        r = (o - split) // qb + 1
                        k, b = ws + r, split + r * qb
                    found = []
                    while b < min(o + 8, n_oct):
                        found.append(k)
                        b += q if k < ws else qb
                        k += 1
                    assert found == brute, (n, shards, o, found, brute)
                    seen_multicut += len(found) > 1
    assert seen_uneven >= 10 and seen_multicut > 0   # both kinds of dealing and entries cut more than once were met
    # the named cases of DESIGN.md 3.2
    p, _ = _plan(16384)
    assert p[1] == 512 and p[4] == 16 and p[5] == 32 and p[6] > 3 * p[7]
    p, _ = _plan(49152)
    assert p[5] == 4 * p[4]                                # 96 rows do not fill the chip with their first workgroups: even
    p, _ = _plan(1048576)
    assert p[0] == 1 and p[1] == 2048
    assert _plan(1000)[0][0] == 0                          # one-sided: no plan
