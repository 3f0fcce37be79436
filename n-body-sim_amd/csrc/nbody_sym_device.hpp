// nbody_sym_device.hpp -- device code shared by the two pair-symmetric kernel families
// (nbody_sym.hip: workgroup tiles, the large systems; nbody_symw.hip: wave tiles, the mid
// sizes): the launch record, the per-pair arithmetic and the 64-step walk of one j group.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nbody_device.hpp"

namespace nbody_sym {

using namespace nbody_device;

struct SymArgs {
  const double* pos;   // all slabs, slab layout
  const double* mass;  // all slabs, flat
  double* R;           // [nR][3][S]  i-side sums of the own rows, per distance chunk
  double* C;           // [Db][3][S]  j-side sums of tile (I, I+d) at row d - c_base, indexed by
                       //             OWN row I - row0; Db = distances per batch
  int64_t S;           // slots per slab
  int64_t own_count;   // bodies in the own slab: its slots from here on are padding (massless, at the origin)
  int nbs;             // blocks per slab = S / B
  int nb;              // blocks in all = num_shards * nbs
  int row0;            // first own block row = own_shard * nbs
  int D;               // largest distance = nb / 2
  int d_lo, d_hi;      // distances this launch covers, inclusive
  int dsplit;          // chunks [d_lo, d_hi] is cut into (grid.y)
  int jsplit;          // parts a tile's j groups are cut into (grid.z): 1, 2, 4 or 8, <= IPT
  int jlocal;          // the same for the launch of phase 1 (diagonal tiles only); host side only
  int r_base;          // R chunk of this launch's chunk 0
  int r_first;         // first R chunk the finishing kernels add (0 = include the local phase's)
  int nR;              // one past the last R chunk they add
  int c_base;          // distance stored in row 0 of C for this batch
  int no_eq;           // 1: never take the equal-mass forms of the tile loop (tests)
  int bshift;          // log2 of the block size B (finishing kernels: block of a slot = off >> bshift)
  // wave tiles only.  A row's work in a launch is the list of its (tile, group) entries, 8
  // "octets" (8 rotation steps) each; wave k of the row (4 per workgroup) takes octets
  // [k q, (k+1) q) of it -- whole entries in the middle, PARTS of entries at the two ends:
  int q;               // octets per wave, >= 8, so an entry is cut at most once
  unsigned q_magic;    // floor(x / q) = (x * q_magic) >> 32 for every x the finishing kernels form
  // Uneven dealing (set_octets): the row's first `ws` waves -- those of the workgroups that are
  // dispatched first, one per CU -- take q octets each, the waves behind them qb <= q: the SIMD's
  // arbiter serves its oldest wave first, so the second wave of a SIMD only fills the issue
  // slots the first leaves, and finishes with it only if its share is that much smaller.
  // Wave k begins at octet k q (k <= ws) or ws q + (k - ws) qb.  Even dealing: ws = all, qb = q.
  int ws;
  int qb;              // >= 2: a later wave's share may be less than an entry, which is then cut more than once
  unsigned qb_magic;
  double* Cb;          // [nbs][wstride][3][64]: j-side sums of the SECOND part of a cut entry, by
                       // the wave that computed it (the first part goes to C like a whole entry)
  int wstride;         // waves per row in Cb's indexing (4 * chunks of the largest launch)
  int fast;            // 1: NBODY_MI355X_FP64_FAST -- equal-mass tiles take kDeferredFast
  double eps2;
  // tuning only (NBODY_MI355X_SYMW_TRACE, nbody_symw.hip): per workgroup {start, end (100 MHz
  // clock), HW_ID, XCC_ID}; nullptr in every normal launch
  unsigned long long* trace;
};

// a pair of coordinates as staged in LDS: {x,y} and {z,m}, in the arithmetic type
template <typename T>
struct alignas(2 * sizeof(T)) P2 {
  T a, b;
};

__device__ __forceinline__ double rotate_from_next_lane(double v) {
  // lane l receives lane (l+1) mod 64's value: DPP wave_rol:1 on both halves
  int lo = __double2loint(v), hi = __double2hiint(v);
  // (every lane is written, so the "old" operand is never used: pass the source itself
  // rather than a constant that would cost a v_mov)
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rotate_from_next_lane(float v) {
  int b = __float_as_int(v);
  b = __builtin_amdgcn_update_dpp(b, b, 0x134, 0xf, 0xf, false);
  return __int_as_float(b);
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// FP32 walks, scaled coordinates (the row's extent in [1, 2)): the bare cube y0^3 of a pair
// closer than 2^-42 is beyond float's range (rsq(2^-84)^3 = 2^126).  Unsoftened, such a pair
// counts as two bodies in one place and contributes nothing; softened, only d^2 == 0 does.
constexpr float kMinR2f = 0x1p-84f;

// The two weights of one pair from its separation: si = m_j r^-3 (applied to body i) and
// sj = m_i r^-3 (applied to body j; only computed when WANT_J).
// double: rsq seed + folded series, r^2 chain seeded with 2^-300 or eps2 (nbody_device.hpp).
// float (the FP32 variant), one body per lane only (two and more take the packed walk below):
// v_rsq_f32 is good to an ulp; r^2 == 0 is an explicit select because 2^-300 does not exist in
// float; the mass is multiplied in BEFORE the cube is completed, (m y0) (y0 y0).
__device__ __forceinline__ double pair_r3(double dx, double dy, double dz, double eps2, double c15, double c1875) {
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
  const double y0 = __builtin_amdgcn_rsq(r2);
  const double u = y0 * y0;
  const double e = __builtin_fma(-r2, u, 1.0);
  const double w0 = y0 * u;
  const double p = __builtin_fma(e, c1875, c15);
  const double we = w0 * e;
  return __builtin_fma(we, p, w0);  // r^-3
}
__device__ __forceinline__ float pair_r3(float, float, float, float, float, float) { return 0.0f; }  // (double only)
// The same with the series cut after the LINEAR term, y0^3 (1 + 3/2 e): one operation fewer
// per pair (17 + rsq), truncation error 15/8 e^2 relative -- e <= 2^-23 for v_rsq_f64's seed
// (tools/ubench_fp64.hip), i.e. <= 2.7e-14 per pair: three orders inside north_star's 1e-10,
// outside this library's own 1e-12 asserts -- hence a precision tier of its own
// (NBODY_MI355X_FP64_FAST), never the default.
__device__ __forceinline__ double pair_r3_fast(double dx, double dy, double dz, double eps2, double c15) {
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
  const double y0 = __builtin_amdgcn_rsq(r2);
  const double u = y0 * y0;
  const double e = __builtin_fma(-r2, u, 1.0);
  const double w0 = y0 * u;
  const double we = w0 * e;
  return __builtin_fma(we, c15, w0);  // r^-3
}
template <bool WANT_J>
__device__ __forceinline__ void pair_weights(double dx, double dy, double dz, double mj, double mi, double eps2,
                                             double c15, double c1875, double* si, double* sj) {
  const double s0 = pair_r3(dx, dy, dz, eps2, c15, c1875);
  *si = s0 * mj;
  if (WANT_J) *sj = s0 * mi;
}
template <bool WANT_J>
__device__ __forceinline__ void pair_weights(float dx, float dy, float dz, float mj, float mi, float eps2,
                                             float, float, float* si, float* sj) {
  // the bare separation decides whether the pair counts: a body met with itself (or two in one
  // place) has d2 == 0 whatever the softening, and must give 0, not (m inf) * 0; unsoftened,
  // pairs closer than 2^-42 of the row's extent (kMinR2f) count as "in one place" too
  const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  const float y = __builtin_amdgcn_rsqf(d2 + eps2);
  const float y0 = (d2 > (eps2 > 0.0f ? 0.0f : kMinR2f)) ? y : 0.0f;
  const float y2 = y0 * y0;
  *si = (mj * y0) * y2;
  if (WANT_J) *sj = (mi * y0) * y2;
}

// Wave tiles, own rows: the i side of a PADDING slot (index >= own_count in the own slab) is moved
// this far out in registers.  Against any real coordinate r^2 ~ 3 * 2^720 (float: 3 * 2^120, in the
// row's scaled units), y0^3 underflows to exactly 0 and with it every term the slot would add to a
// real body's sums -- so a row whose REAL bodies have one mass keeps the equal-mass form (the slot's
// mass, 0, is never looked at there) instead of dropping to the general form for a handful of
// padding slots: 20 instead of 18 operations per pair for that row, which an evenly dealt launch
// absorbs and an unevenly dealt one (one round, shares fixed up front) waits for -- 20 000 bodies:
// 168.9 us against 162.9 before.  Two padding slots meet at separation 0: weight * 0, as a body
// with itself.
template <typename T> __device__ __forceinline__ T far_away();
template <> __device__ __forceinline__ double far_away<double>() { return 0x1p360; }
template <> __device__ __forceinline__ float far_away<float>() { return 0x1p60f; }

// LDS layout of a staged tile: each group of 64 j bodies takes kGrp = 72 entries, of which
// entries 64..71 repeat 0..7.  A lane that walks its group from entry r (0..63) can then
// read 8 consecutive steps at immediate offsets without wrapping, so the inner loop has no
// address arithmetic (2 integer instructions per 8 steps instead of 7 per 2).
constexpr int kGrp = 72;

// How the masses enter one pair (workgroup- or wave-uniform, chosen per staged group):
//   kGeneral   si = m_j r^-3, sj = m_i r^-3: 20 FP64 operations per pair.
//   kDeferred  (double) every body of the row AND of the staged groups has the one mass
//              m_ref: both sides take the bare r^-3 and the mass is multiplied in once per
//              sum instead of once per pair -- the j sums when a wave hands them on, the
//              i sums (which stay in units of 1/m_ref for the whole row) when the row is
//              stored: 18 operations per pair.  The benchmark scenes (Plummer sphere,
//              uniform cube) are equal-mass systems, and so are all but one row of a ring.
//   kRowUnit   (double) the row has one mass but the staged bodies do not (a padded block,
//              the ring's planet).  The j side is kDeferred's: a body's acceleration does not
//              depend on its own mass, so sum_i m_ref r^-3 d is right whatever m_j is -- bare
//              r^-3, times m_ref per sum.  The i side, in units of 1/m_ref, takes r^-3 times
//              m_j / m_ref, which is what such a row STAGES in the mass slot: 19 operations
//              (round 3; before, m_j r^-3 / m_ref and m_i r^-3 per pair: 21).
//   kSameMass  (float) as kDeferred's condition, but the mass stays in the product (r^-3
//              alone can leave float's range) and one product serves both sides: 19.
// kDeferred / kRowUnit round once more per sum than kGeneral (the final product with m_ref):
// results agree to an ulp or two of each sum, not bit for bit.
//   kDeferredFast  (double, NBODY_MI355X_FP64_FAST) kDeferred with pair_r3_fast: 17 operations.
enum { kGeneral = 0, kDeferred = 1, kRowUnit = 2, kSameMass = 3, kDeferredFast = 4 };

// The FP32 walk with PACKED arithmetic: two of the lane's bodies per instruction.
// Measured on gfx950 (tools/ubench_valu_rates.hip, ns per wave instruction per SIMD):
// v_fma_f32 1.6, v_pk_fma_f32 2.0 for two FMAs; v_mul/v_sub_f32 1.1, v_pk_mul/v_pk_add_f32
// 1.9 for two -- the 9 FMAs of a pair are what packing pays for.  The j body arrives from LDS
// as two 64-bit pairs {x,y} {z,m}, and a packed instruction can take either half of a pair
// for both of its lanes (op_sel), so broadcasting x_j, y_j, z_j, m_j costs nothing; the
// compiler folds the splats below into op_sel (checked in the ISA: no v_mov in the loop).
// The j-side sums travel as TWO partial sums per component (even / odd bodies of the lane),
// added when they are home.  Element for element the operations are the scalar path's
// (IEEE fma / mul / add), in the same order on the i side.
#ifndef NBODY_PK_W
#define NBODY_PK_W 2
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// ZERO: what is done about r^2 == 0 (a body met with itself, or two in one place), which must
// give no force:
//   kZeroNone    nothing -- softened (eps2 >= FLT_MIN), r^2 > 0 for every pair;
//   kZeroSelect  such a pair's r^2 becomes +inf, whose rsq is 0: 2 instructions a pair;
//   kZeroRedo    nothing in the walk: rsq(0) = inf makes that pair's term inf * 0 = NaN (or
//                +-inf, where a tiny difference squared to 0) and the body's i-side sum
//                stays non-finite from there on.  The caller looks at the group's sums
//                afterwards, discards them and walks again with kZeroSelect if one is not
//                finite.  Off the diagonal tile that takes two bodies in one place (or
//                input that is not finite, which then simply comes out again).
//   kZeroSelectSoft  softened, where kZeroNone is not safe (below): the r^2 chain starts at 0, a
//                pair with d^2 == 0 becomes +inf as above, every other gets d^2 + eps2: 3 a pair.
// kZeroNone is only taken where the self pair's weight is certain to be finite (its separation
// is exactly 0, and inf * 0 = NaN would poison the body's sum): the bare cube eps2'^-3/2 must
// stay in range (eps2' >= kMinR2f in the row's scaled units) and no mass may multiply it
// (kSameMass: the common mass is applied per sum, in double).  A scene with mixed masses in SI
// units (m ~ 1e27 in the product) overflows long before that, so it takes kZeroSelectSoft.
enum { kZeroNone = 0, kZeroSelect = 1, kZeroRedo = 2, kZeroSelectSoft = 3 };

template <int H, bool SYM, int MODE, int ZERO>
__device__ __forceinline__ void walk_pk(const P2<float>* xy, const P2<float>* zm, int gb, int lane,
                                        const f2 (&xi2)[H], const f2 (&yi2)[H], const f2 (&zi2)[H],
                                        const f2 (&mi2)[H], f2 (&ax2)[H], f2 (&ay2)[H], f2 (&az2)[H], f2 e2,
                                        int k0, int k1, f2* fxo, f2* fyo, f2* fzo) {
  // W packed pairs go through each stage together: neighbours in the instruction stream are
  // then independent (a packed result is not ready for the very next instruction), at W
  // times the live temporaries
  constexpr int W = H >= NBODY_PK_W ? NBODY_PK_W : H;
  f2 fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
  int rot = (lane + 8 * k0) & 63;
  P2<float> bxy[2], bzm[2];
  bxy[0] = xy[gb + rot];
  bzm[0] = zm[gb + rot];
  for (int t8 = k0; t8 < k1; t8++) {
    const int rot_next = (rot + 8) & 63;
    const P2<float>* pxy = xy + gb + rot;
    const P2<float>* pzm = zm + gb + rot;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      bxy[(u + 1) & 1] = u < 7 ? pxy[u + 1] : xy[gb + rot_next];
      bzm[(u + 1) & 1] = u < 7 ? pzm[u + 1] : zm[gb + rot_next];
      const f2 jxy = __builtin_bit_cast(f2, bxy[u & 1]);
      const f2 jzm = __builtin_bit_cast(f2, bzm[u & 1]);
#pragma unroll
      for (int k0 = 0; k0 < H; k0 += W) {
        f2 dx[W], dy[W], dz[W], r2[W], y0[W], si[W], sj[W];
#pragma unroll
        for (int k = 0; k < W; k++) {
          dx[k] = splat(jxy.x) - xi2[k0 + k];
          dy[k] = splat(jxy.y) - yi2[k0 + k];
          dz[k] = splat(jzm.x) - zi2[k0 + k];
        }
#pragma unroll
        for (int k = 0; k < W; k++) r2[k] = ZERO == kZeroSelectSoft ? dx[k] * dx[k] : fma2(dx[k], dx[k], e2);
#pragma unroll
        for (int k = 0; k < W; k++) r2[k] = fma2(dy[k], dy[k], r2[k]);
#pragma unroll
        for (int k = 0; k < W; k++) r2[k] = fma2(dz[k], dz[k], r2[k]);
#pragma unroll
        for (int k = 0; k < W; k++) {
          if (ZERO == kZeroSelect) {
            r2[k].x = (r2[k].x > kMinR2f) ? r2[k].x : __builtin_inff();
            r2[k].y = (r2[k].y > kMinR2f) ? r2[k].y : __builtin_inff();
          } else if (ZERO == kZeroSelectSoft) {
            const f2 soft = r2[k] + e2;
            r2[k].x = (r2[k].x > 0.0f) ? soft.x : __builtin_inff();
            r2[k].y = (r2[k].y > 0.0f) ? soft.y : __builtin_inff();
          }
          y0[k] = f2{__builtin_amdgcn_rsqf(r2[k].x), __builtin_amdgcn_rsqf(r2[k].y)};
        }
#pragma unroll
        for (int k = 0; k < W; k++) {
          const f2 y3 = (y0[k] * y0[k]) * y0[k];  // (scaled coordinates: O(1), see pow2_scale)
          if (MODE == kSameMass) {
            si[k] = sj[k] = y3;
          } else {
            si[k] = splat(jzm.y) * y3;
            if (SYM) sj[k] = mi2[k0 + k] * y3;
          }
        }
#pragma unroll
        for (int k = 0; k < W; k++) {
          ax2[k0 + k] = fma2(si[k], dx[k], ax2[k0 + k]);
          ay2[k0 + k] = fma2(si[k], dy[k], ay2[k0 + k]);
          az2[k0 + k] = fma2(si[k], dz[k], az2[k0 + k]);
        }
        if (SYM) {
#pragma unroll
          for (int k = 0; k < W; k++) {
            fx = fma2(-sj[k], dx[k], fx);
            fy = fma2(-sj[k], dy[k], fy);
            fz = fma2(-sj[k], dz[k], fz);
          }
        }
      }
      if (SYM) {  // both partial sums follow their body
        fx = f2{rotate_from_next_lane(fx.x), rotate_from_next_lane(fx.y)};
        fy = f2{rotate_from_next_lane(fy.x), rotate_from_next_lane(fy.y)};
        fz = f2{rotate_from_next_lane(fz.x), rotate_from_next_lane(fz.y)};
      }
    }
    rot = rot_next;
  }
  *fxo = fx;
  *fyo = fy;
  *fzo = fz;
}

template <int IPT, bool SYM, int MODE, int SOFT>  // SOFT: 0 unsoftened, 1 kZeroNone is safe, 2 kZeroSelectSoft
__device__ __forceinline__ void group_steps_pk(const P2<float>* xy, const P2<float>* zm, int gb, int lane,
                                               const float (&xi)[IPT], const float (&yi)[IPT],
                                               const float (&zi)[IPT], const float (&mi)[IPT], float (&ax)[IPT],
                                               float (&ay)[IPT], float (&az)[IPT], float eps2, int k0, int k1,
                                               float* fxo, float* fyo, float* fzo) {
  static_assert(IPT % 2 == 0, "two bodies per packed instruction");
  static_assert(MODE == kGeneral || MODE == kSameMass, "the float forms");
  constexpr int H = IPT / 2;
  f2 xi2[H], yi2[H], zi2[H], mi2[H], gx[H], gy[H], gz[H];
#pragma unroll
  for (int k = 0; k < H; k++) {
    xi2[k] = f2{xi[2 * k], xi[2 * k + 1]};
    yi2[k] = f2{yi[2 * k], yi[2 * k + 1]};
    zi2[k] = f2{zi[2 * k], zi[2 * k + 1]};
    mi2[k] = f2{mi[2 * k], mi[2 * k + 1]};
    gx[k] = gy[k] = gz[k] = f2{0, 0};  // this group's share of the i-side sums
  }
  const f2 e2 = splat(eps2);
  f2 fx, fy, fz;
  // the diagonal tile meets every body with itself: selects from the start
  constexpr int kFirst = SOFT == 1 ? kZeroNone : SOFT == 2 ? kZeroSelectSoft : (SYM ? kZeroRedo : kZeroSelect);
  walk_pk<H, SYM, MODE, kFirst>(xy, zm, gb, lane, xi2, yi2, zi2, mi2, gx, gy, gz, e2, k0, k1, &fx, &fy, &fz);
  if constexpr (kFirst == kZeroRedo) {
    f2 all = gx[0] + gy[0] + gz[0];  // finite iff every term is (inf - inf = NaN)
#pragma unroll
    for (int k = 1; k < H; k++) all += gx[k] + gy[k] + gz[k];
    const float probe = all.x + all.y;
    if (__builtin_expect(__any(!(probe - probe == 0.0f)), 0)) {
#pragma unroll
      for (int k = 0; k < H; k++) gx[k] = gy[k] = gz[k] = f2{0, 0};
      walk_pk<H, SYM, MODE, kZeroSelect>(xy, zm, gb, lane, xi2, yi2, zi2, mi2, gx, gy, gz, e2, k0, k1, &fx, &fy, &fz);
    }
  }
#pragma unroll
  for (int k = 0; k < H; k++) {
    ax[2 * k] += gx[k].x; ax[2 * k + 1] += gx[k].y;
    ay[2 * k] += gy[k].x; ay[2 * k + 1] += gy[k].y;
    az[2 * k] += gz[k].x; az[2 * k + 1] += gz[k].y;
  }
  *fxo = fx.x + fx.y;
  *fyo = fy.x + fy.y;
  *fzo = fz.x + fz.y;
}

// One group of 64 j bodies (LDS entries gb .. gb+71) against this wave's 64*IPT bodies: at
// step t the lane meets body (lane + t) mod 64 of the group.  SYM: the j-side sums travel
// with their body through the lanes, one DPP wave rotation per step (6 v_mov_b32_dpp, 3.3
// cycles each, per IPT pairs); 64 rotations later they are home: lane l holds body l's.
// [k0, k1) selects a part of the walk in octets (8 steps each; the whole walk is [0, 8)): steps
// 8 k0 .. 8 k1 - 1, after which lane l holds the sums of body (l + 8 k1) mod 64 over those steps
// (the wave tiles cut an entry between two waves to level the load, nbody_symw.hip).
// (Sending each step's sums to the body's LDS entry with a return-less ds_add_f64 instead
// was tried: it removes the moves, but an LDS atomic with 64-bit data holds up the SIMD's
// VALU issue for ~12 cycles, 3 of them cost more than the 6 moves -- tools/ubench_lds.hip,
// DESIGN.md 3.1b.)  Without SYM (a block against itself) only the i side is kept.
template <int IPT, bool SYM, int MODE, typename T>
__device__ __forceinline__ void group_steps(const P2<T>* xy, const P2<T>* zm, int gb, int lane, const T (&xi)[IPT],
                                            const T (&yi)[IPT], const T (&zi)[IPT], const T (&mi)[IPT],
                                            T (&ax)[IPT], T (&ay)[IPT], T (&az)[IPT], T eps2, T c15, T c1875,
                                            T inv_m_ref, int k0, int k1, T* fxo, T* fyo, T* fzo) {
  if constexpr (sizeof(T) == sizeof(float) && IPT % 2 == 0 && (MODE == kGeneral || MODE == kSameMass)) {  // FP32: packed
    if (eps2 > 0.0f) {  // softened: see kZeroNone / kZeroSelectSoft above
      if (MODE == kSameMass && eps2 >= kMinR2f)
        group_steps_pk<IPT, SYM, MODE, 1>(xy, zm, gb, lane, xi, yi, zi, mi, ax, ay, az, eps2, k0, k1, fxo, fyo, fzo);
      else
        group_steps_pk<IPT, SYM, MODE, 2>(xy, zm, gb, lane, xi, yi, zi, mi, ax, ay, az, eps2, k0, k1, fxo, fyo, fzo);
    } else {
      group_steps_pk<IPT, SYM, MODE, 0>(xy, zm, gb, lane, xi, yi, zi, mi, ax, ay, az, eps2, k0, k1, fxo, fyo, fzo);
    }
    return;
  }
  T fx = 0, fy = 0, fz = 0;
  int rot = (lane + 8 * k0) & 63;
  // the body of the NEXT step is read while this step's pairs are evaluated: two register
  // sets used alternately (8 steps per trip: the roles are the same at every trip's start)
  P2<T> bxy[2], bzm[2];
  bxy[0] = xy[gb + rot];
  bzm[0] = zm[gb + rot];
  for (int t8 = k0; t8 < k1; t8++) {
    const int rot_next = (rot + 8) & 63;
    const P2<T>* pxy = xy + gb + rot;
    const P2<T>* pzm = zm + gb + rot;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      // (after the last step this reads the group's first body again: a valid address)
      bxy[(u + 1) & 1] = u < 7 ? pxy[u + 1] : xy[gb + rot_next];
      bzm[(u + 1) & 1] = u < 7 ? pzm[u + 1] : zm[gb + rot_next];
      const P2<T> jxy = bxy[u & 1];
      const P2<T> jzm = bzm[u & 1];
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const T dx = jxy.a - xi[k];
        const T dy = jxy.b - yi[k];
        const T dz = jzm.a - zi[k];
        T si = 0, sj = 0;
        if constexpr (MODE == kDeferred) {
          si = sj = pair_r3(dx, dy, dz, eps2, c15, c1875);
        } else if constexpr (MODE == kDeferredFast) {
          if constexpr (sizeof(T) == sizeof(double)) si = sj = pair_r3_fast(dx, dy, dz, eps2, c15);
        } else if constexpr (MODE == kRowUnit) {
          if constexpr (sizeof(T) == sizeof(double)) {  // (staged: m_j / m_ref)
            sj = pair_r3(dx, dy, dz, eps2, c15, c1875);
            si = sj * jzm.b;
          }
        } else if constexpr (MODE == kSameMass) {
          pair_weights<false>(dx, dy, dz, jzm.b, mi[k], eps2, c15, c1875, &si, &sj);
          sj = si;
        } else {
          pair_weights<SYM>(dx, dy, dz, jzm.b, mi[k], eps2, c15, c1875, &si, &sj);
        }
        ax[k] = fma_t(si, dx, ax[k]);
        ay[k] = fma_t(si, dy, ay[k]);
        az[k] = fma_t(si, dz, az[k]);
        if (SYM) {
          fx = fma_t(-sj, dx, fx);
          fy = fma_t(-sj, dy, fy);
          fz = fma_t(-sj, dz, fz);
        }
      }
      if (SYM) {  // the sums follow their body to the lane that meets it next
        fx = rotate_from_next_lane(fx);
        fy = rotate_from_next_lane(fy);
        fz = rotate_from_next_lane(fz);
      }
    }
    rot = rot_next;
  }
  *fxo = fx;
  *fyo = fy;
  *fzo = fz;
}

// first slot of block J in the slab layout: coordinates via *x (y at +S, z at +2S), mass via *m
template <int B>
__device__ __forceinline__ void block_base(const SymArgs& a, int J, const double** x, const double** m) {
  const int slab = J / a.nbs;
  const int64_t off = static_cast<int64_t>(J - slab * a.nbs) * B;
  *x = a.pos + static_cast<int64_t>(slab) * 3 * a.S + off;
  *m = a.mass + static_cast<int64_t>(slab) * a.S + off;
}

// FP32 variant: where a row's coordinates are measured from.  A float holds a coordinate to
// 6e-8 of its MAGNITUDE, so narrowing absolute positions makes close pairs of a system that
// sits far from the origin cancel (a cluster of size 1 at distance 2e3: ulp 1.2e-4).  The
// reference's float instantiation differences full-precision inputs (particle_simulation.h:
// 257-270); here the row's bodies and every j body staged against them are taken relative
// to one point, in double, and only the (small) relative coordinate is narrowed.  The point
// is the centroid of the row's real bodies -- for bodies in no spatial order that is the
// system's centre, for sorted ones the block's own neighbourhood.  Padding slots (mass 0 at
// the origin) are not counted; a centroid that is not finite (NaN / inf input) falls back to
// 0, i.e. absolute coordinates, so that garbage spreads exactly as before.  Every workgroup
// of a row adds in the same fixed order: the same origin, bit for bit, run to run.
__device__ __forceinline__ double uniform(double v) {  // a wave-uniform value into SGPRs
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// ... and the SCALE they are measured in: a power of two c (exact in double and float) that
// brings the row's extent -- the largest |coordinate - origin| of its real bodies -- into
// [1, 2).  In scaled coordinates r'^-3 = (c r)^-3 is O(1) for the pairs that matter, so the
// bare cube can be formed in float (unscaled, r^-3 leaves float's range beyond r = 4e12 --
// metres to Neptune) and the masses stay out of the pair: sums are in units of c^2 (and of
// the common mass, where there is one) and are converted once, in double, when stored.
// 1 for a row without extent, or with one that is not finite.
__device__ __forceinline__ double pow2_scale(double ext) {
  int e = ((__double2hiint(ext) >> 20) & 0x7ff) - 1023;
  if (!(ext > 0.0) || e == 1024 || e == -1023) return 1.0;  // 0, NaN, inf, subnormal
  e = e > 500 ? 500 : (e < -500 ? -500 : e);                // c * c must stay a double
  return __hiloint2double((1023 - e) << 20, 0);
}
__device__ __forceinline__ double body_extent(double x, double y, double z, double m, double ox, double oy,
                                              double oz) {
  const bool padding = m == 0.0 && x == 0.0 && y == 0.0 && z == 0.0;
  const double ex = __builtin_fabs(x - ox), ey = __builtin_fabs(y - oy), ez = __builtin_fabs(z - oz);
  const double e = __builtin_fmax(ex, __builtin_fmax(ey, ez));
  return (padding || !(e - e == 0.0)) ? 0.0 : e;  // padding and non-finite bodies do not set the scale
}
template <int IPT>
__device__ __forceinline__ void row_origin(const double* px, const double* pm, int64_t S, int tid, double* ox,
                                           double* oy, double* oz, double* scale) {
  __shared__ double cen[4][5];
  double sx = 0.0, sy = 0.0, sz = 0.0, cnt = 0.0;
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 256 + tid;
    const double x = px[idx], y = px[S + idx], z = px[2 * S + idx];
    const bool padding = pm[idx] == 0.0 && x == 0.0 && y == 0.0 && z == 0.0;
    sx += x; sy += y; sz += z;
    cnt += padding ? 0.0 : 1.0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sx += __shfl_down(sx, o, 64);
    sy += __shfl_down(sy, o, 64);
    sz += __shfl_down(sz, o, 64);
    cnt += __shfl_down(cnt, o, 64);
  }
  if ((tid & 63) == 0) {
    cen[tid >> 6][0] = sx; cen[tid >> 6][1] = sy; cen[tid >> 6][2] = sz; cen[tid >> 6][3] = cnt;
  }
  __syncthreads();
  const double n = cen[0][3] + cen[1][3] + cen[2][3] + cen[3][3];
  double cx = (cen[0][0] + cen[1][0] + cen[2][0] + cen[3][0]) / n;
  double cy = (cen[0][1] + cen[1][1] + cen[2][1] + cen[3][1]) / n;
  double cz = (cen[0][2] + cen[1][2] + cen[2][2] + cen[3][2]) / n;
  const bool ok = n > 0.0 && (cx - cx == 0.0) && (cy - cy == 0.0) && (cz - cz == 0.0);
  // workgroup-uniform: kept in scalar registers (the float IPT = 4 build sits at the VGPR
  // count that still allows two waves per SIMD; six more VGPRs halve its occupancy)
  cx = uniform(ok ? cx : 0.0);
  cy = uniform(ok ? cy : 0.0);
  cz = uniform(ok ? cz : 0.0);
  double ext = 0.0;
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 256 + tid;
    ext = __builtin_fmax(ext, body_extent(px[idx], px[S + idx], px[2 * S + idx], pm[idx], cx, cy, cz));
  }
  for (int o = 32; o > 0; o >>= 1) ext = __builtin_fmax(ext, __shfl_down(ext, o, 64));
  if ((tid & 63) == 0) cen[tid >> 6][4] = ext;
  __syncthreads();
  ext = __builtin_fmax(__builtin_fmax(cen[0][4], cen[1][4]), __builtin_fmax(cen[2][4], cen[3][4]));
  *ox = cx;
  *oy = cy;
  *oz = cz;
  *scale = uniform(pow2_scale(ext));
}


// nbody_symw.hip: launches sym_wave_kernel<ipt, float|double> (ipt 1, 2, 4; 8 in double only)
hipError_t launch_wave_tiles(int ipt, bool fp32, const SymArgs& sa, dim3 grid, hipStream_t stream);

}  // namespace nbody_sym
