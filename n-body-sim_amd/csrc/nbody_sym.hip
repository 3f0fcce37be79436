// nbody_sym.hip -- the pair-symmetric form of the small step (Newton's third law).
//
// The one-sided kernel (nbody_kernels.hip) evaluates every ORDERED pair: 16 FP64
// instructions + v_rsq_f64 = 20 FMA issue slots, which caps it at 50 % of the FP64 vector
// peak under the 20-FLOP-per-interaction convention.  Here every
// UNORDERED pair is evaluated once and applied to both bodies:
//     s0 = r^-3 ;  a_i += (m_j s0) d ;  a_j -= (m_i s0) d          20 FP64 + rsq per pair
// i.e. 12 issue slots per ordered pair.  The price is that the j side of a pair belongs to
// another lane, another wave and usually another workgroup:
//
//   * bodies are cut into blocks of B = 256*IPT slots; workgroup (I, s) owns block row I
//     and a sub-range of the cyclic block distances d = 0 .. nb/2: tile (I, J = I+d mod nb).
//     Every unordered block pair is met exactly once (d = nb/2, nb even: only rows
//     I < nb/2); every row has the same amount of work.
//   * i side: the lane's IPT bodies and their sums live in registers for the whole row.
//   * j side, inside a wave: a group of 64 j bodies circulates through the 64 lanes --
//     at step t lane l meets body (l+t) mod 64, read from LDS at a per-lane address
//     (conflict-free ds_read_b128), and the three running j sums travel with the body,
//     one DPP wave rotation per step (6 v_mov_b32_dpp per IPT pairs).  After 64 steps the
//     sums are back in their home lanes holding the whole wave's contribution.
//   * j side, across the 4 waves: the waves work on 4 different groups per phase
//     (barrier per phase) and add their group sums into one LDS array in a fixed order.
//   * j side, across workgroups: the tile's B sums are STORED (not atomically added) to
//     a workspace slab C[d][row][.], the row's i sums to R[s][.]; a finishing kernel adds
//     the partials of every body in a fixed order and integrates.  No atomics anywhere:
//     results are bit-reproducible run to run.
//   * d = 0 (the diagonal tile) is evaluated one-sided.
//
// Launch shape (make_plan below): grid = (own rows, dsplit distance chunks, jsplit parts of
// a tile's j groups).  dsplit and jsplit only exist to put >= 2 waves' worth of workgroups
// on each of the 256 CUs when there are few block rows; each extra part costs one more
// R slab to add up.  When the C slabs for all nb/2+1 distances do not fit in the caller's
// workspace the distances are walked in batches and the finishing kernel gathers the
// batch's C rows into the running sums between batches (sym_gather_kernel), so any N runs
// in a bounded workspace.  args->phase splits the step for the multi-rank path: 1 = tiles
// that need only the own slab (the diagonal), 2 = the rest + finish, 0 = both.
//
// The kernels are templated on the arithmetic type T: double is the product path; float is
// NBODY_MI355X_FP32 (state, sums and integration stay FP64, only the pair arithmetic and
// the staged tile are FP32; the reference's float instantiation keeps
// everything in float).
//
// Summation order differs from the reference's j = 0..N-1; the per-pair arithmetic is
// the same 4-ulp r^-3 as the one-sided kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "nbody_device.hpp"
#include "nbody_internal.h"

using namespace nbody_device;
using nbody_internal::set_error;

namespace {

struct SymArgs {
  const double* pos;   // all slabs, slab layout
  const double* mass;  // all slabs, flat
  double* R;           // [nR][3][S]  i-side sums of the own rows, per distance chunk
  double* C;           // [Db][3][S]  j-side sums of tile (I, I+d) at row d - c_base, indexed by
                       //             OWN row I - row0; Db = distances per batch
  int64_t S;           // slots per slab
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
  double eps2;
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

// The two weights of one pair from its separation: si = m_j r^-3 (applied to body i) and
// sj = m_i r^-3 (applied to body j; only computed when WANT_J).
// double: rsq seed + folded series, r^2 chain seeded with 2^-600 or eps2 (nbody_device.hpp).
// float (the FP32 variant): v_rsq_f32 is good to an ulp; r^2 == 0 is an explicit select
// because 2^-600 does not exist in float; and the mass is multiplied in BEFORE the cube is
// completed, (m y0) (y0 y0), so that r^-3 alone never underflows (r = 1e13 m: 1e-39).
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
  const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
  const float y = __builtin_amdgcn_rsqf(r2);
  const float y0 = (r2 > 0.0f) ? y : 0.0f;
  const float y2 = y0 * y0;
  *si = (mj * y0) * y2;
  if (WANT_J) *sj = (mi * y0) * y2;
}

// LDS layout of a staged tile: each group of 64 j bodies takes kGrp = 72 entries, of which
// entries 64..71 repeat 0..7.  A lane that walks its group from entry r (0..63) can then
// read 8 consecutive steps at immediate offsets without wrapping, so the inner loop has no
// address arithmetic (2 integer instructions per 8 steps instead of 7 per 2).
constexpr int kGrp = 72;

// j groups of a tile that are staged in LDS at a time: 16 groups = 1024 bodies = 63 KB in
// double, two workgroups per CU.  Blocks of up to 1024 bodies (IPT <= 4) are staged whole;
// a 2048-body block (IPT = 8) goes through in two halves.
constexpr int kStageGroups = 16;
template <int IPT>
constexpr int staged_groups() { return 4 * IPT < kStageGroups ? 4 * IPT : kStageGroups; }

// The workgroup's j groups of the staged tile against this wave's 64*IPT bodies.  SYM = false is
// the diagonal tile (a block against itself): i side only, no barriers needed.
// The j-side sums travel with their body through the lanes: one DPP wave rotation per step,
// 6 v_mov_b32_dpp (3.3 cycles each) per IPT pairs.  (Sending each step's sums to the body's
// LDS entry with a return-less ds_add_f64 instead was tried: it removes the moves, but an
// LDS atomic with 64-bit data holds up the SIMD's VALU issue for ~12 cycles, 3 of them cost
// more than the 6 moves -- tools/ubench_lds.hip)
// MODE selects how the masses enter (workgroup-uniform, chosen per LDS fill):
//   kGeneral   si = m_j r^-3, sj = m_i r^-3: 20 FP64 operations per pair.
//   kDeferred  (double) every body of the row AND of the staged groups has the one mass
//              m_ref: both sides take the bare r^-3 and the mass is multiplied in once per
//              sum instead of once per pair -- the j sums when a wave adds them to LDS, the
//              i sums (which stay in units of 1/m_ref for the whole row) when the row is
//              stored: 18 operations per pair.  The benchmark scenes (Plummer sphere,
//              uniform cube) are equal-mass systems, and so are all but one row of a ring.
//   kRowUnit   (double) the row has one mass but the staged bodies do not (a padded block,
//              the ring's planet): the i sums are in units of 1/m_ref, so the i-side weight
//              is (m_j r^-3) / m_ref; j side as kGeneral.  Rare; 21 operations.
//   kSameMass  (float) as kDeferred's condition, but the mass stays in the product (r^-3
//              alone can leave float's range) and one product serves both sides: 19.
// kDeferred / kRowUnit round once more per sum than kGeneral (the final product with m_ref):
// results agree to an ulp or two of each sum, not bit for bit.
enum { kGeneral = 0, kDeferred = 1, kRowUnit = 2, kSameMass = 3 };
template <int IPT, bool SYM, int MODE, typename T>
__device__ __forceinline__ void tile_phases(const P2<T>* xy, const P2<T>* zm, T (*fJ)[kGrp * staged_groups<IPT>()],
                                            int lane, int w, int ng, const T (&xi)[IPT], const T (&yi)[IPT],
                                            const T (&zi)[IPT], const T (&mi)[IPT], T (&ax)[IPT],
                                            T (&ay)[IPT], T (&az)[IPT], T eps2, T c15, T c1875, T m_ref, T inv_m_ref) {
  // the staged groups 0 .. ng-1 (ng a multiple of 4)
  for (int ph = 0; ph < ng; ph++) {
    const int grp = (ph + (ng >> 2) * w) % ng;  // the 4 waves sit on 4 different groups
    const int gb = grp * kGrp;
    T fx = 0, fy = 0, fz = 0;
    int rot = lane;  // at step t the lane meets body (lane + t) mod 64 of the group
    // the body of the NEXT step is read while this step's pairs are evaluated: two register
    // sets used alternately (8 steps per trip: the roles are the same at every trip's start)
    P2<T> bxy[2], bzm[2];
    bxy[0] = xy[gb + rot];
    bzm[0] = zm[gb + rot];
    for (int t8 = 0; t8 < 8; t8++) {
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
          T si, sj = 0;
          if constexpr (MODE == kDeferred) {
            si = sj = pair_r3(dx, dy, dz, eps2, c15, c1875);
          } else if constexpr (MODE == kRowUnit) {
            pair_weights<SYM>(dx, dy, dz, jzm.b, mi[k], eps2, c15, c1875, &si, &sj);
            si *= inv_m_ref;
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
    if (SYM) {  // 64 rotations later the sums are home: lane l holds body l of the group
      if constexpr (MODE == kDeferred) {
        fx *= m_ref;
        fy *= m_ref;
        fz *= m_ref;
      }
      fJ[0][gb + lane] += fx;
      fJ[1][gb + lane] += fy;
      fJ[2][gb + lane] += fz;
      __syncthreads();  // next phase another wave adds to this group
    }
  }
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
template <int IPT>
__device__ __forceinline__ void row_origin(const double* px, const double* pm, int64_t S, int tid, double* ox,
                                           double* oy, double* oz) {
  __shared__ double cen[4][4];
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
  const double cx = (cen[0][0] + cen[1][0] + cen[2][0] + cen[3][0]) / n;
  const double cy = (cen[0][1] + cen[1][1] + cen[2][1] + cen[3][1]) / n;
  const double cz = (cen[0][2] + cen[1][2] + cen[2][2] + cen[3][2]) / n;
  const bool ok = n > 0.0 && (cx - cx == 0.0) && (cy - cy == 0.0) && (cz - cz == 0.0);
  // workgroup-uniform: kept in scalar registers (the float IPT = 4 build sits at the VGPR
  // count that still allows two waves per SIMD; six more VGPRs halve its occupancy)
  *ox = uniform(ok ? cx : 0.0);
  *oy = uniform(ok ? cy : 0.0);
  *oz = uniform(ok ? cz : 0.0);
}

template <int IPT, typename T>
__global__ __launch_bounds__(256) void sym_tiles_kernel(const SymArgs a) {
  constexpr int B = 256 * IPT;
  constexpr int G = staged_groups<IPT>();  // groups per LDS fill
  constexpr int L = G * kGrp;                                         // LDS entries
  __shared__ P2<T> xy[L];
  __shared__ P2<T> zm[L];
  __shared__ T fJ[3][L];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int il = blockIdx.x, s = blockIdx.y;  // own row (local index), distance chunk
  const int ng = (4 * IPT) / a.jsplit;        // j groups of a tile this workgroup takes ...
  const int g0 = blockIdx.z * ng;             // ... starting here (64 bodies per group)
  const int I = a.row0 + il;
  const int64_t S = a.S;
  // double: 2^-600 stands in for "no softening" (nbody_device.hpp); float: plain 0 + a select
  const T eps2 = static_cast<T>((a.eps2 > 0.0) ? a.eps2 : (sizeof(T) == sizeof(double) ? kTinyR2 : 0.0));
  T c15 = static_cast<T>(1.5), c1875 = static_cast<T>(1.875);
  asm volatile("" : "+v"(c15), "+v"(c1875));

  T xi[IPT], yi[IPT], zi[IPT], mi[IPT], ax[IPT], ay[IPT], az[IPT];
  // float only: the origin the row's and the staged bodies' coordinates are taken relative
  // to BEFORE they are narrowed (row_origin); 0 in double, where nothing is subtracted
  double ox = 0.0, oy = 0.0, oz = 0.0;
  {
    const double *px, *pm;
    block_base<B>(a, I, &px, &pm);
    if constexpr (sizeof(T) == sizeof(float)) row_origin<IPT>(px, pm, S, tid, &ox, &oy, &oz);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int idx = k * 256 + tid;
      if constexpr (sizeof(T) == sizeof(float)) {
        xi[k] = static_cast<T>(px[idx] - ox);
        yi[k] = static_cast<T>(px[S + idx] - oy);
        zi[k] = static_cast<T>(px[2 * S + idx] - oz);
      } else {
        xi[k] = static_cast<T>(px[idx]);
        yi[k] = static_cast<T>(px[S + idx]);
        zi[k] = static_cast<T>(px[2 * S + idx]);
      }
      mi[k] = static_cast<T>(pm[idx]);
      ax[k] = ay[k] = az[k] = 0;
    }
  }
  // do all bodies of this row have one (finite, non-zero) mass?  Workgroup-uniform answer.
  __shared__ T m_row;
  if (tid == 0) m_row = mi[0];
  __syncthreads();
  const T m_ref = m_row;
  // not 0, inf or NaN -- nor so small that 1 / m_ref (kRowUnit) overflows
  const T inv_probe = 1 / m_ref;
  bool row_same = (m_ref != 0) && (m_ref - m_ref == 0) && (inv_probe - inv_probe == 0);
#pragma unroll
  for (int k = 0; k < IPT; k++) row_same = row_same && (mi[k] == m_ref);
  const bool row_uniform = __syncthreads_and(row_same) && !a.no_eq;
  // double: a uniform row keeps its i sums in units of 1/m_ref (kDeferred / kRowUnit)
  constexpr bool kDouble = sizeof(T) == sizeof(double);
  const T inv_m_ref = row_uniform ? inv_probe : 0;

  // distances [d_lo, d_hi] cut into dsplit chunks whose sizes differ by at most one
  const int nd = a.d_hi - a.d_lo + 1;
  const int d_begin = a.d_lo + static_cast<int>(static_cast<int64_t>(s) * nd / a.dsplit);
  const int d_end = a.d_lo + static_cast<int>(static_cast<int64_t>(s + 1) * nd / a.dsplit);

  for (int d = d_begin; d < d_end; d++) {
    if (2 * d == a.nb && I >= a.nb / 2) continue;  // the antipodal tile belongs to the low rows
    int J = I + d;
    if (J >= a.nb) J -= a.nb;
    const bool sym = (d != 0);
    const double *px, *pm;
    block_base<B>(a, J, &px, &pm);
    double* Cd = a.C + static_cast<int64_t>(d - a.c_base) * 3 * S + static_cast<int64_t>(il) * B;

    for (int c0 = 0; c0 < ng; c0 += G) {  // the workgroup's groups, one LDS fill at a time
      const int cg = ng - c0 < G ? ng - c0 : G;   // a multiple of 4 (jsplit <= IPT)
      const int j_lo = (g0 + c0) * 64, j_n = cg * 64;  // = bodies [j_lo, j_lo + j_n) of block J

      __syncthreads();  // every wave is done with the previous fill
      bool fill_same = true;  // ... and do the staged bodies have that same mass?
      for (int o = tid; o < j_n; o += 256) {
        const int idx = j_lo + o;
        const int e = o & 63, q = (o >> 6) * kGrp + e;
        P2<T> vxy, vzm;
        if constexpr (sizeof(T) == sizeof(float)) {  // relative to the row's origin, narrowed after
          vxy = P2<T>{static_cast<T>(px[idx] - ox), static_cast<T>(px[S + idx] - oy)};
          vzm = P2<T>{static_cast<T>(px[2 * S + idx] - oz), static_cast<T>(pm[idx])};
        } else {
          vxy = P2<T>{static_cast<T>(px[idx]), static_cast<T>(px[S + idx])};
          vzm = P2<T>{static_cast<T>(px[2 * S + idx]), static_cast<T>(pm[idx])};
        }
        fill_same = fill_same && (vzm.b == m_ref);
        xy[q] = vxy;
        zm[q] = vzm;
        fJ[0][q] = 0;
        fJ[1][q] = 0;
        fJ[2][q] = 0;
        if (e < kGrp - 64) {  // the group's first entries again behind its end
          xy[q + 64] = vxy;
          zm[q + 64] = vzm;
        }
      }
      const bool equal_mass = __syncthreads_and(fill_same) && row_uniform;  // (also the barrier after the fill)

#define NBODY_TILE(SYM_, MODE_) \
  tile_phases<IPT, SYM_, MODE_, T>(xy, zm, fJ, lane, w, cg, xi, yi, zi, mi, ax, ay, az, eps2, c15, c1875, m_ref, inv_m_ref)
      if (sym) {
        if (kDouble) {
          if (equal_mass) NBODY_TILE(true, kDeferred);
          else if (row_uniform) NBODY_TILE(true, kRowUnit);
          else NBODY_TILE(true, kGeneral);
        } else {
          if (equal_mass) NBODY_TILE(true, kSameMass);
          else NBODY_TILE(true, kGeneral);
        }
        for (int o = tid; o < j_n; o += 256) {
          const int idx = j_lo + o;
          const int q = (o >> 6) * kGrp + (o & 63);
          Cd[idx] = fJ[0][q];
          Cd[S + idx] = fJ[1][q];
          Cd[2 * S + idx] = fJ[2][q];
        }
      } else {  // the diagonal tile (1 of nb/2 + 1): no form of its own for equal masses
        if (kDouble && row_uniform) NBODY_TILE(false, kRowUnit);
        else NBODY_TILE(false, kGeneral);
      }
#undef NBODY_TILE
    }
  }

  double* Rs = a.R + static_cast<int64_t>(a.r_base + s * a.jsplit + static_cast<int>(blockIdx.z)) * 3 * S +
               static_cast<int64_t>(il) * B;
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 256 + tid;
    if (kDouble && row_uniform) {  // back from units of 1/m_ref
      ax[k] *= m_ref;
      ay[k] *= m_ref;
      az[k] *= m_ref;
    }
    Rs[idx] = ax[k];
    Rs[S + idx] = ay[k];
    Rs[2 * S + idx] = az[k];
  }
}

// Everything this rank has computed IN THIS BATCH of distances for global slot
// g = (slab, off), added in a fixed order: the own rows' i-side chunks R[r_first..nR) (own
// slab only) and, for every own row I that met g's block J at a distance d = J - I (mod nb)
// inside [max(d_lo,1), d_hi], the j-side sums C[d - c_base][I][.].
// The list of partials is dealt over kParts threads per slot (part p takes every kParts-th
// chunk and row), whose sums are then added in part order through LDS: 4x the loads in
// flight on these memory-bound kernels (a slot has up to ~300 partials of 24 bytes), still one
// fixed order of additions.
constexpr int kParts = 4;
template <int B>
__device__ __forceinline__ void sum_partials(const SymArgs& sa, int slab, int64_t off, bool own_slab, int part,
                                             double* ax, double* ay, double* az) {
  const int64_t S = sa.S;
  const int J = slab * sa.nbs + static_cast<int>(off / B);
  const int jl = static_cast<int>(off % B);
  double x = 0.0, y = 0.0, z = 0.0;
  // (unrolled so that several partials' loads are in flight; the order of the adds is unchanged)
  if (own_slab) {
#pragma unroll 4
    for (int s = sa.r_first + part; s < sa.nR; s += kParts) {
      const double* Rs = sa.R + static_cast<int64_t>(s) * 3 * S + off;
      x += Rs[0];
      y += Rs[S];
      z += Rs[2 * S];
    }
  }
  const int d_min = sa.d_lo > 1 ? sa.d_lo : 1;
#pragma unroll 4
  for (int il = part; il < sa.nbs; il += kParts) {
    const int I = sa.row0 + il;
    int d = J - I;
    if (d < 0) d += sa.nb;
    if (d < d_min || d > sa.d_hi || (2 * d == sa.nb && I >= sa.nb / 2)) continue;
    const double* Cd = sa.C + static_cast<int64_t>(d - sa.c_base) * 3 * S + static_cast<int64_t>(il) * B + jl;
    x += Cd[0];
    y += Cd[S];
    z += Cd[2 * S];
  }
  *ax = x; *ay = y; *az = z;
}

// The kParts partial sums of the workgroup's 64 slots, added in part order; valid for part 0.
// blockDim = (64, kParts): threadIdx.x = slot within the workgroup, threadIdx.y = part.
__device__ __forceinline__ void combine_parts(double* ax, double* ay, double* az) {
  __shared__ double red[3][kParts][64];
  const int l = threadIdx.x, p = threadIdx.y;
  red[0][p][l] = *ax;
  red[1][p][l] = *ay;
  red[2][p][l] = *az;
  __syncthreads();
  if (p == 0) {
    double x = red[0][0][l], y = red[1][0][l], z = red[2][0][l];
#pragma unroll
    for (int q = 1; q < kParts; q++) {
      x += red[0][q][l];
      y += red[1][q][l];
      z += red[2][q][l];
    }
    *ax = x; *ay = y; *az = z;
  }
}

// one slab, one batch: the partial sums are the complete sums -> integrate directly
template <int IPT>
__global__ __launch_bounds__(256) void sym_finish_kernel(const SymArgs sa, const nbody_mi355x_step_args a) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  const bool real = g < a.own_count;
  const int64_t S = sa.S;
  double ax = 0.0, ay = 0.0, az = 0.0;
  if (real) sum_partials<256 * IPT>(sa, 0, g, true, threadIdx.y, &ax, &ay, &az);
  combine_parts(&ax, &ay, &az);
  if (real && threadIdx.y == 0)
    integrate_and_store(a, g, a.pos_cur[g], a.pos_cur[S + g], a.pos_cur[2 * S + g], ax, ay, az);
}

// several slabs (ranks) and/or several batches: this rank's partial sums of this batch for
// EVERY slot into out[num_shards][3][S] (first batch: stored; later batches: added).
template <int IPT>
__global__ __launch_bounds__(256) void sym_gather_kernel(const SymArgs sa, int own_shard, int num_shards,
                                                         bool first, double* __restrict__ out) {
  const int64_t S = sa.S;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  const bool real = g < S * num_shards;
  const int slab = real ? static_cast<int>(g / S) : 0;
  const int64_t off = real ? g - static_cast<int64_t>(slab) * S : 0;
  double ax = 0.0, ay = 0.0, az = 0.0;
  if (real) sum_partials<256 * IPT>(sa, slab, off, slab == own_shard, threadIdx.y, &ax, &ay, &az);
  combine_parts(&ax, &ay, &az);
  if (!real || threadIdx.y != 0) return;
  double* o = out + static_cast<int64_t>(slab) * 3 * S + off;
  if (first) {
    o[0] = ax;
    o[S] = ay;
    o[2 * S] = az;
  } else {
    o[0] += ax;
    o[S] += ay;
    o[2 * S] += az;
  }
}

// one slab, several batches: integrate from the accumulated sums
__global__ __launch_bounds__(256) void sym_integrate_kernel(const double* __restrict__ acc,
                                                            const nbody_mi355x_step_args a) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (g >= a.own_count) return;
  const int64_t S = a.shard_slots;
  integrate_and_store(a, g, a.pos_cur[g], a.pos_cur[S + g], a.pos_cur[2 * S + g], acc[g], acc[S + g],
                      acc[2 * S + g]);
}

struct Plan {
  int ipt, nbs, nb, D, dsplit, jsplit, jlocal;
  int64_t bytes;      // preferred workspace: every distance in one batch, capped (kBatchCap)
  int64_t row_bytes;  // 3 * S doubles: one R chunk / one distance of C / the batch accumulator
};

// Above this the distances are processed in batches that reuse the C rows (the workspace
// of one batch would otherwise grow as N^2 / block: 206 GB at N = 4M on one GPU).
constexpr int64_t kBatchCap = int64_t{16} << 30;

// rows of the workspace that are not C: R chunks (jlocal of phase 1 + ds * js) and, on one
// slab, the accumulator that batches add into
int64_t fixed_rows(const Plan& p, int num_shards) {
  return p.jlocal + static_cast<int64_t>(p.dsplit) * p.jsplit + (num_shards == 1 ? 1 : 0);
}

// One candidate block size: fills *p and returns its modelled time (arbitrary units), or a
// negative value if it does not apply.
double plan_for(const nbody_mi355x_step_args& a, int ipt, Plan* p) {
  const int64_t B = 256 * ipt;
  if (a.num_shards < 1 || a.shard_slots <= 0 || a.shard_slots % B != 0) return -1.0;
  p->ipt = ipt;
  p->nbs = static_cast<int>(a.shard_slots / B);
  const int64_t nb = static_cast<int64_t>(p->nbs) * a.num_shards;
  if (nb < 2 || nb > (1 << 24)) return -1.0;
  p->nb = static_cast<int>(nb);
  p->D = p->nb / 2;
  // Work items: every row has D+1 equal tiles; a row is cut into ds distance chunks and a
  // tile into jsplit parts -> nbs*ds*jsplit workgroups, dispatched dynamically by the
  // hardware as others retire.  Measured on MI355X (N = 65 536 .. 262 144): throughput
  // rises with the number of chunks up to ~16-32 and is flat beyond, and what is left is
  // the tail of one work item; so: up to 32 chunks, and tiles split until there are ~32
  // items per CU.  Model time = work / 256 CUs + one item at full CU residency.
  if (ipt == 8 && a.precision == NBODY_MI355X_FP32) return -1.0;  // (not instantiated for float)
  const int per_cu = ipt >= 4 ? 2 : (ipt == 2 ? 4 : 8);
  // 6 DPP moves per IPT pairs on top of 96 cycles of FP64 work per pair
  const double pair_cost = ipt == 8 ? 0.97 : (ipt == 4 ? 1.00 : (ipt == 2 ? 1.05 : 1.15));
  const int nd = p->D + 1;
  const int ds_cap = ipt == 8 ? 32 : 16;  // (measured: smaller blocks gain nothing beyond 16)
  int ds = nd < ds_cap ? nd : ds_cap;
  if (const char* env = getenv("NBODY_MI355X_SYM_DSPLIT")) {  // tuning experiments only
    const int v = atoi(env);
    if (v > 0) ds = v;
  }
  if (ds > p->D) ds = p->D;
  if (ds < 1) ds = 1;
  p->dsplit = ds;
  int js = 1;
  while (js < ipt && static_cast<int64_t>(p->nbs) * nd * js < 64 * 256) js *= 2;
  if (const char* env = getenv("NBODY_MI355X_SYM_JSPLIT")) {  // tuning experiments only
    const int v = atoi(env);
    if ((v == 1 || v == 2 || v == 4 || v == 8) && v <= ipt) js = v;
  }
  p->jsplit = js;
  // phase 1 (multi-rank runs: the diagonal tiles, while the all-gather is in flight) has one
  // tile per own row: with few rows per rank the tiles are split so that the launch still
  // puts two workgroups on every CU (8 ranks at N = 1M: 64 rows -> 512 workgroups of 1/8 tile;
  // unsplit, that launch ran 64 workgroups for a whole tile's time: +0.6 ms per step)
  int jl = 1;
  while (jl < ipt && static_cast<int64_t>(p->nbs) * jl < 2 * 256) jl *= 2;
  p->jlocal = a.num_shards > 1 ? jl : 1;
  const double tile = static_cast<double>(B) * static_cast<double>(B) * pair_cost;
  const double best = static_cast<double>(p->nbs) * nd * tile / 256.0 + tile / js * per_cu;
  p->row_bytes = static_cast<int64_t>(sizeof(double)) * 3 * a.shard_slots;
  const int64_t full = p->row_bytes * (fixed_rows(*p, a.num_shards) + p->D);
  const int64_t least = p->row_bytes * (fixed_rows(*p, a.num_shards) + 1);
  p->bytes = full <= kBatchCap ? full : (least > kBatchCap ? least : kBatchCap);
  return best;
}

// The plan is a pure function of (shard_slots, num_shards, variant): the workspace query
// and the step agree on it.
bool make_plan(const nbody_mi355x_step_args& a, Plan* p) {
  if (a.variant == NBODY_MI355X_VARIANT_SYM1) return plan_for(a, 1, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM2) return plan_for(a, 2, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM4) return plan_for(a, 4, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM8) return plan_for(a, 8, p) >= 0.0;
  // automatic: the largest block that still leaves enough tiles per rank to keep 512
  // resident workgroups busy to the end.  Bigger blocks cost fewer DPP moves per pair, but
  // a launch of few big tiles is mostly ramp and tail.  Thresholds from sweeps on MI355X
  // (Ginteractions/s, best block per size): 16 384 bodies: 256-body blocks 1665 (1024: 1326);
  // 32 768: 512 2371 (2048: 2264); 65 536: 1024 2802 (2048: 2744); 131 072: 1024 2880
  // (2048: 2783); 262 144 and up: 2048.  The model value of plan_for only breaks ties.
  for (int ipt : {8, 4, 2, 1}) {
    Plan c;
    if (plan_for(a, ipt, &c) < 0.0) continue;
    const int64_t tiles = static_cast<int64_t>(c.nbs) * (c.D + 1);
    if (ipt > 1 && tiles < (ipt == 8 ? 8192 : 2048)) {
      Plan smaller;  // only if a smaller block applies at all (shard_slots divisible, >= 2 blocks)
      bool any = false;
      for (int q = ipt / 2; q >= 1 && !any; q /= 2) any = plan_for(a, q, &smaller) >= 0.0;
      if (any) continue;
    }
    *p = c;
    return true;
  }
  return false;
}

template <int IPT, typename T>
hipError_t launch(SymArgs sa, const nbody_mi355x_step_args& a, hipStream_t stream, int batch, double* accum) {
  hipError_t e = hipSuccess;
  // phase 1 = what needs the own slab only (the diagonal tiles), kept in R[0];
  // phase 2 = the rest, then output; phase 0 = both
  if (a.phase == 1) {
    sa.d_lo = 0; sa.d_hi = 0; sa.dsplit = 1; sa.jsplit = sa.jlocal; sa.r_base = 0; sa.c_base = 1;
    hipLaunchKernelGGL((sym_tiles_kernel<IPT, T>), dim3(sa.nbs, 1, sa.jlocal), dim3(256), 0, stream, sa);
    return hipGetLastError();
  }
  const int d_first = a.phase == 2 ? 1 : 0;
  sa.r_base = sa.jlocal;
  double* out = a.acc_out != nullptr ? a.acc_out : accum;
  for (int lo = d_first, hi; lo <= sa.D; lo = hi + 1) {
    // `batch` rows of C hold the distances max(lo,1) .. hi (d = 0 needs none)
    hi = (lo > 1 ? lo : 1) + batch - 1;
    if (hi > sa.D) hi = sa.D;
    const bool first = lo == d_first, only = first && hi == sa.D;
    sa.d_lo = lo;
    sa.d_hi = hi;
    sa.c_base = lo > 1 ? lo : 1;
    sa.r_first = (first && a.phase == 2) ? 0 : sa.jlocal;  // the local phase's chunks join the first batch
    hipLaunchKernelGGL((sym_tiles_kernel<IPT, T>), dim3(sa.nbs, sa.dsplit, sa.jsplit), dim3(256), 0, stream, sa);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (only && a.acc_out == nullptr) {
      hipLaunchKernelGGL((sym_finish_kernel<IPT>), dim3(static_cast<unsigned>((a.own_count + 63) / 64)),
                         dim3(64, kParts), 0, stream, sa, a);
      return hipGetLastError();
    }
    const int64_t total = sa.S * a.num_shards;
    hipLaunchKernelGGL((sym_gather_kernel<IPT>), dim3(static_cast<unsigned>((total + 63) / 64)), dim3(64, kParts), 0,
                       stream, sa, a.own_shard, a.num_shards, first, out);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.acc_out == nullptr) {
    hipLaunchKernelGGL(sym_integrate_kernel, dim3(static_cast<unsigned>((a.own_count + 255) / 256)), dim3(256),
                       0, stream, accum, a);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace

namespace nbody_internal {

bool sym_supported(const nbody_mi355x_step_args& a, int64_t* preferred_bytes, int64_t* least_bytes) {
  Plan p;
  if (!make_plan(a, &p)) return false;
  if (preferred_bytes) *preferred_bytes = p.bytes;
  if (least_bytes) *least_bytes = p.row_bytes * (fixed_rows(p, a.num_shards) + 1);
  return true;
}

int sym_step(const nbody_mi355x_step_args& a, hipStream_t stream) {
  Plan p;
  if (!make_plan(a, &p)) {
    set_error("symmetric step: shard_slots must be a multiple of the block (256 .. 2048 by variant) "
              "and the system at least two blocks");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a.acc_in) {
    set_error("symmetric step: acc_in is for one-sided steps");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a.phase < 0 || a.phase > 2) {
    set_error("symmetric step: phase %d outside [0,2]", a.phase);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a.num_shards > 1 && a.phase != 1 && a.acc_out == nullptr) {
    set_error("symmetric step over %d slabs: acc_out [num_shards][3][shard_slots] is required "
              "(the partial sums must be reduce-scattered before integrating)", a.num_shards);
    return NBODY_MI355X_ERR_INVALID;
  }
  const int64_t fixed = fixed_rows(p, a.num_shards);
  const int64_t rows = a.workspace ? a.workspace_bytes / p.row_bytes : 0;
  if (rows < fixed + 1) {
    set_error("symmetric step: workspace of at least %lld bytes needed (%lld preferred), %lld given",
              (long long)(p.row_bytes * (fixed + 1)), (long long)p.bytes, (long long)a.workspace_bytes);
    return NBODY_MI355X_ERR_INVALID;
  }
  int64_t batch = rows - fixed;  // distances whose j-side sums fit at once
  if (batch > p.D) batch = p.D;
  SymArgs sa;
  sa.pos = a.pos_cur;
  sa.mass = a.mass;
  sa.R = static_cast<double*>(a.workspace);
  sa.C = sa.R + static_cast<int64_t>(p.jlocal + p.dsplit * p.jsplit) * 3 * a.shard_slots;
  double* accum = a.num_shards == 1 ? sa.C + batch * 3 * a.shard_slots : nullptr;
  sa.S = a.shard_slots;
  sa.nbs = p.nbs;
  sa.nb = p.nb;
  sa.row0 = a.own_shard * p.nbs;
  sa.D = p.D;
  sa.dsplit = p.dsplit;
  sa.jsplit = p.jsplit;
  sa.jlocal = p.jlocal;
  sa.r_first = p.jlocal;
  sa.nR = p.jlocal + p.dsplit * p.jsplit;
  sa.d_lo = sa.d_hi = sa.r_base = 0;
  sa.c_base = 1;
  sa.eps2 = a.eps2;
  const char* no_eq = getenv("NBODY_MI355X_SYM_NOEQ");  // tests: compare the two forms bit for bit
  sa.no_eq = (no_eq && atoi(no_eq) != 0) ? 1 : 0;
  const int b = static_cast<int>(batch);
  hipError_t e;
  if (a.precision == NBODY_MI355X_FP32) {
    e = p.ipt == 1 ? launch<1, float>(sa, a, stream, b, accum)
        : p.ipt == 2 ? launch<2, float>(sa, a, stream, b, accum) : launch<4, float>(sa, a, stream, b, accum);
  } else {
    e = p.ipt == 1 ? launch<1, double>(sa, a, stream, b, accum)
        : p.ipt == 2 ? launch<2, double>(sa, a, stream, b, accum)
        : p.ipt == 4 ? launch<4, double>(sa, a, stream, b, accum) : launch<8, double>(sa, a, stream, b, accum);
  }
  if (e != hipSuccess) {
    set_error("symmetric step launch: %s", hipGetErrorString(e));
    return NBODY_MI355X_ERR_LAUNCH;
  }
  return NBODY_MI355X_OK;
}

}  // namespace nbody_internal

// ------------------------------------------------------------------ no allocation
// nbody_mi355x_step_f64_static: the pair-symmetric small step for callers that cannot
// allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc).  The partial-sum workspace is this file-scope device array; every GPU the
// library's code is loaded on holds its own copy, at its own address, until the process ends.
// In host code the array's NAME is only a handle for hipGetSymbolAddress -- the address a
// kernel can use is what that call returns, once per device, with that device current.
static __device__ __align__(256) unsigned char scratch[NBODY_MI355X_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxStaticDevices = 64;
std::mutex g_static_mutex;
void* g_static_ptr[kMaxStaticDevices];  // per device: where `scratch` lives there (nullptr: not yet asked)

// the current device's copy of `scratch` (resolved on the first call on that device)
int static_scratch(void** ptr) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxStaticDevices) {
    (void)hipGetLastError();
    set_error("step_f64_static: no current device (%s)", e != hipSuccess ? hipGetErrorString(e) : "ordinal out of range");
    return NBODY_MI355X_ERR_NO_DEVICE;
  }
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (g_static_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      set_error("step_f64_static: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
      return NBODY_MI355X_ERR_NO_DEVICE;
    }
    g_static_ptr[dev] = p;
  }
  *ptr = g_static_ptr[dev];
  return NBODY_MI355X_OK;
}
}  // namespace

extern "C" int nbody_mi355x_step_f64_static(const nbody_mi355x_step_args* args, void* stream) {
  int rc = nbody_internal::check_step_args(args);
  if (rc != NBODY_MI355X_OK) return rc;
  nbody_mi355x_step_args a = *args;
  const bool sym_variant = a.variant == NBODY_MI355X_VARIANT_SYM || a.variant == NBODY_MI355X_VARIANT_SYM1 ||
                           a.variant == NBODY_MI355X_VARIANT_SYM2 || a.variant == NBODY_MI355X_VARIANT_SYM4 ||
                           a.variant == NBODY_MI355X_VARIANT_SYM8;
  if (a.variant != 0 && !sym_variant) {
    set_error("step_f64_static: variant %d is a one-sided kernel, which needs no scratch: call nbody_mi355x_step_f64",
              a.variant);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (!sym_variant) a.variant = NBODY_MI355X_VARIANT_SYM;
  int64_t preferred = 0, least = 0;
  if (!nbody_internal::sym_supported(a, &preferred, &least)) {
    set_error("step_f64_static: the pair-symmetric step does not apply to %d slab(s) of %lld slots "
              "(at least two blocks of 256 slots are needed)", a.num_shards, (long long)a.shard_slots);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (least > static_cast<int64_t>(NBODY_MI355X_STATIC_SCRATCH_BYTES)) {
    set_error("step_f64_static: %d slab(s) of %lld slots need a workspace of at least %lld bytes; the static "
              "scratch holds %lld (NBODY_MI355X_STATIC_SCRATCH_BYTES): use nbody_mi355x_step_f64 with a workspace",
              a.num_shards, (long long)a.shard_slots, (long long)least, (long long)NBODY_MI355X_STATIC_SCRATCH_BYTES);
    return NBODY_MI355X_ERR_INVALID;
  }
  void* ptr = nullptr;
  rc = static_scratch(&ptr);
  if (rc != NBODY_MI355X_OK) return rc;
  // Nothing in the workspace has to start at zero: every R chunk, C row and accumulator entry
  // the finishing kernels read is stored by a kernel of THIS call first (launch()), so what
  // the previous call -- of any size -- left behind is never seen.
  a.workspace = ptr;
  a.workspace_bytes = static_cast<int64_t>(NBODY_MI355X_STATIC_SCRATCH_BYTES);
  return nbody_internal::sym_step(a, static_cast<hipStream_t>(stream));
}
