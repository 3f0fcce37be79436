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
// This file holds the WORKGROUP-tile kernel (the large systems), the finishing kernels and the
// launch plans of both pair-symmetric families; the WAVE-tile kernel for the mid sizes, which
// shares the arithmetic (nbody_sym_device.hpp) and everything downstream of the tiles, is
// nbody_symw.hip.  make_plan picks the family and the block size.
//
// Launch shape of the workgroup tiles (plan_for below): grid = (own rows, dsplit distance chunks, jsplit parts of
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

#include "nbody_internal.h"
#include "nbody_sym_device.hpp"

using namespace nbody_device;
using namespace nbody_sym;
using nbody_internal::set_error;

namespace {

// j groups of a tile that are staged in LDS at a time: 16 groups = 1024 bodies = 63 KB in
// double, two workgroups per CU.  Blocks of up to 1024 bodies (IPT <= 4) are staged whole;
// a 2048-body block (IPT = 8) goes through in two halves.
constexpr int kStageGroups = 16;
template <int IPT>
constexpr int staged_groups() { return 4 * IPT < kStageGroups ? 4 * IPT : kStageGroups; }

// The workgroup's j groups of the staged tile against this wave's 64*IPT bodies (group_steps,
// nbody_sym_device.hpp).  SYM = false is the diagonal tile (a block against itself): i side
// only, no barriers needed.  MODE: how the masses enter, chosen per LDS fill.
template <int IPT, bool SYM, int MODE, typename T>
__device__ __forceinline__ void tile_phases(const P2<T>* xy, const P2<T>* zm, T (*fJ)[kGrp * staged_groups<IPT>()],
                                            int lane, int w, int ng, const T (&xi)[IPT], const T (&yi)[IPT],
                                            const T (&zi)[IPT], const T (&mi)[IPT], T (&ax)[IPT],
                                            T (&ay)[IPT], T (&az)[IPT], T eps2, T c15, T c1875, T m_ref, T inv_m_ref) {
  // the staged groups 0 .. ng-1 (ng a multiple of 4)
  for (int ph = 0; ph < ng; ph++) {
    const int grp = (ph + (ng >> 2) * w) % ng;  // the 4 waves sit on 4 different groups
    const int gb = grp * kGrp;
    T fx, fy, fz;
    group_steps<IPT, SYM, MODE, T>(xy, zm, gb, lane, xi, yi, zi, mi, ax, ay, az, eps2, c15, c1875, inv_m_ref, 0, 8, &fx,
                                   &fy, &fz);
    if (SYM) {  // the sums are home: lane l holds body l of the group
      if constexpr (MODE == kDeferred || MODE == kDeferredFast || MODE == kRowUnit) {
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

template <int IPT, typename T, bool FAST = false>
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
  T c15 = static_cast<T>(1.5), c1875 = static_cast<T>(1.875);
  asm volatile("" : "+v"(c15), "+v"(c1875));

  T xi[IPT], yi[IPT], zi[IPT], mi[IPT], ax[IPT], ay[IPT], az[IPT];
  // float only: the origin the row's and the staged bodies' coordinates are taken relative
  // to, and the power of two they are multiplied by, BEFORE they are narrowed (row_origin);
  // 0 and 1 in double, where nothing is subtracted or scaled
  double ox = 0.0, oy = 0.0, oz = 0.0, sc = 1.0;
  {
    const double *px, *pm;
    block_base<B>(a, I, &px, &pm);
    if constexpr (sizeof(T) == sizeof(float)) row_origin<IPT>(px, pm, S, tid, &ox, &oy, &oz, &sc);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int idx = k * 256 + tid;
      if constexpr (sizeof(T) == sizeof(float)) {
        xi[k] = static_cast<T>((px[idx] - ox) * sc);
        yi[k] = static_cast<T>((px[S + idx] - oy) * sc);
        zi[k] = static_cast<T>((px[2 * S + idx] - oz) * sc);
      } else {
        xi[k] = static_cast<T>(px[idx]);
        yi[k] = static_cast<T>(px[S + idx]);
        zi[k] = static_cast<T>(px[2 * S + idx]);
      }
      mi[k] = static_cast<T>(pm[idx]);
      ax[k] = ay[k] = az[k] = 0;
    }
  }
  // double: 2^-300 stands in for "no softening" (nbody_device.hpp); float: plain 0 (r^2 == 0
  // is dealt with in the walk, nbody_sym_device.hpp) or, softened, eps2 in the row's scaled units
  const T eps2 = static_cast<T>((a.eps2 > 0.0) ? a.eps2 * sc * sc : (sizeof(T) == sizeof(double) ? kTinyR2 : 0.0));
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
  // float: a uniform row's i sums are in units of its mass AND of the scale: staged masses
  // are divided by m_ref (so the general form's m_j r^-3 lands in the same unit), and the
  // sums are converted when they are stored, in double
  const double m_unit = (!kDouble && row_uniform) ? static_cast<double>(m_ref) : 1.0;
  const double stage_m = 1.0 / m_unit;
  const double unit_i = sc * sc * m_unit;

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
          vxy = P2<T>{static_cast<T>((px[idx] - ox) * sc), static_cast<T>((px[S + idx] - oy) * sc)};
          vzm = P2<T>{static_cast<T>((px[2 * S + idx] - oz) * sc), static_cast<T>(pm[idx] * stage_m)};
        } else {
          vxy = P2<T>{static_cast<T>(px[idx]), static_cast<T>(px[S + idx])};
          vzm = P2<T>{static_cast<T>(px[2 * S + idx]),   // (a uniform row stages m_j / m_ref: kRowUnit)
                      static_cast<T>(row_uniform ? pm[idx] * inv_m_ref : pm[idx])};
        }
        fill_same = fill_same && (static_cast<T>(pm[idx]) == m_ref);
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
          if (equal_mass) {
            if constexpr (FAST) NBODY_TILE(true, kDeferredFast);
            else NBODY_TILE(true, kDeferred);
          } else if (row_uniform) NBODY_TILE(true, kRowUnit);
          else NBODY_TILE(true, kGeneral);
        } else {
          if (equal_mass) NBODY_TILE(true, kSameMass);
          else NBODY_TILE(true, kGeneral);
        }
        // float: out of the scaled units (of the common mass too where the fill had one)
        const double unit_j = kDouble ? 1.0 : sc * sc * (equal_mass ? static_cast<double>(m_ref) : 1.0);
        for (int o = tid; o < j_n; o += 256) {
          const int idx = j_lo + o;
          const int q = (o >> 6) * kGrp + (o & 63);
          if constexpr (kDouble) {
            Cd[idx] = fJ[0][q];
            Cd[S + idx] = fJ[1][q];
            Cd[2 * S + idx] = fJ[2][q];
          } else {
            Cd[idx] = fJ[0][q] * unit_j;
            Cd[S + idx] = fJ[1][q] * unit_j;
            Cd[2 * S + idx] = fJ[2][q] * unit_j;
          }
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
    if constexpr (kDouble) {
      Rs[idx] = ax[k];
      Rs[S + idx] = ay[k];
      Rs[2 * S + idx] = az[k];
    } else {
      Rs[idx] = ax[k] * unit_i;
      Rs[S + idx] = ay[k] * unit_i;
      Rs[2 * S + idx] = az[k] * unit_i;
    }
  }
}

// Everything this rank has computed IN THIS BATCH of distances for global slot
// g = (slab, off), added in a fixed order: the own rows' i-side chunks R[r_first..nR) (own
// slab only) and, for every own row I that met g's block J at a distance d = J - I (mod nb)
// inside [max(d_lo,1), d_hi], the j-side sums C[d - c_base][I][.].
// The list of partials is dealt over PARTS threads per slot (part p takes every PARTS-th
// chunk and row), whose sums are then added in part order through LDS: more loads in flight on
// these memory-bound kernels (a slot has up to ~300 partials of 24 bytes), still one fixed order
// of additions.  PARTS by launch size (parts_for): a small system's finishing launch is one
// round of waves whose time is the latency of its dependent loads -- 16 384 bodies: 41 partials
// per slot, 4 parts: 6.9 us; 16 parts: every thread two or three loads deep.
template <int PARTS>
__device__ __forceinline__ void sum_partials(const SymArgs& sa, int slab, int64_t off, bool own_slab, int part,
                                             double* ax, double* ay, double* az) {
  const int64_t S = sa.S;
  const int B = 1 << sa.bshift;  // block size of the kernel family that produced the partials
  const int J = slab * sa.nbs + static_cast<int>(off >> sa.bshift);
  const int jl = static_cast<int>(off & (B - 1));
  double x = 0.0, y = 0.0, z = 0.0;
  // (unrolled so that several partials' loads are in flight; the order of the adds is unchanged)
  if (own_slab) {
#pragma unroll 4
    for (int s = sa.r_first + part; s < sa.nR; s += PARTS) {
      const double* Rs = sa.R + static_cast<int64_t>(s) * 3 * S + off;
      x += Rs[0];
      y += Rs[S];
      z += Rs[2 * S];
    }
  }
  const int d_min = sa.d_lo > 1 ? sa.d_lo : 1;
#pragma unroll 4
  for (int il = part; il < sa.nbs; il += PARTS) {
    const int I = sa.row0 + il;
    int d = J - I;
    if (d < 0) d += sa.nb;
    if (d < d_min || d > sa.d_hi || (2 * d == sa.nb && I >= sa.nb / 2)) continue;
    const double* Cd = sa.C + static_cast<int64_t>(d - sa.c_base) * 3 * S + (static_cast<int64_t>(il) << sa.bshift) + jl;
    x += Cd[0];
    y += Cd[S];
    z += Cd[2 * S];
    if (sa.q > 0) {
      // wave tiles: was this entry of row I cut between waves?  Its octets are [8 e, 8 e + 8) of
      // the row's list; wave k begins at octet k q (the first ws waves) or ws q + (k - ws) qb
      // (those behind them).  The first boundary behind 8 e is that of wave k = floor(8 e / q) + 1
      // (or its like in the second range); every boundary that falls before 8 e + 8 (and inside
      // the row's list: the high rows' lists are shorter by the antipodal tile) begins a part
      // whose sums are in Cb[row][that wave].  q >= 8: at most one such part, except among the
      // later waves of an uneven dealing, whose shares may be smaller than an entry.
      int dh = sa.d_hi;
      if (2 * dh == sa.nb && I >= sa.nb / 2) dh--;
      const unsigned o = 8u * static_cast<unsigned>(((d - sa.d_lo) << (sa.bshift - 6)) + (jl >> 6));
      const unsigned split = static_cast<unsigned>(sa.ws) * static_cast<unsigned>(sa.q);
      unsigned k, b;
      if (o < split) {
        k = __umulhi(o, sa.q_magic) + 1u;
        b = k * static_cast<unsigned>(sa.q);
      } else {
        const unsigned r = __umulhi(o - split, sa.qb_magic) + 1u;
        k = static_cast<unsigned>(sa.ws) + r;
        b = split + r * static_cast<unsigned>(sa.qb);
      }
      unsigned end = static_cast<unsigned>((dh - sa.d_lo + 1) << (sa.bshift - 6)) * 8u;
      end = end < o + 8u ? end : o + 8u;
      for (; b < end; k++) {
        const double* Cb = sa.Cb + (static_cast<int64_t>(il) * sa.wstride + k) * 192 + (jl & 63);
        x += Cb[0];
        y += Cb[64];
        z += Cb[128];
        b += static_cast<unsigned>(k < static_cast<unsigned>(sa.ws) ? sa.q : sa.qb);
      }
    }
  }
  *ax = x; *ay = y; *az = z;
}

// The PARTS partial sums of the workgroup's 64 slots, added in part order; valid for part 0.
// blockDim = (64, PARTS): threadIdx.x = slot within the workgroup, threadIdx.y = part.
template <int PARTS>
__device__ __forceinline__ void combine_parts(double* ax, double* ay, double* az) {
  __shared__ double red[3][PARTS][64];
  const int l = threadIdx.x, p = threadIdx.y;
  red[0][p][l] = *ax;
  red[1][p][l] = *ay;
  red[2][p][l] = *az;
  __syncthreads();
  if (p == 0) {
    double x = red[0][0][l], y = red[1][0][l], z = red[2][0][l];
#pragma unroll
    for (int q = 1; q < PARTS; q++) {
      x += red[0][q][l];
      y += red[1][q][l];
      z += red[2][q][l];
    }
    *ax = x; *ay = y; *az = z;
  }
}

// one slab, one batch: the partial sums are the complete sums -> integrate directly
template <int PARTS>
__global__ __launch_bounds__(64 * PARTS) void sym_finish_kernel(const SymArgs sa, const nbody_mi355x_step_args a) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  const bool real = g < a.own_count;
  const int64_t S = sa.S;
  double ax = 0.0, ay = 0.0, az = 0.0;
  if (real) sum_partials<PARTS>(sa, 0, g, true, threadIdx.y, &ax, &ay, &az);
  combine_parts<PARTS>(&ax, &ay, &az);
  if (real && threadIdx.y == 0)
    integrate_and_store(a, g, a.pos_cur[g], a.pos_cur[S + g], a.pos_cur[2 * S + g], ax, ay, az);
}

// several slabs (ranks) and/or several batches: this rank's partial sums of this batch for
// EVERY slot into out[num_shards][3][S] (first batch: stored; later batches: added).
template <int PARTS>
__global__ __launch_bounds__(64 * PARTS) void sym_gather_kernel(const SymArgs sa, int own_shard, int num_shards,
                                                                bool first, double* __restrict__ out) {
  const int64_t S = sa.S;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  const bool real = g < S * num_shards;
  const int slab = real ? static_cast<int>(g / S) : 0;
  const int64_t off = real ? g - static_cast<int64_t>(slab) * S : 0;
  double ax = 0.0, ay = 0.0, az = 0.0;
  if (real) sum_partials<PARTS>(sa, slab, off, slab == own_shard, threadIdx.y, &ax, &ay, &az);
  combine_parts<PARTS>(&ax, &ay, &az);
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

int env_int(const char* name) {  // tuning experiments only
  const char* env = getenv(name);
  return env ? atoi(env) : 0;
}

// threads per slot of the finishing kernels: enough waves to fill the chip with loads in flight
// (1024 SIMDs x several waves), fewer parts where there are slots enough
int parts_for(int64_t slots) {
  if (const int v = env_int("NBODY_MI355X_SYM_PARTS"); v == 4 || v == 8 || v == 16) return v;  // (A/B tests)
  // (beyond ~32 768 slots the launch is bandwidth-bound whatever the parts: 65 536 bodies, 116 MB
  // of partials: 20.0 us with 4 parts, 22.1 with 8)
  return slots <= 32768 ? 16 : 4;
}
void launch_finish(int parts, unsigned groups, hipStream_t stream, const SymArgs& sa, const nbody_mi355x_step_args& a) {
  if (parts == 16) hipLaunchKernelGGL(sym_finish_kernel<16>, dim3(groups), dim3(64, 16), 0, stream, sa, a);
  else if (parts == 8) hipLaunchKernelGGL(sym_finish_kernel<8>, dim3(groups), dim3(64, 8), 0, stream, sa, a);
  else hipLaunchKernelGGL(sym_finish_kernel<4>, dim3(groups), dim3(64, 4), 0, stream, sa, a);
}
void launch_gather(int parts, unsigned groups, hipStream_t stream, const SymArgs& sa, int own_shard, int num_shards,
                   bool first, double* out) {
  if (parts == 16)
    hipLaunchKernelGGL(sym_gather_kernel<16>, dim3(groups), dim3(64, 16), 0, stream, sa, own_shard, num_shards, first, out);
  else if (parts == 8)
    hipLaunchKernelGGL(sym_gather_kernel<8>, dim3(groups), dim3(64, 8), 0, stream, sa, own_shard, num_shards, first, out);
  else
    hipLaunchKernelGGL(sym_gather_kernel<4>, dim3(groups), dim3(64, 4), 0, stream, sa, own_shard, num_shards, first, out);
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
  bool wave;  // wave tiles (nbody_symw.hip): blocks of 64 * ipt slots; else workgroup tiles: 256 * ipt
  int ipt, nbs, nb, D, dsplit, jsplit, jlocal;
  int rchunks;  // R chunks the main launch writes per row (dsplit * jsplit / wave tiles: workgroups per row)
  int cbrows;   // wave tiles: rows of the workspace that hold Cb (second parts of cut entries)
  int ca;       // wave tiles: chunks of a row whose workgroups are dispatched first (uneven dealing, set_octets); 0: even
  double rho;   //   ... and the share of a later wave relative to theirs
  int64_t block;      // slots per block
  int64_t bytes;      // preferred workspace: every distance in one batch, capped (kBatchCap)
  int64_t row_bytes;  // 3 * S doubles: one R chunk / one distance of C / the batch accumulator
};

// Above this the distances are processed in batches that reuse the C rows (the workspace
// of one batch would otherwise grow as N^2 / block: 7.3 GB at N = 1M, 206 GB at N = 4M on one
// GPU).  Every batch is a launch of its own with a ramp and a tail; measured at N = 1M
// (ms per step by cap): 1 GiB 362.6, 2 GiB 352.7, 4 GiB 351.5, unbatched (7.3 GB) 351.1.
// Larger systems need room for a few distances per chunk of a row to keep the launches long:
// the cap is never below four times the plan's minimum.
constexpr int64_t kBatchCap = int64_t{4} << 30;

// rows of the workspace that are not C: R chunks (jlocal of phase 1 + the main launch's) and,
// on one slab, the accumulator that batches add into
int64_t fixed_rows(const Plan& p, int num_shards) {
  return p.jlocal + static_cast<int64_t>(p.rchunks) + p.cbrows + (num_shards == 1 ? 1 : 0);
}

void plan_bytes(const nbody_mi355x_step_args& a, Plan* p) {
  p->row_bytes = static_cast<int64_t>(sizeof(double)) * 3 * a.shard_slots;
  const int64_t full = p->row_bytes * (fixed_rows(*p, a.num_shards) + p->D);
  const int64_t least = p->row_bytes * (fixed_rows(*p, a.num_shards) + 1);
  int64_t cap = kBatchCap > 4 * least ? kBatchCap : 4 * least;
  if (const int mb = env_int("NBODY_MI355X_SYM_WS_CAP_MB"); mb > 0) cap = static_cast<int64_t>(mb) << 20;
  p->bytes = full <= cap ? full : (least > cap ? least : cap / p->row_bytes * p->row_bytes);  // whole rows
}

// One candidate block size of the workgroup-tile kernels: fills *p and returns its modelled
// time (arbitrary units), or a negative value if it does not apply.
double plan_for(const nbody_mi355x_step_args& a, int ipt, Plan* p) {
  const int64_t B = 256 * ipt;
  if (a.num_shards < 1 || a.shard_slots <= 0 || a.shard_slots % B != 0) return -1.0;
  p->wave = false;
  p->cbrows = 0;
  p->ca = 0;
  p->rho = 1.0;
  p->block = B;
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
  if (ipt == 8 && a.precision == NBODY_MI355X_FP32 && env_int("NBODY_MI355X_F32_NO8")) return -1.0;  // (A/B tests)
  const int per_cu = ipt >= 4 ? 2 : (ipt == 2 ? 4 : 8);
  // 6 DPP moves per IPT pairs on top of 96 cycles of FP64 work per pair
  const double pair_cost = ipt == 8 ? 0.97 : (ipt == 4 ? 1.00 : (ipt == 2 ? 1.05 : 1.15));
  const int nd = p->D + 1;
  const int ds_cap = ipt == 8 ? 32 : 16;  // (measured: smaller blocks gain nothing beyond 16)
  int ds = nd < ds_cap ? nd : ds_cap;
  if (const int v = env_int("NBODY_MI355X_SYM_DSPLIT"); v > 0) ds = v;
  if (ds > p->D) ds = p->D;
  if (ds < 1) ds = 1;
  p->dsplit = ds;
  int js = 1;
  while (js < ipt && static_cast<int64_t>(p->nbs) * nd * js < 64 * 256) js *= 2;
  if (const int v = env_int("NBODY_MI355X_SYM_JSPLIT"); (v == 1 || v == 2 || v == 4 || v == 8) && v <= ipt) js = v;
  p->jsplit = js;
  p->rchunks = ds * js;
  // phase 1 (multi-rank runs: the diagonal tiles, while the all-gather is in flight) has one
  // tile per own row: with few rows per rank the tiles are split so that the launch still
  // puts two workgroups on every CU (8 ranks at N = 1M: 64 rows -> 512 workgroups of 1/8 tile;
  // unsplit, that launch ran 64 workgroups for a whole tile's time: +0.6 ms per step)
  int jl = 1;
  while (jl < ipt && static_cast<int64_t>(p->nbs) * jl < 2 * 256) jl *= 2;
  p->jlocal = a.num_shards > 1 ? jl : 1;
  const double tile = static_cast<double>(B) * static_cast<double>(B) * pair_cost;
  const double best = static_cast<double>(p->nbs) * nd * tile / 256.0 + tile / js * per_cu;
  plan_bytes(a, p);
  return best;
}

// The wave-tile kernels (nbody_symw.hip) with blocks of 64 * ipt slots.  A row's work is the
// list of its (tile, group) entries -- (D+1) tiles of ipt groups of 64 j bodies, 8 octets of
// rotation steps each -- dealt in OCTETS to the 4 * chunks waves of the row's `chunks`
// workgroups (all holding the row's bodies), so every wave of the launch has the same work to
// within one octet (round 2 dealt whole entries: level only where they happened to divide
// evenly over the chip).  What is left to choose is `chunks`, by a score fitted to a sweep on the
// chip (tools/sweep_wave_plan.py -> profiles/r03_wave_tile_sweep.csv: 12 sizes x 3 block sizes x
// 15 values of chunks; equal work items make the dispatch-round model accurate -- round 2's
// whole-entry items defeated it):
//   * U = wgs / (256 ceil(wgs / 256)), wgs = nbs * chunks: the share of the last dispatch round
//     that is used -- the launch is level where wgs is (just under) a multiple of the 256 CUs;
//   * octets / (4 chunks q), q = ceil(octets / (4 chunks)): what rounding a wave's share up to
//     whole octets wastes;
//   * - 0.0005 per chunk: an R row (24 bytes per slot) to store and to add again;
//   * - 0.02 / rounds: the end of a launch of few, long work items is ragged (the two waves of
//     a SIMD do not finish together), smaller items even it out;
//   * - 0.01 under two workgroups per CU: a lone wave per SIMD leaves its loads exposed.
// Against the best of the sweep this rule loses 0.5 % on average, 1.7 % at worst over the sizes
// it decides.  NBODY_MI355X_SYMW_CHUNKS overrides it.
constexpr int64_t kWaveTileMaxSlots = 262143;  // wave tiles up to here, workgroup tiles beyond
// Uneven dealing (set_octets): a later wave's share of octets relative to an older wave's, in percent
// (0: even dealing), from a sweep on the chip (tools/sweep_skew.sh, 10 ... 40 %: flat from 10 to 20,
// worse above) and from the trace: a SIMD's arbiter serves its older wave first, which then runs at
// 0.87 of the SIMD's rate whatever its neighbour does; the younger gets the 0.13 that is left.
constexpr int kSkewPercent = 15;
constexpr int kSkewPercentSmall = 25;  // blocks of up to 256 bodies, and the FP32 tier: fewer independent chains per
                                       // wave (or packed ones) leave the neighbour more -- best of 10 ... 50 %
int set_octets(SymArgs* sa, int64_t octets, int chunks, int old_chunks = 0, double rho = 1.0);  // (below)
constexpr int64_t kSkewMinOctets = 20;  // octets per older wave from which the dealing is uneven
bool plan_wave(const nbody_mi355x_step_args& a, int ipt, Plan* p) {
  const int64_t B = 64 * ipt;
  if (a.num_shards < 1 || a.shard_slots <= 0 || a.shard_slots % B != 0) return false;
  if (ipt == 8 && a.precision == NBODY_MI355X_FP32 && env_int("NBODY_MI355X_F32_NO8")) return false;  // (A/B tests)
  p->wave = true;
  p->block = B;
  p->ipt = ipt;
  p->nbs = static_cast<int>(a.shard_slots / B);
  const int64_t nb = static_cast<int64_t>(p->nbs) * a.num_shards;
  if (nb < 2 || nb > (1 << 24)) return false;
  p->nb = static_cast<int>(nb);
  p->D = p->nb / 2;
  const int64_t octets = static_cast<int64_t>(p->D + 1) * ipt * 8;  // per row (the low rows')
  constexpr int64_t kCUs = 256;
  int best = 1;
  double best_score = -1.0;
  for (int c = 1; c <= 64; c++) {
    const int64_t q = (octets + 4 * c - 1) / (4 * c);
    if (q < 8) break;
    const int64_t wgs = static_cast<int64_t>(p->nbs) * c;
    const int64_t rounds = (wgs + kCUs - 1) / kCUs;
    const double score = static_cast<double>(wgs) / static_cast<double>(kCUs * rounds) *
                             (static_cast<double>(octets) / static_cast<double>(4 * c * q)) -
                         0.0005 * c - 0.02 / static_cast<double>(rounds) - (wgs < 2 * kCUs ? 0.01 : 0.0);
    if (score > best_score) best = c, best_score = score;
  }
  p->ca = 0;
  p->rho = 1.0;
  // Uneven dealing (see set_octets): one launch of exactly two workgroups per CU, the first
  // 256 / nbs chunks of every row (dispatched first: the older wave of every SIMD) with the
  // larger share.  Needs enough octets per SIMD that the smaller share is still >= 8.
  {
    int pct = (a.precision == NBODY_MI355X_FP32 || ipt <= 4) ? kSkewPercentSmall : kSkewPercent;
    if (const char* v = getenv("NBODY_MI355X_SYMW_SKEW"); v != nullptr && v[0] != 0) pct = atoi(v);
    // (the first ca chunks of every row must fill the chip, 256 / nbs workgroups per row: all of it
    // for a long launch -- 24 576 / 40 960 bodies, 240 of 256 CUs: 0.6 / 1.1 % slower than even dealing
    // in three or more dispatch rounds -- and at least 15/16 of it for a short one, up to 64 octets
    // per SIMD, which even dealing cannot spread over rounds -- 10 240 ... 20 480 bodies: +3 ... 7 %)
    const int ca = p->nbs <= kCUs ? static_cast<int>(kCUs / p->nbs) : 0;
    int64_t min_octets = kSkewMinOctets;
    if (const int v = env_int("NBODY_MI355X_SYMW_SKEW_MIN"); v > 0) min_octets = v;
    const bool fills = ca * p->nbs == kCUs || (16 * ca * p->nbs >= 15 * kCUs && octets * p->nbs <= 64 * 4 * kCUs);
    if (pct > 0 && pct < 100 && ca >= 1 && fills && octets >= static_cast<int64_t>(4 * ca) * min_octets) {
      SymArgs probe;  // (the shares must come out as asked: at least 8 octets for an older wave, 2 for a later one)
      if (set_octets(&probe, octets, 2 * ca, ca, pct / 100.0) == 2 * ca && probe.ws == 4 * ca) {
        p->ca = ca;
        p->rho = pct / 100.0;
        best = 2 * ca;
      }
    }
  }
  if (const int v = env_int("NBODY_MI355X_SYMW_CHUNKS"); v >= 1 && (octets + 4 * v - 1) / (4 * v) >= 8) best = v, p->ca = 0;
  p->rchunks = best;
  p->dsplit = p->jsplit = 1;
  p->jlocal = (ipt + 3) / 4;  // phase 1: the ipt entries of the diagonal tile, one per wave
  p->cbrows = (4 * best + ipt - 1) / ipt;  // nbs * 4 chunks * 192 doubles in rows of 3 S = nbs * 192 ipt
  plan_bytes(a, p);
  return true;
}

// The plan is a pure function of (shard_slots, num_shards, variant, precision): the workspace
// query and the step agree on it.
bool make_plan(const nbody_mi355x_step_args& a, Plan* p) {
  if (a.variant == NBODY_MI355X_VARIANT_SYM1) return plan_for(a, 1, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM2) return plan_for(a, 2, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM4) return plan_for(a, 4, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYM8) return plan_for(a, 8, p) >= 0.0;
  if (a.variant == NBODY_MI355X_VARIANT_SYMW1) return plan_wave(a, 1, p);
  if (a.variant == NBODY_MI355X_VARIANT_SYMW2) return plan_wave(a, 2, p);
  if (a.variant == NBODY_MI355X_VARIANT_SYMW4) return plan_wave(a, 4, p);
  if (a.variant == NBODY_MI355X_VARIANT_SYMW8) return plan_wave(a, 8, p);
  // automatic.  Up to kWaveTileMaxSlots slots in all: wave tiles, whose work items are small
  // enough to level the load of a short launch; block size by system size.  Measured on
  // MI355X (Ginteractions/s, equal masses; workgroup tiles of the best block in brackets):
  // 8192 bodies: 128-body blocks 1707, 256: 1657, 64: 1467 (1177); 16 384: 256: 2487, 512: 2297
  // (1893); 32 768: 256: 2861, 512: 2830 (2578); 65 536: 512: 3025, 256: 2978 (2857); 131 072:
  // 512: 3019 (2985); 198 656: (2692, 1024-body blocks: too few 2048-body tiles); 262 144: 2938
  // (2936 .. 2966); 524 288: 2969 (3012); 1 048 576: 2971 (3040) --
  // the j-side sums are stored after meeting 512 bodies instead of 2048, traffic that grows
  // as N^2, so the large systems keep workgroup tiles.
  {
    const int64_t total = a.shard_slots * a.num_shards;
    if (total <= kWaveTileMaxSlots) {
      // block size by system size (round 3's sweeps, a row's list dealt in octets, unevenly where
      // that applies; tools/sweep_skew_blocks.sh, us per step with 256- / 512-body blocks): 12 288:
      // 63.0 / 67.0; 13 312: 77.8 / 76.6; 14 336: 87.0 / 84.3; 16 384: 103.5 / 102.1; 18 432: 134.0 /
      // 127.7 -- 512-body blocks from 13 312 slots; at the small end 64-body blocks up to 2560
      // slots (2304: 11.0 / 13.3 with 128) and 128-body blocks up to 4608 (3072: 13.7, 64: 13.9; 4096:
      // 15.9, 256: 18.6; 5120: 21.9 / 19.5 with 256); a smaller block where the larger one does not
      // divide the slab
      for (int ipt = total <= 2560 ? 1 : (total <= 4608 ? 2 : (total < 13312 ? 4 : 8)); ipt >= 1; ipt /= 2)
        if (plan_wave(a, ipt, p)) return true;
    }
  }
  // Beyond: workgroup tiles, the largest block that still leaves enough tiles per rank to keep
  // 512 resident workgroups busy to the end.  Bigger blocks cost fewer DPP moves per pair, but
  // a launch of few big tiles is mostly ramp and tail.  Thresholds from sweeps on MI355X
  // (Ginteractions/s, best block per size): 65 536: 1024 2802 (2048: 2744); 131 072: 1024 2880
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
hipError_t launch_wg_tiles(const SymArgs& sa, dim3 grid, hipStream_t stream) {
  if constexpr (IPT == 8 && sizeof(T) == sizeof(double)) {  // the FAST tier: 2048-body blocks only (the large systems)
    if (sa.fast) {
      hipLaunchKernelGGL((sym_tiles_kernel<IPT, T, true>), grid, dim3(256), 0, stream, sa);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((sym_tiles_kernel<IPT, T>), grid, dim3(256), 0, stream, sa);
  return hipGetLastError();
}

// the tile kernel of the plan's family, block size and arithmetic
hipError_t launch_tiles(const Plan& p, bool fp32, const SymArgs& sa, dim3 grid, hipStream_t stream) {
  if (p.wave) return nbody_sym::launch_wave_tiles(p.ipt, fp32, sa, grid, stream);
  if (fp32)
    return p.ipt == 1 ? launch_wg_tiles<1, float>(sa, grid, stream)
           : p.ipt == 2 ? launch_wg_tiles<2, float>(sa, grid, stream)
           : p.ipt == 4 ? launch_wg_tiles<4, float>(sa, grid, stream) : launch_wg_tiles<8, float>(sa, grid, stream);
  return p.ipt == 1 ? launch_wg_tiles<1, double>(sa, grid, stream)
         : p.ipt == 2 ? launch_wg_tiles<2, double>(sa, grid, stream)
         : p.ipt == 4 ? launch_wg_tiles<4, double>(sa, grid, stream) : launch_wg_tiles<8, double>(sa, grid, stream);
}

// wave tiles: `octets` of a row's list over the 4 * chunks waves of `chunks` workgroups: at
// least 8 octets per wave (fewer workgroups if need be; returned), and the multiplier the
// finishing kernels divide by q with -- exact for x q < 2^32 (x: an octet index, < 2^21)
int set_octets(SymArgs* sa, int64_t octets, int chunks, int old_chunks, double rho) {
  const auto magic = [](int64_t q) { return static_cast<unsigned>((uint64_t{1} << 32) / static_cast<uint64_t>(q)) + 1u; };
  if (old_chunks > 0 && old_chunks < chunks) {
    // uneven: the waves of the first old_chunks workgroups of a row take qa octets, the others
    // qb ~ rho qa: at least 2 (the multiplier above), possibly less than an entry -- an entry among
    // them is then cut more than once, which the finishing kernels walk (sum_partials)
    const int64_t ws = 4 * old_chunks, wy = 4 * (chunks - old_chunks);
    int64_t qa = static_cast<int64_t>((static_cast<double>(octets) + ws + rho * wy - 1) / (ws + rho * wy));
    int64_t qb = octets > ws * qa ? (octets - ws * qa + wy - 1) / wy : 0;
    if (qb < 2) {
      qb = 2;
      qa = octets > wy * 2 ? (octets - wy * 2 + ws - 1) / ws : 0;
    }
    if (qa >= qb && qa >= 8) {
      sa->q = static_cast<int>(qa);
      sa->q_magic = magic(qa);
      sa->ws = static_cast<int>(ws);
      sa->qb = static_cast<int>(qb);
      sa->qb_magic = magic(qb);
      return chunks;
    }
  }
  while (chunks > 1 && (octets + 4 * chunks - 1) / (4 * chunks) < 8) chunks--;
  int64_t q = (octets + 4 * chunks - 1) / (4 * chunks);
  if (q < 8) q = 8;
  sa->q = sa->qb = static_cast<int>(q);
  sa->q_magic = sa->qb_magic = magic(q);
  sa->ws = 4 * chunks;
  return chunks;
}

hipError_t launch(const Plan& p, SymArgs sa, const nbody_mi355x_step_args& a, hipStream_t stream, int batch,
                  double* accum) {
  hipError_t e = hipSuccess;
  const bool fp32 = a.precision == NBODY_MI355X_FP32;
  // phase 1 = what needs the own slab only (the diagonal tiles), kept in R[0 .. jlocal);
  // phase 2 = the rest, then output; phase 0 = both
  if (a.phase == 1) {
    sa.d_lo = 0; sa.d_hi = 0; sa.dsplit = 1; sa.jsplit = sa.jlocal; sa.r_base = 0; sa.c_base = 1;
    if (p.wave) set_octets(&sa, p.ipt * 8, sa.jlocal);  // (no j-side sums on the diagonal: nothing is cut)
    return launch_tiles(p, fp32, sa, p.wave ? dim3(sa.nbs, sa.jlocal, 1) : dim3(sa.nbs, 1, sa.jlocal), stream);
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
    dim3 grid(sa.nbs, sa.dsplit, sa.jsplit);
    if (p.wave) {  // this launch's part of every row's list over its share of the row's workgroups
      const int64_t octets = static_cast<int64_t>(hi - lo + 1) * p.ipt * 8, all = static_cast<int64_t>(sa.D + 1) * p.ipt * 8;
      int chunks = static_cast<int>((p.rchunks * octets + all - 1) / all);
      chunks = chunks < 1 ? 1 : (chunks > p.rchunks ? p.rchunks : chunks);
      // (uneven dealing only where one launch walks the whole list: its two workgroups per CU start together)
      chunks = only && p.ca > 0 ? set_octets(&sa, octets, p.rchunks, p.ca, p.rho) : set_octets(&sa, octets, chunks);
      grid = dim3(sa.nbs, chunks, 1);
      sa.nR = sa.r_base + chunks;
    }
    e = launch_tiles(p, fp32, sa, grid, stream);
    if (e != hipSuccess) return e;
    const int64_t total = sa.S * a.num_shards;
    if (only && a.acc_out == nullptr) {
      launch_finish(parts_for(a.own_count), static_cast<unsigned>((a.own_count + 63) / 64), stream, sa, a);
      return hipGetLastError();
    }
    launch_gather(parts_for(total), static_cast<unsigned>((total + 63) / 64), stream, sa, a.own_shard, a.num_shards, first, out);
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

int sym_resolve(const nbody_mi355x_step_args& a) {
  Plan p;
  if (!make_plan(a, &p)) return -1;
  return (p.wave ? 20 : 10) + p.ipt;  // NBODY_MI355X_VARIANT_SYMW* / _SYM*
}

// nbody_mi355x_step_plan: the plan of a whole, unbatched step (phase 0) as launch() would set it up
bool sym_plan_query(const nbody_mi355x_step_args& a, int32_t* out) {
  Plan p;
  if (!make_plan(a, &p)) return false;
  for (int k = 0; k < NBODY_MI355X_PLAN_FIELDS; k++) out[k] = 0;
  out[0] = p.wave ? 2 : 1;
  out[1] = static_cast<int32_t>(p.block);
  out[2] = p.nbs;
  out[3] = p.D;
  out[9] = p.jlocal + p.rchunks;
  out[10] = p.cbrows;
  if (p.wave) {
    const int64_t octets = static_cast<int64_t>(p.D + 1) * p.ipt * 8;
    SymArgs sa;
    const int chunks = p.ca > 0 ? set_octets(&sa, octets, p.rchunks, p.ca, p.rho) : set_octets(&sa, octets, p.rchunks);
    out[4] = chunks;
    out[5] = sa.ws;
    out[6] = sa.q;
    out[7] = sa.qb;
    out[8] = static_cast<int32_t>(octets);
  } else {
    out[4] = p.dsplit * p.jsplit;
  }
  return true;
}

// launches of the tile kernel a whole step (phase 0 or 2) takes with this workspace: the
// iterations of the batch loop in launch(); 0 if the plan or the workspace does not apply
int sym_batches(const nbody_mi355x_step_args& a) {
  Plan p;
  if (!make_plan(a, &p)) return 0;
  const int64_t fixed = fixed_rows(p, a.num_shards);
  const int64_t rows = a.workspace ? a.workspace_bytes / p.row_bytes : 0;
  if (rows < fixed + 1) return 0;
  int64_t batch = rows - fixed;
  if (batch > p.D) batch = p.D;
  int n = 0;
  for (int64_t lo = a.phase == 2 ? 1 : 0, hi; lo <= p.D; lo = hi + 1, n++) {
    hi = (lo > 1 ? lo : 1) + batch - 1;
    if (hi > p.D) hi = p.D;
  }
  return n;
}

int sym_step(const nbody_mi355x_step_args& a, hipStream_t stream) {
  Plan p;
  if (!make_plan(a, &p)) {
    set_error("symmetric step: shard_slots must be a multiple of the block (64 .. 2048 by variant) "
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
  sa.Cb = sa.R + static_cast<int64_t>(p.jlocal + p.rchunks) * 3 * a.shard_slots;
  sa.C = sa.Cb + static_cast<int64_t>(p.cbrows) * 3 * a.shard_slots;
  double* accum = a.num_shards == 1 ? sa.C + batch * 3 * a.shard_slots : nullptr;
  sa.S = a.shard_slots;
  sa.own_count = a.own_count;
  sa.nbs = p.nbs;
  sa.nb = p.nb;
  sa.row0 = a.own_shard * p.nbs;
  sa.D = p.D;
  sa.dsplit = p.dsplit;
  sa.jsplit = p.jsplit;
  sa.jlocal = p.jlocal;
  sa.r_first = p.jlocal;
  sa.nR = p.jlocal + p.rchunks;
  sa.d_lo = sa.d_hi = sa.r_base = 0;
  sa.c_base = 1;
  sa.eps2 = a.eps2;
  sa.bshift = 0;
  while ((int64_t{1} << sa.bshift) < p.block) sa.bshift++;
  sa.trace = nullptr;
  sa.fast = a.precision == NBODY_MI355X_FP64_FAST ? 1 : 0;
  sa.q = sa.qb = sa.ws = 0;  // (wave tiles: set per launch, set_octets)
  sa.q_magic = sa.qb_magic = 0;
  sa.wstride = 4 * p.rchunks;
  sa.no_eq = env_int("NBODY_MI355X_SYM_NOEQ") != 0 ? 1 : 0;  // tests: compare the two forms bit for bit
  const hipError_t e = launch(p, sa, a, stream, static_cast<int>(batch), accum);
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
  const bool sym_variant = nbody_internal::is_sym_variant(a.variant);
  if (a.variant != 0 && !sym_variant) {
    set_error("step_f64_static: variant %d is a one-sided kernel, which needs no scratch: call nbody_mi355x_step_f64",
              a.variant);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (!sym_variant) a.variant = NBODY_MI355X_VARIANT_SYM;
  int64_t preferred = 0, least = 0;
  if (!nbody_internal::sym_supported(a, &preferred, &least)) {
    set_error("step_f64_static: the pair-symmetric step does not apply to %d slab(s) of %lld slots "
              "(at least two blocks of the variant's size are needed)", a.num_shards, (long long)a.shard_slots);
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
