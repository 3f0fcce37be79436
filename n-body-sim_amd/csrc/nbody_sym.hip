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
// Summation order differs from the reference's j = 0..N-1; the per-pair arithmetic is
// the same 4-ulp r^-3 as the one-sided kernel.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nbody_device.hpp"
#include "nbody_internal.h"

using namespace nbody_device;
using nbody_internal::set_error;

namespace {

struct SymArgs {
  const double* pos;   // one slab: x[S], y[S], z[S]
  const double* mass;  // [S]
  double* R;           // [dsplit][3][S]   i-side sums of row-chunk (I, s)
  double* C;           // [D][3][S]        j-side sums of tile (I, I+d), indexed by SOURCE row I
  int64_t S;
  int nb;      // blocks = S / B
  int D;       // largest distance = nb / 2
  int dsplit;  // distance chunks per row
  double eps2;
};

struct alignas(16) D2 {
  double a, b;
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

// All 4*IPT j groups of the staged tile against this wave's 64*IPT bodies.  SYM = false is
// the diagonal tile (a block against itself): i side only, no barriers needed.
template <int IPT, bool SYM>
__device__ __forceinline__ void tile_phases(const D2* xy, const D2* zm, double (*fJ)[256 * IPT], int lane,
                                            int w, const double (&xi)[IPT], const double (&yi)[IPT],
                                            const double (&zi)[IPT], const double (&mi)[IPT],
                                            double (&ax)[IPT], double (&ay)[IPT], double (&az)[IPT],
                                            double eps2, double c15, double c1875) {
  constexpr int NG = 4 * IPT;
  for (int ph = 0; ph < NG; ph++) {
    const int grp = (ph + IPT * w) % NG;  // the 4 waves sit on 4 different groups
    const int base = grp * 64;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    int rot = lane;
#pragma unroll 2
    for (int t = 0; t < 64; t++) {
      const D2 pxy = xy[base + rot];
      const D2 pzm = zm[base + rot];
      rot = (rot + 1) & 63;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const double dx = pxy.a - xi[k];
        const double dy = pxy.b - yi[k];
        const double dz = pzm.a - zi[k];
        const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
        const double y0 = __builtin_amdgcn_rsq(r2);
        const double u = y0 * y0;
        const double e = __builtin_fma(-r2, u, 1.0);
        const double w0 = y0 * u;
        const double p = __builtin_fma(e, c1875, c15);
        const double we = w0 * e;
        const double s0 = __builtin_fma(we, p, w0);  // r^-3
        const double si = s0 * pzm.b;
        ax[k] = __builtin_fma(si, dx, ax[k]);
        ay[k] = __builtin_fma(si, dy, ay[k]);
        az[k] = __builtin_fma(si, dz, az[k]);
        if (SYM) {
          const double sj = s0 * mi[k];
          fx = __builtin_fma(-sj, dx, fx);
          fy = __builtin_fma(-sj, dy, fy);
          fz = __builtin_fma(-sj, dz, fz);
        }
      }
      if (SYM) {  // the sums follow their body to the lane that meets it next
        fx = rotate_from_next_lane(fx);
        fy = rotate_from_next_lane(fy);
        fz = rotate_from_next_lane(fz);
      }
    }
    if (SYM) {  // 64 rotations later the sums are home: lane l holds body base + l
      fJ[0][base + lane] += fx;
      fJ[1][base + lane] += fy;
      fJ[2][base + lane] += fz;
      __syncthreads();  // next phase another wave adds to this group
    }
  }
}

template <int IPT>
__global__ __launch_bounds__(256) void sym_tiles_kernel(const SymArgs a) {
  constexpr int B = 256 * IPT;
  __shared__ D2 xy[B];
  __shared__ D2 zm[B];
  __shared__ double fJ[3][B];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int I = blockIdx.x, s = blockIdx.y;
  const int64_t S = a.S;
  const double eps2 = (a.eps2 > 0.0) ? a.eps2 : kTinyR2;
  double c15 = 1.5, c1875 = 1.875;
  asm volatile("" : "+v"(c15), "+v"(c1875));

  double xi[IPT], yi[IPT], zi[IPT], mi[IPT], ax[IPT], ay[IPT], az[IPT];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int64_t g = static_cast<int64_t>(I) * B + k * 256 + tid;
    xi[k] = a.pos[g];
    yi[k] = a.pos[S + g];
    zi[k] = a.pos[2 * S + g];
    mi[k] = a.mass[g];
    ax[k] = ay[k] = az[k] = 0.0;
  }

  const int nd = a.D + 1;  // distances 0 .. D
  const int chunk = (nd + a.dsplit - 1) / a.dsplit;
  const int d_begin = s * chunk;
  const int d_end = (d_begin + chunk < nd) ? d_begin + chunk : nd;

  for (int d = d_begin; d < d_end; d++) {
    if (2 * d == a.nb && I >= a.nb / 2) continue;  // the antipodal tile belongs to the low rows
    int J = I + d;
    if (J >= a.nb) J -= a.nb;
    const bool sym = (d != 0);

    __syncthreads();  // every wave is done with the previous tile's LDS
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int idx = k * 256 + tid;
      const int64_t g = static_cast<int64_t>(J) * B + idx;
      xy[idx] = D2{a.pos[g], a.pos[S + g]};
      zm[idx] = D2{a.pos[2 * S + g], a.mass[g]};
      fJ[0][idx] = 0.0;
      fJ[1][idx] = 0.0;
      fJ[2][idx] = 0.0;
    }
    __syncthreads();

    if (sym) tile_phases<IPT, true>(xy, zm, fJ, lane, w, xi, yi, zi, mi, ax, ay, az, eps2, c15, c1875);
    else tile_phases<IPT, false>(xy, zm, fJ, lane, w, xi, yi, zi, mi, ax, ay, az, eps2, c15, c1875);

    if (sym) {
      double* Cd = a.C + static_cast<int64_t>(d - 1) * 3 * S + static_cast<int64_t>(I) * B;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int idx = k * 256 + tid;
        Cd[idx] = fJ[0][idx];
        Cd[S + idx] = fJ[1][idx];
        Cd[2 * S + idx] = fJ[2][idx];
      }
    }
  }

  double* Rs = a.R + static_cast<int64_t>(s) * 3 * S + static_cast<int64_t>(I) * B;
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 256 + tid;
    Rs[idx] = ax[k];
    Rs[S + idx] = ay[k];
    Rs[2 * S + idx] = az[k];
  }
}

// a_g = sum_s R[s][g] + sum_d C[d][row J(g) - d][g mod B], fixed order, then the epilogue.
template <int IPT>
__global__ __launch_bounds__(256) void sym_finish_kernel(const SymArgs sa, const nbody_mi355x_step_args a) {
  constexpr int B = 256 * IPT;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (g >= a.own_count) return;
  const int64_t S = sa.S;
  const int J = static_cast<int>(g / B);
  const int jl = static_cast<int>(g - static_cast<int64_t>(J) * B);
  double ax = 0.0, ay = 0.0, az = 0.0;
  for (int s = 0; s < sa.dsplit; s++) {
    const double* Rs = sa.R + static_cast<int64_t>(s) * 3 * S;
    ax += Rs[g];
    ay += Rs[S + g];
    az += Rs[2 * S + g];
  }
  for (int d = 1; d <= sa.D; d++) {
    int I = J - d;
    if (I < 0) I += sa.nb;
    if (2 * d == sa.nb && I >= sa.nb / 2) continue;
    const double* Cd = sa.C + static_cast<int64_t>(d - 1) * 3 * S + static_cast<int64_t>(I) * B + jl;
    ax += Cd[0];
    ay += Cd[S];
    az += Cd[2 * S];
  }
  integrate_and_store(a, g, a.pos_cur[g], a.pos_cur[S + g], a.pos_cur[2 * S + g], ax, ay, az);
}

struct Plan {
  int ipt, nb, D, dsplit;
  int64_t bytes;
};

bool make_plan(const nbody_mi355x_step_args& a, Plan* p) {
  p->ipt = (a.variant == 12) ? 2 : 4;
  const int64_t B = 256 * p->ipt;
  if (a.num_shards != 1 || a.shard_slots % B != 0 || a.shard_slots / B < 2) return false;
  p->nb = static_cast<int>(a.shard_slots / B);
  p->D = p->nb / 2;
  const int nd = p->D + 1;
  int ds = (1024 + p->nb - 1) / p->nb;  // aim at >= 1024 workgroups
  if (ds > nd) ds = nd;
  if (ds < 1) ds = 1;
  p->dsplit = ds;
  p->bytes = static_cast<int64_t>(sizeof(double)) * 3 * a.shard_slots * (static_cast<int64_t>(ds) + p->D);
  return true;
}

template <int IPT>
hipError_t launch(const SymArgs& sa, const nbody_mi355x_step_args& a, hipStream_t stream) {
  hipLaunchKernelGGL((sym_tiles_kernel<IPT>), dim3(sa.nb, sa.dsplit), dim3(256), 0, stream, sa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned blocks = static_cast<unsigned>((a.own_count + 255) / 256);
  hipLaunchKernelGGL((sym_finish_kernel<IPT>), dim3(blocks), dim3(256), 0, stream, sa, a);
  return hipGetLastError();
}

}  // namespace

namespace nbody_internal {

bool sym_supported(const nbody_mi355x_step_args& a, int64_t* bytes) {
  Plan p;
  if (!make_plan(a, &p)) return false;
  if (bytes) *bytes = p.bytes;
  return true;
}

int sym_step(const nbody_mi355x_step_args& a, hipStream_t stream) {
  Plan p;
  if (!make_plan(a, &p)) {
    set_error("symmetric step: needs one slab whose slot count is a multiple of %d and at least two blocks",
              256 * ((a.variant == 12) ? 2 : 4));
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a.acc_in || a.acc_out || a.j0_begin != 0 || a.j1_begin != a.j1_end || a.precision != NBODY_MI355X_FP64) {
    set_error("symmetric step: split steps, j sub-ranges and the FP32 variant are one-sided only");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (!a.workspace || a.workspace_bytes < p.bytes) {
    set_error("symmetric step: workspace of %lld bytes needed, %lld given", (long long)p.bytes,
              (long long)a.workspace_bytes);
    return NBODY_MI355X_ERR_INVALID;
  }
  SymArgs sa;
  sa.pos = a.pos_cur;
  sa.mass = a.mass;
  sa.R = static_cast<double*>(a.workspace);
  sa.C = sa.R + static_cast<int64_t>(p.dsplit) * 3 * a.shard_slots;
  sa.S = a.shard_slots;
  sa.nb = p.nb;
  sa.D = p.D;
  sa.dsplit = p.dsplit;
  sa.eps2 = a.eps2;
  const hipError_t e = (p.ipt == 2) ? launch<2>(sa, a, stream) : launch<4>(sa, a, stream);
  if (e != hipSuccess) {
    set_error("symmetric step launch: %s", hipGetErrorString(e));
    return NBODY_MI355X_ERR_LAUNCH;
  }
  return NBODY_MI355X_OK;
}

}  // namespace nbody_internal
