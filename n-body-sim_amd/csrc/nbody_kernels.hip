// nbody_kernels.hip -- gfx950 (MI355X) kernels of the all-pairs hot path and the
// stateless (device-pointer) half of the C ABI declared in include/nbody_mi355x.h.
//
// What is computed is the reference's MakeSmallStep (particle_simulation.h:248-322;
// CUDA twin particle_simulation_GPU.cu:60-142):
//     a_i = sum_j m_j (r_j - r_i) / |r_j - r_i|^3        (pairs with r^2 == 0 skipped)
//     v_i' = v_i + (dt * a_i) * G ;  r_i' = r_i + dt * v_i (old v)
// How it is computed is MI355X-first and shares nothing with the reference's
// one-thread-per-body global-memory kernel:
//   * slab SoA layout in HBM (see the header), coalesced 8-byte loads;
//   * a 256-thread workgroup owns 256*IPT i-bodies held in registers, walks
//     the j bodies in tiles of 256 staged through LDS as {x,y,z,m} records
//     (register-prefetched, two LDS buffers, one barrier per tile); every lane
//     reads the same j record per step, i.e. conflict-free LDS broadcasts
//     (2 x ds_read_b128 per j, amortised over IPT interactions);
//   * r^-3 from v_rsq_f64 plus a folded series correction, no division, no
//     sqrt, no branch: the r^2 == 0 rule is kept exactly by seeding the r^2 FMA
//     chain with 2^-300 (see pair_f64), which costs no instruction;
//   * the Euler update and the optional snapshot are fused into the epilogue.
// FP64 vector FMA is the bounding unit (no MFMA: this is not a contraction).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nbody_device.hpp"
#include "nbody_internal.h"

using namespace nbody_device;

namespace {

template <typename Acc>
struct AccOps;
template <>
struct AccOps<double> {
  static __device__ __forceinline__ void pair(double xi, double yi, double zi, const JBody& b,
                                              double eps2, double c15, double c1875, double& ax,
                                              double& ay, double& az) {
    pair_f64(xi, yi, zi, b.x, b.y, b.z, b.m, eps2, c15, c1875, ax, ay, az);
  }
};
template <>
struct AccOps<float> {
  static __device__ __forceinline__ void pair(double xi, double yi, double zi, const JBody& b,
                                              double eps2, double, double, float& ax, float& ay,
                                              float& az) {
    pair_f32(xi, yi, zi, b.x, b.y, b.z, static_cast<float>(b.m), static_cast<float>(eps2), ax, ay, az);
  }
};

// Fetch the j body that thread `tid` stages for the tile starting at slot jt.
__device__ __forceinline__ JBody load_j(const nbody_mi355x_step_args& a, int64_t jt, int64_t jend,
                                        int tid) {
  JBody b;
  const int64_t shard = jt / a.shard_slots;  // wave-uniform: a tile never straddles slabs
  const int64_t off = jt - shard * a.shard_slots + tid;
  const double* base = a.pos_cur + shard * 3 * a.shard_slots + off;
  const bool ok = (jt + tid) < jend;
  // Slots in [jend, tile end) still lie inside the slab allocation (shard_slots is a
  // multiple of the tile), so the loads are in bounds; they are masked to a massless
  // body at the origin, which contributes exactly zero.
  const double x = base[0];
  const double y = base[a.shard_slots];
  const double z = base[2 * a.shard_slots];
  const double m = a.mass[jt + tid];
  b.x = ok ? x : 0.0;
  b.y = ok ? y : 0.0;
  b.z = ok ? z : 0.0;
  b.m = ok ? m : 0.0;
  return b;
}

template <int IPT, typename Acc, int UNROLL>
__global__ __launch_bounds__(kTile) void step_kernel(const nbody_mi355x_step_args a) {
  __shared__ JBody tile[2][kTile];
  const int tid = threadIdx.x;
  const int64_t S = a.shard_slots;
  // double path: 2^-300 stands in for "no softening" (see kTinyR2); float path: plain 0
  const double eps2 = (a.eps2 > 0.0) ? a.eps2 : (sizeof(Acc) == sizeof(double) ? kTinyR2 : 0.0);
  double c15 = 1.5, c1875 = 1.875;  // opaque to the optimiser: stay in VGPRs (see pair_f64)
  asm volatile("" : "+v"(c15), "+v"(c1875));

  // own bodies: local slot li[k] = block*256*IPT + k*256 + tid (coalesced per k)
  const int64_t li0 = static_cast<int64_t>(blockIdx.x) * (kTile * IPT) + tid;
  const double* own = a.pos_cur + static_cast<int64_t>(a.own_shard) * 3 * S;
  double xi[IPT], yi[IPT], zi[IPT];
  Acc ax[IPT], ay[IPT], az[IPT];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int64_t li = li0 + k * kTile;
    // lanes past the slab end clamp their loads (results are discarded below)
    const int64_t lc = li < S ? li : S - 1;
    xi[k] = own[lc];
    yi[k] = own[S + lc];
    zi[k] = own[2 * S + lc];
    if (a.acc_in != nullptr) {
      ax[k] = static_cast<Acc>(a.acc_in[lc]);
      ay[k] = static_cast<Acc>(a.acc_in[S + lc]);
      az[k] = static_cast<Acc>(a.acc_in[2 * S + lc]);
    } else {
      ax[k] = ay[k] = az[k] = Acc(0);
    }
  }

  const int64_t t0 = (a.j0_end - a.j0_begin + kTile - 1) / kTile;
  const int64_t t1 = (a.j1_end - a.j1_begin + kTile - 1) / kTile;
  const int64_t ntiles = t0 + t1;

  auto tile_start = [&](int64_t t, int64_t& jend) -> int64_t {
    if (t < t0) {
      jend = a.j0_end;
      return a.j0_begin + t * kTile;
    }
    jend = a.j1_end;
    return a.j1_begin + (t - t0) * kTile;
  };

  if (ntiles > 0) {
    int64_t jend;
    int64_t jt = tile_start(0, jend);
    tile[0][tid] = load_j(a, jt, jend, tid);
    __syncthreads();
    for (int64_t t = 0; t < ntiles; t++) {
      const int buf = static_cast<int>(t & 1);
      JBody nxt;
      const bool more = (t + 1) < ntiles;
      if (more) {
        jt = tile_start(t + 1, jend);
        nxt = load_j(a, jt, jend, tid);  // in flight while this tile is computed
      }
#pragma unroll UNROLL
      for (int jj = 0; jj < kTile; jj++) {
        const JBody b = tile[buf][jj];  // same address in every lane: LDS broadcast
#pragma unroll
        for (int k = 0; k < IPT; k++) AccOps<Acc>::pair(xi[k], yi[k], zi[k], b, eps2, c15, c1875, ax[k], ay[k], az[k]);
      }
      if (more) tile[buf ^ 1][tid] = nxt;  // buf^1 was last read before the previous barrier
      __syncthreads();
    }
  }

#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int64_t li = li0 + k * kTile;
    if (li >= a.own_count) continue;
    const double axd = static_cast<double>(ax[k]);
    const double ayd = static_cast<double>(ay[k]);
    const double azd = static_cast<double>(az[k]);
    if (a.acc_out != nullptr) {  // first half of a split step: hand the raw sums on
      a.acc_out[li] = axd;
      a.acc_out[S + li] = ayd;
      a.acc_out[2 * S + li] = azd;
      continue;
    }
    integrate_and_store(a, li, xi[k], yi[k], zi[k], axd, ayd, azd);
  }
}

// ---------------------------------------------------------------- small systems
// Below a few thousand bodies one lane per body leaves the chip empty (N = 500 is eight
// waves) and a launch is one long serial j loop.  Here a body is shared by a DPP row of 16
// lanes: lane r of the row takes the j bodies r, r+16, ... of each staged tile, the 16
// partial sums are added across the row (4 row_shr steps) and lane 15 integrates.  A
// workgroup is 16 bodies, so N = 500 is 32 workgroups of 4 waves and the j loop is 16 times
// shorter.  Summation order: 16 interleaved partial sums per body (not the reference's
// single chain); same per-pair arithmetic.
__device__ __forceinline__ double row_sum16(double v) {
  // after the four steps lane 15 of each 16-lane row holds the row's sum
#define NBODY_ROW_STEP(ctrl)                                                         \
  {                                                                                  \
    int lo = __double2loint(v), hi = __double2hiint(v);                              \
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);                   \
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);                   \
    v += __hiloint2double(hi, lo);                                                   \
  }
  NBODY_ROW_STEP(0x111)  // row_shr:1
  NBODY_ROW_STEP(0x112)  // row_shr:2
  NBODY_ROW_STEP(0x114)  // row_shr:4
  NBODY_ROW_STEP(0x118)  // row_shr:8
#undef NBODY_ROW_STEP
  return v;
}

__global__ __launch_bounds__(kTile) void step_small_kernel(const nbody_mi355x_step_args a) {
  __shared__ JBody tile[2][kTile];
  const int tid = threadIdx.x;
  const int r = tid & 15;                                  // j slice of this lane
  const int64_t S = a.shard_slots;
  const double eps2 = (a.eps2 > 0.0) ? a.eps2 : kTinyR2;
  double c15 = 1.5, c1875 = 1.875;
  asm volatile("" : "+v"(c15), "+v"(c1875));

  const int64_t li = static_cast<int64_t>(blockIdx.x) * 16 + (tid >> 4);  // own body of this row
  const int64_t lc = li < S ? li : S - 1;
  const double* own = a.pos_cur + static_cast<int64_t>(a.own_shard) * 3 * S;
  const double xi = own[lc], yi = own[S + lc], zi = own[2 * S + lc];
  double ax = 0.0, ay = 0.0, az = 0.0;

  const int64_t ntiles = (a.j0_end - a.j0_begin + kTile - 1) / kTile;
  if (ntiles > 0) {
    tile[0][tid] = load_j(a, a.j0_begin, a.j0_end, tid);
    __syncthreads();
    for (int64_t t = 0; t < ntiles; t++) {
      const int buf = static_cast<int>(t & 1);
      JBody nxt;
      const bool more = (t + 1) < ntiles;
      if (more) nxt = load_j(a, a.j0_begin + (t + 1) * kTile, a.j0_end, tid);
#pragma unroll 4
      for (int jj = 0; jj < kTile / 16; jj++) {
        const JBody b = tile[buf][jj * 16 + r];
        pair_f64(xi, yi, zi, b.x, b.y, b.z, b.m, eps2, c15, c1875, ax, ay, az);
      }
      if (more) tile[buf ^ 1][tid] = nxt;
      __syncthreads();
    }
  }
  ax = row_sum16(ax);
  ay = row_sum16(ay);
  az = row_sum16(az);
  if (r == 15 && li < a.own_count) integrate_and_store(a, li, xi, yi, zi, ax, ay, az);
}

// The same idea taken to the end for the smallest systems (the saturn ring's N = 500 is
// BASELINE config 2): a whole WAVE per body.  Lane l takes the j bodies l, l+64, ... straight
// from global memory (64 consecutive doubles per load, the arrays are L2-resident; no LDS
// staging, no barrier), so a small step of 500 bodies is 8 pairs per lane on 125 workgroups
// instead of 32 pairs per lane on 32, and nothing in the kernel waits on anything but its
// own loads.  The 64 partial sums are added with the 4 row_shr steps above and 3 adds across
// the rows' last lanes.  Lanes past the end of the j range take a valid slot with weight 0.
__device__ __forceinline__ double wave_sum64(double v) {
  v = row_sum16(v);  // lanes 15, 31, 47, 63 hold their rows' sums
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 4; r++)
    s += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r + 15), __builtin_amdgcn_readlane(lo, 16 * r + 15));
  return s;  // wave-uniform
}

// JPL: j bodies per lane held in registers per trip; IPW: own bodies per wave (each j body
// fetched serves IPW pairs).  The NEXT trip's j bodies are requested before this trip's pairs
// are evaluated (two register sets used alternately), so that the L2 latency of a trip's 4 JPL
// loads lies under the previous trip's arithmetic instead of in front of its own.
template <int JPL>
struct JSet {
  double x[JPL], y[JPL], z[JPL], m[JPL];
};
template <int JPL>
__device__ __forceinline__ void wave_fetch(const nbody_mi355x_step_args& a, int64_t g, int lane, JSet<JPL>& b) {
  const int64_t S = a.shard_slots;
#pragma unroll
  for (int u = 0; u < JPL; u++) {
    // the slab of a 64-slot group is wave-uniform (S is a multiple of 64): scalar arithmetic,
    // and none at all on one slab
    int64_t gu = g + 64 * u;
    if (gu >= a.j0_end) gu = a.j0_begin;  // (a whole group past the end: any valid group, weights zeroed)
    const int64_t shard = a.num_shards == 1 ? 0 : gu / S;
    const double* base = a.pos_cur + shard * 2 * S + gu + lane;  // = slab * 3S + (gu - slab * S) + lane
    const bool in = g + 64 * u + lane < a.j0_end;
    b.x[u] = base[0];
    b.y[u] = base[S];
    b.z[u] = base[2 * S];
    b.m[u] = in ? a.mass[gu + lane] : 0.0;
  }
}

template <int JPL, int IPW>
__global__ __launch_bounds__(kTile) void step_wave_kernel(const nbody_mi355x_step_args a) {
  const int lane = threadIdx.x & 63;
  const int64_t S = a.shard_slots;
  const double eps2 = (a.eps2 > 0.0) ? a.eps2 : kTinyR2;
  double c15 = 1.5, c1875 = 1.875;
  asm volatile("" : "+v"(c15), "+v"(c1875));
  // own bodies of this wave: IPW consecutive slots
  const int64_t li0 = (static_cast<int64_t>(blockIdx.x) * (kTile / 64) + (threadIdx.x >> 6)) * IPW;
  const double* own = a.pos_cur + static_cast<int64_t>(a.own_shard) * 3 * S;
  double xi[IPW], yi[IPW], zi[IPW], ax[IPW], ay[IPW], az[IPW];
#pragma unroll
  for (int k = 0; k < IPW; k++) {
    const int64_t lc = li0 + k < S ? li0 + k : S - 1;
    xi[k] = own[lc]; yi[k] = own[S + lc]; zi[k] = own[2 * S + lc];
    ax[k] = ay[k] = az[k] = 0.0;
  }
  const int64_t jend = (a.j0_end + 63) & ~int64_t{63};
  constexpr int64_t kTrip = 64 * JPL;
  auto pairs = [&](const JSet<JPL>& b) {
#pragma unroll
    for (int u = 0; u < JPL; u++)
#pragma unroll
      for (int k = 0; k < IPW; k++)
        pair_f64(xi[k], yi[k], zi[k], b.x[u], b.y[u], b.z[u], b.m[u], eps2, c15, c1875, ax[k], ay[k], az[k]);
  };
  JSet<JPL> b0, b1;
  int64_t g = a.j0_begin;
  if (g < jend) wave_fetch<JPL>(a, g, lane, b0);
  while (g < jend) {
    if (g + kTrip < jend) wave_fetch<JPL>(a, g + kTrip, lane, b1);
    pairs(b0);
    g += kTrip;
    if (g >= jend) break;
    if (g + kTrip < jend) wave_fetch<JPL>(a, g + kTrip, lane, b0);
    pairs(b1);
    g += kTrip;
  }
#pragma unroll
  for (int k = 0; k < IPW; k++) {
    const double sx = wave_sum64(ax[k]), sy = wave_sum64(ay[k]), sz = wave_sum64(az[k]);
    if (lane == 0 && li0 + k < a.own_count) integrate_and_store(a, li0 + k, xi[k], yi[k], zi[k], sx, sy, sz);
  }
}

// ------------------------------------------------------------------ tiny systems
// The reference's two-body and eight-body scenes (earth_moon.cc, slingshoot.cc,
// earth_7moons.cc) take 324 000 small steps per big step; at one kernel per small step
// (3.2 us, the floor of a dependent launch) a big step of the earth-moon scene would take a
// second of pure launch latency.  Up to 64 bodies ONE WAVE carries the whole system: lane i
// keeps body i's position and velocity in registers across `substeps` small steps, the
// positions go round through LDS (two buffers alternating, one barrier per small step) and
// are read back as broadcasts in the reference's order j = 0 .. N-1 (same summation order
// as MakeSmallStep, particle_simulation.h:256-275).  In place: the state is read from and
// written back to pos_cur / vel.
__global__ __launch_bounds__(64) void step_tiny_kernel(const nbody_mi355x_step_args a, int64_t substeps, int kdk_prime) {
  __shared__ JBody body[2][64];
  const int i = threadIdx.x;
  const int n = static_cast<int>(a.own_count);
  const int64_t S = a.shard_slots;
  const double eps2 = (a.eps2 > 0.0) ? a.eps2 : kTinyR2;
  double c15 = 1.5, c1875 = 1.875;
  asm volatile("" : "+v"(c15), "+v"(c1875));
  const bool real = i < n;
  const int ic = real ? i : 0;
  double x = a.pos_cur[ic], y = a.pos_cur[S + ic], z = a.pos_cur[2 * S + ic];
  double vx = a.vel[ic], vy = a.vel[S + ic], vz = a.vel[2 * S + ic];
  const double m = real ? a.mass[ic] : 0.0;
  const bool kd = a.integrator == NBODY_MI355X_KICK_DRIFT;
  const bool kdk = a.integrator == NBODY_MI355X_KDK;
  const double h = 0.5 * a.dt;
  double ax = 0.0, ay = 0.0, az = 0.0;
  // the sums over all bodies at the positions every lane has just published in buf
  auto forces = [&](const JBody* buf) {
    ax = ay = az = 0.0;
    for (int j = 0; j < n; j++) {
      const JBody b = buf[j];
      pair_f64(x, y, z, b.x, b.y, b.z, b.m, eps2, c15, c1875, ax, ay, az);
    }
  };
  if (kdk) {
    // kick-drift-kick: a(r) comes in from the previous big step (kdk_acc) or, the first time,
    // from one extra evaluation; every small step then costs one evaluation, at r'
    if (kdk_prime) {
      body[1][i] = JBody{x, y, z, m};
      __syncthreads();
      forces(body[1]);
    } else {
      ax = a.kdk_acc[ic]; ay = a.kdk_acc[S + ic]; az = a.kdk_acc[2 * S + ic];
    }
  }
  for (int64_t k = 0; k < substeps; k++) {
    JBody* buf = body[k & 1];
    if (kdk) {
      vx = __builtin_fma(h * ax, a.G, vx);
      vy = __builtin_fma(h * ay, a.G, vy);
      vz = __builtin_fma(h * az, a.G, vz);
      x = __builtin_fma(a.dt, vx, x);
      y = __builtin_fma(a.dt, vy, y);
      z = __builtin_fma(a.dt, vz, z);
    }
    buf[i] = JBody{x, y, z, m};
    __syncthreads();  // (buffer k&1 is next written two small steps on, past the next barrier)
    forces(buf);
    if (kdk) {
      vx = __builtin_fma(h * ax, a.G, vx);
      vy = __builtin_fma(h * ay, a.G, vy);
      vz = __builtin_fma(h * az, a.G, vz);
      continue;
    }
    // ref: particle_simulation.h:288-300 (see integrate_and_store)
    const double nvx = __builtin_fma(a.dt * ax, a.G, vx);
    const double nvy = __builtin_fma(a.dt * ay, a.G, vy);
    const double nvz = __builtin_fma(a.dt * az, a.G, vz);
    x = __builtin_fma(a.dt, kd ? nvx : vx, x);
    y = __builtin_fma(a.dt, kd ? nvy : vy, y);
    z = __builtin_fma(a.dt, kd ? nvz : vz, z);
    vx = nvx; vy = nvy; vz = nvz;
  }
  if (!real) return;
  double* pos = const_cast<double*>(a.pos_cur);
  pos[i] = x; pos[S + i] = y; pos[2 * S + i] = z;
  a.vel[i] = vx; a.vel[S + i] = vy; a.vel[2 * S + i] = vz;
  if (kdk && a.kdk_acc != nullptr) {
    a.kdk_acc[i] = ax; a.kdk_acc[S + i] = ay; a.kdk_acc[2 * S + i] = az;
  }
  if (a.snapshot != nullptr) {
    double* o = a.snapshot + 9 * i;
    o[0] = x; o[1] = y; o[2] = z;
    o[3] = vx; o[4] = vy; o[5] = vz;
    o[6] = ax * a.G; o[7] = ay * a.G; o[8] = az * a.G;
  }
}

// ------------------------------------------------------------ kick-drift-kick: the drift
// vel += (dt/2 * a) * G with the sums the last kick stored, pos_next = pos_cur + dt * vel
__global__ __launch_bounds__(256) void kdk_drift_kernel(const nbody_mi355x_step_args a) {
  const int64_t li = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (li >= a.own_count) return;
  const int64_t S = a.shard_slots;
  const double h = 0.5 * a.dt;
  const double* own = a.pos_cur + static_cast<int64_t>(a.own_shard) * 3 * S;
  double* nown = a.pos_next + static_cast<int64_t>(a.own_shard) * 3 * S;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double v = __builtin_fma(h * a.kdk_acc[c * S + li], a.G, a.vel[c * S + li]);
    a.vel[c * S + li] = v;
    nown[c * S + li] = __builtin_fma(a.dt, v, own[c * S + li]);
  }
}

// ------------------------------------------------------------ layout converters
__global__ void pack_kernel(const double* __restrict__ aos, int64_t count, double* __restrict__ pos,
                            double* __restrict__ vel, int64_t S) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double* p = aos + 6 * i;
  pos[i] = p[0]; pos[S + i] = p[1]; pos[2 * S + i] = p[2];
  vel[i] = p[3]; vel[S + i] = p[4]; vel[2 * S + i] = p[5];
}

__global__ void unpack_kernel(const double* __restrict__ pos, const double* __restrict__ vel,
                              int64_t count, int64_t S, double* __restrict__ aos) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double* p = aos + 6 * i;
  p[0] = pos[i]; p[1] = pos[S + i]; p[2] = pos[2 * S + i];
  p[3] = vel[i]; p[4] = vel[S + i]; p[5] = vel[2 * S + i];
}

// ------------------------------------------------------------ energy diagnostic
__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kTile / 64; i++) s += red[i];
  return s;  // valid in thread 0
}

__global__ __launch_bounds__(kTile) void energy_partial_kernel(
    const double* __restrict__ pos, const double* __restrict__ mass, const double* __restrict__ vel,
    int64_t S, int num_shards, int own_shard, int64_t own_count, double G,
    double* __restrict__ scratch) {
  __shared__ JBody tile[kTile];
  __shared__ double red[kTile / 64];
  const int tid = threadIdx.x;
  const int64_t li = static_cast<int64_t>(blockIdx.x) * kTile + tid;
  const int64_t lc = li < S ? li : S - 1;
  const bool real = li < own_count;
  const double* own = pos + static_cast<int64_t>(own_shard) * 3 * S;
  const double xi = own[lc], yi = own[S + lc], zi = own[2 * S + lc];
  const int64_t gi = static_cast<int64_t>(own_shard) * S + lc;
  const double mi = real ? mass[gi] : 0.0;
  double ke = 0.0;
  if (real) {
    const double vx = vel[lc], vy = vel[S + lc], vz = vel[2 * S + lc];
    ke = 0.5 * mi * (vx * vx + vy * vy + vz * vz);
  }
  // sum over ALL j != i of m_j / r_ij; each unordered pair is met twice across the
  // grid, hence the factor 1/2 below.
  double phi = 0.0;
  const int64_t total = static_cast<int64_t>(num_shards) * S;
  for (int64_t jt = 0; jt < total; jt += kTile) {
    const int64_t shard = jt / S;
    const int64_t off = jt - shard * S + tid;
    const double* base = pos + shard * 3 * S + off;
    JBody b;
    b.x = base[0]; b.y = base[S]; b.z = base[2 * S]; b.m = mass[jt + tid];
    __syncthreads();
    tile[tid] = b;
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < kTile; jj++) {
      const JBody o = tile[jj];
      const double dx = o.x - xi, dy = o.y - yi, dz = o.z - zi;
      const double r2 = dx * dx + dy * dy + dz * dz;
      phi += (r2 != 0.0) ? o.m / sqrt(r2) : 0.0;
    }
  }
  const double pe = -0.5 * G * mi * phi;
  const double bke = block_sum(ke, red);
  const double bpe = block_sum(pe, red);
  if (tid == 0) {
    scratch[2 * blockIdx.x] = bke;
    scratch[2 * blockIdx.x + 1] = bpe;
  }
}

__global__ __launch_bounds__(kTile) void energy_final_kernel(const double* __restrict__ scratch,
                                                             int64_t nblocks,
                                                             double* __restrict__ result) {
  __shared__ double red[kTile / 64];
  double ke = 0.0, pe = 0.0;
  for (int64_t b = threadIdx.x; b < nblocks; b += kTile) {
    ke += scratch[2 * b];
    pe += scratch[2 * b + 1];
  }
  const double ske = block_sum(ke, red);
  const double spe = block_sum(pe, red);
  if (threadIdx.x == 0) {
    result[0] = ske;
    result[1] = spe;
  }
}

template <int IPT, typename Acc, int UNROLL>
hipError_t launch_step(const nbody_mi355x_step_args& a, hipStream_t stream) {
  const int64_t per_block = static_cast<int64_t>(kTile) * IPT;
  const int64_t blocks = (a.own_count + per_block - 1) / per_block;
  if (blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL((step_kernel<IPT, Acc, UNROLL>), dim3(static_cast<unsigned>(blocks)), dim3(kTile),
                     0, stream, a);
  return hipGetLastError();
}

// The small-system kernel handles whole, unsplit steps over one j range.
// (Precision: these kernels compute in double.  A caller that ALLOWS float forces,
// NBODY_MI355X_FP32, is served by them too where they are the fastest path: at these sizes a
// step is bound by latency and launch, not by arithmetic, and the float kernels are slower.)
bool small_applies(const nbody_mi355x_step_args& a, int64_t max_bodies = 4096) {
  return a.acc_in == nullptr && a.acc_out == nullptr && a.j1_begin == a.j1_end && a.own_count <= max_bodies &&
         a.own_count > 0;
}

// a wave per IPW bodies: whole steps over one j range.  The default up to
// NBODY_MI355X_WAVE_MAX_BODIES bodies; up to kWaveFallbackBodies where the pair-symmetric kernels
// have no workspace.  Shapes and thresholds from tools/sweep_mid.py on MI355X (microseconds per
// small step, graph-replayed; pair-symmetric wave tiles of the best block size in brackets):
// 500 bodies 3.9, 1024 5.1 (11.3), 2048 9.5 (12.1), 2304 16.6 (11.0), 2560 16.8 (13.2), 3072 16.5
// (13.7), 4096 20.5 (15.9), 5120 39.8 (19.5) -- above 2048 bodies the pair-symmetric kernels' halved
// arithmetic outweighs their fixed cost (two launches, partial sums through HBM).  (Round 2 drew
// the line at 4096 with whole entries per wave; a row's list dealt in octets moved it.)
constexpr int64_t kWaveFallbackBodies = 4096;
constexpr int64_t kWaveMaxBodies = NBODY_MI355X_WAVE_MAX_BODIES;
template <int JPL, int IPW>
hipError_t launch_wave_t(const nbody_mi355x_step_args& a, hipStream_t stream) {
  const int64_t waves = (a.own_count + IPW - 1) / IPW;
  const unsigned blocks = static_cast<unsigned>((waves + kTile / 64 - 1) / (kTile / 64));
  hipLaunchKernelGGL((step_wave_kernel<JPL, IPW>), dim3(blocks), dim3(kTile), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_wave(const nbody_mi355x_step_args& a, hipStream_t stream) {
  if (!small_applies(a, 65536)) return hipErrorInvalidValue;
  if (a.own_count <= 1024) return launch_wave_t<8, 1>(a, stream);
  if (a.own_count <= 2560) return launch_wave_t<8, 2>(a, stream);
  return launch_wave_t<8, 4>(a, stream);
}

hipError_t launch_small(const nbody_mi355x_step_args& a, hipStream_t stream) {
  if (!small_applies(a)) return hipErrorInvalidValue;
  const unsigned blocks = static_cast<unsigned>((a.own_count + 15) / 16);
  hipLaunchKernelGGL(step_small_kernel, dim3(blocks), dim3(kTile), 0, stream, a);
  return hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------ error text
namespace nbody_internal {
thread_local char g_last_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
  va_end(ap);
}
}  // namespace nbody_internal

using nbody_internal::set_error;

int nbody_internal::tiny_run(const nbody_mi355x_step_args& a, int64_t substeps, bool kdk_prime, hipStream_t stream) {
  if (a.own_count < 1 || a.own_count > kTinyMaxBodies || a.num_shards != 1 || a.own_shard != 0 ||
      !a.pos_cur || !a.vel || !a.mass || a.shard_slots < a.own_count ||
      substeps < 1 || !(a.eps2 >= 0.0) || a.integrator == NBODY_MI355X_KDK_CONTINUE ||
      (a.integrator == NBODY_MI355X_KDK && !a.kdk_acc)) {
    set_error("tiny_run: not a one-slab system of at most %lld bodies", (long long)kTinyMaxBodies);
    return NBODY_MI355X_ERR_INVALID;
  }
  hipLaunchKernelGGL(step_tiny_kernel, dim3(1), dim3(64), 0, stream, a, substeps, kdk_prime ? 1 : 0);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("tiny_run launch: %s", hipGetErrorString(e));
    return NBODY_MI355X_ERR_LAUNCH;
  }
  return NBODY_MI355X_OK;
}

extern "C" const char* nbody_mi355x_last_error(void) { return nbody_internal::g_last_error; }

extern "C" int nbody_mi355x_device_count(int* count) {
  if (!count) return NBODY_MI355X_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *count = 0;
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return NBODY_MI355X_ERR_NO_DEVICE;
  }
  *count = n;
  return NBODY_MI355X_OK;
}

// The pair-symmetric kernels apply from this many bodies; by default they take over above
// NBODY_MI355X_WAVE_MAX_BODIES (launch_wave), or where a step is split or spans two j ranges.
static constexpr int64_t kSymMinBodies = 2048;

// Host-side shape checks: everything the kernel's indexing assumes is verified
// here, before any launch.
int nbody_internal::check_step_args(const nbody_mi355x_step_args* a) {
  if (!a) { set_error("step: null args"); return NBODY_MI355X_ERR_INVALID; }
  if (!a->pos_cur || !a->mass) { set_error("step: null pos_cur/mass"); return NBODY_MI355X_ERR_INVALID; }
  if (a->shard_slots <= 0 || a->shard_slots % kTile != 0) {
    set_error("step: shard_slots=%lld must be a positive multiple of %d", (long long)a->shard_slots, kTile);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->num_shards <= 0 || a->own_shard < 0 || a->own_shard >= a->num_shards) {
    set_error("step: own_shard=%d outside [0,%d)", a->own_shard, a->num_shards);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->own_count < 0 || a->own_count > a->shard_slots) {
    set_error("step: own_count=%lld outside [0,%lld]", (long long)a->own_count, (long long)a->shard_slots);
    return NBODY_MI355X_ERR_INVALID;
  }
  const int64_t total = a->shard_slots * a->num_shards;
  const int64_t r[4] = {a->j0_begin, a->j0_end, a->j1_begin, a->j1_end};
  for (int k = 0; k < 2; k++) {
    const int64_t b = r[2 * k], e = r[2 * k + 1];
    if (b < 0 || e < b || e > total || b % kTile != 0) {
      set_error("step: j range %d = [%lld,%lld) invalid (total %lld, begin must be a multiple of %d)", k,
                (long long)b, (long long)e, (long long)total, kTile);
      return NBODY_MI355X_ERR_INVALID;
    }
  }
  const bool writes_pos = a->acc_out == nullptr && a->integrator != NBODY_MI355X_KDK;  // (KDK: second kick only)
  if (a->acc_out == nullptr && (!a->vel || (writes_pos && !a->pos_next))) {
    set_error("step: integrating step needs pos_next and vel");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->pos_next == a->pos_cur && writes_pos) {
    set_error("step: pos_next must not alias pos_cur");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (!(a->eps2 >= 0.0)) { set_error("step: eps2 must be >= 0"); return NBODY_MI355X_ERR_INVALID; }
  if (a->precision != NBODY_MI355X_FP64 && a->precision != NBODY_MI355X_FP32 && a->precision != NBODY_MI355X_FP64_FAST) {
    set_error("step: unknown precision %d", a->precision);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->integrator < NBODY_MI355X_EULER || a->integrator > NBODY_MI355X_KDK_CONTINUE) {
    set_error("step: unknown integrator %d", a->integrator);
    return NBODY_MI355X_ERR_INVALID;
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_kdk_drift_f64(const nbody_mi355x_step_args* a, void* stream) {
  if (!a || !a->pos_cur || !a->pos_next || !a->vel || !a->kdk_acc || a->pos_next == a->pos_cur) {
    set_error("kdk_drift: pos_cur, pos_next (distinct), vel and kdk_acc are required");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->shard_slots <= 0 || a->shard_slots % kTile != 0 || a->num_shards <= 0 || a->own_shard < 0 ||
      a->own_shard >= a->num_shards || a->own_count < 0 || a->own_count > a->shard_slots) {
    set_error("kdk_drift: bad slab layout (shard_slots=%lld, own_shard=%d of %d, own_count=%lld)",
              (long long)a->shard_slots, a->own_shard, a->num_shards, (long long)a->own_count);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (a->own_count == 0) return NBODY_MI355X_OK;
  hipLaunchKernelGGL(kdk_drift_kernel, dim3(static_cast<unsigned>((a->own_count + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("kdk_drift launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

using nbody_internal::is_sym_variant;

// Which kernel build a step with these (validated) args runs: a concrete variant code --
// NBODY_MI355X_VARIANT_SYM1 .. SYMW8 (pair-symmetric, block size resolved), _WAVE, or a
// one-sided build 1 .. 6.  The ONE place the choice is made: nbody_mi355x_step_f64 dispatches
// on the result, nbody_mi355x_step_variant reports it.
static int resolve_variant(const nbody_mi355x_step_args* args) {
  if (is_sym_variant(args->variant)) return nbody_internal::sym_resolve(*args);  // (-1: does not apply)
  const bool wave_ok = small_applies(*args, kWaveFallbackBodies) && args->j0_end - args->j0_begin <= kWaveFallbackBodies;
  const bool wave_size = wave_ok && args->own_count <= kWaveMaxBodies;
  if (args->variant == 0 && !wave_size && args->workspace != nullptr &&
      args->acc_in == nullptr && args->acc_out == nullptr && args->own_count >= kSymMinBodies &&
      args->j0_begin == 0 && args->j1_begin == args->j1_end &&
      args->j0_end >= args->own_count) {
    // default for large single-slab systems: the pair-symmetric kernels
    nbody_mi355x_step_args a = *args;
    a.variant = NBODY_MI355X_VARIANT_SYM;
    // (any workspace from the plan's minimum up will do: fewer C rows mean more batches)
    int64_t preferred = 0, least = 0;
    if (nbody_internal::sym_supported(a, &preferred, &least) && least <= a.workspace_bytes)
      return nbody_internal::sym_resolve(a);
  }
  if (wave_ok && (args->variant == 0 || args->variant > 7)) return NBODY_MI355X_VARIANT_WAVE;
  // a step with acc_in and no j range only integrates sums that were reduced in double: it
  // takes a double build whatever the precision, as the one-slab finish kernel does --
  // narrowing them first would make 1-shard and N-shard FP32 runs differ for nothing
  const bool integrate_only = args->acc_in != nullptr && args->j0_begin == args->j0_end && args->j1_begin == args->j1_end;
  if (args->precision == NBODY_MI355X_FP32 && !integrate_only)
    return (args->variant == 1 || args->variant == 4) ? args->variant : 2;
  if (args->variant >= 1 && args->variant <= 7) return args->variant;
  // enough workgroups to give every SIMD >= 2 waves with 2 bodies per lane?
  return args->own_count >= 2 * 256 * 4 * 2 * 64 ? 2 : 1;
}

extern "C" int nbody_mi355x_step_variant(const nbody_mi355x_step_args* args, int32_t* variant) {
  if (!variant) { set_error("step_variant: null argument"); return NBODY_MI355X_ERR_INVALID; }
  const int rc = nbody_internal::check_step_args(args);
  if (rc != NBODY_MI355X_OK) return rc;
  const int v = resolve_variant(args);
  if (v < 0) {
    set_error("step_variant: the pair-symmetric kernels of variant %d do not apply to %d slab(s) of %lld slots",
              args->variant, args->num_shards, (long long)args->shard_slots);
    return NBODY_MI355X_ERR_INVALID;
  }
  *variant = v;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_step_plan(const nbody_mi355x_step_args* args, int32_t* plan) {
  if (!plan) { set_error("step_plan: null argument"); return NBODY_MI355X_ERR_INVALID; }
  const int rc = nbody_internal::check_step_args(args);
  if (rc != NBODY_MI355X_OK) return rc;
  const int v = resolve_variant(args);
  for (int k = 0; k < NBODY_MI355X_PLAN_FIELDS; k++) plan[k] = 0;
  if (v > 0 && is_sym_variant(v)) {
    nbody_mi355x_step_args a = *args;
    a.variant = v;
    if (!nbody_internal::sym_plan_query(a, plan)) {
      set_error("step_plan: no pair-symmetric plan for variant %d", v);
      return NBODY_MI355X_ERR_INVALID;
    }
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_step_tile_launches(const nbody_mi355x_step_args* args, int32_t* launches) {
  if (!launches) { set_error("step_tile_launches: null argument"); return NBODY_MI355X_ERR_INVALID; }
  const int rc = nbody_internal::check_step_args(args);
  if (rc != NBODY_MI355X_OK) return rc;
  const int v = resolve_variant(args);
  *launches = 1;
  if (v > 0 && is_sym_variant(v)) {
    nbody_mi355x_step_args a = *args;
    a.variant = v;
    const int b = nbody_internal::sym_batches(a);
    *launches = args->phase == 1 ? 1 : (b > 0 ? b : 1);
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_step_f64(const nbody_mi355x_step_args* args, void* stream) {
  int rc = nbody_internal::check_step_args(args);
  if (rc != NBODY_MI355X_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e;
  const int v = resolve_variant(args);
  if (v < 0 || is_sym_variant(v)) {
    nbody_mi355x_step_args a = *args;
    if (v > 0) a.variant = v;
    return nbody_internal::sym_step(a, s);  // (v < 0: reports why it does not apply)
  }
  const bool fp32 = args->precision == NBODY_MI355X_FP32 &&
                    !(args->acc_in != nullptr && args->j0_begin == args->j0_end && args->j1_begin == args->j1_end);
  if (v == NBODY_MI355X_VARIANT_WAVE) {
    e = launch_wave(*args, s);
  } else if (fp32) {
    switch (v) {
      case 1: e = launch_step<1, float, 8>(*args, s); break;
      case 4: e = launch_step<4, float, 4>(*args, s); break;
      default: e = launch_step<2, float, 8>(*args, s); break;
    }
  } else {
    switch (v) {
      case 1: e = launch_step<1, double, 8>(*args, s); break;
      case 2: e = launch_step<2, double, 4>(*args, s); break;
      case 3: e = launch_step<1, double, 4>(*args, s); break;
      case 4: e = launch_step<4, double, 2>(*args, s); break;
      case 5: e = launch_step<2, double, 8>(*args, s); break;
      default: e = launch_small(*args, s); break;  // 6
    }
  }
  if (e != hipSuccess) {
    set_error("step launch: %s", hipGetErrorString(e));
    return NBODY_MI355X_ERR_LAUNCH;
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_step_workspace_bytes(const nbody_mi355x_step_args* args, int64_t* bytes) {
  if (!args || !bytes) { set_error("step_workspace_bytes: null argument"); return NBODY_MI355X_ERR_INVALID; }
  int64_t b = 0;
  nbody_mi355x_step_args a = *args;
  if (!is_sym_variant(a.variant)) a.variant = NBODY_MI355X_VARIANT_SYM;
  *bytes = nbody_internal::sym_supported(a, &b, nullptr) ? b : 0;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_step_workspace_min_bytes(const nbody_mi355x_step_args* args, int64_t* bytes) {
  if (!args || !bytes) { set_error("step_workspace_min_bytes: null argument"); return NBODY_MI355X_ERR_INVALID; }
  int64_t pref = 0, least = 0;
  nbody_mi355x_step_args a = *args;
  if (!is_sym_variant(a.variant)) a.variant = NBODY_MI355X_VARIANT_SYM;
  *bytes = nbody_internal::sym_supported(a, &pref, &least) ? least : 0;
  return NBODY_MI355X_OK;
}

namespace nbody_internal {
// What a handle takes for the symmetric kernels out of `budget` bytes: the preferred size if
// it fits, else the budget itself down to the minimum, else 0 (one-sided kernel).
// Half of the free device memory; NBODY_MI355X_WS_BUDGET_MB lowers it (a way to leave room
// for what the caller allocates after the handle -- and how the tests stand in for a full GPU).
int64_t workspace_budget(size_t free_bytes) {
  int64_t budget = static_cast<int64_t>(free_bytes / 2);
  if (const char* e = getenv("NBODY_MI355X_WS_BUDGET_MB")) {
    const long long mb = atoll(e);
    if (mb >= 0 && (mb << 20) < budget) budget = mb << 20;
  }
  return budget;
}

int64_t workspace_to_take(const nbody_mi355x_step_args& probe, int64_t budget) {
  int64_t pref = 0, least = 0;
  if (nbody_mi355x_step_workspace_bytes(&probe, &pref) != NBODY_MI355X_OK || pref <= 0) return 0;
  if (nbody_mi355x_step_workspace_min_bytes(&probe, &least) != NBODY_MI355X_OK || least <= 0) return 0;
  if (pref <= budget) return pref;
  return least <= budget ? budget : 0;
}
}  // namespace nbody_internal

extern "C" int nbody_mi355x_pack_f64(const double* aos, int64_t count, double* pos_slab, double* vel,
                                     int64_t shard_slots, void* stream) {
  if (!aos || !pos_slab || !vel || count < 0 || count > shard_slots) {
    set_error("pack: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (count == 0) return NBODY_MI355X_OK;
  const unsigned blocks = static_cast<unsigned>((count + 255) / 256);
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), aos,
                     count, pos_slab, vel, shard_slots);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("pack launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_unpack_f64(const double* pos_slab, const double* vel, int64_t count,
                                       int64_t shard_slots, double* aos, void* stream) {
  if (!aos || !pos_slab || !vel || count < 0 || count > shard_slots) {
    set_error("unpack: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (count == 0) return NBODY_MI355X_OK;
  const unsigned blocks = static_cast<unsigned>((count + 255) / 256);
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pos_slab, vel, count, shard_slots, aos);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("unpack launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_energy_f64(const double* pos, const double* mass, const double* vel,
                                       int64_t shard_slots, int32_t num_shards, int32_t own_shard,
                                       int64_t own_count, double G, double* scratch, double* result,
                                       void* stream) {
  if (!pos || !mass || !vel || !scratch || !result || shard_slots <= 0 || shard_slots % kTile != 0 ||
      num_shards <= 0 || own_shard < 0 || own_shard >= num_shards || own_count < 0 ||
      own_count > shard_slots) {
    set_error("energy: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t blocks = (own_count + kTile - 1) / kTile;
  if (blocks > 0) {
    hipLaunchKernelGGL(energy_partial_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kTile), 0, s, pos,
                       mass, vel, shard_slots, num_shards, own_shard, own_count, G, scratch);
  }
  hipLaunchKernelGGL(energy_final_kernel, dim3(1), dim3(kTile), 0, s, scratch, blocks, result);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("energy launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}
