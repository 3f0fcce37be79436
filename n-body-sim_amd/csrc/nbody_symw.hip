// nbody_symw.hip -- the pair-symmetric small step with WAVE tiles: the mid-size systems
// (2049 bodies .. 262 143 slots on a GPU), where a launch of workgroup tiles (nbody_sym.hip)
// is a few hundred big work items on 1024 SIMDs -- mostly ramp and tail -- and smaller
// workgroup tiles pay for it with 6 DPP moves per pair instead of per 8.
//
// Same arithmetic, same workspace layout, same finishing kernels as nbody_sym.hip; what
// differs is who shares what:
//   * a block is 64*IPT slots (64 .. 512); ALL FOUR waves of a workgroup hold the same block
//     row in registers (IPT bodies per lane), so IPT -- the number of pairs one LDS read and
//     one set of DPP moves serve -- no longer fixes the size of a work item;
//   * the work of row I is the list of its (tile, group) entries: tiles (I, I+d), d = 0 .. D,
//     each IPT groups of 64 j bodies, each group a walk of 64 rotation steps = 8 OCTETS.  The
//     list is dealt to the row's waves (4 per workgroup, `chunks` workgroups per row) in octets:
//     wave k takes octets [k q, (k+1) q) -- whole entries in the middle and a PART of an entry
//     at either end (q >= 8: an entry is cut at most once).  With whole entries per wave a
//     launch is level only where the entries happen to divide evenly over the chip's SIMDs
//     (16 384 bodies: 8.125 entries per SIMD -> some do 9: 10 % lost; round 2's sizes between
//     the powers of two 3-8 % under their neighbours); in octets every wave has the same work
//     to 1/8 entry.  A wave stages its group in its OWN 72 LDS entries (no workgroup barrier
//     anywhere in the loop), walks its part of it (group_steps) and STORES the group's j-side
//     sums -- to C[d][I][group] (whole entries and first parts) or, for the second part of a
//     cut entry, to Cb[I][wave]; after steps [8 k0, 8 k1) lane l holds body (l + 8 k1) mod 64;
//   * the i-side sums of the 4 waves are added through LDS in wave order at the end and
//     stored once per workgroup to R[c][I].
// How many workgroups a row gets (`chunks`) comes from a model of the launch (plan_wave,
// nbody_sym.hip): equal work items make the dispatch-round model exact.
// The price is C traffic that grows as N^2 / (64*IPT) (each j sum is stored after meeting
// 64*IPT bodies instead of 256*IPT): that is why the large systems keep workgroup tiles.
// No atomics: bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nbody_sym_device.hpp"

namespace nbody_sym {

namespace {

// float only: the centroid of the row's real bodies (see row_origin, nbody_sym_device.hpp);
// here every wave holds the whole row, so the sum is a wave reduction and needs no LDS.
template <int IPT>
__device__ __forceinline__ void wave_origin(const double* px, const double* pm, int64_t S, int lane, double* ox,
                                            double* oy, double* oz, double* scale) {
  double sx = 0.0, sy = 0.0, sz = 0.0, cnt = 0.0;
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 64 + lane;
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
  const double n = uniform(cnt);
  double cx = uniform(sx) / n, cy = uniform(sy) / n, cz = uniform(sz) / n;
  const bool ok = n > 0.0 && (cx - cx == 0.0) && (cy - cy == 0.0) && (cz - cz == 0.0);
  cx = ok ? cx : 0.0;
  cy = ok ? cy : 0.0;
  cz = ok ? cz : 0.0;
  double ext = 0.0;  // the row's extent about that point sets the scale (pow2_scale)
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int idx = k * 64 + lane;
    ext = __builtin_fmax(ext, body_extent(px[idx], px[S + idx], px[2 * S + idx], pm[idx], cx, cy, cz));
  }
  for (int o = 32; o > 0; o >>= 1) ext = __builtin_fmax(ext, __shfl_down(ext, o, 64));
  *ox = cx;
  *oy = cy;
  *oz = cz;
  *scale = uniform(pow2_scale(ext));
}

template <int IPT, typename T, bool FAST = false>
__global__ __launch_bounds__(256) void sym_wave_kernel(const SymArgs a) {
  constexpr int B = 64 * IPT;
  constexpr bool kDouble = sizeof(T) == sizeof(double);
  __shared__ P2<T> xy[4][kGrp];
  __shared__ P2<T> zm[4][kGrp];
  __shared__ double red[3][3][B];  // the i-side sums of waves 1 .. 3

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int il = blockIdx.x, c = blockIdx.y;  // own row (local index), chunk of its list
  const int I = a.row0 + il;
  const int64_t S = a.S;
  T c15 = static_cast<T>(1.5), c1875 = static_cast<T>(1.875);
  asm volatile("" : "+v"(c15), "+v"(c1875));
  if (a.trace != nullptr && tid == 0) {  // (tuning runs only: where and when this workgroup ran)
    unsigned long long* t = a.trace + 12 * (static_cast<size_t>(c) * gridDim.x + il);
    t[0] = wall_clock64();
    t[2] = __builtin_amdgcn_s_getreg((4) | (31 << 11));   // HW_REG_HW_ID
    t[3] = __builtin_amdgcn_s_getreg((20) | (31 << 11));  // HW_REG_XCC_ID
    t[10] = __builtin_readcyclecounter();                 // shader clock (s_memtime): with t[0], t[1] the clock it ran at
  }

  T xi[IPT], yi[IPT], zi[IPT], mi[IPT], ax[IPT], ay[IPT], az[IPT];
  double ox = 0.0, oy = 0.0, oz = 0.0, sc = 1.0;  // float only: origin and scale, see wave_origin
  {
    const double *px, *pm;
    block_base<B>(a, I, &px, &pm);
    if constexpr (!kDouble) wave_origin<IPT>(px, pm, S, lane, &ox, &oy, &oz, &sc);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int idx = k * 64 + lane;
      if constexpr (!kDouble) {
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
      if (static_cast<int64_t>(il) * B + idx >= a.own_count) xi[k] = yi[k] = zi[k] = far_away<T>();  // padding: see far_away
    }
  }
  // double: 2^-300 stands in for "no softening" (nbody_device.hpp); float: plain 0 (r^2 == 0
  // is dealt with in the walk, nbody_sym_device.hpp) or, softened, eps2 in the row's scaled units
  const T eps2 = static_cast<T>((a.eps2 > 0.0) ? a.eps2 * sc * sc : (kDouble ? kTinyR2 : 0.0));
  // do all bodies of this row have one (finite, non-zero) mass?  Every wave holds the whole
  // row: a wave-uniform answer that is the same in all four.
  const T m_ref = __shfl(mi[0], 0, 64);
  const T inv_probe = 1 / m_ref;  // not 0, inf or NaN -- nor so small that 1 / m_ref overflows
  bool row_same = (m_ref != 0) && (m_ref - m_ref == 0) && (inv_probe - inv_probe == 0);
#pragma unroll
  for (int k = 0; k < IPT; k++)  // (padding slots, out of everybody's reach, have no say)
    row_same = row_same && (mi[k] == m_ref || static_cast<int64_t>(il) * B + k * 64 + lane >= a.own_count);
  const bool row_uniform = __all(row_same) && !a.no_eq;
  const T inv_m_ref = row_uniform ? inv_probe : 0;
  // float: units of the row's scale and (uniform row) mass, as in sym_tiles_kernel
  const double m_unit = (!kDouble && row_uniform) ? static_cast<double>(m_ref) : 1.0;
  const double stage_m = 1.0 / m_unit;
  const double unit_i = sc * sc * m_unit;

  // this launch's part of the row's list: distances d_lo .. dh, IPT entries of 8 octets each;
  // this wave's octets of it (wave-uniform: kept in scalar registers)
  int dh = a.d_hi;
  if (2 * dh == a.nb && I >= a.nb / 2) dh--;  // the antipodal tile belongs to the low rows
  const int n_octets = (dh - a.d_lo + 1) * IPT * 8;
  const int wr = __builtin_amdgcn_readfirstlane(c * 4 + w);  // this wave among the row's
  int o_begin = wr < a.ws ? wr * a.q : a.ws * a.q + (wr - a.ws) * a.qb;
  int o_end = o_begin + (wr < a.ws ? a.q : a.qb);
  if (o_begin > n_octets) o_begin = n_octets;
  if (o_end > n_octets) o_end = n_octets;

  P2<T>* wxy = xy[w];
  P2<T>* wzm = zm[w];
  // (tuning runs only: where wave 0's time goes)
  const bool timed = a.trace != nullptr;
  unsigned long long t_loop = timed ? wall_clock64() : 0, t_stage = 0, t_walk = 0, t_store = 0, t_done = 0, n_done = 0, t0 = t_loop;
  for (int o = o_begin; o < o_end;) {
    const int e = o >> 3, k0 = o & 7;                        // entry, first octet of it taken here
    const int k1 = o_end - 8 * e < 8 ? o_end - 8 * e : 8;    // one past the last
    o = 8 * e + k1;
    const int d = a.d_lo + e / IPT, g = e % IPT;
    int J = I + d;
    if (J >= a.nb) J -= a.nb;
    const double *px, *pm;
    block_base<B>(a, J, &px, &pm);
    const int idx = g * 64 + lane;
    P2<T> vxy, vzm;
    if constexpr (!kDouble) {  // relative to the row's origin, narrowed after
      vxy = P2<T>{static_cast<T>((px[idx] - ox) * sc), static_cast<T>((px[S + idx] - oy) * sc)};
      vzm = P2<T>{static_cast<T>((px[2 * S + idx] - oz) * sc), static_cast<T>(pm[idx] * stage_m)};
    } else {  // (a uniform row stages m_j / m_ref: kRowUnit, nbody_sym_device.hpp)
      vxy = P2<T>{static_cast<T>(px[idx]), static_cast<T>(px[S + idx])};
      vzm = P2<T>{static_cast<T>(px[2 * S + idx]), static_cast<T>(row_uniform ? pm[idx] * inv_m_ref : pm[idx])};
    }
    // padding of the OWN slab (known by its index): out of everybody's reach, as on the i side (far_away),
    // so that the group keeps the equal-mass form; its j-side sums are nobody's
    const int jl_own = J - a.row0;
    const bool pad_j = jl_own >= 0 && jl_own < a.nbs && static_cast<int64_t>(jl_own) * B + idx >= a.own_count;
    if (pad_j) {
      vxy = P2<T>{far_away<T>(), far_away<T>()};
      vzm.a = far_away<T>();
    }
    // (the wave's previous group has been read to the end: LDS serves a wave's accesses in order)
    wxy[lane] = vxy;
    wzm[lane] = vzm;
    if (lane < kGrp - 64) {  // the group's first entries again behind its end
      wxy[lane + 64] = vxy;
      wzm[lane + 64] = vzm;
    }
    const bool equal_mass = __all(static_cast<T>(pm[idx]) == m_ref || pad_j) && row_uniform;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (timed) { const unsigned long long t = wall_clock64(); t_stage += t - t0; t0 = t; }

    T fx = 0, fy = 0, fz = 0;
#define NBODY_GROUP(SYM_, MODE_)                                                                                  \
  group_steps<IPT, SYM_, MODE_, T>(wxy, wzm, 0, lane, xi, yi, zi, mi, ax, ay, az, eps2, c15, c1875, inv_m_ref, k0, \
                                   k1, &fx, &fy, &fz)
    if (d != 0) {
      if (kDouble) {
        if (equal_mass) {
          if constexpr (FAST) NBODY_GROUP(true, kDeferredFast);  // (NBODY_MI355X_FP64_FAST, 512-body blocks)
          else NBODY_GROUP(true, kDeferred);
          fx *= m_ref;
          fy *= m_ref;
          fz *= m_ref;
        } else if (row_uniform) {
          NBODY_GROUP(true, kRowUnit);
          fx *= m_ref;
          fy *= m_ref;
          fz *= m_ref;
        } else {
          NBODY_GROUP(true, kGeneral);
        }
      } else {
        if (equal_mass) NBODY_GROUP(true, kSameMass);
        else NBODY_GROUP(true, kGeneral);
      }
      // the sums of the steps walked: lane l holds those of body (l + 8 k1) mod 64 of the group
      // (a whole walk, or one that ends with the entry: its own body l).  A part that BEGINS
      // the entry goes where a whole entry's sums go, C[d][I][group]; a part that begins
      // inside it is the second part of a cut entry: Cb[I][this wave]
      if (timed) { const unsigned long long t = wall_clock64() + (fx == static_cast<T>(12345.5) ? 1 : 0); t_walk += t - t0; t0 = t; }
      const int body = (lane + 8 * k1) & 63;
      double* out = k0 == 0 ? a.C + static_cast<int64_t>(d - a.c_base) * 3 * S + static_cast<int64_t>(il) * B + g * 64
                            : a.Cb + (static_cast<int64_t>(il) * a.wstride + wr) * 192;
      const int64_t stride = k0 == 0 ? S : 64;
      if constexpr (kDouble) {
        out[body] = fx;
        out[stride + body] = fy;
        out[2 * stride + body] = fz;
      } else {  // out of the scaled units (of the common mass too where the group had it)
        const double unit_j = sc * sc * (equal_mass ? static_cast<double>(m_ref) : 1.0);
        out[body] = fx * unit_j;
        out[stride + body] = fy * unit_j;
        out[2 * stride + body] = fz * unit_j;
      }
    } else {  // a group of the row's own block: i side only, every ordered pair
      if (kDouble && row_uniform) NBODY_GROUP(false, kRowUnit);
      else NBODY_GROUP(false, kGeneral);
    }
#undef NBODY_GROUP
    __builtin_amdgcn_wave_barrier();
    if (timed) { const unsigned long long t = wall_clock64(); (d != 0 ? t_store : t_walk) += t - t0; t0 = t; n_done++; }
  }
  if (timed) t_done = wall_clock64();

  // the four waves' i-side sums, added in wave order
  if (w > 0) {
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      red[w - 1][0][k * 64 + lane] = ax[k];
      red[w - 1][1][k * 64 + lane] = ay[k];
      red[w - 1][2][k * 64 + lane] = az[k];
    }
  }
  __syncthreads();
  if (w == 0) {
    double* Rs = a.R + static_cast<int64_t>(a.r_base + c) * 3 * S + static_cast<int64_t>(il) * B;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int idx = k * 64 + lane;
      double x = ax[k], y = ay[k], z = az[k];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        x += red[q][0][idx];
        y += red[q][1][idx];
        z += red[q][2][idx];
      }
      if (kDouble && row_uniform) {  // back from units of 1/m_ref
        x *= m_ref;
        y *= m_ref;
        z *= m_ref;
      }
      if constexpr (!kDouble) {
        x *= unit_i;
        y *= unit_i;
        z *= unit_i;
      }
      Rs[idx] = x;
      Rs[S + idx] = y;
      Rs[2 * S + idx] = z;
    }
  }
  if (a.trace != nullptr && tid == 0) {
    unsigned long long* t = a.trace + 12 * (static_cast<size_t>(c) * gridDim.x + il);
    t[1] = wall_clock64();
    t[4] = t_loop;   // the row is in registers, the list begins
    t[5] = t_stage;  // summed over wave 0's entries: from the entry's start to its bodies staged in LDS
    t[6] = t_walk;   //   ... the walk
    t[7] = t_store;  //   ... the j-side sums stored
    t[8] = t_done;   // the list is done: what follows is the i-side sums through LDS and the R store
    t[9] = n_done;   // entries (or parts) wave 0 walked
    t[11] = __builtin_readcyclecounter();
  }
}

template <int IPT, typename T>
hipError_t launch_one(const SymArgs& sa, dim3 grid, hipStream_t stream) {
  if constexpr (IPT == 8 && sizeof(T) == sizeof(double)) {  // the FAST tier: 512-body blocks only
    if (sa.fast) {
      hipLaunchKernelGGL((sym_wave_kernel<IPT, T, true>), grid, dim3(256), 0, stream, sa);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((sym_wave_kernel<IPT, T>), grid, dim3(256), 0, stream, sa);
  return hipGetLastError();
}

}  // namespace

// NBODY_MI355X_SYMW_TRACE=FILE (tuning only): every wave-tile launch is followed by a
// synchronisation and appends one line per workgroup to FILE:
//   launch  row  chunk  start  end (100 MHz ticks)  HW_ID  XCC_ID  list-begins  sum(stage)  sum(walk)  sum(store)  list-done  entries  shader-clocks
// -- where the dispatcher put each workgroup and when it ran.  Allocates, blocks the host: never
// set it in a measured run.
hipError_t launch_wave_tiles_plain(int ipt, bool fp32, const SymArgs& sa, dim3 grid, hipStream_t stream);
hipError_t launch_wave_tiles(int ipt, bool fp32, const SymArgs& sa_in, dim3 grid, hipStream_t stream) {
  SymArgs sa = sa_in;
  sa.trace = nullptr;
  const char* path = getenv("NBODY_MI355X_SYMW_TRACE");
  if (path == nullptr || path[0] == 0) return launch_wave_tiles_plain(ipt, fp32, sa, grid, stream);
  static int launch_no = 0;
  const size_t count = static_cast<size_t>(grid.x) * grid.y;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&sa.trace), count * 12 * sizeof(unsigned long long));
  if (e != hipSuccess) return e;
  e = launch_wave_tiles_plain(ipt, fp32, sa, grid, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) {
    std::vector<unsigned long long> h(count * 12);
    e = hipMemcpy(h.data(), sa.trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (FILE* f = e == hipSuccess ? fopen(path, "a") : nullptr) {
      for (size_t k = 0; k < count; k++)
        fprintf(f, "%d %zu %zu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", launch_no, k % grid.x, k / grid.x, h[12 * k],
                h[12 * k + 1], h[12 * k + 2], h[12 * k + 3], h[12 * k + 4], h[12 * k + 5], h[12 * k + 6], h[12 * k + 7],
                h[12 * k + 8], h[12 * k + 9], h[12 * k + 11] - h[12 * k + 10]);
      fclose(f);
    }
  }
  launch_no++;
  (void)hipFree(sa.trace);
  return e;
}

hipError_t launch_wave_tiles_plain(int ipt, bool fp32, const SymArgs& sa, dim3 grid, hipStream_t stream) {
  if (fp32) {
    switch (ipt) {
      case 1: return launch_one<1, float>(sa, grid, stream);
      case 2: return launch_one<2, float>(sa, grid, stream);
      case 4: return launch_one<4, float>(sa, grid, stream);
      case 8: return launch_one<8, float>(sa, grid, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (ipt) {
    case 1: return launch_one<1, double>(sa, grid, stream);
    case 2: return launch_one<2, double>(sa, grid, stream);
    case 4: return launch_one<4, double>(sa, grid, stream);
    case 8: return launch_one<8, double>(sa, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nbody_sym
