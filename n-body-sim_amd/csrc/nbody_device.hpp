// nbody_device.hpp -- device code shared by the kernels of libnbody_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nbody_mi355x.h"

namespace nbody_device {

constexpr int kTile = NBODY_MI355X_TILE;  // threads per workgroup == j bodies per tile

struct alignas(32) JBody {
  double x, y, z, m;
};

// 2^-300.  With eps2 == 0 the kernel adds this to r^2 inside the first FMA of
// the r^2 chain.  For any pair a simulation can hold apart (r > 6e-38 in the
// unit of length) r^2 + 2^-300 rounds to r^2, so the value is bit-for-bit the
// unsoftened one; for r^2 == 0 (i == j, or coincident bodies) it makes
// rsq finite (2^150), m * 2^450 stays finite for |m| < 2^574 (1e172: any mass a
// double-precision scene holds, SI units included), and the pair contributes
// s * 0 = 0: exactly the reference's `if (r_squared != 0)`
// (particle_simulation.h:269) without a compare or a select.  (A smaller seed
// buys nothing and costs mass range: 2^-600 overflowed m * y0^3 from 2^124 = 2e37 kg up.)
constexpr double kTinyR2 = 0x1p-300;

// One interaction in double: 16 FP64 VALU operations + v_rsq_f64:
//   3 sub, 3 fma (r^2), rsq, 7 for m * r^-3 (below), 3 fma (accumulate).
// y0 = v_rsq_f64(r2) is only a seed (the ISA promises about 2^-23 relative).  With
// u = y0^2 and e = 1 - r2*u (one FMA, so e carries just u's rounding) the exact value is
//   r2^(-3/2) = y0^3 * (1 - e)^(-3/2) = y0^3 * (1 + 3/2 e + 15/8 e^2 + 35/16 e^3 ...)
// and the series is cut after e^2: the truncation error is 2.2 |e|^3, below
// 2^-53 for any seed better than 2^-19.  c15/c1875 are 1.5 and 1.875 held in
// registers by the caller (neither is an inline constant of the ISA; as literals
// they cost two v_mov per interaction).
__device__ __forceinline__ void pair_f64(double xi, double yi, double zi, double xj, double yj,
                                         double zj, double mj, double eps2, double c15,
                                         double c1875, double& ax, double& ay, double& az) {
  const double dx = xj - xi;
  const double dy = yj - yi;
  const double dz = zj - zi;
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
  const double y0 = __builtin_amdgcn_rsq(r2);
  const double u = y0 * y0;
  const double e = __builtin_fma(-r2, u, 1.0);
  const double q = mj * y0;
  const double w = q * u;                       // m * y0^3
  const double p = __builtin_fma(e, c1875, c15);
  const double we = w * e;
  const double s = __builtin_fma(we, p, w);     // m * r^-3
  ax = __builtin_fma(s, dx, ax);
  ay = __builtin_fma(s, dy, ay);
  az = __builtin_fma(s, dz, az);
}

// Mixed variant: separations in double (so that close pairs of bodies far from the
// origin do not cancel catastrophically), r^2, r^-3 and the running sums in float.
// v_rsq_f32 is accurate to ~1 ulp, no correction step.  The 2^-600 trick of the double
// path does not fit float's exponent range (m * y0^3 would overflow for r^2 -> 0), so the
// r^2 == 0 rule is an explicit select here (one v_cndmask_b32).
__device__ __forceinline__ void pair_f32(double xi, double yi, double zi, double xj, double yj,
                                         double zj, float mj, float eps2, float& ax, float& ay,
                                         float& az) {
  const float dx = static_cast<float>(xj - xi);
  const float dy = static_cast<float>(yj - yi);
  const float dz = static_cast<float>(zj - zi);
  // (the select is on the BARE separation: the self pair has d2 == 0 whatever the softening, and
  // its weight m eps2^-3/2 may be beyond float -- inf * 0 would be NaN)
  const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  const float y0 = __builtin_amdgcn_rsqf(d2 + eps2);
  const float s3 = (mj * y0) * (y0 * y0);
  const float s = (d2 > 0.0f) ? s3 : 0.0f;
  ax = __builtin_fmaf(s, dx, ax);
  ay = __builtin_fmaf(s, dy, ay);
  az = __builtin_fmaf(s, dz, az);
}


// The fused epilogue of a small step for own-slab slot li: forward Euler and the optional
// snapshot.  ref: particle_simulation.h:288-300 -- `v + (dt*a)*G` and `r + dt*v` are each
// one FMA, which is also what GCC -O3 -mfma makes of the reference's expressions;
// snapshot: ref: particle_simulation.h:304-315 / particle_simulation_GPU.cu:127-141.
__device__ __forceinline__ void integrate_and_store(const nbody_mi355x_step_args& a, int64_t li,
                                                    double xi, double yi, double zi, double axd,
                                                    double ayd, double azd) {
  const int64_t S = a.shard_slots;
  const double vx = a.vel[li], vy = a.vel[S + li], vz = a.vel[2 * S + li];
  if (a.integrator >= NBODY_MI355X_KDK) {
    // kick-drift-kick leapfrog: (xi, yi, zi) are the positions AFTER this step's drift, the
    // sums are a(r'): the second kick completes the step -- {r', v', a} is the synchronised
    // state -- and KDK_CONTINUE goes straight on with the next step's first kick and drift
    const double h = 0.5 * a.dt;
    const double svx = __builtin_fma(h * axd, a.G, vx);
    const double svy = __builtin_fma(h * ayd, a.G, vy);
    const double svz = __builtin_fma(h * azd, a.G, vz);
    if (a.integrator == NBODY_MI355X_KDK_CONTINUE) {
      const double nvx = __builtin_fma(h * axd, a.G, svx);
      const double nvy = __builtin_fma(h * ayd, a.G, svy);
      const double nvz = __builtin_fma(h * azd, a.G, svz);
      a.vel[li] = nvx;
      a.vel[S + li] = nvy;
      a.vel[2 * S + li] = nvz;
      double* nown = a.pos_next + static_cast<int64_t>(a.own_shard) * 3 * S;
      nown[li] = __builtin_fma(a.dt, nvx, xi);
      nown[S + li] = __builtin_fma(a.dt, nvy, yi);
      nown[2 * S + li] = __builtin_fma(a.dt, nvz, zi);
    } else {
      a.vel[li] = svx;
      a.vel[S + li] = svy;
      a.vel[2 * S + li] = svz;
    }
    if (a.kdk_acc != nullptr) {
      a.kdk_acc[li] = axd;
      a.kdk_acc[S + li] = ayd;
      a.kdk_acc[2 * S + li] = azd;
    }
    if (a.snapshot != nullptr) {
      double* o = a.snapshot + 9 * li;
      o[0] = xi; o[1] = yi; o[2] = zi;
      o[3] = svx; o[4] = svy; o[5] = svz;
      o[6] = axd * a.G; o[7] = ayd * a.G; o[8] = azd * a.G;
    }
    return;
  }
  const double nvx = __builtin_fma(a.dt * axd, a.G, vx);
  const double nvy = __builtin_fma(a.dt * ayd, a.G, vy);
  const double nvz = __builtin_fma(a.dt * azd, a.G, vz);
  // the reference drifts with the OLD velocity; KICK_DRIFT with the one just kicked
  const bool kd = a.integrator == NBODY_MI355X_KICK_DRIFT;
  const double nx = __builtin_fma(a.dt, kd ? nvx : vx, xi);
  const double ny = __builtin_fma(a.dt, kd ? nvy : vy, yi);
  const double nz = __builtin_fma(a.dt, kd ? nvz : vz, zi);
  a.vel[li] = nvx;
  a.vel[S + li] = nvy;
  a.vel[2 * S + li] = nvz;
  double* nown = a.pos_next + static_cast<int64_t>(a.own_shard) * 3 * S;
  nown[li] = nx;
  nown[S + li] = ny;
  nown[2 * S + li] = nz;
  if (a.snapshot != nullptr) {
    double* o = a.snapshot + 9 * li;
    o[0] = nx; o[1] = ny; o[2] = nz;
    o[3] = nvx; o[4] = nvy; o[5] = nvz;
    o[6] = axd * a.G; o[7] = ayd * a.G; o[8] = azd * a.G;
  }
}

}  // namespace nbody_device
