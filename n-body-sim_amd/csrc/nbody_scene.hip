// nbody_scene.hip -- the benchmark scenes generated AT SCALE: on the device, straight into the
// slab layout, and by the same code on the host's threads ("initial-condition
// layer on device / at scale").
//
// The reference builds a scene on the host, one SetParticle() at a time, from ONE libstdc++
// random stream (saturn_ring.cc:42-63, particle_initializer.h:120-175) -- serial by
// construction, and nbody_ics.cpp keeps that shape (one xoshiro256** stream per scene).  Here
// every body has a stream of its own, seeded from (seed, body index) through splitmix64, so
// that body i is a pure function of (scene, i): any range of bodies can be produced anywhere,
// in any order -- a GPU thread per body, or one host thread per range -- and comes out the
// same up to the last-bit differences between the device's and the host's pow / log / cos.
// The recipes are those of nbody_ics.cpp (Plummer sphere by Aarseth-Henon-Wielen rejection,
// cold uniform cube, the reference's ring of NonCircularOrbitMoon bodies).
// A scene made here is NOT the scene nbody_mi355x_ic_* makes from the same seed (another way
// of drawing), only statistically the same.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "nbody_internal.h"

using nbody_internal::set_error;

namespace {

struct BodyRng {  // splitmix64-seeded xoshiro256**, as nbody_ics.cpp
  uint64_t s[4];
  __host__ __device__ BodyRng(uint64_t seed, uint64_t body) {
    uint64_t x = seed * 0xD6E8FEB86659FD93ull + body;  // one stream per (seed, body)
    for (int i = 0; i < 4; i++) {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      s[i] = z ^ (z >> 31);
    }
  }
  __host__ __device__ static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  __host__ __device__ uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  __host__ __device__ double u01() { return static_cast<double>(next() >> 11) * 0x1p-53; }  // [0,1)
  __host__ __device__ double u01_open() {  // (0,1): a zero is drawn again (2^-53 each time)
    double u = u01();
    for (int k = 0; k < 64 && u == 0.0; k++) u = u01();
    return u == 0.0 ? 0x1p-53 : u;
  }
  __host__ __device__ double normal(double sigma) {  // Box-Muller, one deviate per call
    const double u = u01_open();
    return sigma * sqrt(-2.0 * log(u)) * cos(2.0 * M_PI * u01());
  }
};

struct Body {
  double r[3], v[3], m;
};

__host__ __device__ inline void isotropic(BodyRng& g, double len, double* out) {
  const double z = 1.0 - 2.0 * g.u01();
  const double phi = 2.0 * M_PI * g.u01();
  const double s = sqrt(1.0 - z * z);
  out[0] = len * s * cos(phi);
  out[1] = len * s * sin(phi);
  out[2] = len * z;
}

// Rejection loops are capped (a cap no draw reaches: the acceptance rates are 0.97 and 0.43),
// so that every GPU lane leaves them whatever the generator returns.
constexpr int kMaxTries = 4096;

__host__ __device__ inline Body scene_body(const nbody_mi355x_scene& sc, int64_t i) {
  BodyRng g(sc.seed, static_cast<uint64_t>(i));
  Body b;
  if (sc.scene == NBODY_MI355X_SCENE_PLUMMER) {
    // units G*M = 1, scale radius 1, truncated at 10 radii (nbody_ics.cpp, ic_plummer)
    double r = 1.0;
    for (int k = 0; k < kMaxTries; k++) {
      const double u = g.u01_open();
      r = 1.0 / sqrt(pow(u, -2.0 / 3.0) - 1.0);
      if (r <= 10.0) break;
      r = 1.0;
    }
    double q = 0.0;
    for (int k = 0; k < kMaxTries; k++) {
      q = g.u01();
      const double y = 0.1 * g.u01();
      if (!(y > q * q * pow(1.0 - q * q, 3.5))) break;
      q = 0.0;
    }
    const double v = q * sqrt(2.0) * pow(1.0 + r * r, -0.25);
    isotropic(g, r, b.r);
    isotropic(g, v, b.v);
    b.m = 1.0 / (sc.G * static_cast<double>(sc.num_particles));
  } else if (sc.scene == NBODY_MI355X_SCENE_UNIFORM_CUBE) {
    for (int k = 0; k < 3; k++) {
      b.r[k] = 2.0 * g.u01() - 1.0;
      b.v[k] = 0.0;
    }
    b.m = 1.0 / (sc.G * static_cast<double>(sc.num_particles));
  } else {  // NBODY_MI355X_SCENE_RING: body 0 the central mass, the others as NonCircularOrbitMoon
    if (i == 0) {
      for (int k = 0; k < 3; k++) b.r[k] = b.v[k] = 0.0;
      b.m = sc.central_mass;
      return b;
    }
    const double G_orbit = 6.67430e-11;  // the initializer's own constant (particle_initializer.h)
    const double dz = g.normal(sc.thickness_sigma);
    const double dvx = g.normal(sc.v_sigma), dvy = g.normal(sc.v_sigma);
    const double phase = 2.0 * M_PI * g.u01();
    const double ms = g.normal(sc.mass_sigma);
    double radius = sc.ring_radius;
    for (int k = 0; k < kMaxTries; k++) {
      radius = sc.ring_radius + g.normal(sc.width_sigma);
      if (radius > 0.0) break;
      radius = sc.ring_radius;
    }
    const double speed = sqrt(G_orbit * sc.central_mass / radius);
    const double c = cos(phase), s = sin(phase);
    const double vx = dvx, vy = speed + dvy;
    b.r[0] = radius * c;
    b.r[1] = radius * s;
    b.r[2] = dz;
    b.v[0] = vx * c - vy * s;
    b.v[1] = vx * s + vy * c;
    b.v[2] = 0.0;
    b.m = ms * ms;
  }
  return b;
}

__global__ __launch_bounds__(256) void scene_fill_kernel(const nbody_mi355x_scene sc, int64_t first, int64_t count,
                                                         double* pos, double* vel, double* mass, int64_t S) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (k >= count) return;
  const Body b = scene_body(sc, first + k);
  for (int c = 0; c < 3; c++) {
    pos[c * S + k] = b.r[c];
    if (vel) vel[c * S + k] = b.v[c];
  }
  mass[k] = b.m;
}

// Sums of a slab's first `count` positions and velocities: ONE workgroup, every thread its
// strided share in index order, then a tree over the threads -- a fixed order of additions.
constexpr int kSumThreads = 1024;
__global__ __launch_bounds__(kSumThreads) void slab_sums_kernel(const double* pos, const double* vel, int64_t count,
                                                                 int64_t S, double* sums6) {
  __shared__ double part[6][kSumThreads];
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t k = threadIdx.x; k < count; k += kSumThreads)
    for (int c = 0; c < 3; c++) {
      a[c] += pos[c * S + k];
      if (vel) a[3 + c] += vel[c * S + k];
    }
  for (int c = 0; c < 6; c++) part[c][threadIdx.x] = a[c];
  __syncthreads();
  for (int o = kSumThreads / 2; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o)
      for (int c = 0; c < 6; c++) part[c][threadIdx.x] += part[c][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 6) sums6[threadIdx.x] = part[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void slab_shift_kernel(double* pos, double* vel, int64_t count, int64_t S, double sx,
                                                         double sy, double sz, double svx, double svy, double svz) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (k >= count) return;
  pos[k] -= sx;
  pos[S + k] -= sy;
  pos[2 * S + k] -= sz;
  if (vel) {
    vel[k] -= svx;
    vel[S + k] -= svy;
    vel[2 * S + k] -= svz;
  }
}

bool scene_ok(const nbody_mi355x_scene* sc, const char* who) {
  if (!sc || sc->num_particles <= 0) { set_error("%s: null scene or no particles", who); return false; }
  switch (sc->scene) {
    case NBODY_MI355X_SCENE_PLUMMER:
    case NBODY_MI355X_SCENE_UNIFORM_CUBE:
      if (!(sc->G > 0.0)) { set_error("%s: G must be positive", who); return false; }
      return true;
    case NBODY_MI355X_SCENE_RING:
      if (!(sc->central_mass > 0.0) || !(sc->ring_radius > 0.0) || sc->width_sigma < 0.0 ||
          sc->thickness_sigma < 0.0 || sc->v_sigma < 0.0 || sc->mass_sigma < 0.0) {
        set_error("%s: ring parameters out of range", who);
        return false;
      }
      return true;
    default:
      set_error("%s: unknown scene %d", who, static_cast<int>(sc->scene));
      return false;
  }
}

bool range_ok(const nbody_mi355x_scene* sc, int64_t first, int64_t count, const char* who) {
  if (first < 0 || count < 0 || first > sc->num_particles - count) {
    set_error("%s: bodies [%lld, %lld) are not inside the scene's %lld", who, (long long)first,
              (long long)(first + count), (long long)sc->num_particles);
    return false;
  }
  return true;
}

}  // namespace

namespace nbody_internal {
bool scene_valid(const nbody_mi355x_scene* sc, const char* who) { return scene_ok(sc, who); }
}  // namespace nbody_internal

extern "C" int nbody_mi355x_scene_fill_f64(const nbody_mi355x_scene* sc, int64_t first, int64_t count,
                                           double* pos_slab, double* vel, double* mass_slab, int64_t shard_slots,
                                           void* stream) {
  if (!scene_ok(sc, "scene_fill") || !range_ok(sc, first, count, "scene_fill")) return NBODY_MI355X_ERR_INVALID;
  if (!pos_slab || !mass_slab || count > shard_slots) {
    set_error("scene_fill: null slab or %lld bodies for %lld slots", (long long)count, (long long)shard_slots);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (count == 0) return NBODY_MI355X_OK;
  const unsigned blocks = static_cast<unsigned>((count + 255) / 256);
  hipLaunchKernelGGL(scene_fill_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *sc, first,
                     count, pos_slab, vel, mass_slab, shard_slots);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("scene_fill launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_slab_sums_f64(const double* pos_slab, const double* vel, int64_t count,
                                          int64_t shard_slots, double* sums6, void* stream) {
  if (!pos_slab || !sums6 || count < 0 || count > shard_slots) {
    set_error("slab_sums: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  hipLaunchKernelGGL(slab_sums_kernel, dim3(1), dim3(kSumThreads), 0, static_cast<hipStream_t>(stream), pos_slab, vel,
                     count, shard_slots, sums6);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("slab_sums launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_slab_shift_f64(double* pos_slab, double* vel, int64_t count, int64_t shard_slots,
                                           const double* shift6, void* stream) {
  if (!pos_slab || !shift6 || count < 0 || count > shard_slots) {
    set_error("slab_shift: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  if (count == 0) return NBODY_MI355X_OK;
  const unsigned blocks = static_cast<unsigned>((count + 255) / 256);
  hipLaunchKernelGGL(slab_shift_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pos_slab, vel,
                     count, shard_slots, shift6[0], shift6[1], shift6[2], shift6[3], shift6[4], shift6[5]);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("slab_shift launch: %s", hipGetErrorString(e)); return NBODY_MI355X_ERR_LAUNCH; }
  return NBODY_MI355X_OK;
}

// The same bodies on the host: Particle<double> records and masses for bodies
// [first, first + count), on up to 16 threads (a body does not depend on its neighbours).
extern "C" int nbody_mi355x_scene_host(const nbody_mi355x_scene* sc, int64_t first, int64_t count, double* particles,
                                       double* mass) {
  if (!scene_ok(sc, "scene_host") || !range_ok(sc, first, count, "scene_host")) return NBODY_MI355X_ERR_INVALID;
  if (!particles || !mass) { set_error("scene_host: null output"); return NBODY_MI355X_ERR_INVALID; }
  const nbody_mi355x_scene scene = *sc;
  auto work = [&scene, first, particles, mass](int64_t lo, int64_t hi) {
    for (int64_t k = lo; k < hi; k++) {
      const Body b = scene_body(scene, first + k);
      for (int c = 0; c < 3; c++) {
        particles[6 * k + c] = b.r[c];
        particles[6 * k + 3 + c] = b.v[c];
      }
      mass[k] = b.m;
    }
  };
  unsigned hw = std::thread::hardware_concurrency();
  int64_t threads = hw == 0 ? 1 : (hw > 16 ? 16 : hw);
  if (count < 16384) threads = 1;
  std::vector<std::thread> pool;
  const int64_t per = (count + threads - 1) / threads;
  for (int64_t t = 1; t < threads; t++) {
    const int64_t lo = t * per, hi = lo + per < count ? lo + per : count;
    if (lo >= hi) continue;
    try {
      pool.emplace_back(work, lo, hi);
    } catch (...) {  // no more threads to be had: no exception crosses the C ABI, the range is done here
      work(lo, hi);
    }
  }
  work(0, per < count ? per : count);
  for (std::thread& t : pool) t.join();
  return NBODY_MI355X_OK;
}
