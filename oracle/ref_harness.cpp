// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY; builds only where /root/reference exists.
//
// A thin extern "C" shim around the REAL reference header
// (/root/reference/src/particle_simulation.h, included from where it lies, never
// copied).  oracle/Makefile compiles it into oracle/_ref/libnbody_ref.so, which is
// used (a) by tests/golden/make_golden.py to produce the golden vectors that pin
// oracle/nbody_oracle.c, and (b) by bench.py's cpu_baseline leg (kind
// "reference": the reference's own SimulateAVX2()).  The built .so travels to
// the GPU box; the reference sources do not.
//
// The reference keeps its results private with no getters
// (particle_simulation.h:711-749), hence the `#define private public` AFTER all
// standard headers (defining it earlier breaks libstdc++'s <sstream>).
#include <immintrin.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <optional>
#include <random>
#include <sstream>
#include <thread>
#include <vector>

#define private public
#include "particle_simulation.h"
#undef private
#include "particle_initializer.h"

namespace {
// every Simulate*() prints 3 lines per big step (particle_simulation.h:612-615)
struct MuteCout {
  std::streambuf* old;
  MuteCout() : old(std::cout.rdbuf(nullptr)) {}
  ~MuteCout() { std::cout.rdbuf(old); std::cout.clear(); }
};

template <typename T, int W>
int run(int mode, int n, long big, long sub, T dt, int threads, const T* particles,
        const T* mass, T* out11, double* seconds) {
  ParticleSimulation<T, W> sim(n, big, sub, dt, threads);
  for (int i = 0; i < n; i++) {
    Particle<T> p;
    for (int c = 0; c < 3; c++) { p.r[c] = particles[6 * i + c]; p.v[c] = particles[6 * i + 3 + c]; }
    sim.SetParticle(p, mass[i]);
  }
  sim.SetMissingParticlesInBatch();
  auto t0 = std::chrono::steady_clock::now();
  {
    MuteCout mute;
    if (mode == 0) sim.SimulateCPU();
    else if constexpr (std::is_same<T, double>::value && W == 4) sim.SimulateAVX2();
    else return -3;
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (out11) {
    for (long k = 0; k < big + 1; k++)
      for (int i = 0; i < n; i++) {
        const OutPutParticle<T>& o = sim.out_put_data_[k][i];
        T* d = out11 + (k * n + i) * 11;
        for (int c = 0; c < 3; c++) { d[c] = o.r[c]; d[3 + c] = o.v[c]; d[6 + c] = o.a[c]; }
        d[9] = o.m; d[10] = o.t;
      }
  }
  return 0;
}
}  // namespace

extern "C" {

// mode 0 = SimulateCPU(), 1 = SimulateAVX2() (needs n % 4 == 0: the reference
// writes out of bounds otherwise, particle_simulation.h:514-540).
// out11: OutPutParticle<double>[big+1][n] = {r,v,a,m,t}, nullable.
int nbody_ref_simulate_f64(int mode, int n, long big, long sub, double dt, int threads,
                           const double* particles, const double* mass, double* out11,
                           double* seconds) {
  if (mode == 1 && (n % 4) != 0) return -2;
  return run<double, 4>(mode, n, big, sub, dt, threads, particles, mass, out11, seconds);
}

// ParticleSimulation<float,8>::SimulateCPU()
int nbody_ref_simulate_f32(int n, long big, long sub, float dt, int threads,
                           const float* particles, const float* mass, float* out11,
                           double* seconds) {
  return run<float, 8>(0, n, big, sub, dt, threads, particles, mass, out11, seconds);
}

// Scene builders: the particle set-up of the reference's example mains, through
// the reference's own particle_initializer.h.  They return N.
static void put(const InPutParticle<double>& in, int i, double* particles, double* mass) {
  for (int c = 0; c < 3; c++) { particles[6 * i + c] = in.particle.r[c]; particles[6 * i + 3 + c] = in.particle.v[c]; }
  mass[i] = in.m;
}

// earth_moon.cc:34-38
int nbody_ref_scene_earth_moon(double* particles, double* mass) {
  put(BodyAtCenter<double>(), 0, particles, mass);
  put(CircularOrbitMoon<double>(), 1, particles, mass);
  return 2;
}

// earth_7moons.cc:34-50
int nbody_ref_scene_earth_7moons(double* particles, double* mass) {
  int i = 0;
  put(BodyAtCenter<double>(), i++, particles, mass);
  put(CircularOrbitMoon<double>(), i++, particles, mass);
  put(CircularOrbitMoon<double>(M_PI / 2, M_PI / 2), i++, particles, mass);
  put(CircularOrbitMoon<double>(M_PI / 2, M_PI / 4, M_PI / 4), i++, particles, mass);
  put(CircularOrbitMoon<double>(2 * M_PI / 3, M_PI / 4, M_PI / 4, false), i++, particles, mass);
  put(CircularOrbitMoon<double>(7 * M_PI / 9, 0.0, M_PI / 8, false), i++, particles, mass);
  put(CircularOrbitMoon<double>(M_PI / 5, 0.0, M_PI / 3), i++, particles, mass);
  put(CircularOrbitMoon<double>(M_PI / 7, 0.0, M_PI / 3), i++, particles, mass);
  return i;
}

// saturn_ring.cc:24-63 (libstdc++ default_random_engine(42) streams: the result
// is libstdc++-specific, which is why it is committed as a fixture).
int nbody_ref_scene_saturn_ring(int num_particles, double* particles, double* mass) {
  const double v_deviation_sigma = 50, disc_thickness_sigma = 50;
  const double ring_radius = 1.12e+8, ring_width_sigma = 3.0e+7, mass_sigma = 0.1;
  std::default_random_engine engine(42);
  std::normal_distribution<double> disc_height_distro(0, disc_thickness_sigma);
  std::normal_distribution<double> disc_radius_distro(0, ring_width_sigma);
  std::uniform_real_distribution<double> phase_distro(0, 2 * M_PI);
  std::normal_distribution<double> v_dev_distro(0, v_deviation_sigma);
  std::normal_distribution<double> mass_distro(0, mass_sigma);
  put(BodyAtCenter<double>(5.683e+26), 0, particles, mass);
  for (int i = 1; i < num_particles; i++) {
    double x_deviation[3] = {0.0, 0.0, disc_height_distro(engine)};
    double y_deviation[3] = {v_dev_distro(engine), v_dev_distro(engine), 0.0};
    InPutParticle<double> in = NonCircularOrbitMoon<double>(
        x_deviation, y_deviation, phase_distro(engine), 0.0, 0.0, true, 5.683e+26,
        pow(mass_distro(engine), 2), ring_radius + disc_radius_distro(engine));
    put(in, i, particles, mass);
  }
  return num_particles;
}

}  // extern "C"
