// nbody_ics.cpp -- synthetic initial conditions for the benchmark scenes (host code).
//
// The reference builds its scenes on the host, one SetParticle() at a time, from
// particle_initializer.h and libstdc++ <random> distributions (saturn_ring.cc:42-63),
// whose streams are not portable across standard libraries.  The large synthetic
// scenes of the measurement plan therefore come from this
// in-library generator: splitmix64-seeded xoshiro256**, 53-bit uniforms, plain libm.
// Output is the reference's record layout: Particle<double>[n] {r[3], v[3]}
// (particle_simulation.h:20-24) and double[n] masses, ready for SetParticle().
#include <cmath>
#include <cstdint>

#include "nbody_internal.h"

using nbody_internal::set_error;

namespace {

struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (int i = 0; i < 4; i++) {  // splitmix64
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      s[i] = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {  // xoshiro256**
    const uint64_t r = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  double u01() { return static_cast<double>(next() >> 11) * 0x1p-53; }  // [0,1)
  double normal(double sigma) {  // Box-Muller, one deviate per call (two uniforms)
    double u;
    do { u = u01(); } while (u == 0.0);
    return sigma * std::sqrt(-2.0 * std::log(u)) * std::cos(2.0 * M_PI * u01());
  }
};

void isotropic(Rng& g, double len, double* out) {
  const double z = 1.0 - 2.0 * g.u01();
  const double phi = 2.0 * M_PI * g.u01();
  const double s = std::sqrt(1.0 - z * z);
  out[0] = len * s * std::cos(phi);
  out[1] = len * s * std::sin(phi);
  out[2] = len * z;
}

}  // namespace

// Plummer sphere in units G*M = 1, scale radius 1, truncated at 10 radii; speeds by
// Aarseth-Henon-Wielen rejection from the isotropic distribution function; equal
// masses 1/(G*n); centre of mass and mean velocity removed.
extern "C" int nbody_mi355x_ic_plummer(int64_t n, uint64_t seed, double G, double* particles,
                                       double* mass) {
  if (n <= 0 || !particles || !mass || !(G > 0.0)) {
    set_error("ic_plummer: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  Rng g(seed);
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    double r;
    do {
      double u;
      do { u = g.u01(); } while (u == 0.0);
      r = 1.0 / std::sqrt(std::pow(u, -2.0 / 3.0) - 1.0);
    } while (!(r <= 10.0));
    double q, y;
    do {
      q = g.u01();
      y = 0.1 * g.u01();
    } while (y > q * q * std::pow(1.0 - q * q, 3.5));
    const double v = q * std::sqrt(2.0) * std::pow(1.0 + r * r, -0.25);
    double* p = particles + 6 * i;
    isotropic(g, r, p);
    isotropic(g, v, p + 3);
    for (int k = 0; k < 6; k++) c[k] += p[k];
    mass[i] = 1.0 / (G * static_cast<double>(n));
  }
  for (int k = 0; k < 6; k++) c[k] /= static_cast<double>(n);
  for (int64_t i = 0; i < n; i++)
    for (int k = 0; k < 6; k++) particles[6 * i + k] -= c[k];
  return NBODY_MI355X_OK;
}

// Cold uniform cube: x,y,z ~ U(-1,1), v = 0, equal masses 1/(G*n).
extern "C" int nbody_mi355x_ic_uniform_cube(int64_t n, uint64_t seed, double G, double* particles,
                                            double* mass) {
  if (n <= 0 || !particles || !mass || !(G > 0.0)) {
    set_error("ic_uniform_cube: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  Rng g(seed);
  for (int64_t i = 0; i < n; i++) {
    double* p = particles + 6 * i;
    for (int k = 0; k < 3; k++) p[k] = 2.0 * g.u01() - 1.0;
    p[3] = p[4] = p[5] = 0.0;
    mass[i] = 1.0 / (G * static_cast<double>(n));
  }
  return NBODY_MI355X_OK;
}

// A ring around a central body: the recipe of the reference's saturn_ring.cc:42-63 for any
// n, with this file's generator instead of libstdc++'s <random>.  Body 0 is the central
// mass at rest at the origin; bodies 1..n-1 are built as the reference's
// NonCircularOrbitMoon does (particle_initializer.h:120-175): at (radius, 0, dz) with
// velocity (dvx, speed + dvy, 0), speed = sqrt(G_orbit * M / radius) with the initializer's
// G_orbit = 6.67430e-11, then turned about Z by a uniform phase.  radius = ring_radius +
// N(0, width_sigma), dz = N(0, thickness_sigma), dvx, dvy = N(0, v_sigma), mass =
// N(0, mass_sigma)^2.
extern "C" int nbody_mi355x_ic_ring(int64_t n, uint64_t seed, double central_mass, double ring_radius,
                                    double width_sigma, double thickness_sigma, double v_sigma, double mass_sigma,
                                    double* particles, double* mass) {
  if (n <= 0 || !particles || !mass || !(central_mass > 0.0) || !(ring_radius > 0.0) || width_sigma < 0.0 ||
      thickness_sigma < 0.0 || v_sigma < 0.0 || mass_sigma < 0.0) {
    set_error("ic_ring: bad arguments");
    return NBODY_MI355X_ERR_INVALID;
  }
  Rng g(seed);
  const double G_orbit = 6.67430e-11;
  for (int k = 0; k < 6; k++) particles[k] = 0.0;
  mass[0] = central_mass;
  for (int64_t i = 1; i < n; i++) {
    const double dz = g.normal(thickness_sigma);
    const double dvx = g.normal(v_sigma), dvy = g.normal(v_sigma);
    const double phase = 2.0 * M_PI * g.u01();
    const double ms = g.normal(mass_sigma);
    double radius;
    do { radius = ring_radius + g.normal(width_sigma); } while (!(radius > 0.0));
    const double speed = std::sqrt(G_orbit * central_mass / radius);
    const double c = std::cos(phase), s = std::sin(phase);
    const double x = radius, y = 0.0, vx = dvx, vy = speed + dvy;
    double* p = particles + 6 * i;
    p[0] = x * c - y * s;
    p[1] = x * s + y * c;
    p[2] = dz;
    p[3] = vx * c - vy * s;
    p[4] = vx * s + vy * c;
    p[5] = 0.0;
    mass[i] = ms * ms;
  }
  return NBODY_MI355X_OK;
}
