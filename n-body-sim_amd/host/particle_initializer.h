// particle_initializer.h -- analytic initial conditions with the reference's call
// signatures (ref: particle_initializer.h:10-175), plus the large synthetic scenes of
// libnbody_mi355x.so behind the same InPutParticle type.
//
// A body on a (perturbed) circular orbit is built in the XY plane at phase 0 and then
// turned about X (azimuthal tilt), Y (polar tilt) and Z (phase), in that order, exactly
// as the reference does; the orbital speed uses G = 6.67430e-11 (ref: :67, :130 -- not
// the simulator's 6.7430e-11).
#pragma once

#include <cmath>
#include <vector>

#include "particle_simulation.h"

template <typename T>
struct InPutParticle {
  Particle<T> particle;
  T m;
};

// ref: particle_initializer.h:17-42 (names as spelled there).  Each is the plane rotation
//   p' = p cos(angle) - q sin(angle),  q' = p sin(angle) + q cos(angle)
// of two components; written out per axis, products in the reference's order, because
// with -mfma the compiler fuses one product of each sum and WHICH one follows the
// expression order -- the scene fixtures are compared bit for bit.
template <typename T>
void RotateArroundX_Achis(T* vector, T angle) {
  const T y = vector[1] * cos(angle) - vector[2] * sin(angle);
  const T z = vector[1] * sin(angle) + vector[2] * cos(angle);
  vector[1] = y;
  vector[2] = z;
}
template <typename T>
void RotateArroundY_Achis(T* vector, T angle) {
  const T x = vector[0] * cos(angle) + vector[2] * sin(angle);
  const T z = (-1) * vector[0] * sin(angle) + vector[2] * cos(angle);
  vector[0] = x;
  vector[2] = z;
}
template <typename T>
void RotateArroundZ_Achis(T* vector, T angle) {
  const T x = vector[0] * cos(angle) - vector[1] * sin(angle);
  const T y = vector[0] * sin(angle) + vector[1] * cos(angle);
  vector[0] = x;
  vector[1] = y;
}

// ref: particle_initializer.h:45-54 -- a body at rest at the origin (default: the earth)
template <typename T>
InPutParticle<T> BodyAtCenter(T mass = 5.974e+24) {
  InPutParticle<T> body{};
  body.m = mass;
  return body;
}

// ref: particle_initializer.h:120-175
template <typename T>
InPutParticle<T> NonCircularOrbitMoon(T* x_deviation, T* v_deviation, T orbital_phase_z = 0.0,
                                      T orbital_plane_angle_polar_y = 0.0,
                                      T orbital_plane_angle_azimutal_x = 0.0, bool clockwise = true,
                                      T central_body_mass = 5.974e+24, T moon_mass = 7.342e+22,
                                      T orbital_radius = 3.844e+8) {
  const T G = 6.67430e-11;
  const T speed = sqrt((G * central_body_mass) / orbital_radius);
  InPutParticle<T> body{};
  T* x = body.particle.r;
  T* v = body.particle.v;
  x[0] = orbital_radius + x_deviation[0];
  x[1] = 0 + x_deviation[1];
  x[2] = 0 + x_deviation[2];
  v[0] = 0 + v_deviation[0];
  v[1] = (clockwise ? speed : -speed) + v_deviation[1];
  v[2] = 0 + v_deviation[2];
  if (orbital_plane_angle_azimutal_x != 0.0) {
    RotateArroundX_Achis(x, orbital_plane_angle_azimutal_x);
    RotateArroundX_Achis(v, orbital_plane_angle_azimutal_x);
  }
  if (orbital_plane_angle_polar_y != 0.0) {
    RotateArroundY_Achis(x, orbital_plane_angle_polar_y);
    RotateArroundY_Achis(v, orbital_plane_angle_polar_y);
  }
  if (orbital_phase_z != 0.0) {
    RotateArroundZ_Achis(x, orbital_phase_z);
    RotateArroundZ_Achis(v, orbital_phase_z);
  }
  body.m = moon_mass;
  return body;
}

// ref: particle_initializer.h:59-112 -- the unperturbed case
template <typename T>
InPutParticle<T> CircularOrbitMoon(T orbital_phase_z = 0.0, T orbital_plane_angle_polar_y = 0.0,
                                   T orbital_plane_angle_azimutal_x = 0.0, bool clockwise = true,
                                   T central_body_mass = 5.974e+24, T moon_mass = 7.342e+22,
                                   T orbital_radius = 3.844e+8) {
  T none[3] = {0, 0, 0};
  return NonCircularOrbitMoon<T>(none, none, orbital_phase_z, orbital_plane_angle_polar_y,
                                 orbital_plane_angle_azimutal_x, clockwise, central_body_mass, moon_mass,
                                 orbital_radius);
}

// ---- synthetic scenes (not in the reference): Plummer sphere / cold uniform cube from the
// library's portable generator, in units G_sim * M_total = 1.
inline std::vector<InPutParticle<double>> SyntheticScene(bool plummer, int64_t n, uint64_t seed,
                                                         double G = NBODY_MI355X_G_REFERENCE) {
  std::vector<double> p(6 * static_cast<size_t>(n)), m(static_cast<size_t>(n));
  const int rc = plummer ? nbody_mi355x_ic_plummer(n, seed, G, p.data(), m.data())
                         : nbody_mi355x_ic_uniform_cube(n, seed, G, p.data(), m.data());
  std::vector<InPutParticle<double>> out;
  if (rc != NBODY_MI355X_OK) return out;
  out.resize(static_cast<size_t>(n));
  for (size_t i = 0; i < out.size(); i++) {
    for (int c = 0; c < 3; c++) {
      out[i].particle.r[c] = p[6 * i + c];
      out[i].particle.v[c] = p[6 * i + 3 + c];
    }
    out[i].m = m[i];
  }
  return out;
}
inline std::vector<InPutParticle<double>> PlummerSphere(int64_t n, uint64_t seed = 42) {
  return SyntheticScene(true, n, seed);
}
inline std::vector<InPutParticle<double>> UniformCube(int64_t n, uint64_t seed = 7) {
  return SyntheticScene(false, n, seed);
}

// saturn_ring.cc's scene (ref: saturn_ring.cc:42-63) for any n, from the library's
// portable generator: body 0 = the central mass, the rest on perturbed circular orbits.
inline std::vector<InPutParticle<double>> Ring(int64_t n, uint64_t seed = 42, double central_mass = 5.683e+26,
                                               double ring_radius = 1.12e+8, double width_sigma = 3.0e+7,
                                               double thickness_sigma = 50, double v_sigma = 50,
                                               double mass_sigma = 0.1) {
  std::vector<double> p(6 * static_cast<size_t>(n)), m(static_cast<size_t>(n));
  std::vector<InPutParticle<double>> out;
  if (nbody_mi355x_ic_ring(n, seed, central_mass, ring_radius, width_sigma, thickness_sigma, v_sigma, mass_sigma,
                           p.data(), m.data()) != NBODY_MI355X_OK)
    return out;
  out.resize(static_cast<size_t>(n));
  for (size_t i = 0; i < out.size(); i++) {
    for (int c = 0; c < 3; c++) {
      out[i].particle.r[c] = p[6 * i + c];
      out[i].particle.v[c] = p[6 * i + 3 + c];
    }
    out[i].m = m[i];
  }
  return out;
}

// ---- the same three scenes AT SCALE: one random stream per body, made on the
// host's threads by nbody_mi355x_scene_host (the code nbody_mi355x_generate runs on the GPU).
// Statistically the scenes above, not bit for bit.  The Plummer sphere is moved into its
// centre-of-mass frame here, as PlummerSphere() is.
inline std::vector<InPutParticle<double>> SceneAtScale(const nbody_mi355x_scene& scene) {
  const size_t n = static_cast<size_t>(scene.num_particles > 0 ? scene.num_particles : 0);
  std::vector<double> p(6 * n), m(n);
  std::vector<InPutParticle<double>> out;
  if (nbody_mi355x_scene_host(&scene, 0, scene.num_particles, p.data(), m.data()) != NBODY_MI355X_OK) return out;
  double c[6] = {0, 0, 0, 0, 0, 0};
  if (scene.scene == NBODY_MI355X_SCENE_PLUMMER) {
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 6; k++) c[k] += p[6 * i + k];
    for (double& v : c) v /= static_cast<double>(n);
  }
  out.resize(n);
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      out[i].particle.r[k] = p[6 * i + k] - c[k];
      out[i].particle.v[k] = p[6 * i + 3 + k] - c[3 + k];
    }
    out[i].m = m[i];
  }
  return out;
}
inline nbody_mi355x_scene SceneRecord(int32_t kind, int64_t n, uint64_t seed) {
  nbody_mi355x_scene s{};
  s.scene = kind;
  s.seed = seed;
  s.num_particles = n;
  s.G = NBODY_MI355X_G_REFERENCE;
  s.central_mass = 5.683e+26;  // saturn_ring.cc:42-63
  s.ring_radius = 1.12e+8;
  s.width_sigma = 3.0e+7;
  s.thickness_sigma = 50;
  s.v_sigma = 50;
  s.mass_sigma = 0.1;
  return s;
}
inline std::vector<InPutParticle<double>> PlummerSphereAtScale(int64_t n, uint64_t seed = 42) {
  return SceneAtScale(SceneRecord(NBODY_MI355X_SCENE_PLUMMER, n, seed));
}
inline std::vector<InPutParticle<double>> UniformCubeAtScale(int64_t n, uint64_t seed = 7) {
  return SceneAtScale(SceneRecord(NBODY_MI355X_SCENE_UNIFORM_CUBE, n, seed));
}
inline std::vector<InPutParticle<double>> RingAtScale(int64_t n, uint64_t seed = 42) {
  return SceneAtScale(SceneRecord(NBODY_MI355X_SCENE_RING, n, seed));
}
