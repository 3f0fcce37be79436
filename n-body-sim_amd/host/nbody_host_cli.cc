// nbody_host_cli.cc -- a small driver over the host API (particle_simulation.h +
// particle_initializer.h): the reference's example scenes and raw-file runs, with the
// simulation path chosen at run time.  The reference hard-codes one scene and one
// Simulate*() call per main() (earth_moon.cc, saturn_ring.cc, ...); here the set-up
// code is the same API calls, only the `Simulate` line is a switch.
//
//   nbody_host_cli scene  NAME OUT.bin
//       write the initial conditions of a reference scene (earth_moon | earth_7moons |
//       slingshoot | saturn_ring) as  int64 n, double[n][6] {r,v}, double[n] m
//   nbody_host_cli run    PATH IN.bin BIG SUB DT THREADS OUT.bin
//       PATH = cpu | avx2 | avx512 | mi355x | mi355x-fp32 | mi355x-fast | mi355x-shardsK | cpu-f32 | avx512-f32 | mi355x-f32host
//       read IN.bin (format above), simulate BIG big steps of SUB small steps, write
//       OutPutParticle rows double[(BIG+1)][n][11] {r,v,a,m,t} (float runs are widened); one JSON
//       line on stdout with the host's wall clock of every big step
//   nbody_host_cli demo   NAME PATH [BIG] [SUB_DIVISOR] [OUTDIR]
//       run a reference scene with its own step parameters (SUB divided by SUB_DIVISOR to
//       shorten it) and write the text files the reference writes
//   nbody_host_cli bench  IC N PATH [BIG] [SUB] [THREADS]
//       IC = plummer | cube | ring (the library's portable generators, seed 42 / 7 / 42), or
//       plummer-scale | cube-scale | ring-scale (one stream per body, made on all host threads);
//       BIG big steps (default 3) of SUB small steps (default 5) through ParticleSimulation
//       and the chosen Simulate path; one JSON line: Ginteractions/s from the host's own
//       wall clock around every big step after the first (which holds the set-up), i.e.
//       what a user of SimulateMI355X() sees, snapshot copy and unpack included
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "particle_initializer.h"
#include "particle_simulation.h"

namespace {

struct Scene {
  std::vector<InPutParticle<double>> bodies;
  int big_steps = 400;
  double t_delta = 0, steps_per_second = 0;  // the reference's step-size recipe
  bool per_timestep_files = false;
};

// step parameters as the example mains derive them (ref: earth_moon.cc:15-21)
void StepParameters(const Scene& s, int64_t* substeps, double* stepsize) {
  const int64_t total = static_cast<int64_t>(std::ceil((s.steps_per_second * s.t_delta) / s.big_steps)) * s.big_steps;
  *substeps = total / s.big_steps;
  *stepsize = s.t_delta / static_cast<double>(total);
}

bool MakeScene(const std::string& name, Scene* s) {
  const double day = 60 * 60 * 24;
  if (name == "earth_moon") {  // ref: earth_moon.cc:13-38
    s->t_delta = 5 * 30 * day; s->steps_per_second = 10;
    s->bodies = {BodyAtCenter<double>(), CircularOrbitMoon<double>()};
  } else if (name == "slingshoot") {  // ref: slingshoot.cc:15-40
    s->t_delta = 1.4 * 30 * day; s->steps_per_second = 20;
    InPutParticle<double> probe = CircularOrbitMoon<double>(0.0, 0.0, 0.0, true, 5.974e+24, 7.342e+22, 1.0e+8);
    probe.particle.r[1] = -1.2e+9;
    s->bodies = {BodyAtCenter<double>(), probe};
  } else if (name == "earth_7moons") {  // ref: earth_7moons.cc:15-50
    s->big_steps = 500; s->t_delta = 5 * 30 * day; s->steps_per_second = 20;
    s->bodies = {BodyAtCenter<double>(),
                 CircularOrbitMoon<double>(),
                 CircularOrbitMoon<double>(M_PI / 2, M_PI / 2),
                 CircularOrbitMoon<double>(M_PI / 2, M_PI / 4, M_PI / 4),
                 CircularOrbitMoon<double>(2 * M_PI / 3, M_PI / 4, M_PI / 4, false),
                 CircularOrbitMoon<double>(7 * M_PI / 9, 0.0, M_PI / 8, false),
                 CircularOrbitMoon<double>(M_PI / 5, 0.0, M_PI / 3),
                 CircularOrbitMoon<double>(M_PI / 7, 0.0, M_PI / 3)};
  } else if (name == "saturn_ring") {  // ref: saturn_ring.cc:15-63 (N = 500, engine seed 42)
    s->t_delta = day; s->steps_per_second = 100; s->per_timestep_files = true;
    const double saturn = 5.683e+26, ring_radius = 1.12e+8;
    std::default_random_engine engine(42);
    std::normal_distribution<double> height(0, 50), radius(0, 3.0e+7), speed(0, 50), sqrt_mass(0, 0.1);
    std::uniform_real_distribution<double> phase(0, 2 * M_PI);
    s->bodies.push_back(BodyAtCenter<double>(saturn));
    for (int i = 1; i < 500; i++) {
      // draw order matters for the stream: height, then the two speed deviations; the
      // remaining three draws happen in g++'s right-to-left argument evaluation of the
      // reference's call (ring radius, mass, phase) -- made explicit here
      double dx[3] = {0.0, 0.0, height(engine)};
      double dv[3] = {speed(engine), speed(engine), 0.0};
      const double r = ring_radius + radius(engine);
      const double m = std::pow(sqrt_mass(engine), 2);
      const double ph = phase(engine);
      s->bodies.push_back(NonCircularOrbitMoon<double>(dx, dv, ph, 0.0, 0.0, true, saturn, m, r));
    }
  } else {
    return false;
  }
  return true;
}

bool ReadIcs(const char* path, std::vector<double>* p, std::vector<double>* m) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  int64_t n = 0;
  bool ok = fread(&n, sizeof n, 1, f) == 1 && n > 0 && n < (int64_t{1} << 31);
  if (ok) {
    p->resize(6 * static_cast<size_t>(n));
    m->resize(static_cast<size_t>(n));
    ok = fread(p->data(), sizeof(double), p->size(), f) == p->size() &&
         fread(m->data(), sizeof(double), m->size(), f) == m->size();
  }
  fclose(f);
  return ok;
}

bool WriteIcs(const char* path, const std::vector<InPutParticle<double>>& b) {
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  const int64_t n = static_cast<int64_t>(b.size());
  fwrite(&n, sizeof n, 1, f);
  for (const auto& q : b) { fwrite(q.particle.r, sizeof(double), 3, f); fwrite(q.particle.v, sizeof(double), 3, f); }
  for (const auto& q : b) fwrite(&q.m, sizeof(double), 1, f);
  return fclose(f) == 0;
}

template <typename T, int W>
std::string Simulate(ParticleSimulation<T, W>& sim, const std::string& path) {
  if (path == "cpu" || path == "cpu-f32") { sim.SimulateCPU(); return "ok"; }
  if constexpr (std::is_same<T, double>::value && W == 4) {
    if (path == "avx2") { sim.SimulateAVX2(); return "ok"; }
  }
  if (path == "avx512" || path == "avx512-f32") return sim.SimulateAVX512() ? "ok" : "no AVX-512F on this CPU";
  if (path == "mi355x" || path == "mi355x-f32host") return sim.SimulateMI355X();
  if (path == "mi355x-fp32") return sim.SimulateMI355X(0, true);
  if (path == "mi355x-fast") { sim.SetMI355XPrecision(NBODY_MI355X_FP64_FAST); return sim.SimulateMI355X(); }
  if (path == "mi355x-kdk") return sim.SimulateMI355X(0, false, 0, NBODY_MI355X_KDK);
  if (path.rfind("mi355x-shards", 0) == 0) {  // mi355x-shardsK: K shards, round-robin over the visible GPUs
    const int k = std::atoi(path.c_str() + 13);
    int ndev = 0;
    if (k < 1 || nbody_mi355x_device_count(&ndev) != NBODY_MI355X_OK || ndev < 1) return "bad shard count or no device";
    std::vector<int> devs;
    for (int i = 0; i < k; i++) devs.push_back(i % ndev);
    return sim.SimulateMI355X(devs);
  }
  return "unknown path " + path;
}

template <typename T, int W>
int RunRaw(const std::string& path, const std::vector<double>& p, const std::vector<double>& m, int64_t big,
           int64_t sub, double dt, int threads, const char* out_path) {
  const int n = static_cast<int>(m.size());
  ParticleSimulation<T, W> sim(n, big, sub, static_cast<T>(dt), threads);
  sim.SetQuiet(true);
  for (int i = 0; i < n; i++) {
    Particle<T> q;
    for (int c = 0; c < 3; c++) { q.r[c] = static_cast<T>(p[6 * i + c]); q.v[c] = static_cast<T>(p[6 * i + 3 + c]); }
    sim.SetParticle(q, static_cast<T>(m[i]));
  }
  sim.SetMissingParticlesInBatch();
  const std::string status = Simulate(sim, path);
  fprintf(stderr, "%s\n", status.c_str());
  if (status != "ok" && status.find("succesfull") == std::string::npos) return 3;
  // one JSON line: the host's own wall clock around every big step (the reference's
  // execution_time_, particle_simulation.h:203-207); the first one holds the set-up
  printf("{\"path\": \"%s\", \"bodies\": %d, \"substeps\": %lld, \"threads\": %d, \"big_step_seconds\": [", path.c_str(), n,
         (long long)sub, threads);
  for (int64_t k = 0; k < big; k++) printf("%s%.9f", k ? ", " : "", sim.BigStepSeconds()[k]);
  printf("]}\n");
  FILE* f = fopen(out_path, "wb");
  if (!f) return 4;
  for (const auto& snap : sim.OutputData())
    for (const auto& o : snap) {
      const double row[11] = {o.r[0], o.r[1], o.r[2], o.v[0], o.v[1], o.v[2], o.a[0], o.a[1], o.a[2], o.m, o.t};
      fwrite(row, sizeof(double), 11, f);
    }
  return fclose(f) == 0 ? 0 : 4;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string cmd = argc > 1 ? argv[1] : "";
  if (cmd == "scene" && argc == 4) {
    Scene s;
    if (!MakeScene(argv[2], &s)) { fprintf(stderr, "unknown scene %s\n", argv[2]); return 2; }
    return WriteIcs(argv[3], s.bodies) ? 0 : 4;
  }
  if (cmd == "run" && argc == 9) {
    std::vector<double> p, m;
    if (!ReadIcs(argv[3], &p, &m)) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
    const std::string path = argv[2];
    const int64_t big = atoll(argv[4]), sub = atoll(argv[5]);
    const double dt = atof(argv[6]);
    const int threads = atoi(argv[7]);
    if (path == "cpu-f32" || path == "avx512-f32" || path == "mi355x-f32host") return RunRaw<float, 8>(path, p, m, big, sub, dt, threads, argv[8]);
    return RunRaw<double, 4>(path, p, m, big, sub, dt, threads, argv[8]);
  }
  if (cmd == "demo" && argc >= 4) {
    Scene s;
    if (!MakeScene(argv[2], &s)) { fprintf(stderr, "unknown scene %s\n", argv[2]); return 2; }
    int64_t sub; double dt;
    StepParameters(s, &sub, &dt);
    const int64_t big = argc > 4 ? atoll(argv[4]) : s.big_steps;
    if (argc > 5) sub = std::max<int64_t>(1, sub / atoll(argv[5]));
    const std::string dir = argc > 6 ? argv[6] : "";
    const std::string path = argv[3];
    const int threads = path == "cpu" && s.bodies.size() < 64 ? 1 : static_cast<int>(std::thread::hardware_concurrency());
    ParticleSimulation<double, 4> my_sim(static_cast<int>(s.bodies.size()), big, sub, dt, threads);
    for (const auto& b : s.bodies) my_sim.SetParticle(b.particle, b.m);
    my_sim.SetMissingParticlesInBatch();
    const std::string status = Simulate(my_sim, path);
    std::cout << status << std::endl;
    if (s.per_timestep_files) my_sim.WriteTimestepFiles(dir); else my_sim.WriteParticleFiles(dir);
    my_sim.PrintAverageStepTime();
    return 0;
  }
  if (cmd == "bench" && argc >= 5) {
    const std::string ic = argv[2], path = argv[4];
    const int64_t n = atoll(argv[3]);
    const int64_t big = argc > 5 ? atoll(argv[5]) : 3, sub = argc > 6 ? atoll(argv[6]) : 5;
    const int threads = argc > 7 ? atoi(argv[7]) : static_cast<int>(std::thread::hardware_concurrency());
    if (n < 1 || n >= (int64_t{1} << 31) || big < 2 || sub < 1) { fprintf(stderr, "bench: N >= 1, BIG >= 2, SUB >= 1\n"); return 2; }
    std::vector<double> p(6 * static_cast<size_t>(n)), m(static_cast<size_t>(n));
    const double G = 6.7430e-11;  // the simulator's constant (ref: particle_simulation.h:713)
    double dt = 1e-3;
    int rc;
    if (ic == "plummer") rc = nbody_mi355x_ic_plummer(n, 42, G, p.data(), m.data());
    else if (ic == "cube") { rc = nbody_mi355x_ic_uniform_cube(n, 7, G, p.data(), m.data()); dt = 1e-4; }
    else if (ic == "ring") { rc = nbody_mi355x_ic_ring(n, 42, 5.683e+26, 1.12e+8, 3.0e+7, 50, 50, 0.1, p.data(), m.data()); dt = 0.01; }
    else if (ic == "plummer-scale" || ic == "cube-scale" || ic == "ring-scale") {
      const std::vector<InPutParticle<double>> bodies = ic[0] == 'p' ? PlummerSphereAtScale(n) : ic[0] == 'c' ? UniformCubeAtScale(n) : RingAtScale(n);
      rc = bodies.size() == static_cast<size_t>(n) ? NBODY_MI355X_OK : NBODY_MI355X_ERR_INVALID;
      for (size_t i = 0; i < bodies.size(); i++) {
        for (int c = 0; c < 3; c++) { p[6 * i + c] = bodies[i].particle.r[c]; p[6 * i + 3 + c] = bodies[i].particle.v[c]; }
        m[i] = bodies[i].m;
      }
      dt = ic[0] == 'p' ? 1e-3 : ic[0] == 'c' ? 1e-4 : 0.01;
    }
    else { fprintf(stderr, "unknown IC %s\n", ic.c_str()); return 2; }
    if (rc != NBODY_MI355X_OK) { fprintf(stderr, "%s\n", nbody_mi355x_last_error()); return 3; }
    ParticleSimulation<double, 4> sim(static_cast<int>(n), big, sub, dt, threads);
    sim.SetQuiet(true);
    for (int64_t i = 0; i < n; i++) {
      Particle<double> q;
      for (int c = 0; c < 3; c++) { q.r[c] = p[6 * i + c]; q.v[c] = p[6 * i + 3 + c]; }
      sim.SetParticle(q, m[i]);
    }
    sim.SetMissingParticlesInBatch();
    const std::string status = Simulate(sim, path);
    fprintf(stderr, "%s\n", status.c_str());
    if (status != "ok" && status.find("succesfull") == std::string::npos) return 3;
    double sec = 0;
    for (int64_t k = 1; k < big; k++) sec += sim.BigStepSeconds()[k];
    const double inter = static_cast<double>(n) * static_cast<double>(n) * static_cast<double>(sub) * static_cast<double>(big - 1);
    printf("{\"ic\": \"%s\", \"bodies\": %lld, \"path\": \"%s\", \"big_steps_timed\": %lld, \"substeps\": %lld, "
           "\"threads\": %d, \"seconds\": %.6f, \"Ginteractions_per_s\": %.2f, \"clock\": \"host wall clock per big step\"}\n",
           ic.c_str(), (long long)n, path.c_str(), (long long)(big - 1), (long long)sub, threads, sec, inter / sec / 1e9);
    return 0;
  }
  fprintf(stderr, "usage: see the comment at the top of nbody_host_cli.cc\n");
  return 2;
}
