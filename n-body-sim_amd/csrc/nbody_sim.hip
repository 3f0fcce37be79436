// nbody_sim.hip -- the handle half of the C ABI (include/nbody_mi355x.h, "level 2"):
// what ParticleSimulation::SimulateMI355X() binds.  It mirrors the stage structure
// of the reference's SimulateGPU() (particle_simulation_GPU.cu:277-303) --
// allocate, copy in, big steps, copy out, free -- but owns one HIP stream, keeps
// the whole substep loop of a big step on the device without host
// synchronisation (the reference syncs after every substep, .cu:155), and
// returns each big step's snapshot as soon as it exists instead of holding
// 72*N*num_big_steps bytes on the device (.cu:29-30).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "nbody_internal.h"

using nbody_internal::set_error;

struct nbody_mi355x_sim {
  nbody_mi355x_config cfg{};
  int64_t n = 0;      // real bodies
  int64_t slots = 0;  // n rounded up to the tile
  double* pos[2] = {nullptr, nullptr};  // [3][slots] each
  double* vel = nullptr;                // [3][slots]
  double* acc = nullptr;                // [3][slots]  KDK: the raw sums at the current positions
  bool acc_valid = false;               //             ... once a step (or the priming call) has stored them
  double* mass = nullptr;               // [slots]
  double* aos = nullptr;                // staging: Particle<double>[n] / DeviceOutputParticle<double>[n]
  double* scratch = nullptr;            // energy partials + result
  void* workspace = nullptr;            // pair-symmetric kernels' partial sums (large N only)
  int64_t workspace_bytes = 0;
  int cur = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_run_seconds = 0.0;
  // launch-bound sizes: kGraphSteps small steps captured once per buffer parity and replayed
  hipGraphExec_t graph[2] = {nullptr, nullptr};
  bool graph_failed = false;
  std::vector<double> host_tmp;  // float <-> double conversion of host records
  // a big step enqueued by run_async and not yet completed by wait
  bool pending = false;
  void* pending_out = nullptr;
  nbody_internal::MultiSim* multi = nullptr;  // several shards / GPUs (nbody_multi.hip); the fields above are unused
};

namespace {

#define HIP_TRY(expr, code, what)                                            \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      (void)hipGetLastError();                                               \
      set_error("%s: %s", what, hipGetErrorString(e_));                      \
      return code;                                                           \
    }                                                                        \
  } while (0)


// Small systems are launch-bound (a small step of N = 500 is a few microseconds of GPU
// work): the substep loop of a big step is replayed from a hipGraph of kGraphSteps steps
// (even, so that the current/next buffers end where they started) instead of being
// launched one kernel at a time from the host.  The reference launches AND synchronises
// per substep (particle_simulation_GPU.cu:149-155).
constexpr int kGraphSteps = 64;
constexpr int64_t kGraphMaxBodies = 16384;

int free_all(nbody_mi355x_sim* s) {
  int rc = NBODY_MI355X_OK;
  auto fr = [&](double*& p) {
    if (p && hipFree(p) != hipSuccess) rc = NBODY_MI355X_ERR_FREE;
    p = nullptr;
  };
  fr(s->pos[0]); fr(s->pos[1]); fr(s->vel); fr(s->acc); fr(s->mass); fr(s->aos); fr(s->scratch);
  if (s->workspace && hipFree(s->workspace) != hipSuccess) rc = NBODY_MI355X_ERR_FREE;
  s->workspace = nullptr;
  for (auto& g : s->graph) {
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  s->ev0 = s->ev1 = nullptr;
  s->stream = nullptr;
  return rc;
}

}  // namespace

static int check_config(const nbody_mi355x_config* cfg) {
  if (cfg->num_particles <= 0) { set_error("create: num_particles must be > 0"); return NBODY_MI355X_ERR_INVALID; }
  if (!(cfg->eps2 >= 0.0)) { set_error("create: eps2 must be >= 0"); return NBODY_MI355X_ERR_INVALID; }
  if (cfg->precision != NBODY_MI355X_FP64 && cfg->precision != NBODY_MI355X_FP32 && cfg->precision != NBODY_MI355X_FP64_FAST) {
    set_error("create: unknown precision %d", cfg->precision);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (cfg->integrator != NBODY_MI355X_EULER && cfg->integrator != NBODY_MI355X_KICK_DRIFT &&
      cfg->integrator != NBODY_MI355X_KDK) {
    set_error("create: unknown integrator %d", cfg->integrator);
    return NBODY_MI355X_ERR_INVALID;
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_create_multi(const nbody_mi355x_config* cfg, const int32_t* devices, int32_t num_devices,
                                         nbody_mi355x_sim** out) {
  if (!cfg || !out || !devices) { set_error("create_multi: null argument"); return NBODY_MI355X_ERR_INVALID; }
  *out = nullptr;
  int rc = check_config(cfg);
  if (rc != NBODY_MI355X_OK) return rc;
  const char* force = getenv("NBODY_MI355X_MULTI_FORCE");  // tests: one shard through the multi-shard code
  if (num_devices == 1 && !(force && atoi(force) != 0)) {  // one shard is the plain handle
    nbody_mi355x_config one = *cfg;
    one.device = devices[0];
    return nbody_mi355x_create(&one, out);
  }
  nbody_mi355x_sim* s = new (std::nothrow) nbody_mi355x_sim();
  if (!s) { set_error("create_multi: out of host memory"); return NBODY_MI355X_ERR_ALLOC; }
  s->cfg = *cfg;
  if (s->cfg.G == 0.0) s->cfg.G = NBODY_MI355X_G_REFERENCE;
  s->n = cfg->num_particles;
  rc = nbody_internal::multi_create(s->cfg, devices, num_devices, &s->multi);
  if (rc != NBODY_MI355X_OK) { delete s; return rc; }
  *out = s;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_create(const nbody_mi355x_config* cfg, nbody_mi355x_sim** out) {
  if (!cfg || !out) { set_error("create: null argument"); return NBODY_MI355X_ERR_INVALID; }
  *out = nullptr;
  const int vrc = check_config(cfg);
  if (vrc != NBODY_MI355X_OK) return vrc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_error("create: no HIP device visible");
    return NBODY_MI355X_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    set_error("create: device %d outside [0,%d)", cfg->device, ndev);
    return NBODY_MI355X_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(cfg->device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  nbody_mi355x_sim* s = new (std::nothrow) nbody_mi355x_sim();
  if (!s) { set_error("create: out of host memory"); return NBODY_MI355X_ERR_ALLOC; }
  s->cfg = *cfg;
  if (s->cfg.G == 0.0) s->cfg.G = NBODY_MI355X_G_REFERENCE;
  s->n = cfg->num_particles;
  s->slots = nbody_internal::slab_slots(s->n, cfg->variant);
  const size_t slab = sizeof(double) * 3 * static_cast<size_t>(s->slots);
  const int64_t eblocks = s->slots / NBODY_MI355X_TILE;
  bool ok = hipMalloc(&s->pos[0], slab) == hipSuccess && hipMalloc(&s->pos[1], slab) == hipSuccess &&
            hipMalloc(&s->vel, slab) == hipSuccess &&
            (cfg->integrator != NBODY_MI355X_KDK || hipMalloc(&s->acc, slab) == hipSuccess) &&
            hipMalloc(&s->mass, sizeof(double) * static_cast<size_t>(s->slots)) == hipSuccess &&
            hipMalloc(&s->aos, sizeof(double) * 9 * static_cast<size_t>(s->n)) == hipSuccess &&
            hipMalloc(&s->scratch, sizeof(double) * static_cast<size_t>(2 * eblocks + 2)) == hipSuccess &&
            hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&s->ev0) == hipSuccess && hipEventCreate(&s->ev1) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    free_all(s);
    delete s;
    set_error("create: device allocation failed for N=%lld", (long long)cfg->num_particles);
    return NBODY_MI355X_ERR_ALLOC;
  }
  // Workspace of the symmetric kernels: the preferred size if that is under half of the free
  // memory, else what is under half of it down to the kernels' minimum (they then walk the
  // distances in more batches); without one the one-sided kernel runs, same results to rounding.
  {
    nbody_mi355x_step_args probe;
    memset(&probe, 0, sizeof probe);
    probe.shard_slots = s->slots;
    probe.num_shards = 1;
    probe.variant = cfg->variant;
    probe.precision = cfg->precision;
    size_t free_b = 0, total_b = 0;
    // (small systems run the wave-per-body kernels by default: no workspace to hold)
    const bool wanted = (cfg->variant == 0 && s->n > NBODY_MI355X_WAVE_MAX_BODIES) ||
                        nbody_internal::is_sym_variant(cfg->variant);
    int64_t take = 0;
    if (wanted && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      take = nbody_internal::workspace_to_take(probe, nbody_internal::workspace_budget(free_b));
    if (take > 0 && hipMalloc(&s->workspace, static_cast<size_t>(take)) == hipSuccess) {
      s->workspace_bytes = take;
    } else {
      (void)hipGetLastError();
      s->workspace = nullptr;
    }
  }
  *out = s;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_destroy(nbody_mi355x_sim* s) {
  if (!s) return NBODY_MI355X_OK;
  if (s->pending) (void)nbody_mi355x_wait(s);
  if (s->multi) {
    const int mrc = nbody_internal::multi_destroy(s->multi);
    delete s;
    return mrc;
  }
  (void)hipSetDevice(s->cfg.device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  int rc = free_all(s);
  delete s;
  if (rc != NBODY_MI355X_OK) set_error("destroy: hipFree failed");
  return rc;
}

extern "C" int nbody_mi355x_upload(nbody_mi355x_sim* s, const void* particles, const void* mass) {
  if (!s || !particles || !mass) { set_error("upload: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (s->pending) { const int wrc = nbody_mi355x_wait(s); if (wrc != NBODY_MI355X_OK) return wrc; }
  if (!s->multi) HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  const size_t n = static_cast<size_t>(s->n);
  const double* hp = static_cast<const double*>(particles);
  const double* hm = static_cast<const double*>(mass);
  if (s->cfg.host_is_float) {
    s->host_tmp.resize(7 * n);
    const float* fp = static_cast<const float*>(particles);
    const float* fm = static_cast<const float*>(mass);
    for (size_t i = 0; i < 6 * n; i++) s->host_tmp[i] = fp[i];
    for (size_t i = 0; i < n; i++) s->host_tmp[6 * n + i] = fm[i];
    hp = s->host_tmp.data();
    hm = s->host_tmp.data() + 6 * n;
  }
  if (s->multi) return nbody_internal::multi_upload(s->multi, hp, hm);
  const size_t slab = sizeof(double) * 3 * static_cast<size_t>(s->slots);
  // padding slots: mass 0 at the origin, in both position buffers, never written again
  HIP_TRY(hipMemsetAsync(s->pos[0], 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset pos0");
  HIP_TRY(hipMemsetAsync(s->pos[1], 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset pos1");
  HIP_TRY(hipMemsetAsync(s->vel, 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset vel");
  HIP_TRY(hipMemsetAsync(s->mass, 0, sizeof(double) * static_cast<size_t>(s->slots), s->stream),
          NBODY_MI355X_ERR_H2D, "memset mass");
  HIP_TRY(hipMemcpyAsync(s->aos, hp, sizeof(double) * 6 * n, hipMemcpyHostToDevice, s->stream),
          NBODY_MI355X_ERR_H2D, "copy particles");
  HIP_TRY(hipMemcpyAsync(s->mass, hm, sizeof(double) * n, hipMemcpyHostToDevice, s->stream),
          NBODY_MI355X_ERR_H2D, "copy masses");
  s->cur = 0;
  s->acc_valid = false;
  int rc = nbody_mi355x_pack_f64(s->aos, s->n, s->pos[0], s->vel, s->slots, s->stream);
  if (rc != NBODY_MI355X_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream), NBODY_MI355X_ERR_H2D, "upload sync");
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_generate(nbody_mi355x_sim* s, const nbody_mi355x_scene* sc) {
  if (!s || !sc) { set_error("generate: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (!nbody_internal::scene_valid(sc, "generate")) return NBODY_MI355X_ERR_INVALID;  // before anything is touched
  if (sc->num_particles != s->n) {
    set_error("generate: a scene of %lld bodies for a handle of %lld", (long long)sc->num_particles, (long long)s->n);
    return NBODY_MI355X_ERR_INVALID;
  }
  if (s->pending) { const int wrc = nbody_mi355x_wait(s); if (wrc != NBODY_MI355X_OK) return wrc; }
  if (s->multi) return nbody_internal::multi_generate(s->multi, *sc);
  HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  const size_t slab = sizeof(double) * 3 * static_cast<size_t>(s->slots);
  // padding slots: mass 0 at the origin, in both position buffers, never written again
  HIP_TRY(hipMemsetAsync(s->pos[0], 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset pos0");
  HIP_TRY(hipMemsetAsync(s->pos[1], 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset pos1");
  HIP_TRY(hipMemsetAsync(s->vel, 0, slab, s->stream), NBODY_MI355X_ERR_H2D, "memset vel");
  HIP_TRY(hipMemsetAsync(s->mass, 0, sizeof(double) * static_cast<size_t>(s->slots), s->stream),
          NBODY_MI355X_ERR_H2D, "memset mass");
  int rc = nbody_mi355x_scene_fill_f64(sc, 0, s->n, s->pos[0], s->vel, s->mass, s->slots, s->stream);
  if (rc != NBODY_MI355X_OK) return rc;
  if (sc->scene == NBODY_MI355X_SCENE_PLUMMER) {  // into the centre-of-mass frame
    double sums[6];
    rc = nbody_mi355x_slab_sums_f64(s->pos[0], s->vel, s->n, s->slots, s->aos, s->stream);  // (aos: scratch here)
    if (rc != NBODY_MI355X_OK) return rc;
    HIP_TRY(hipMemcpyAsync(sums, s->aos, sizeof sums, hipMemcpyDeviceToHost, s->stream), NBODY_MI355X_ERR_D2H,
            "copy sums");
    HIP_TRY(hipStreamSynchronize(s->stream), NBODY_MI355X_ERR_D2H, "generate sync");
    for (double& v : sums) v /= static_cast<double>(s->n);
    rc = nbody_mi355x_slab_shift_f64(s->pos[0], s->vel, s->n, s->slots, sums, s->stream);
    if (rc != NBODY_MI355X_OK) return rc;
  }
  s->cur = 0;
  s->acc_valid = false;
  HIP_TRY(hipStreamSynchronize(s->stream), NBODY_MI355X_ERR_LAUNCH, "generate sync");
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_run(nbody_mi355x_sim* s, int64_t substeps, void* snapshot_out) {
  const int rc = nbody_mi355x_run_async(s, substeps, snapshot_out);
  return rc != NBODY_MI355X_OK ? rc : nbody_mi355x_wait(s);
}

extern "C" int nbody_mi355x_wait(nbody_mi355x_sim* s) {
  if (!s) { set_error("wait: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (!s->pending) return NBODY_MI355X_OK;
  s->pending = false;
  if (s->multi) {
    const int mrc = nbody_internal::multi_wait(s->multi, &s->last_run_seconds);
    if (mrc != NBODY_MI355X_OK) return mrc;
  } else {
    HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    hipError_t e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("run: %s", hipGetErrorString(e));
      return NBODY_MI355X_ERR_LAUNCH;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_run_seconds = ms * 1e-3;
  }
  if (s->pending_out && s->cfg.host_is_float) {
    float* o = static_cast<float*>(s->pending_out);
    for (size_t i = 0; i < 9 * static_cast<size_t>(s->n); i++) o[i] = static_cast<float>(s->host_tmp[i]);
  }
  s->pending_out = nullptr;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) { set_error("host_alloc: bad arguments"); return NBODY_MI355X_ERR_INVALID; }
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, static_cast<size_t>(bytes), hipHostMallocDefault), NBODY_MI355X_ERR_ALLOC, "hipHostMalloc");
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_host_free(void* p) {
  if (!p) return NBODY_MI355X_OK;
  HIP_TRY(hipHostFree(p), NBODY_MI355X_ERR_FREE, "hipHostFree");
  return NBODY_MI355X_OK;
}

// a whole small step of the (one-device) handle, buffers left to the caller
static nbody_mi355x_step_args step_args_of(const nbody_mi355x_sim* s) {
  nbody_mi355x_step_args a;
  memset(&a, 0, sizeof a);
  a.mass = s->mass;
  a.vel = s->vel;
  a.shard_slots = s->slots;
  a.num_shards = 1;
  a.own_shard = 0;
  a.own_count = s->n;
  a.j0_begin = 0;
  a.j0_end = s->n;  // padding slots need not be visited on one GPU
  a.dt = s->cfg.stepsize;
  a.G = s->cfg.G;
  a.eps2 = s->cfg.eps2;
  a.precision = s->cfg.precision;
  a.variant = s->cfg.variant;
  a.integrator = s->cfg.integrator;
  a.kdk_acc = s->acc;
  a.workspace = s->workspace;
  a.workspace_bytes = s->workspace_bytes;
  return a;
}

static bool takes_tiny_path(const nbody_mi355x_sim* s) {
  return s->n <= nbody_internal::kTinyMaxBodies &&
         (s->cfg.variant == 0 || s->cfg.variant == NBODY_MI355X_VARIANT_TINY);
}

extern "C" int nbody_mi355x_kernel_in_use(nbody_mi355x_sim* s, int32_t* variant, int64_t* workspace_bytes) {
  if (!s || !variant || !workspace_bytes) { set_error("kernel_in_use: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (s->multi) return nbody_internal::multi_kernel_in_use(s->multi, variant, workspace_bytes);
  *workspace_bytes = s->workspace ? s->workspace_bytes : 0;
  if (takes_tiny_path(s)) { *variant = NBODY_MI355X_VARIANT_TINY; return NBODY_MI355X_OK; }
  nbody_mi355x_step_args a = step_args_of(s);
  a.pos_cur = s->pos[0];
  a.pos_next = s->pos[1];
  return nbody_mi355x_step_variant(&a, variant);
}

extern "C" int nbody_mi355x_run_async(nbody_mi355x_sim* s, int64_t substeps, void* snapshot_out) {
  if (!s || substeps <= 0) { set_error("run: bad arguments"); return NBODY_MI355X_ERR_INVALID; }
  if (s->pending) { set_error("run: the previous run_async has not been waited for"); return NBODY_MI355X_ERR_INVALID; }
  if (s->multi) {
    const size_t n9 = 9 * static_cast<size_t>(s->n);
    double* out = static_cast<double*>(snapshot_out);
    if (snapshot_out && s->cfg.host_is_float) { s->host_tmp.resize(n9); out = s->host_tmp.data(); }
    const int mrc = nbody_internal::multi_enqueue(s->multi, substeps, out);
    // (on an error part of the step may be enqueued: wait() drains it)
    s->pending = true;
    s->pending_out = mrc == NBODY_MI355X_OK ? snapshot_out : nullptr;
    if (mrc != NBODY_MI355X_OK) { (void)nbody_internal::multi_wait(s->multi, &s->last_run_seconds); s->pending = false; }
    return mrc;
  }
  HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  nbody_mi355x_step_args a = step_args_of(s);
  HIP_TRY(hipEventRecord(s->ev0, s->stream), NBODY_MI355X_ERR_LAUNCH, "event record");
  int64_t k = 0;
  // at most 64 bodies: the whole big step is ONE launch of one wave (tiny_run), in place on
  // pos[cur] -- the reference's two- and eight-body scenes take 324 000 small steps per big step
  const bool kdk = s->cfg.integrator == NBODY_MI355X_KDK;
  int rc = NBODY_MI355X_OK;
  if (takes_tiny_path(s)) {
    a.pos_cur = s->pos[s->cur];
    a.pos_next = nullptr;
    a.snapshot = snapshot_out ? s->aos : nullptr;
    rc = nbody_internal::tiny_run(a, substeps, kdk && !s->acc_valid, s->stream);
    if (rc == NBODY_MI355X_OK) {
      k = substeps;
      s->acc_valid = kdk;
    }
  }
  // kick-drift-kick (not in the reference): the sums at the current positions are carried from
  // big step to big step in s->acc -- one extra evaluation (a KDK call with dt = 0) the first
  // time -- so that a big step is: drift; substeps-1 x {forces, kick, kick, drift} in one
  // epilogue; {forces, kick} with the snapshot.  The state between big steps is synchronised.
  if (rc == NBODY_MI355X_OK && kdk && k == 0) {
    if (!s->acc_valid) {
      nbody_mi355x_step_args prime = a;
      prime.dt = 0.0;
      prime.pos_cur = s->pos[s->cur];
      prime.pos_next = nullptr;
      rc = nbody_mi355x_step_f64(&prime, s->stream);
      s->acc_valid = rc == NBODY_MI355X_OK;
    }
    if (rc == NBODY_MI355X_OK) {
      a.pos_cur = s->pos[s->cur];
      a.pos_next = s->pos[s->cur ^ 1];
      rc = nbody_mi355x_kdk_drift_f64(&a, s->stream);
      if (rc == NBODY_MI355X_OK) s->cur ^= 1;
    }
    a.integrator = NBODY_MI355X_KDK_CONTINUE;  // every small step but the last
  }
  // the snapshot step, and the last step of a KDK big step, are launched plainly
  const int64_t replayable = substeps - ((snapshot_out || kdk) ? 1 : 0);
  if (rc == NBODY_MI355X_OK && k == 0 && s->n <= kGraphMaxBodies && replayable >= 2 * kGraphSteps && !s->graph_failed) {
    hipGraphExec_t& exec = s->graph[s->cur];
    if (!exec) {
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      int crc = NBODY_MI355X_OK;
      if (ok) {
        a.snapshot = nullptr;
        int c = s->cur;
        for (int g = 0; g < kGraphSteps && crc == NBODY_MI355X_OK; g++, c ^= 1) {
          a.pos_cur = s->pos[c];
          a.pos_next = s->pos[c ^ 1];
          crc = nbody_mi355x_step_f64(&a, s->stream);
        }
        ok = hipStreamEndCapture(s->stream, &graph) == hipSuccess && crc == NBODY_MI355X_OK && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) {  // not fatal: plain launches below do the same work
        (void)hipGetLastError();
        exec = nullptr;
        s->graph_failed = true;
      }
    }
    while (exec && k + kGraphSteps <= replayable) {
      if (hipGraphLaunch(exec, s->stream) != hipSuccess) {
        (void)hipGetLastError();
        set_error("graph launch failed");
        rc = NBODY_MI355X_ERR_LAUNCH;
        break;
      }
      k += kGraphSteps;  // an even number of steps: s->cur is unchanged
    }
  }
  for (; rc == NBODY_MI355X_OK && k < substeps; k++) {
    const bool last = k == substeps - 1;
    a.pos_cur = s->pos[s->cur];
    a.pos_next = s->pos[s->cur ^ 1];
    a.snapshot = (snapshot_out && last) ? s->aos : nullptr;
    if (kdk) a.integrator = last ? NBODY_MI355X_KDK : NBODY_MI355X_KDK_CONTINUE;
    rc = nbody_mi355x_step_f64(&a, s->stream);
    if (rc != NBODY_MI355X_OK) break;
    if (!(kdk && last)) s->cur ^= 1;  // ref: pointer swap after every small step, .cu:157-162
  }
  if (rc != NBODY_MI355X_OK) {
    // a launch failed with part of the big step enqueued: leave the handle consistent -- the
    // work in flight is marked pending so that wait() (or the caller's next call) drains it,
    // no snapshot is delivered, and the KDK carry-over is recomputed next time
    const char* text = nbody_mi355x_last_error();
    char keep[512];
    snprintf(keep, sizeof keep, "%s", text ? text : "");
    (void)hipEventRecord(s->ev1, s->stream);
    s->pending = true;
    s->pending_out = nullptr;
    s->acc_valid = false;
    (void)nbody_mi355x_wait(s);
    set_error("%s", keep);
    return rc;
  }
  HIP_TRY(hipEventRecord(s->ev1, s->stream), NBODY_MI355X_ERR_LAUNCH, "event record");
  if (snapshot_out) {
    const size_t n9 = 9 * static_cast<size_t>(s->n);
    if (s->cfg.host_is_float) {
      s->host_tmp.resize(n9);
      HIP_TRY(hipMemcpyAsync(s->host_tmp.data(), s->aos, sizeof(double) * n9, hipMemcpyDeviceToHost, s->stream),
              NBODY_MI355X_ERR_D2H, "copy snapshot");
    } else {
      HIP_TRY(hipMemcpyAsync(snapshot_out, s->aos, sizeof(double) * n9, hipMemcpyDeviceToHost, s->stream),
              NBODY_MI355X_ERR_D2H, "copy snapshot");
    }
  }
  s->pending = true;
  s->pending_out = snapshot_out;
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_download(nbody_mi355x_sim* s, void* particles) {
  if (!s || !particles) { set_error("download: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (s->pending) { const int wrc = nbody_mi355x_wait(s); if (wrc != NBODY_MI355X_OK) return wrc; }
  if (s->multi) {
    const size_t n6 = 6 * static_cast<size_t>(s->n);
    double* out = static_cast<double*>(particles);
    if (s->cfg.host_is_float) { s->host_tmp.resize(n6); out = s->host_tmp.data(); }
    const int mrc = nbody_internal::multi_download(s->multi, out);
    if (mrc == NBODY_MI355X_OK && s->cfg.host_is_float) {
      float* o = static_cast<float*>(particles);
      for (size_t i = 0; i < n6; i++) o[i] = static_cast<float>(s->host_tmp[i]);
    }
    return mrc;
  }
  HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  int rc = nbody_mi355x_unpack_f64(s->pos[s->cur], s->vel, s->n, s->slots, s->aos, s->stream);
  if (rc != NBODY_MI355X_OK) return rc;
  const size_t n6 = 6 * static_cast<size_t>(s->n);
  void* dst = particles;
  if (s->cfg.host_is_float) { s->host_tmp.resize(n6); dst = s->host_tmp.data(); }
  HIP_TRY(hipMemcpyAsync(dst, s->aos, sizeof(double) * n6, hipMemcpyDeviceToHost, s->stream),
          NBODY_MI355X_ERR_D2H, "copy state");
  HIP_TRY(hipStreamSynchronize(s->stream), NBODY_MI355X_ERR_D2H, "download sync");
  if (s->cfg.host_is_float) {
    float* o = static_cast<float*>(particles);
    for (size_t i = 0; i < n6; i++) o[i] = static_cast<float>(s->host_tmp[i]);
  }
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_energy(nbody_mi355x_sim* s, double* ke, double* pe) {
  if (!s || !ke || !pe) { set_error("energy: null argument"); return NBODY_MI355X_ERR_INVALID; }
  if (s->pending) { const int wrc = nbody_mi355x_wait(s); if (wrc != NBODY_MI355X_OK) return wrc; }
  if (s->multi) return nbody_internal::multi_energy(s->multi, ke, pe);
  HIP_TRY(hipSetDevice(s->cfg.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
  const int64_t eblocks = s->slots / NBODY_MI355X_TILE;
  double* result = s->scratch + 2 * eblocks;
  int rc = nbody_mi355x_energy_f64(s->pos[s->cur], s->mass, s->vel, s->slots, 1, 0, s->n, s->cfg.G,
                                   s->scratch, result, s->stream);
  if (rc != NBODY_MI355X_OK) return rc;
  double h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, result, sizeof h, hipMemcpyDeviceToHost, s->stream), NBODY_MI355X_ERR_D2H,
          "copy energy");
  HIP_TRY(hipStreamSynchronize(s->stream), NBODY_MI355X_ERR_D2H, "energy sync");
  *ke = h[0];
  *pe = h[1];
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_last_exchange_seconds(nbody_mi355x_sim* s, double* positions, double* partial_sums) {
  if (!s || !positions || !partial_sums) { set_error("last_exchange_seconds: null argument"); return NBODY_MI355X_ERR_INVALID; }
  *positions = *partial_sums = 0.0;
  if (s->pending) { set_error("last_exchange_seconds: a run_async is pending: wait first"); return NBODY_MI355X_ERR_INVALID; }
  if (!s->multi) return NBODY_MI355X_OK;  // one device: nothing is exchanged
  return nbody_internal::multi_exchange_seconds(s->multi, positions, partial_sums);
}

extern "C" int nbody_mi355x_exchange_backend(nbody_mi355x_sim* s, int32_t* backend) {
  if (!s || !backend) { set_error("exchange_backend: null argument"); return NBODY_MI355X_ERR_INVALID; }
  *backend = !s->multi ? NBODY_MI355X_EXCHANGE_NONE
                       : (nbody_internal::multi_uses_rccl(s->multi) ? NBODY_MI355X_EXCHANGE_RCCL : NBODY_MI355X_EXCHANGE_PEER_COPIES);
  return NBODY_MI355X_OK;
}

extern "C" int nbody_mi355x_last_run_seconds(nbody_mi355x_sim* s, double* seconds) {
  if (!s || !seconds) { set_error("last_run_seconds: null argument"); return NBODY_MI355X_ERR_INVALID; }
  *seconds = s->last_run_seconds;
  return NBODY_MI355X_OK;
}
