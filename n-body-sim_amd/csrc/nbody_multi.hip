// nbody_multi.hip -- the handle API over several GPUs driven by ONE host thread
// (nbody_mi355x_create_multi; SURVEY 8b "device list", 8e).  The reference has one GPU
// (particle_simulation_GPU.cu:13-35); the partition is the one of its CPU threads
// (particle_simulation.h:98-110): shard k owns the k-th contiguous i-range, remainder to the
// low shards.  The Python driver (nbody_mi355x/sharded.py) is the same scheme with one
// process per GPU and RCCL through torch.distributed; here every shard has a COMPUTE stream
// and a COMM stream, nothing blocks the host inside a big step, and a small step is
//
//   pair-symmetric path (large systems):
//     A1 compute: the tiles that need only the own slab (phase 1, the diagonal) -- while the
//        comm stream is still fetching the other shards' positions of the previous step
//     A2 compute: waits for that exchange, then all other tiles -> partial sums for EVERY
//        slot (acc_all)
//     B  compute: "reduce-scatter" -- pulls the other shards' partial sums for ITS slab, adds
//        them in shard order (deterministic), integrates its bodies into its slab of pos[next]
//     C  comm:    "all-gather" -- pulls the other shards' new slabs into its pos[next]
//   one-sided path (small systems, or no room for the workspace): A+B are one step call
//   after the exchange.
//
// The two exchanges are RCCL collectives -- ncclReduceScatter of the partial sums and an
// in-place ncclAllGather of the new positions, one communicator over the listed devices
// (ncclCommInitAll) -- whenever every shard sits on a GPU of its own and librccl can be
// resolved at run time (from the process: torch's copy if one is mapped; else /opt/rocm/lib:
// no link-time dependency) AND the communicator passes a self-check at create: a small
// patterned buffer is all-gathered and reduce-scattered both ways, by RCCL and by the
// event-ordered peer copies, and the results must agree bit for bit (rccl_self_check).  Where
// any of that fails -- or NBODY_MI355X_MULTI_RCCL=0 -- the exchanges are the peer copies, and
// nbody_mi355x_last_error() says why after the (successful) create.  NBODY_MI355X_MULTI_RCCL=1
// asks for the communicator also where the default would not build one (a single shard).
// RCCL adds the partial sums in ITS order (a ring's), the peer copies in shard order: each is
// bit-reproducible run to run, the two agree to rounding.
//
// Why the collectives alone order the shards (no waits on peers' events in that mode): a
// collective reads and writes rank q's buffers only inside q's own collective kernel, which
// sits in q's stream behind q's waits -- a sender pushes into a peer's user buffer only after
// that peer's kernel has posted it.  So the hazards the peer copies close with events
// (gather_positions, multi_step) are closed by stream order on each side.
//
// Ordering argument: see multi_step and gather_positions.  A device ordinal may appear more
// than once: several shards on one GPU, which is how this path is tested on a one-GPU machine
// (tests/test_gpu_multi.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "nbody_internal.h"

using nbody_internal::set_error;

namespace nbody_internal {

struct Shard {
  int device = 0;
  int64_t count = 0, begin = 0;
  double* pos[2] = {nullptr, nullptr};  // [P][3][S] each: every shard's slab
  double* mass = nullptr;               // [P][S]
  double* vel = nullptr;                // [3][S]   own bodies
  double* acc = nullptr;                // [3][S]   own bodies' complete sums
  double* kacc = nullptr;               // [3][S]   KDK: the sums the last kick was made with
  double* acc_all = nullptr;            // [P][3][S] this shard's partial sums for every slot
  double* stage = nullptr;              // [P][3][S] the other shards' partial sums for the own slab
  double* snap = nullptr;               // [count][9] DeviceOutputParticle rows / [count][6] state rows
  double* scratch = nullptr;            // energy partials + result
  void* workspace = nullptr;
  int64_t workspace_bytes = 0;
  hipStream_t stream = nullptr;  // compute
  hipStream_t comm = nullptr;    // the exchange of positions
  hipEvent_t ev_partials = nullptr, ev_pos = nullptr, ev_gathered = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  // around the two exchanges of the latest small step (timing on): positions on comm, sums on stream
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr, ev_r0 = nullptr, ev_r1 = nullptr;
  void* nccl = nullptr;          // ncclComm_t of this shard (RCCL path only)
};

// RCCL, resolved at run time (no link-time dependency; see the file header)
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclDouble = 8, kNcclSum = 0;  // rccl.h: ncclFloat64 = 8, ncclSum = 0

// Named ranges for a rocprofv3 --marker-trace of a multi-GPU run (the reference's only
// timing hooks are wall-clock timers around a big step, particle_simulation.h:203-207): big
// step, phase 1, exchange, phase 2, reduce-scatter.  roctx is resolved at run time like RCCL
// -- the profiler's copy if it is mapped, else the tracer library torch links, else
// /opt/rocm/lib; without any of them the ranges are no-ops -- no link-time dependency.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* off = getenv("NBODY_MI355X_ROCTX");
    if (off && atoi(off) == 0) return;
    void* lib = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4"})
      if (!lib) lib = dlopen(name, RTLD_NOW);
    if (!lib) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
inline const Roctx& roctx() {
  static const Roctx r;
  return r;
}
struct RangeScope {  // host-side range: what is ENQUEUED inside it (the kernels carry their own names)
  bool on;
  explicit RangeScope(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~RangeScope() {
    if (on) roctx().pop();
  }
};

struct MultiSim {
  nbody_mi355x_config cfg{};
  int64_t n = 0, S = 0;
  int P = 0;
  int cur = 0;
  bool sym = false;
  bool kacc_valid = false;  // KDK: kacc holds the sums at the current positions
  bool rccl = false;        // the exchanges are RCCL collectives (self-checked at create), else peer copies
  bool sums_timed = false;  // the latest small step exchanged partial sums (ev_r0 / ev_r1 are set)
  bool pos_timed = false;   // ... and positions (ev_x0 / ev_x1)
  Rccl nccl;
  int sym_variant = NBODY_MI355X_VARIANT_SYM;
  std::vector<Shard> sh;
  std::vector<double> host;  // slab-layout image for uploads
};

namespace {

#define M_TRY(expr, code, what)                                              \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      (void)hipGetLastError();                                               \
      set_error("multi: %s: %s", what, hipGetErrorString(e_));               \
      return code;                                                           \
    }                                                                        \
  } while (0)


using nbody_internal::is_sym_variant;

// out[i] = sum over shards q = 0..P-1, in that order, of q's partial sums for this slab
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ own,
                                                              const double* __restrict__ stage, int P, int r,
                                                              int64_t len, double* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= len) return;
  double s = 0.0;
  for (int q = 0; q < P; q++) s += (q == r) ? own[i] : stage[static_cast<int64_t>(q) * len + i];
  out[i] = s;
}

// dst (on shard `to`) <- src (on shard `from`), on a stream of `to`
hipError_t pull(const Shard& to, hipStream_t stream, void* dst, const Shard& from, const void* src, size_t bytes) {
  if (to.device == from.device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
  return hipMemcpyPeerAsync(dst, to.device, src, from.device, bytes, stream);
}

#define NCCL_TRY(m, expr, what)                                                          \
  do {                                                                                   \
    const int r_ = (expr);                                                               \
    if (r_ != 0) {                                                                       \
      set_error("multi: %s: %s", what, (m).nccl.GetErrorString ? (m).nccl.GetErrorString(r_) : "RCCL error"); \
      return NBODY_MI355X_ERR_LAUNCH;                                                    \
    }                                                                                    \
  } while (0)

bool load_rccl(Rccl* r) {
  // a copy already mapped into the process (torch ships its own) wins: one RCCL, one HIP runtime
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
  if (!lib) return false;
  r->lib = lib;
  r->CommInitAll = reinterpret_cast<decltype(r->CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
  r->CommDestroy = reinterpret_cast<decltype(r->CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  r->AllGather = reinterpret_cast<decltype(r->AllGather)>(dlsym(lib, "ncclAllGather"));
  r->ReduceScatter = reinterpret_cast<decltype(r->ReduceScatter)>(dlsym(lib, "ncclReduceScatter"));
  r->GroupStart = reinterpret_cast<decltype(r->GroupStart)>(dlsym(lib, "ncclGroupStart"));
  r->GroupEnd = reinterpret_cast<decltype(r->GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
  r->GetErrorString = reinterpret_cast<decltype(r->GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  return r->CommInitAll && r->CommDestroy && r->AllGather && r->ReduceScatter && r->GroupStart && r->GroupEnd;
}

void free_shard(Shard& s) {
  (void)hipSetDevice(s.device);
  if (s.stream) (void)hipStreamSynchronize(s.stream);
  if (s.comm) (void)hipStreamSynchronize(s.comm);
  for (double** p : {&s.pos[0], &s.pos[1], &s.mass, &s.vel, &s.acc, &s.kacc, &s.acc_all, &s.stage, &s.snap, &s.scratch}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (s.workspace) (void)hipFree(s.workspace);
  s.workspace = nullptr;
  for (hipEvent_t* e : {&s.ev_partials, &s.ev_pos, &s.ev_gathered, &s.ev_t0, &s.ev_t1, &s.ev_x0, &s.ev_x1, &s.ev_r0, &s.ev_r1}) {
    if (*e) (void)hipEventDestroy(*e);
    *e = nullptr;
  }
  if (s.stream) (void)hipStreamDestroy(s.stream);
  if (s.comm) (void)hipStreamDestroy(s.comm);
  s.stream = s.comm = nullptr;
}

nbody_mi355x_step_args base_args(const MultiSim& m, const Shard& s, int r) {
  nbody_mi355x_step_args a;
  memset(&a, 0, sizeof a);
  a.pos_cur = s.pos[m.cur];
  a.pos_next = s.pos[m.cur ^ 1];
  a.mass = s.mass;
  a.vel = s.vel;
  a.shard_slots = m.S;
  a.num_shards = m.P;
  a.own_shard = r;
  a.own_count = s.count;
  a.dt = m.cfg.stepsize;
  a.G = m.cfg.G;
  a.eps2 = m.cfg.eps2;
  a.precision = m.cfg.precision;
  a.integrator = m.cfg.integrator;
  a.kdk_acc = s.kacc;
  return a;
}

// The exchange of positions ("all-gather"), on every shard's COMM stream: shard q fetches the
// other shards' slabs of pos[buf], each after its owner's ev_pos, and records ev_gathered.
// What it overwrites -- q's copies of the other slabs -- was last read by q's own kernels one
// step ago, before q's ev_pos of this step in stream order: the comm stream waits for that too.
// What it reads -- r's own slab of pos[buf] -- is next written by r two steps on, after r has
// waited (in B, or in the one-sided kernel's place in the stream) on events q records after
// ev_gathered.
int gather_positions(MultiSim& m, int buf) {
  const int P = m.P;
  const int64_t slab = 3 * m.S;
  const size_t slab_bytes = sizeof(double) * static_cast<size_t>(slab);
  RangeScope range("nbody exchange: positions (all-gather)");
  for (int q = 0; q < P; q++) {
    Shard& s = m.sh[q];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    M_TRY(hipStreamWaitEvent(s.comm, s.ev_pos, 0), NBODY_MI355X_ERR_LAUNCH, "wait own positions");
    if (!m.rccl)
      for (int r = 0; r < P; r++)
        if (r != q) M_TRY(hipStreamWaitEvent(s.comm, m.sh[r].ev_pos, 0), NBODY_MI355X_ERR_LAUNCH, "wait positions");
    M_TRY(hipEventRecord(s.ev_x0, s.comm), NBODY_MI355X_ERR_LAUNCH, "event record");
    if (m.rccl) continue;
    for (int r = 0; r < P; r++) {
      if (r == q) continue;
      M_TRY(pull(s, s.comm, s.pos[buf] + r * slab, m.sh[r], m.sh[r].pos[buf] + r * slab, slab_bytes),
            NBODY_MI355X_ERR_LAUNCH, "pull positions");
    }
    M_TRY(hipEventRecord(s.ev_gathered, s.comm), NBODY_MI355X_ERR_LAUNCH, "event record");
    M_TRY(hipEventRecord(s.ev_x1, s.comm), NBODY_MI355X_ERR_LAUNCH, "event record");
  }
  if (m.rccl) {  // one in-place all-gather: every shard's own slab into its place everywhere
    NCCL_TRY(m, m.nccl.GroupStart(), "ncclGroupStart");
    for (int q = 0; q < P; q++) {
      Shard& s = m.sh[q];
      NCCL_TRY(m, m.nccl.AllGather(s.pos[buf] + q * slab, s.pos[buf], static_cast<size_t>(slab), kNcclDouble, s.nccl, s.comm),
               "ncclAllGather");
    }
    NCCL_TRY(m, m.nccl.GroupEnd(), "ncclGroupEnd");
    for (int q = 0; q < P; q++) {
      Shard& s = m.sh[q];
      M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
      M_TRY(hipEventRecord(s.ev_gathered, s.comm), NBODY_MI355X_ERR_LAUNCH, "event record");
      M_TRY(hipEventRecord(s.ev_x1, s.comm), NBODY_MI355X_ERR_LAUNCH, "event record");
    }
  }
  m.pos_timed = true;
  return NBODY_MI355X_OK;
}

// One small step of every shard, enqueued on the shards' streams.  integrator: the epilogue
// (EULER / KICK_DRIFT / KDK_CONTINUE write new positions, which are then exchanged and become
// current; KDK is the closing kick of a big step: velocities and sums only, nothing to exchange).
int multi_step(MultiSim& m, bool snapshot, int integrator, double dt) {
  const int P = m.P;
  const int64_t S = m.S, slab = 3 * S;
  const size_t slab_bytes = sizeof(double) * static_cast<size_t>(slab);
  const int nxt = m.cur ^ 1;
  m.sums_timed = m.pos_timed = false;
  if (m.sym) {
    // A1: the diagonal tiles need the own slab only, which the shard wrote itself: they run
    //     while the comm stream is still fetching the other slabs of pos[cur].  (Their sums
    //     stay in the workspace; the workspace's previous contents were consumed by the
    //     gather kernel of the last step's A2, earlier on this stream.)
    {
      RangeScope range("nbody phase 1: diagonal tiles (own slab, under the exchange)");
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
        nbody_mi355x_step_args a = base_args(m, s, r);
        a.variant = m.sym_variant;
        a.workspace = s.workspace;
        a.workspace_bytes = s.workspace_bytes;
        a.phase = 1;
        const int rc = nbody_mi355x_step_f64(&a, s.stream);
        if (rc != NBODY_MI355X_OK) return rc;
      }
    }
    // A2: everything else, once the exchange is complete.  acc_all of shard r was last read by
    //     its peers in B of the previous step; r's wait on ev_gathered is a wait on every
    //     peer's ev_pos of that step, recorded after those reads.
    {
      RangeScope range("nbody phase 2: all other tiles");
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
        M_TRY(hipStreamWaitEvent(s.stream, s.ev_gathered, 0), NBODY_MI355X_ERR_LAUNCH, "wait exchange");
        nbody_mi355x_step_args a = base_args(m, s, r);
        a.acc_out = s.acc_all;
        a.variant = m.sym_variant;
        a.workspace = s.workspace;
        a.workspace_bytes = s.workspace_bytes;
        a.phase = 2;
        const int rc = nbody_mi355x_step_f64(&a, s.stream);
        if (rc != NBODY_MI355X_OK) return rc;
        M_TRY(hipEventRecord(s.ev_partials, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
      }
    }
    // B: "reduce-scatter", then integrate.  r's slab of pos[next] was last read by its peers
    //    in C two steps ago; each peer recorded ev_partials (waited on here) after its wait on
    //    the ev_gathered that closed that read.  (RCCL: see the file header.)
    {
      RangeScope range("nbody exchange: partial sums (reduce-scatter)");
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
        if (!m.rccl)
          for (int q = 0; q < P; q++)
            if (q != r) M_TRY(hipStreamWaitEvent(s.stream, m.sh[q].ev_partials, 0), NBODY_MI355X_ERR_LAUNCH, "wait partials");
        M_TRY(hipEventRecord(s.ev_r0, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
      }
      if (m.rccl) {
        NCCL_TRY(m, m.nccl.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < P; r++) {
          Shard& s = m.sh[r];
          NCCL_TRY(m, m.nccl.ReduceScatter(s.acc_all, s.acc, static_cast<size_t>(slab), kNcclDouble, kNcclSum, s.nccl, s.stream),
                   "ncclReduceScatter");
        }
        NCCL_TRY(m, m.nccl.GroupEnd(), "ncclGroupEnd");
      }
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
        if (!m.rccl) {
          for (int q = 0; q < P; q++) {
            if (q == r) continue;
            M_TRY(pull(s, s.stream, s.stage + q * slab, m.sh[q], m.sh[q].acc_all + r * slab, slab_bytes),
                  NBODY_MI355X_ERR_LAUNCH, "pull partial sums");
          }
          hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((slab + 255) / 256)), dim3(256), 0,
                             s.stream, s.acc_all + r * slab, s.stage, P, r, slab, s.acc);
          M_TRY(hipGetLastError(), NBODY_MI355X_ERR_LAUNCH, "reduce launch");
        }
        M_TRY(hipEventRecord(s.ev_r1, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
      }
      m.sums_timed = true;
    }
    {
      RangeScope range("nbody integrate");
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
        nbody_mi355x_step_args a = base_args(m, s, r);
        a.integrator = integrator;
        a.dt = dt;
        a.acc_in = s.acc;  // no j range: acc_in -> the integrator's update (+ snapshot)
        a.variant = 1;
        a.snapshot = snapshot ? s.snap : nullptr;
        const int rc = nbody_mi355x_step_f64(&a, s.stream);
        if (rc != NBODY_MI355X_OK) return rc;
        M_TRY(hipEventRecord(s.ev_pos, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
      }
    }
  } else {
    // one-sided: every shard sums over all slabs and integrates in one call, after the
    // exchange.  r's slab of pos[next] was last read by its peers in C two steps ago; r's wait
    // on ev_gathered of the step in between was a wait on ev_pos events its peers recorded
    // after their own waits on the ev_gathered that closed that read.
    RangeScope range("nbody one-sided step (after the exchange)");
    for (int r = 0; r < P; r++) {
      Shard& s = m.sh[r];
      M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
      M_TRY(hipStreamWaitEvent(s.stream, s.ev_gathered, 0), NBODY_MI355X_ERR_LAUNCH, "wait exchange");
      nbody_mi355x_step_args a = base_args(m, s, r);
      a.integrator = integrator;
      a.dt = dt;
      a.j0_begin = 0;
      a.j0_end = static_cast<int64_t>(P) * S;
      a.variant = is_sym_variant(m.cfg.variant) ? 0 : m.cfg.variant;
      a.snapshot = snapshot ? s.snap : nullptr;
      const int rc = nbody_mi355x_step_f64(&a, s.stream);
      if (rc != NBODY_MI355X_OK) return rc;
      M_TRY(hipEventRecord(s.ev_pos, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
    }
  }
  if (integrator == NBODY_MI355X_KDK) return NBODY_MI355X_OK;  // positions unchanged
  // C: the exchange of the new positions, on the comm streams
  const int prc = gather_positions(m, nxt);
  if (prc != NBODY_MI355X_OK) return prc;
  m.cur = nxt;
  return NBODY_MI355X_OK;
}

int sync_all(MultiSim& m, int code, const char* what) {
  for (Shard& s : m.sh) {
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    M_TRY(hipStreamSynchronize(s.stream), code, what);
    M_TRY(hipStreamSynchronize(s.comm), code, what);
  }
  return NBODY_MI355X_OK;
}

// The communicator's self-check at create (see the file header): both exchanges on a small
// patterned buffer, by RCCL and by the event-ordered peer copies, must agree BIT FOR BIT.
// The pattern is small integers, whose sums are exact in any order, so the two reduce-scatters
// must give the same bits although they add in different orders.  Uses pos[0] / acc_all /
// stage / acc of the shards as scratch (before any upload); returns false with the reason in
// `why` when anything disagrees or fails.  Each shard's streams are idle on return.
bool rccl_self_check(MultiSim& m, char* why, size_t why_len) {
  const int P = m.P;
  const int64_t len = 384;  // doubles per rank: 3 KiB -- all-gather P * len, reduce-scatter P * len in, len out
  if (3 * m.S < len) { snprintf(why, why_len, "slab too small for the self-check"); return false; }
  std::vector<double> host(static_cast<size_t>(P) * len), got_rccl, got_peer;
  auto fail = [&](const char* what, hipError_t e) {
    (void)hipGetLastError();
    snprintf(why, why_len, "self-check: %s: %s", what, hipGetErrorString(e));
    return false;
  };
  // scratch: a = pos[0] (P * len: gather target / reduce input), b = pos[1] (second target), c = vel (len: reduce output)
  for (int r = 0; r < P; r++) {
    Shard& s = m.sh[r];
    hipError_t e = hipSetDevice(s.device);
    if (e != hipSuccess) return fail("hipSetDevice", e);
    // rank r's contribution for slot (q, i): an integer below 2^20 that differs by rank, slot and index
    for (int q = 0; q < P; q++)
      for (int64_t i = 0; i < len; i++) host[static_cast<size_t>(q) * len + i] = static_cast<double>((r + 1) * 4099 + q * 131 + i * 7 + ((i * (r + 3)) % 5));
    e = hipMemcpyAsync(s.pos[0], host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, s.stream);
    if (e == hipSuccess) e = hipMemsetAsync(s.pos[1], 0xff, sizeof(double) * host.size(), s.stream);
    if (e == hipSuccess) e = hipMemsetAsync(s.vel, 0xff, sizeof(double) * len, s.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess) return fail("fill", e);
  }
  auto read_back = [&](std::vector<double>* out, bool gather) -> bool {
    out->assign(static_cast<size_t>(P) * (gather ? P * len : len), 0.0);
    for (int r = 0; r < P; r++) {
      Shard& s = m.sh[r];
      hipError_t e = hipSetDevice(s.device);
      if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
      const size_t n = static_cast<size_t>(gather ? P * len : len);
      if (e == hipSuccess) e = hipMemcpy(out->data() + r * n, gather ? s.pos[1] : s.vel, sizeof(double) * n, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { fail("read back", e); return false; }
    }
    return true;
  };
  // ---- all-gather of each rank's own block (block r of its pos[0]) into pos[1], both ways
  for (int pass = 0; pass < 2; pass++) {
    const bool rccl = pass == 0;
    if (rccl) {
      if (m.nccl.GroupStart() != 0) { snprintf(why, why_len, "self-check: ncclGroupStart failed"); return false; }
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        const int rc = m.nccl.AllGather(s.pos[0] + r * len, s.pos[1], static_cast<size_t>(len), kNcclDouble, s.nccl, s.stream);
        if (rc != 0) { (void)m.nccl.GroupEnd(); snprintf(why, why_len, "self-check: ncclAllGather: %s", m.nccl.GetErrorString ? m.nccl.GetErrorString(rc) : "error"); return false; }
      }
      if (m.nccl.GroupEnd() != 0) { snprintf(why, why_len, "self-check: ncclGroupEnd failed"); return false; }
    } else {
      for (int q = 0; q < P; q++) {
        Shard& s = m.sh[q];
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipMemsetAsync(s.pos[1], 0xff, sizeof(double) * static_cast<size_t>(P) * len, s.stream);
        for (int r = 0; r < P && e == hipSuccess; r++)
          e = pull(s, s.stream, s.pos[1] + r * len, m.sh[r], m.sh[r].pos[0] + r * len, sizeof(double) * len);
        if (e != hipSuccess) return fail("peer all-gather", e);
      }
    }
    if (!read_back(rccl ? &got_rccl : &got_peer, true)) return false;
  }
  if (got_rccl.size() != got_peer.size() || memcmp(got_rccl.data(), got_peer.data(), sizeof(double) * got_rccl.size()) != 0) {
    snprintf(why, why_len, "self-check: ncclAllGather and the peer copies disagree");
    return false;
  }
  // ---- reduce-scatter of pos[0] ([P][len] per rank) into vel, both ways
  for (int pass = 0; pass < 2; pass++) {
    const bool rccl = pass == 0;
    if (rccl) {
      if (m.nccl.GroupStart() != 0) { snprintf(why, why_len, "self-check: ncclGroupStart failed"); return false; }
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        const int rc = m.nccl.ReduceScatter(s.pos[0], s.vel, static_cast<size_t>(len), kNcclDouble, kNcclSum, s.nccl, s.stream);
        if (rc != 0) { (void)m.nccl.GroupEnd(); snprintf(why, why_len, "self-check: ncclReduceScatter: %s", m.nccl.GetErrorString ? m.nccl.GetErrorString(rc) : "error"); return false; }
      }
      if (m.nccl.GroupEnd() != 0) { snprintf(why, why_len, "self-check: ncclGroupEnd failed"); return false; }
    } else {
      for (int r = 0; r < P; r++) {
        Shard& s = m.sh[r];
        hipError_t e = hipSetDevice(s.device);
        // stage in pos[1]: [P][len], block q = rank q's contribution for rank r
        for (int q = 0; q < P && e == hipSuccess; q++)
          if (q != r) e = pull(s, s.stream, s.pos[1] + q * len, m.sh[q], m.sh[q].pos[0] + r * len, sizeof(double) * len);
        if (e != hipSuccess) return fail("peer reduce-scatter", e);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((len + 255) / 256)), dim3(256), 0, s.stream,
                           s.pos[0] + r * len, s.pos[1], P, r, len, s.vel);
        e = hipGetLastError();
        if (e != hipSuccess) return fail("reduce launch", e);
      }
    }
    if (!read_back(rccl ? &got_rccl : &got_peer, false)) return false;
  }
  if (memcmp(got_rccl.data(), got_peer.data(), sizeof(double) * got_rccl.size()) != 0) {
    snprintf(why, why_len, "self-check: ncclReduceScatter and the peer copies disagree");
    return false;
  }
  if (const char* inject = getenv("NBODY_MI355X_MULTI_SELFCHECK_FAIL"); inject && atoi(inject) != 0) {
    snprintf(why, why_len, "self-check: failure injected by NBODY_MI355X_MULTI_SELFCHECK_FAIL (tests)");
    return false;  // (exercises the fall-back: communicators destroyed, peer copies, the reason reported)
  }
  // and against the closed form (both could be wrong together): sum over ranks of the pattern
  for (int r = 0; r < P; r++)
    for (int64_t i = 0; i < len; i++) {
      double want = 0.0;
      for (int q = 0; q < P; q++) want += static_cast<double>((q + 1) * 4099 + r * 131 + i * 7 + ((i * (q + 3)) % 5));
      if (got_rccl[static_cast<size_t>(r) * len + i] != want) {
        snprintf(why, why_len, "self-check: reduce-scatter sums are wrong (rank %d, element %lld)", r, (long long)i);
        return false;
      }
    }
  return true;
}

}  // namespace

int multi_destroy(MultiSim* m) {
  if (!m) return NBODY_MI355X_OK;
  for (Shard& s : m->sh) {
    if (s.nccl && m->nccl.CommDestroy) {
      (void)hipSetDevice(s.device);
      if (s.stream) (void)hipStreamSynchronize(s.stream);
      if (s.comm) (void)hipStreamSynchronize(s.comm);
      (void)m->nccl.CommDestroy(s.nccl);
      s.nccl = nullptr;
    }
  }
  for (Shard& s : m->sh) free_shard(s);
  delete m;
  return NBODY_MI355X_OK;
}

int multi_create(const nbody_mi355x_config& cfg, const int32_t* devices, int32_t num, MultiSim** out) {
  *out = nullptr;
  set_error("%s", "");  // (a successful create may leave a note here: staged peer copies, RCCL not taken)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_error("create_multi: no HIP device visible");
    return NBODY_MI355X_ERR_NO_DEVICE;
  }
  if (num < 1 || num > 64 || num > cfg.num_particles) {
    set_error("create_multi: %d shards for %lld bodies", num, (long long)cfg.num_particles);
    return NBODY_MI355X_ERR_INVALID;
  }
  for (int k = 0; k < num; k++)
    if (devices[k] < 0 || devices[k] >= ndev) {
      set_error("create_multi: device %d outside [0,%d)", devices[k], ndev);
      return NBODY_MI355X_ERR_INVALID;
    }
  MultiSim* m = new (std::nothrow) MultiSim();
  if (!m) { set_error("create_multi: out of host memory"); return NBODY_MI355X_ERR_ALLOC; }
  m->cfg = cfg;
  m->n = cfg.num_particles;
  m->P = num;
  m->sh.resize(num);
  const int64_t base = m->n / num, rem = m->n % num;  // ref: particle_simulation.h:98-110
  for (int r = 0; r < num; r++) {
    m->sh[r].device = devices[r];
    m->sh[r].count = base + (r < rem ? 1 : 0);
    m->sh[r].begin = base * r + std::min<int64_t>(r, rem);
  }
  const int64_t big = m->sh[0].count;
  m->S = slab_slots(big, cfg.variant, num);
  const int64_t S = m->S;

  // pair-symmetric kernels: when asked for, or by default for systems they pay off on
  nbody_mi355x_step_args probe;
  memset(&probe, 0, sizeof probe);
  probe.shard_slots = S;
  probe.num_shards = num;
  probe.precision = cfg.precision;
  probe.variant = is_sym_variant(cfg.variant) ? cfg.variant : NBODY_MI355X_VARIANT_SYM;
  int64_t need = 0;
  const bool wanted = is_sym_variant(cfg.variant) || (cfg.variant == 0 && m->n >= 2048);
  m->sym = wanted && nbody_mi355x_step_workspace_bytes(&probe, &need) == NBODY_MI355X_OK && need > 0;
  m->sym_variant = probe.variant;

  // peers that are different GPUs: direct access where the platform offers it.  Without it
  // the pulls still work, staged by the runtime through host memory -- slower, so it is said:
  // the text stays in nbody_mi355x_last_error() after a successful create.
  int no_peer = 0;
  bool distinct = true;
  for (int r = 0; r < num; r++)
    for (int q = 0; q < num; q++) {
      if (r == q) continue;
      if (devices[r] == devices[q]) { distinct = false; continue; }
      int can = 0;
      hipError_t e = hipSetDevice(devices[r]);
      if (e == hipSuccess) e = hipDeviceCanAccessPeer(&can, devices[r], devices[q]);
      if (e == hipSuccess && can) {
        e = hipDeviceEnablePeerAccess(devices[q], 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
      }
      (void)hipGetLastError();
      if (e != hipSuccess || !can) {
        no_peer++;
        set_error("create_multi: no direct peer access from device %d to device %d (%s): exchanges between them are "
                  "staged by the runtime", devices[r], devices[q], e != hipSuccess ? hipGetErrorString(e) : "not offered");
      }
    }
  (void)no_peer;

  const size_t all = sizeof(double) * 3 * static_cast<size_t>(S) * num, one = sizeof(double) * 3 * static_cast<size_t>(S);
  const int64_t eblocks = S / NBODY_MI355X_TILE;
  bool ok = true;
  for (int r = 0; r < num && ok; r++) {
    Shard& s = m->sh[r];
    ok = hipSetDevice(s.device) == hipSuccess && hipMalloc(&s.pos[0], all) == hipSuccess &&
         hipMalloc(&s.pos[1], all) == hipSuccess &&
         hipMalloc(&s.mass, sizeof(double) * static_cast<size_t>(S) * num) == hipSuccess &&
         hipMalloc(&s.vel, one) == hipSuccess && hipMalloc(&s.acc, one) == hipSuccess &&
         (cfg.integrator != NBODY_MI355X_KDK || hipMalloc(&s.kacc, one) == hipSuccess) &&
         hipMalloc(&s.snap, sizeof(double) * 9 * static_cast<size_t>(s.count)) == hipSuccess &&
         hipMalloc(&s.scratch, sizeof(double) * static_cast<size_t>(2 * eblocks + 2)) == hipSuccess &&
         hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&s.comm, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&s.ev_partials, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.ev_pos, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.ev_gathered, hipEventDisableTiming) == hipSuccess &&
         hipEventCreate(&s.ev_t0) == hipSuccess && hipEventCreate(&s.ev_t1) == hipSuccess &&
         hipEventCreate(&s.ev_x0) == hipSuccess && hipEventCreate(&s.ev_x1) == hipSuccess &&
         hipEventCreate(&s.ev_r0) == hipSuccess && hipEventCreate(&s.ev_r1) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    multi_destroy(m);
    set_error("create_multi: device allocation failed for N=%lld over %d shards", (long long)cfg.num_particles, num);
    return NBODY_MI355X_ERR_ALLOC;
  }
  // The symmetric path's buffers are taken only if they fit comfortably on EVERY shard's
  // device; otherwise all shards run the one-sided kernel (same results to rounding).
  for (int r = 0; r < num && m->sym; r++) {
    Shard& s = m->sh[r];
    size_t free_b = 0, total_b = 0;
    // the preferred workspace, or what is left of half the free memory down to the minimum
    int64_t take = 0;
    if (hipSetDevice(s.device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const int64_t budget = nbody_internal::workspace_budget(free_b) - 2 * static_cast<int64_t>(all);
      take = budget > 0 ? nbody_internal::workspace_to_take(probe, budget) : 0;
    }
    if (take > 0 && hipMalloc(&s.acc_all, all) == hipSuccess && hipMalloc(&s.stage, all) == hipSuccess &&
        hipMalloc(&s.workspace, static_cast<size_t>(take)) == hipSuccess) {
      s.workspace_bytes = take;
      continue;
    }
    (void)hipGetLastError();
    m->sym = false;
    for (Shard& t : m->sh) {
      (void)hipSetDevice(t.device);
      if (t.acc_all) (void)hipFree(t.acc_all);
      if (t.stage) (void)hipFree(t.stage);
      if (t.workspace) (void)hipFree(t.workspace);
      t.acc_all = t.stage = nullptr;
      t.workspace = nullptr;
      t.workspace_bytes = 0;
    }
  }
  // RCCL for the two exchanges (see the file header): by default whenever every shard has a GPU
  // of its own (and there is something to exchange), behind the self-check; anything missing
  // -> the peer copies, and the reason is left in last_error
  const char* want = getenv("NBODY_MI355X_MULTI_RCCL");
  const int mode = want ? (atoi(want) != 0 ? 1 : 0) : -1;  // 1 asked for, 0 refused, -1 default
  if (mode == 1 || (mode == -1 && distinct && num > 1)) {
    if (!distinct) {
      set_error("create_multi: NBODY_MI355X_MULTI_RCCL needs every shard on its own GPU; using peer copies");
    } else if (!load_rccl(&m->nccl)) {
      set_error("create_multi: librccl could not be loaded (%s); using peer copies", dlerror());
    } else {
      std::vector<void*> comms(num, nullptr);
      std::vector<int> devs(devices, devices + num);
      const int rc = m->nccl.CommInitAll(comms.data(), num, devs.data());
      if (rc != 0) {
        set_error("create_multi: ncclCommInitAll: %s; using peer copies",
                  m->nccl.GetErrorString ? m->nccl.GetErrorString(rc) : "error");
      } else {
        for (int r = 0; r < num; r++) m->sh[r].nccl = comms[r];
        char why[200] = "";
        if (rccl_self_check(*m, why, sizeof why)) {
          m->rccl = true;
        } else {
          for (int r = 0; r < num; r++) {
            (void)hipSetDevice(m->sh[r].device);
            (void)hipStreamSynchronize(m->sh[r].stream);
            (void)m->nccl.CommDestroy(m->sh[r].nccl);
            m->sh[r].nccl = nullptr;
          }
          set_error("create_multi: RCCL %s; using peer copies", why);
        }
      }
    }
  }
  *out = m;
  return NBODY_MI355X_OK;
}

int multi_upload(MultiSim* m, const double* particles, const double* mass) {
  const int P = m->P;
  const int64_t S = m->S;
  // host image of the slab layout; padding slots: mass 0 at the origin
  m->host.assign(static_cast<size_t>(P) * 3 * S + static_cast<size_t>(P) * S + 3 * static_cast<size_t>(S), 0.0);
  double* hpos = m->host.data();
  double* hmass = hpos + static_cast<size_t>(P) * 3 * S;
  double* hvel = hmass + static_cast<size_t>(P) * S;
  for (int r = 0; r < P; r++) {
    const Shard& s = m->sh[r];
    for (int64_t i = 0; i < s.count; i++) {
      const double* p = particles + 6 * (s.begin + i);
      for (int c = 0; c < 3; c++) hpos[(static_cast<size_t>(r) * 3 + c) * S + i] = p[c];
      hmass[static_cast<size_t>(r) * S + i] = mass[s.begin + i];
    }
  }
  for (int r = 0; r < P; r++) {
    Shard& s = m->sh[r];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    for (int c = 0; c < 3; c++)
      for (int64_t i = 0; i < S; i++) hvel[c * S + i] = i < s.count ? particles[6 * (s.begin + i) + 3 + c] : 0.0;
    const size_t all = sizeof(double) * 3 * static_cast<size_t>(S) * P;
    M_TRY(hipMemcpyAsync(s.pos[0], hpos, all, hipMemcpyHostToDevice, s.stream), NBODY_MI355X_ERR_H2D, "copy positions");
    M_TRY(hipMemcpyAsync(s.pos[1], hpos, all, hipMemcpyHostToDevice, s.stream), NBODY_MI355X_ERR_H2D, "copy positions");
    M_TRY(hipMemcpyAsync(s.mass, hmass, sizeof(double) * static_cast<size_t>(S) * P, hipMemcpyHostToDevice, s.stream),
          NBODY_MI355X_ERR_H2D, "copy masses");
    M_TRY(hipMemcpyAsync(s.vel, hvel, sizeof(double) * 3 * static_cast<size_t>(S), hipMemcpyHostToDevice, s.stream),
          NBODY_MI355X_ERR_H2D, "copy velocities");
    M_TRY(hipStreamSynchronize(s.stream), NBODY_MI355X_ERR_H2D, "upload sync");  // hvel is reused
  }
  m->cur = 0;
  m->kacc_valid = false;
  m->host.clear();
  m->host.shrink_to_fit();
  return NBODY_MI355X_OK;
}

// The scene made where it is held: every device fills all P slabs of its position and mass
// buffers (O(N) per device, no exchange) and its own slab's velocities.
int multi_generate(MultiSim* m, const nbody_mi355x_scene& sc) {
  const int P = m->P;
  const int64_t S = m->S;
  const size_t all = sizeof(double) * 3 * static_cast<size_t>(S) * P;
  for (int o = 0; o < P; o++) {
    Shard& s = m->sh[o];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    // padding slots: mass 0 at the origin
    M_TRY(hipMemsetAsync(s.pos[0], 0, all, s.stream), NBODY_MI355X_ERR_H2D, "memset positions");
    M_TRY(hipMemsetAsync(s.mass, 0, sizeof(double) * static_cast<size_t>(S) * P, s.stream), NBODY_MI355X_ERR_H2D,
          "memset masses");
    M_TRY(hipMemsetAsync(s.vel, 0, sizeof(double) * 3 * static_cast<size_t>(S), s.stream), NBODY_MI355X_ERR_H2D,
          "memset velocities");
    for (int r = 0; r < P; r++) {
      const int rc = nbody_mi355x_scene_fill_f64(&sc, m->sh[r].begin, m->sh[r].count, s.pos[0] + static_cast<size_t>(r) * 3 * S,
                                                 r == o ? s.vel : nullptr, s.mass + static_cast<size_t>(r) * S, S, s.stream);
      if (rc != NBODY_MI355X_OK) return rc;
    }
  }
  if (sc.scene == NBODY_MI355X_SCENE_PLUMMER) {
    // centre-of-mass frame: every device sums its own slab (positions and velocities), the host
    // adds the P partial sums in shard order, every device shifts what it holds
    std::vector<double> part(static_cast<size_t>(P) * 6);
    for (int o = 0; o < P; o++) {
      Shard& s = m->sh[o];
      M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
      const int rc = nbody_mi355x_slab_sums_f64(s.pos[0] + static_cast<size_t>(o) * 3 * S, s.vel, s.count, S, s.acc, s.stream);
      if (rc != NBODY_MI355X_OK) return rc;
      M_TRY(hipMemcpyAsync(&part[static_cast<size_t>(o) * 6], s.acc, 6 * sizeof(double), hipMemcpyDeviceToHost, s.stream),
            NBODY_MI355X_ERR_D2H, "copy sums");
    }
    double shift[6] = {0, 0, 0, 0, 0, 0};
    for (int o = 0; o < P; o++) {
      M_TRY(hipSetDevice(m->sh[o].device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
      M_TRY(hipStreamSynchronize(m->sh[o].stream), NBODY_MI355X_ERR_D2H, "generate sync");
      for (int c = 0; c < 6; c++) shift[c] += part[static_cast<size_t>(o) * 6 + c];
    }
    for (double& v : shift) v /= static_cast<double>(m->n);
    for (int o = 0; o < P; o++) {
      Shard& s = m->sh[o];
      M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
      for (int r = 0; r < P; r++) {
        const int rc = nbody_mi355x_slab_shift_f64(s.pos[0] + static_cast<size_t>(r) * 3 * S, r == o ? s.vel : nullptr,
                                                   m->sh[r].count, S, shift, s.stream);
        if (rc != NBODY_MI355X_OK) return rc;
      }
    }
  }
  for (int o = 0; o < P; o++) {
    Shard& s = m->sh[o];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    M_TRY(hipMemcpyAsync(s.pos[1], s.pos[0], all, hipMemcpyDeviceToDevice, s.stream), NBODY_MI355X_ERR_H2D,
          "copy positions");
    M_TRY(hipStreamSynchronize(s.stream), NBODY_MI355X_ERR_LAUNCH, "generate sync");
  }
  m->cur = 0;
  m->kacc_valid = false;
  return NBODY_MI355X_OK;
}

// Enqueues a big step on every shard's stream and returns; multi_wait completes it.
int multi_enqueue(MultiSim* m, int64_t substeps, double* snapshot_out) {
  RangeScope range("nbody big step (enqueue)");
  for (Shard& s : m->sh) {
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    M_TRY(hipEventRecord(s.ev_t0, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
  }
  const double dt = m->cfg.stepsize;
  if (m->cfg.integrator == NBODY_MI355X_KDK) {
    // kick-drift-kick (see nbody_mi355x_run_async): sums carried in kacc (primed by a KDK step
    // with dt = 0); drift + exchange; substeps-1 fused steps; the closing kick with the snapshot
    // (any error from here on leaves the shards half-kicked or half-drifted and kacc stale:
    // kacc_valid goes false on every error return; the state then needs upload / generate)
    if (!m->kacc_valid) {
      const int rc = multi_step(*m, false, NBODY_MI355X_KDK, 0.0);
      if (rc != NBODY_MI355X_OK) return rc;
      m->kacc_valid = true;
    }
    const int nxt = m->cur ^ 1;
    for (int r = 0; r < m->P; r++) {
      Shard& s = m->sh[r];
      hipError_t e = hipSetDevice(s.device);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        m->kacc_valid = false;
        set_error("multi: hipSetDevice: %s", hipGetErrorString(e));
        return NBODY_MI355X_ERR_NO_DEVICE;
      }
      // r's slab of pos[nxt] was last read by its peers two exchanges ago; since then r has
      // waited on an event each of them recorded after that read (ev_partials in B, or ev_pos
      // in the exchange in between)
      nbody_mi355x_step_args a = base_args(*m, s, r);
      const int rc = nbody_mi355x_kdk_drift_f64(&a, s.stream);
      if (rc != NBODY_MI355X_OK) { m->kacc_valid = false; return rc; }
      e = hipEventRecord(s.ev_pos, s.stream);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        m->kacc_valid = false;
        set_error("multi: event record: %s", hipGetErrorString(e));
        return NBODY_MI355X_ERR_LAUNCH;
      }
    }
    const int prc = gather_positions(*m, nxt);
    if (prc != NBODY_MI355X_OK) { m->kacc_valid = false; return prc; }
    m->cur = nxt;
    for (int64_t k = 0; k + 1 < substeps; k++) {
      const int rc = multi_step(*m, false, NBODY_MI355X_KDK_CONTINUE, dt);
      if (rc != NBODY_MI355X_OK) { m->kacc_valid = false; return rc; }
    }
    const int rc = multi_step(*m, snapshot_out != nullptr, NBODY_MI355X_KDK, dt);
    if (rc != NBODY_MI355X_OK) { m->kacc_valid = false; return rc; }
  } else {
    for (int64_t k = 0; k < substeps; k++) {
      const int rc = multi_step(*m, snapshot_out != nullptr && k == substeps - 1, m->cfg.integrator, dt);
      if (rc != NBODY_MI355X_OK) return rc;
    }
  }
  for (Shard& s : m->sh) {
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    // (the big step is complete when the last exchange is: it belongs inside the timed span)
    M_TRY(hipStreamWaitEvent(s.stream, s.ev_gathered, 0), NBODY_MI355X_ERR_LAUNCH, "wait exchange");
    M_TRY(hipEventRecord(s.ev_t1, s.stream), NBODY_MI355X_ERR_LAUNCH, "event record");
    if (snapshot_out)
      M_TRY(hipMemcpyAsync(snapshot_out + 9 * s.begin, s.snap, sizeof(double) * 9 * static_cast<size_t>(s.count),
                           hipMemcpyDeviceToHost, s.stream),
            NBODY_MI355X_ERR_D2H, "copy snapshot");
  }
  return NBODY_MI355X_OK;
}

int multi_wait(MultiSim* m, double* seconds) {
  const int rc = sync_all(*m, NBODY_MI355X_ERR_LAUNCH, "run");
  if (rc != NBODY_MI355X_OK) return rc;
  double longest = 0.0;
  for (Shard& s : m->sh) {
    float ms = 0.f;
    if (hipSetDevice(s.device) == hipSuccess && hipEventElapsedTime(&ms, s.ev_t0, s.ev_t1) == hipSuccess)
      longest = std::max(longest, ms * 1e-3);
  }
  *seconds = longest;
  return NBODY_MI355X_OK;
}

// Device time of the two exchanges of the LAST small step of the last completed run, the
// longest over the shards: positions (comm stream: from "every producer's slab is ready" to
// "gathered") and partial sums (compute stream: pulls + reduction, or the reduce-scatter).
int multi_exchange_seconds(MultiSim* m, double* positions, double* sums) {
  *positions = *sums = 0.0;
  for (Shard& s : m->sh) {
    float ms = 0.f;
    if (hipSetDevice(s.device) != hipSuccess) continue;
    if (m->pos_timed && hipEventElapsedTime(&ms, s.ev_x0, s.ev_x1) == hipSuccess) *positions = std::max(*positions, ms * 1e-3);
    if (m->sums_timed && hipEventElapsedTime(&ms, s.ev_r0, s.ev_r1) == hipSuccess) *sums = std::max(*sums, ms * 1e-3);
  }
  (void)hipGetLastError();
  return NBODY_MI355X_OK;
}

bool multi_uses_rccl(const MultiSim* m) { return m->rccl; }

int multi_download(MultiSim* m, double* particles) {
  const int64_t slab = 3 * m->S;
  for (int r = 0; r < m->P; r++) {
    Shard& s = m->sh[r];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    const int rc = nbody_mi355x_unpack_f64(s.pos[m->cur] + r * slab, s.vel, s.count, m->S, s.snap, s.stream);
    if (rc != NBODY_MI355X_OK) return rc;
    M_TRY(hipMemcpyAsync(particles + 6 * s.begin, s.snap, sizeof(double) * 6 * static_cast<size_t>(s.count),
                         hipMemcpyDeviceToHost, s.stream),
          NBODY_MI355X_ERR_D2H, "copy state");
  }
  return sync_all(*m, NBODY_MI355X_ERR_D2H, "download");
}

int multi_kernel_in_use(MultiSim* m, int32_t* variant, int64_t* workspace_bytes) {
  const Shard& s = m->sh[0];
  *workspace_bytes = s.workspace ? s.workspace_bytes : 0;
  nbody_mi355x_step_args a = base_args(*m, s, 0);
  if (m->sym) {
    a.acc_out = s.acc_all;
    a.variant = m->sym_variant;
    a.workspace = s.workspace;
    a.workspace_bytes = s.workspace_bytes;
  } else {
    a.j0_begin = 0;
    a.j0_end = static_cast<int64_t>(m->P) * m->S;
    a.variant = is_sym_variant(m->cfg.variant) ? 0 : m->cfg.variant;
  }
  return nbody_mi355x_step_variant(&a, variant);
}

int multi_energy(MultiSim* m, double* ke, double* pe) {
  const int64_t eblocks = m->S / NBODY_MI355X_TILE;
  std::vector<double> h(2 * static_cast<size_t>(m->P), 0.0);
  for (int r = 0; r < m->P; r++) {
    Shard& s = m->sh[r];
    M_TRY(hipSetDevice(s.device), NBODY_MI355X_ERR_NO_DEVICE, "hipSetDevice");
    double* result = s.scratch + 2 * eblocks;
    const int rc = nbody_mi355x_energy_f64(s.pos[m->cur], s.mass, s.vel, m->S, m->P, r, s.count, m->cfg.G, s.scratch,
                                           result, s.stream);
    if (rc != NBODY_MI355X_OK) return rc;
    M_TRY(hipMemcpyAsync(&h[2 * r], result, 2 * sizeof(double), hipMemcpyDeviceToHost, s.stream), NBODY_MI355X_ERR_D2H,
          "copy energy");
  }
  const int rc = sync_all(*m, NBODY_MI355X_ERR_D2H, "energy");
  if (rc != NBODY_MI355X_OK) return rc;
  *ke = *pe = 0.0;
  for (int r = 0; r < m->P; r++) {  // shard order: deterministic
    *ke += h[2 * r];
    *pe += h[2 * r + 1];
  }
  return NBODY_MI355X_OK;
}

}  // namespace nbody_internal
