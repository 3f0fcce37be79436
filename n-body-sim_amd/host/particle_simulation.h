// particle_simulation.h -- host-side mirror of the reference's simulation API with the
// MI355X entry point added.
//
// Same class template, constructor, SetParticle / SetMissingParticlesInBatch, Simulate*,
// Write*Files and PrintAverageStepTime as the reference's src/particle_simulation.h
// (ref: particle_simulation.h:78-753), so that a scene file written against the reference
// compiles against this header unchanged; the one new member is
//
//     std::string SimulateMI355X(int device = 0, bool fp32_forces = false,
//                                float_t softening_squared = 0, int integrator = NBODY_MI355X_EULER);
//     std::string SimulateMI355X(const std::vector<int>& devices, ...same options...);
//
// the sibling of the reference's SimulateGPU(int blocksize) (ref: particle_simulation.h:559,
// particle_simulation_GPU.cu:277-303).  It hands the particle records to
// libnbody_mi355x.so through the C ABI of include/nbody_mi355x.h and fills the same
// out_put_data_ snapshots.  This header is compiled by g++ and never sees a HIP header.
//
// The implementation is new: one state block in the reference's record layouts, a
// persistent worker pool instead of a thread spawn per small step (the reference's own
// complaint, ref: particle_simulation.h:73-77), an AVX2 path that is safe for
// N % 4 != 0 (the reference writes out of bounds there, ref: :514-540), and read
// access to the results (the reference has none, ref: :711-749).  The arithmetic of
// SimulateCPU()/SimulateAVX2() follows the reference expression by expression, which is
// what the parity tests check against the golden vectors.
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "nbody_mi355x.h"
#include "timer.h"

// ---- record layouts: identical to the reference's (ref: particle_simulation.h:20-58);
// they are the data contract of the boundary.
template <typename T>
struct Particle {
  T r[3];
  T v[3];
};

template <typename T>
struct DeviceOutputParticle {
  T r[3];
  T v[3];
  T a[3];
};

template <typename T>
struct OutPutParticle {
  T r[3];
  T v[3];
  T a[3];
  T m;
  T t;
};

template <typename T, int SIMD_float_t_width>
struct alignas(32) ParticleBatch {
  T r_x[SIMD_float_t_width];
  T r_y[SIMD_float_t_width];
  T r_z[SIMD_float_t_width];
  T v_x[SIMD_float_t_width];
  T v_y[SIMD_float_t_width];
  T v_z[SIMD_float_t_width];
};

template <typename T, int SIMD_float_t_width>
struct alignas(32) ParticleMassBatch {
  T m[SIMD_float_t_width];
};

namespace nbody_host_detail {

// Fork/join over a fixed set of workers that live as long as the simulation object.
class WorkerPool {
 public:
  explicit WorkerPool(int workers) : n_(workers < 1 ? 1 : workers) {
    for (int id = 1; id < n_; id++) threads_.emplace_back([this, id] { Loop(id); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
      epoch_++;
    }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int size() const { return n_; }
  // runs job(id) for id in [0, size()), id 0 on the calling thread; returns when all are done
  void Run(const std::function<void(int)>& job) {
    if (n_ == 1) { job(0); return; }
    {
      std::lock_guard<std::mutex> g(mu_);
      job_ = &job;
      pending_ = n_ - 1;
      epoch_++;
    }
    wake_.notify_all();
    job(0);
    std::unique_lock<std::mutex> g(mu_);
    done_.wait(g, [this] { return pending_ == 0; });
  }

 private:
  void Loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> g(mu_);
        wake_.wait(g, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (quit_) return;
        job = job_;
      }
      (*job)(id);
      {
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable wake_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int pending_ = 0;
  uint64_t epoch_ = 0;
  bool quit_ = false;
};

// contiguous share of `total` items for worker `id` of `parts`: equal shares, the
// remainder one each to the low ids (ref: particle_simulation.h:98-123, "5 particles,
// 3 threads -> 2,2,1")
inline void Share(int total, int parts, int id, int* first, int* count) {
  const int base = total / parts, extra = total % parts;
  *count = base + (id < extra ? 1 : 0);
  *first = id * base + (id < extra ? id : extra);
}

// ---- AVX-512 building blocks for SimulateAVX512() (the reference's own TODO, ref:
// particle_simulation.h:430-431).  Compiled for avx512f per function, so the header still
// builds with the reference's flags (-mavx2) and the binary still runs on AVX2-only hosts.
#define NBODY_HOST_AVX512 __attribute__((target("avx512f")))
#define NBODY_HOST_AVX512_INLINE __attribute__((target("avx512f"), always_inline)) static inline

template <typename T>
struct Vec512;
template <>
struct Vec512<double> {
  using reg = __m512d;
  static constexpr int kLanes = 8;
  NBODY_HOST_AVX512_INLINE reg Set1(double v) { return _mm512_set1_pd(v); }
  NBODY_HOST_AVX512_INLINE reg Load(const double* p) { return _mm512_load_pd(p); }
  NBODY_HOST_AVX512_INLINE reg Sub(reg a, reg b) { return _mm512_sub_pd(a, b); }
  NBODY_HOST_AVX512_INLINE reg Mul(reg a, reg b) { return _mm512_mul_pd(a, b); }
  NBODY_HOST_AVX512_INLINE reg Fma(reg a, reg b, reg c) { return _mm512_fmadd_pd(a, b, c); }
  // m / (r2 * sqrt(r2)) in the lanes with r2 > 0, exactly 0 in the others: the reference's
  // "skip pairs with r^2 == 0" (ref: :269, :427-460) without a branch
  NBODY_HOST_AVX512_INLINE reg Weight(reg m, reg r2) {
    const __mmask8 k = _mm512_cmp_pd_mask(r2, _mm512_setzero_pd(), _CMP_GT_OQ);
    return _mm512_maskz_div_pd(k, m, _mm512_mul_pd(r2, _mm512_sqrt_pd(r2)));
  }
  NBODY_HOST_AVX512_INLINE double Sum(reg a) { return _mm512_reduce_add_pd(a); }
};
template <>
struct Vec512<float> {
  using reg = __m512;
  static constexpr int kLanes = 16;
  NBODY_HOST_AVX512_INLINE reg Set1(float v) { return _mm512_set1_ps(v); }
  NBODY_HOST_AVX512_INLINE reg Load(const float* p) { return _mm512_load_ps(p); }
  NBODY_HOST_AVX512_INLINE reg Sub(reg a, reg b) { return _mm512_sub_ps(a, b); }
  NBODY_HOST_AVX512_INLINE reg Mul(reg a, reg b) { return _mm512_mul_ps(a, b); }
  NBODY_HOST_AVX512_INLINE reg Fma(reg a, reg b, reg c) { return _mm512_fmadd_ps(a, b, c); }
  NBODY_HOST_AVX512_INLINE reg Weight(reg m, reg r2) {
    const __mmask16 k = _mm512_cmp_ps_mask(r2, _mm512_setzero_ps(), _CMP_GT_OQ);
    return _mm512_maskz_div_ps(k, m, _mm512_mul_ps(r2, _mm512_sqrt_ps(r2)));
  }
  NBODY_HOST_AVX512_INLINE float Sum(reg a) { return _mm512_reduce_add_ps(a); }
};

// 64-byte aligned array of T, zero-initialised
template <typename T>
struct AlignedArray {
  explicit AlignedArray(size_t n = 0) { Reset(n); }
  AlignedArray(const AlignedArray&) = delete;
  AlignedArray& operator=(const AlignedArray&) = delete;
  ~AlignedArray() { std::free(p_); }
  void Reset(size_t n) {
    std::free(p_);
    p_ = nullptr;
    if (n == 0) return;
    const size_t bytes = (n * sizeof(T) + 63) / 64 * 64;
    p_ = static_cast<T*>(std::aligned_alloc(64, bytes));
    if (!p_) throw std::bad_alloc();
    std::memset(p_, 0, bytes);
  }
  T* data() { return p_; }
  const T* data() const { return p_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }

 private:
  T* p_ = nullptr;
};

}  // namespace nbody_host_detail

template <typename float_t, int SIMD_float_t_width>
class ParticleSimulation {
  using Batch = ParticleBatch<float_t, SIMD_float_t_width>;
  using MassBatch = ParticleMassBatch<float_t, SIMD_float_t_width>;

 public:
  // ref: particle_simulation.h:81-145
  ParticleSimulation(int num_particles, int64_t num_big_steps, int64_t num_substeps_per_big_step,
                     float_t stepsize, int num_threads)
      : num_particles_(num_particles),
        num_big_steps_(num_big_steps),
        num_substeps_per_big_step_(num_substeps_per_big_step),
        stepsize_(stepsize),
        num_threads_(num_threads < 1 ? 1 : num_threads),
        num_batches_((num_particles + SIMD_float_t_width - 1) / SIMD_float_t_width),
        state_{std::vector<Particle<float_t>>(num_particles), std::vector<Particle<float_t>>(num_particles)},
        mass_(num_particles),
        batch_mass_(new MassBatch[num_batches_]()),
        out_put_data_(num_big_steps + 1, std::vector<OutPutParticle<float_t>>(num_particles)),
        execution_time_(num_big_steps, 0.0),
        pool_(num_threads_) {
    batches_[0].reset(new Batch[num_batches_]());  // value-initialised: padding lanes are 0
    batches_[1].reset(new Batch[num_batches_]());
  }

  // ref: particle_simulation.h:148-174 -- one particle into both layouts
  void SetParticle(Particle<float_t> particle, float_t mass) {
    const int id = particle_set_counter_++;
    state_[cur_][id] = particle;
    mass_[id] = mass;
    Batch& b = batches_[bcur_][id / SIMD_float_t_width];
    const int lane = id % SIMD_float_t_width;
    b.r_x[lane] = particle.r[0]; b.r_y[lane] = particle.r[1]; b.r_z[lane] = particle.r[2];
    b.v_x[lane] = particle.v[0]; b.v_y[lane] = particle.v[1]; b.v_z[lane] = particle.v[2];
    batch_mass_[id / SIMD_float_t_width].m[lane] = mass;
  }

  // ref: particle_simulation.h:179-192.  Lanes of the last batch that hold no particle
  // are made massless bodies at the origin.  (The reference only does this while
  // particles are still missing; the buffers here start zeroed, so calling it after all
  // particles are set is a harmless no-op, as it is there.)
  void SetMissingParticlesInBatch() {
    if (particle_set_counter_ >= num_particles_ || num_batches_ == 0) return;
    Batch& b = batches_[bcur_][num_batches_ - 1];
    for (int lane = particle_set_counter_ % SIMD_float_t_width; lane < SIMD_float_t_width; lane++) {
      b.r_x[lane] = b.r_y[lane] = b.r_z[lane] = 0;
      b.v_x[lane] = b.v_y[lane] = b.v_z[lane] = 0;
      batch_mass_[num_batches_ - 1].m[lane] = 0;
    }
  }

  // ref: particle_simulation.h:197-212
  void SimulateCPU() {
    RunHost([this](bool last) { SmallStepAllWorkers(last); });
  }

  // ref: particle_simulation.h:329-344.  <double,4> only, like the reference's (its
  // kernel is hard-wired to __m256d).
  void SimulateAVX2() {
    static_assert(std::is_same<float_t, double>::value && SIMD_float_t_width == 4,
                  "SimulateAVX2 is implemented for ParticleSimulation<double,4>");
    RunHost([this](bool last) { SmallStepAllWorkersAVX2(last); });
    // keep the flat records in step with the batched ones
    for (int i = 0; i < num_particles_; i++) {
      const Batch& b = batches_[bcur_][i / 4];
      const int l = i % 4;
      state_[cur_][i] = Particle<float_t>{{b.r_x[l], b.r_y[l], b.r_z[l]}, {b.v_x[l], b.v_y[l], b.v_z[l]}};
    }
  }

  // No counterpart in the reference (its TODO, ref: :430-431): the same small step with
  // 512-bit vectors -- 8 doubles or 16 floats of j bodies per instruction, for double AND
  // float objects of any batch width (the reference's SIMD path is <double,4> only), 4 i
  // bodies per pass over the j bodies so that each load is used four times, the
  // r^2 == 0 rule as a lane mask instead of a scalar fallback, any N (no N % 4 caveat).
  // Arithmetic per pair as SimulateAVX2 (r^2 by two FMAs on dx*dx, m / (r^2 sqrt(r^2)),
  // FMA accumulation, G * sum); the partial sums are 8 / 16 instead of 4, so results agree
  // with SimulateAVX2 / SimulateCPU to rounding (~1e-13 after 100 steps), not bit for bit.
  // Returns false (and does nothing) if the CPU has no AVX-512F.
  bool SimulateAVX512() {
    if (!__builtin_cpu_supports("avx512f")) return false;
    using V = nbody_host_detail::Vec512<float_t>;
    const size_t padded = (static_cast<size_t>(num_particles_) + V::kLanes - 1) / V::kLanes * V::kLanes;
    for (int b = 0; b < 2; b++)
      for (int c = 0; c < 3; c++) soa_r_[b][c].Reset(padded);  // padding: massless bodies at the origin
    for (int c = 0; c < 3; c++) soa_v_[c].Reset(padded);
    soa_m_.Reset(padded);
    scur_ = 0;
    for (int i = 0; i < num_particles_; i++) {
      for (int c = 0; c < 3; c++) {
        soa_r_[0][c][i] = state_[cur_][i].r[c];
        soa_v_[c][i] = state_[cur_][i].v[c];
      }
      soa_m_[i] = mass_[i];
    }
    RunHost([this](bool last) {
      pool_.Run([this, last](int id) { SmallStepAVX512(id, last); });
      scur_ ^= 1;
    });
    // keep the flat and the batched records in step with the vector ones
    for (int i = 0; i < num_particles_; i++) {
      Particle<float_t>& q = state_[cur_][i];
      for (int c = 0; c < 3; c++) {
        q.r[c] = soa_r_[scur_][c][i];
        q.v[c] = soa_v_[c][i];
      }
      Batch& b = batches_[bcur_][i / SIMD_float_t_width];
      const int l = i % SIMD_float_t_width;
      b.r_x[l] = q.r[0]; b.r_y[l] = q.r[1]; b.r_z[l] = q.r[2];
      b.v_x[l] = q.v[0]; b.v_y[l] = q.v[1]; b.v_z[l] = q.v[2];
    }
    return true;
  }

  // The reference's accelerator entry point by its own name (ref: particle_simulation.h:559,
  // defined in particle_simulation_GPU.cu:277-303, which a GPU scene #includes instead of this
  // header, saturn_ring.cu:6-8): same call shape, runs the MI355X path.  `blocksize` was the
  // CUDA launch's block size (.cu:147); launch shapes are the library's own business here.
  std::string SimulateGPU(int blocksize = 32) {
    (void)blocksize;
    return SimulateMI355X();
  }

  // The MI355X path.  Observable effects are those of the reference's SimulateGPU()
  // (ref: particle_simulation_GPU.cu:277-303): out_put_data_[0] = initial data with
  // a = 0, t = 0; out_put_data_[k] = state after big step k with a = G*sum, m, and
  // t = stepsize * substeps * k (ref: .cu:239-266); execution_time_ per big step;
  // progress print per big step; a "Device execution time" line; a status string.
  // integrator: NBODY_MI355X_EULER (the default: the reference's forward Euler),
  // NBODY_MI355X_KICK_DRIFT (symplectic Euler, same cost) or NBODY_MI355X_KDK (kick-drift-kick
  // leapfrog, second order, still one force evaluation per small step).
  std::string SimulateMI355X(int device = 0, bool fp32_forces = false, float_t softening_squared = 0,
                             int integrator = NBODY_MI355X_EULER) {
    return SimulateMI355X(std::vector<int>{device}, fp32_forces, softening_squared, integrator);
  }

  // The same over several GPUs driven by this one host thread: devices[k] owns the k-th
  // contiguous i-range of bodies, the partition of the CPU paths' threads (ref:
  // particle_simulation.h:98-110).  An ordinal may repeat (several shards on one GPU).
  std::string SimulateMI355X(const std::vector<int>& devices, bool fp32_forces = false,
                             float_t softening_squared = 0, int integrator = NBODY_MI355X_EULER) {
    IntervallTimer device_execution_timer;
    nbody_mi355x_config cfg{};
    cfg.num_particles = num_particles_;
    cfg.stepsize = static_cast<double>(stepsize_);
    cfg.G = static_cast<double>(G_);
    cfg.eps2 = static_cast<double>(softening_squared);
    cfg.precision = mi355x_precision_ >= 0 ? mi355x_precision_ : (fp32_forces ? NBODY_MI355X_FP32 : NBODY_MI355X_FP64);
    cfg.device = devices.empty() ? 0 : devices[0];
    cfg.host_is_float = std::is_same<float_t, float>::value ? 1 : 0;
    cfg.integrator = integrator;
    nbody_mi355x_sim* sim = nullptr;
    const std::vector<int32_t> devs(devices.begin(), devices.end());
    int rc = devs.size() > 1 ? nbody_mi355x_create_multi(&cfg, devs.data(), static_cast<int32_t>(devs.size()), &sim)
                             : nbody_mi355x_create(&cfg, &sim);
    if (rc != NBODY_MI355X_OK) return Failure(rc, nullptr);
    if (devs.size() > 1 && !quiet_) {  // how the shards exchange, and the library's note if RCCL was not taken
      int32_t backend = NBODY_MI355X_EXCHANGE_NONE;
      const std::string note = nbody_mi355x_last_error();
      if (nbody_mi355x_exchange_backend(sim, &backend) == NBODY_MI355X_OK)
        std::cout << "MI355X: " << devs.size() << " shards, exchanges by "
                  << (backend == NBODY_MI355X_EXCHANGE_RCCL ? "RCCL collectives (self-check passed)" : "peer copies")
                  << (note.empty() ? "" : " -- ") << note << std::endl;
    }
    rc = nbody_mi355x_upload(sim, state_[cur_].data(), mass_.data());
    if (rc != NBODY_MI355X_OK) return Failure(rc, sim);
    {  // say so when the library had to fall back to the slower kernel family
      int32_t variant = 0;
      int64_t workspace_bytes = 0;
      if (nbody_mi355x_kernel_in_use(sim, &variant, &workspace_bytes) == NBODY_MI355X_OK && workspace_bytes == 0 &&
          num_particles_ > NBODY_MI355X_WAVE_MAX_BODIES && variant >= 1 && variant <= 6)
        std::cout << "MI355X: no room on the device for the pair-symmetric kernels' workspace; running the "
                     "one-sided kernel (variant " << variant << ", about half the rate)." << std::endl;
    }

    SaveFirstStep();
    IntervallTimer sim_timer;
    // Two page-locked snapshot buffers: while big step k+1 runs on the device, big step
    // k's records are unpacked into out_put_data_ here (the reference unpacks everything
    // after the last step, ref: .cu:239-266, holding all snapshots on the device).
    DeviceOutputParticle<float_t>* snapshot[2] = {nullptr, nullptr};
    bool pinned[2] = {false, false};
    std::vector<DeviceOutputParticle<float_t>> pageable[2];  // if page-locked memory is refused
    const int64_t snap_bytes = static_cast<int64_t>(sizeof(DeviceOutputParticle<float_t>)) * num_particles_;
    for (int b = 0; b < 2; b++) {
      void* p = nullptr;
      pinned[b] = nbody_mi355x_host_alloc(snap_bytes, &p) == NBODY_MI355X_OK;
      if (pinned[b]) {
        snapshot[b] = static_cast<DeviceOutputParticle<float_t>*>(p);
      } else {
        pageable[b].resize(num_particles_);
        snapshot[b] = pageable[b].data();
      }
    }
    auto unpin = [&]() {
      for (int b = 0; b < 2; b++)
        if (pinned[b]) { nbody_mi355x_host_free(snapshot[b]); pinned[b] = false; }
    };
    IntervallTimer big_step_timer;
    if (num_big_steps_ > 0) {
      rc = nbody_mi355x_run_async(sim, num_substeps_per_big_step_, snapshot[1]);
      if (rc != NBODY_MI355X_OK) { unpin(); return Failure(rc, sim); }
    }
    for (int64_t k = 1; k <= num_big_steps_; k++) {
      rc = nbody_mi355x_wait(sim);  // big step k is complete, its snapshot in snapshot[k & 1]
      if (rc != NBODY_MI355X_OK) { unpin(); return Failure(rc, sim); }
      execution_time_[k - 1] = static_cast<double>(big_step_timer.getTimeInMicroseconds()) * 1.0e-06;
      big_step_timer = IntervallTimer();
      if (k < num_big_steps_) {
        rc = nbody_mi355x_run_async(sim, num_substeps_per_big_step_, snapshot[(k + 1) & 1]);
        if (rc != NBODY_MI355X_OK) { unpin(); return Failure(rc, sim); }
      }
      const DeviceOutputParticle<float_t>* snap = snapshot[k & 1];
      for (int i = 0; i < num_particles_; i++) {
        OutPutParticle<float_t>& o = out_put_data_[k][i];
        for (int c = 0; c < 3; c++) {
          o.r[c] = snap[i].r[c];
          o.v[c] = snap[i].v[c];
          o.a[c] = snap[i].a[c];
        }
        o.m = mass_[i];
        o.t = stepsize_ * num_substeps_per_big_step_ * k;
      }
      PrintSimulationProgress(static_cast<int>(k));
    }
    total_execution_time_ = sim_timer.getTimeInSeconds();
    // leave the object holding the final state, as the CPU paths do
    const DeviceOutputParticle<float_t>* last = snapshot[num_big_steps_ & 1];
    for (int i = 0; i < num_particles_ && num_big_steps_ > 0; i++)
      for (int c = 0; c < 3; c++) {
        state_[cur_][i].r[c] = last[i].r[c];
        state_[cur_][i].v[c] = last[i].v[c];
      }
    unpin();
    rc = nbody_mi355x_destroy(sim);
    if (rc != NBODY_MI355X_OK) return Failure(rc, nullptr);
    std::cout << "Device execution time: " << device_execution_timer.getTimeInSeconds() << "s" << std::endl;
    return {"Simulation on MI355X succesfull."};
  }

  // ref: particle_simulation.h:620-651 -- one file per particle, one row per snapshot:
  //   t  x y z  vx vy vz  ax ay az  m     (default ostream precision, as the reference)
  void WriteParticleFiles(std::string destination_directory) {
    for (int p = 0; p < num_particles_; p++) {
      std::ofstream f(FileName(destination_directory, "N_Body_Particle_", p));
      if (!f.is_open()) { std::cout << "Unable to open file \n"; continue; }
      for (int64_t k = 0; k <= num_big_steps_; k++) WriteRow(f, k, out_put_data_[k][p]);
    }
  }

  // ref: particle_simulation.h:655-686 -- one file per snapshot, one row per particle.
  // The reference stops one short and never writes the final snapshot (:656); it is
  // written here (files 0 .. num_big_steps).
  void WriteTimestepFiles(std::string destination_directory) {
    for (int64_t k = 0; k <= num_big_steps_; k++) {
      std::ofstream f(FileName(destination_directory, "N_Body_Timestep_", static_cast<int>(k)));
      if (!f.is_open()) { std::cout << "Unable to open file \n"; continue; }
      for (int p = 0; p < num_particles_; p++) WriteRow(f, k, out_put_data_[k][p]);
    }
  }

  // ref: particle_simulation.h:690-709
  void PrintAverageStepTime() {
    std::cout << "Total simulation time: " << total_execution_time_ / 60 << " min "
              << total_execution_time_ % 60 << " s" << std::endl;
    double mean = 0;
    for (double t : execution_time_) mean += t;
    mean /= static_cast<double>(num_big_steps_);
    double var = 0;
    for (double t : execution_time_) var += (t - mean) * (t - mean);
    std::cout << "Average execution time for a big step: " << mean << " +- "
              << std::sqrt(var / static_cast<double>(num_big_steps_ - 1)) << std::endl;
  }

  // ---- read access (not in the reference) ----
  const std::vector<std::vector<OutPutParticle<float_t>>>& OutputData() const { return out_put_data_; }
  const std::vector<double>& BigStepSeconds() const { return execution_time_; }
  const Particle<float_t>* CurrentParticles() const { return state_[cur_].data(); }
  const float_t* Masses() const { return mass_.data(); }
  int NumParticles() const { return num_particles_; }
  void SetQuiet(bool quiet) { quiet_ = quiet; }
  // The arithmetic tier SimulateMI355X() asks the library for, overriding its fp32_forces flag:
  // NBODY_MI355X_FP64 (the parity path), _FP32 or _FP64_FAST (include/nbody_mi355x.h); -1: the flag decides.
  void SetMI355XPrecision(int tier) { mi355x_precision_ = tier; }

 private:
  static constexpr float_t G_ = 6.7430e-11;  // ref: particle_simulation.h:713 (sic)

  // big-step loop shared by the two host paths (ref: :197-219, :329-352)
  template <typename SmallStep>
  void RunHost(SmallStep small_step) {
    IntervallTimer sim_timer;
    SaveFirstStep();
    for (int64_t k = 0; k < num_big_steps_; k++) {
      IntervallTimer big_step_timer;
      current_big_step_++;
      for (int64_t s = 0; s + 1 < num_substeps_per_big_step_; s++) small_step(false);
      small_step(true);
      execution_time_[k] = static_cast<double>(big_step_timer.getTimeInMicroseconds()) * 1.0e-06;
      PrintSimulationProgress(static_cast<int>(current_big_step_));
    }
    total_execution_time_ = sim_timer.getTimeInSeconds();
  }

  // ref: particle_simulation.h:222-240
  void SmallStepAllWorkers(bool is_last_small_step) {
    pool_.Run([this, is_last_small_step](int id) { SmallStep(id, is_last_small_step); });
    cur_ ^= 1;  // current is read-only and next write-only during a step: swap by index
  }

  // ref: particle_simulation.h:248-322, expression by expression
  void SmallStep(int worker, bool is_last_small_step) {
    int first, count;
    nbody_host_detail::Share(num_particles_, num_threads_, worker, &first, &count);
    const Particle<float_t>* cur = state_[cur_].data();
    Particle<float_t>* next = state_[cur_ ^ 1].data();
    for (int i = first; i < first + count; i++) {
      float_t a[3] = {0, 0, 0};
      for (int j = 0; j < num_particles_; j++) {
        float_t delta_x = cur[j].r[0] - cur[i].r[0];
        float_t delta_y = cur[j].r[1] - cur[i].r[1];
        float_t delta_z = cur[j].r[2] - cur[i].r[2];
        float_t r_squared = (delta_x * delta_x) + (delta_y * delta_y) + (delta_z * delta_z);
        if (r_squared != 0) {
          // unqualified sqrt as in the reference (ref: :270): for float_t = float this
          // is ::sqrt(double) under g++/libstdc++, and the golden vectors pin that
          float_t tmp = mass_[j] / (r_squared * sqrt(r_squared));
          a[0] += (tmp * delta_x);
          a[1] += (tmp * delta_y);
          a[2] += (tmp * delta_z);
        }
      }
      for (int c = 0; c < 3; c++) {
        next[i].v[c] = cur[i].v[c] + (stepsize_ * a[c] * G_);
        next[i].r[c] = cur[i].r[c] + (stepsize_ * cur[i].v[c]);
      }
      if (is_last_small_step) {
        OutPutParticle<float_t>& o = out_put_data_[current_big_step_][i];
        for (int c = 0; c < 3; c++) {
          o.r[c] = next[i].r[c];
          o.v[c] = next[i].v[c];
          o.a[c] = a[c] * G_;
        }
        o.m = mass_[i];
        o.t = stepsize_ * current_big_step_ * num_substeps_per_big_step_;
      }
    }
  }

  // ref: particle_simulation.h:355-374
  void SmallStepAllWorkersAVX2(bool is_last_small_step) {
    pool_.Run([this, is_last_small_step](int id) { SmallStepAVX2(id, is_last_small_step); });
    bcur_ ^= 1;
  }

  // ref: particle_simulation.h:382-544 (4 j bodies per __m256d; batches with a zero
  // distance in any lane take the per-lane path)
  void SmallStepAVX2(int worker, bool is_last_small_step) {
    int first, count;
    nbody_host_detail::Share(num_batches_, num_threads_, worker, &first, &count);
    const Batch* cur = batches_[bcur_].get();
    Batch* next = batches_[bcur_ ^ 1].get();
    const __m256d zeros = _mm256_setzero_pd();
    for (int b = first; b < first + count; b++) {
      for (int lane = 0; lane < 4; lane++) {
        const __m256d cx = _mm256_set1_pd(cur[b].r_x[lane]);
        const __m256d cy = _mm256_set1_pd(cur[b].r_y[lane]);
        const __m256d cz = _mm256_set1_pd(cur[b].r_z[lane]);
        __m256d ax = zeros, ay = zeros, az = zeros;
        for (int o = 0; o < num_batches_; o++) {
          const __m256d mass = _mm256_load_pd(batch_mass_[o].m);
          const __m256d dx = _mm256_sub_pd(_mm256_load_pd(cur[o].r_x), cx);
          const __m256d dy = _mm256_sub_pd(_mm256_load_pd(cur[o].r_y), cy);
          const __m256d dz = _mm256_sub_pd(_mm256_load_pd(cur[o].r_z), cz);
          const __m256d r2 = _mm256_fmadd_pd(dz, dz, _mm256_fmadd_pd(dy, dy, _mm256_mul_pd(dx, dx)));
          if (_mm256_movemask_pd(_mm256_cmp_pd(r2, zeros, _CMP_GT_OQ)) == 0xf) {
            const __m256d tmp = _mm256_div_pd(mass, _mm256_mul_pd(r2, _mm256_sqrt_pd(r2)));
            ax = _mm256_fmadd_pd(tmp, dx, ax);
            ay = _mm256_fmadd_pd(tmp, dy, ay);
            az = _mm256_fmadd_pd(tmp, dz, az);
          } else {
            alignas(32) double r[4], m[4], d[3][4], acc[3][4];
            _mm256_store_pd(r, r2); _mm256_store_pd(m, mass);
            _mm256_store_pd(d[0], dx); _mm256_store_pd(d[1], dy); _mm256_store_pd(d[2], dz);
            _mm256_store_pd(acc[0], ax); _mm256_store_pd(acc[1], ay); _mm256_store_pd(acc[2], az);
            for (int j = 0; j < 4; j++) {
              if (r[j] != 0) {
                const double tmp1 = m[j] / (r[j] * sqrt(r[j]));
                for (int c = 0; c < 3; c++) acc[c][j] = tmp1 * d[c][j] + acc[c][j];
              }
            }
            ax = _mm256_load_pd(acc[0]); ay = _mm256_load_pd(acc[1]); az = _mm256_load_pd(acc[2]);
          }
        }
        // hadd, then element 0 + element 2, then * G (ref: :466-475)
        alignas(32) double h[4];
        double fa[3];
        _mm256_store_pd(h, _mm256_hadd_pd(ax, ax)); fa[0] = G_ * (h[0] + h[2]);
        _mm256_store_pd(h, _mm256_hadd_pd(ay, ay)); fa[1] = G_ * (h[0] + h[2]);
        _mm256_store_pd(h, _mm256_hadd_pd(az, az)); fa[2] = G_ * (h[0] + h[2]);
        next[b].v_x[lane] = cur[b].v_x[lane] + (stepsize_ * fa[0]);
        next[b].v_y[lane] = cur[b].v_y[lane] + (stepsize_ * fa[1]);
        next[b].v_z[lane] = cur[b].v_z[lane] + (stepsize_ * fa[2]);
        next[b].r_x[lane] = cur[b].r_x[lane] + (stepsize_ * cur[b].v_x[lane]);
        next[b].r_y[lane] = cur[b].r_y[lane] + (stepsize_ * cur[b].v_y[lane]);
        next[b].r_z[lane] = cur[b].r_z[lane] + (stepsize_ * cur[b].v_z[lane]);
        const int id = b * 4 + lane;
        if (is_last_small_step && id < num_particles_) {  // padding lanes own no output row
          OutPutParticle<float_t>& out = out_put_data_[current_big_step_][id];
          out.r[0] = next[b].r_x[lane]; out.r[1] = next[b].r_y[lane]; out.r[2] = next[b].r_z[lane];
          out.v[0] = next[b].v_x[lane]; out.v[1] = next[b].v_y[lane]; out.v[2] = next[b].v_z[lane];
          out.a[0] = fa[0]; out.a[1] = fa[1]; out.a[2] = fa[2];
          out.m = mass_[id];
          out.t = stepsize_ * current_big_step_ * num_substeps_per_big_step_;
        }
      }
    }
  }

  // One worker's share of a small step with 512-bit vectors (see SimulateAVX512).
  NBODY_HOST_AVX512 void SmallStepAVX512(int worker, bool is_last_small_step) {
    using V = nbody_host_detail::Vec512<float_t>;
    using reg = typename V::reg;
    constexpr int kRows = 4;  // i bodies per pass over the j bodies
    int first, count;
    nbody_host_detail::Share(num_particles_, num_threads_, worker, &first, &count);
    const size_t padded = (static_cast<size_t>(num_particles_) + V::kLanes - 1) / V::kLanes * V::kLanes;
    const float_t* x = soa_r_[scur_][0].data();
    const float_t* y = soa_r_[scur_][1].data();
    const float_t* z = soa_r_[scur_][2].data();
    const float_t* m = soa_m_.data();
    for (int i0 = first; i0 < first + count; i0 += kRows) {
      const int rows = std::min(kRows, first + count - i0);
      reg cx[kRows], cy[kRows], cz[kRows], ax[kRows], ay[kRows], az[kRows];
      for (int k = 0; k < kRows; k++) {
        const int i = i0 + std::min(k, rows - 1);  // a short last pass repeats its last body
        cx[k] = V::Set1(x[i]); cy[k] = V::Set1(y[i]); cz[k] = V::Set1(z[i]);
        ax[k] = ay[k] = az[k] = V::Set1(0);
      }
      for (size_t j = 0; j < padded; j += V::kLanes) {
        const reg xj = V::Load(x + j), yj = V::Load(y + j), zj = V::Load(z + j), mj = V::Load(m + j);
        for (int k = 0; k < kRows; k++) {
          const reg dx = V::Sub(xj, cx[k]), dy = V::Sub(yj, cy[k]), dz = V::Sub(zj, cz[k]);
          const reg r2 = V::Fma(dz, dz, V::Fma(dy, dy, V::Mul(dx, dx)));
          const reg w = V::Weight(mj, r2);
          ax[k] = V::Fma(w, dx, ax[k]);
          ay[k] = V::Fma(w, dy, ay[k]);
          az[k] = V::Fma(w, dz, az[k]);
        }
      }
      for (int k = 0; k < rows; k++) {
        const int i = i0 + k;
        const float_t fa[3] = {G_ * V::Sum(ax[k]), G_ * V::Sum(ay[k]), G_ * V::Sum(az[k])};
        for (int c = 0; c < 3; c++) {
          const float_t v_old = soa_v_[c][i];
          soa_v_[c][i] = v_old + (stepsize_ * fa[c]);
          soa_r_[scur_ ^ 1][c][i] = soa_r_[scur_][c][i] + (stepsize_ * v_old);
        }
        if (is_last_small_step) {
          OutPutParticle<float_t>& o = out_put_data_[current_big_step_][i];
          for (int c = 0; c < 3; c++) {
            o.r[c] = soa_r_[scur_ ^ 1][c][i];
            o.v[c] = soa_v_[c][i];
            o.a[c] = fa[c];
          }
          o.m = mass_[i];
          o.t = stepsize_ * current_big_step_ * num_substeps_per_big_step_;
        }
      }
    }
  }

  // ref: particle_simulation.h:589-599
  void SaveFirstStep() {
    for (int i = 0; i < num_particles_; i++) {
      OutPutParticle<float_t>& o = out_put_data_[0][i];
      for (int c = 0; c < 3; c++) {
        o.r[c] = state_[cur_][i].r[c];
        o.v[c] = state_[cur_][i].v[c];
        o.a[c] = 0;
      }
      o.m = mass_[i];
      o.t = 0;
    }
  }

  // ref: particle_simulation.h:601-616 (same three lines per big step)
  void PrintSimulationProgress(int current_big_step) {
    if (quiet_) return;
    double mean = 0;
    for (int k = 0; k < current_big_step; k++) mean += execution_time_[k];
    mean /= current_big_step;
    const int remaining = static_cast<int>((num_big_steps_ - current_big_step) * mean);
    std::cout << (static_cast<float_t>(current_big_step) / num_big_steps_) * 100 << "% done\nThis big step toke: "
              << execution_time_[current_big_step - 1] << "s\n"
              << "Approximate remaining time: " << remaining / 3600 << "h " << (remaining % 3600) / 60
              << "min " << remaining % 60 << "s " << std::endl;
  }

  std::string Failure(int rc, nbody_mi355x_sim* sim) {
    const std::string detail = nbody_mi355x_last_error();
    if (sim) nbody_mi355x_destroy(sim);
    const char* stage = "MI355X run failed.";
    switch (rc) {  // the reference's four stage messages (ref: .cu:280-295) + the new ones
      case NBODY_MI355X_ERR_ALLOC: stage = "Failed memory allocation on device."; break;
      case NBODY_MI355X_ERR_H2D: stage = "Failed memory copy to device."; break;
      case NBODY_MI355X_ERR_D2H: stage = "Failed to copy results to host."; break;
      case NBODY_MI355X_ERR_FREE: stage = "Failed to free device memory."; break;
      case NBODY_MI355X_ERR_LAUNCH: stage = "Kernel launch or execution failed."; break;
      case NBODY_MI355X_ERR_NO_DEVICE: stage = "No usable MI355X device."; break;
      case NBODY_MI355X_ERR_INVALID: stage = "Invalid simulation parameters."; break;
      default: break;
    }
    return std::string(stage) + " (" + detail + ")";
  }

  static std::string FileName(const std::string& dir, const char* stem, int id) {
    std::ostringstream s;
    s << dir << stem << std::setw(5) << std::setfill('0') << id << ".txt";
    return s.str();
  }

  void WriteRow(std::ofstream& f, int64_t k, const OutPutParticle<float_t>& o) {
    f << (k * num_substeps_per_big_step_ * stepsize_) << "  " << o.r[0] << " " << o.r[1] << " " << o.r[2]
      << "  " << o.v[0] << " " << o.v[1] << " " << o.v[2] << "  " << o.a[0] << " " << o.a[1] << " "
      << o.a[2] << "  " << o.m << "\n";
  }

  int num_particles_;
  int64_t num_big_steps_;
  int64_t num_substeps_per_big_step_;
  float_t stepsize_;
  int num_threads_;
  int num_batches_;
  int64_t current_big_step_ = 0;
  int particle_set_counter_ = 0;
  int cur_ = 0, bcur_ = 0;  // which of the two buffers is "current"
  std::vector<Particle<float_t>> state_[2];
  std::vector<float_t> mass_;
  std::unique_ptr<Batch[]> batches_[2];
  std::unique_ptr<MassBatch[]> batch_mass_;
  // SimulateAVX512(): plain coordinate runs, 64-byte aligned, padded to whole vectors
  nbody_host_detail::AlignedArray<float_t> soa_r_[2][3], soa_v_[3], soa_m_;
  int scur_ = 0;
  std::vector<std::vector<OutPutParticle<float_t>>> out_put_data_;
  std::vector<double> execution_time_;
  int64_t total_execution_time_ = 0;
  bool quiet_ = false;
  int mi355x_precision_ = -1;
  nbody_host_detail::WorkerPool pool_;
};
