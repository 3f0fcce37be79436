// ubench_lds.hip -- per-CU throughput of the LDS instructions the pair-symmetric kernel's
// j side can be built from: return-less ds_add_f64 / ds_add_f32 (conflict-free, one entry per
// lane), ds_read_b128, ds_read_b64 + ds_write_b64 (read-modify-write by hand).  Reported in
// shader cycles per wave instruction per CU, calibrated against a v_fma_f64 chain
// (4 cycles per wave instruction per SIMD) measured in the same process.
// Build: make -C tools.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 1 << 16;
constexpr int UN = 8;

enum Op { FMA64, ADD_F64, ADD_F32, READ_B128, RMW_B64, ADD_F64_X3_READ_X2, NOPS, MIX_NONE, MIX_ADD, MIX_READ, MIX_DPP };
static const char* kNames[NOPS] = {"v_fma_f64 (calibration)", "ds_add_f64 (no return)", "ds_add_f32 (no return)",
                                   "ds_read_b128", "ds_read_b64 + v_add_f64 + ds_write_b64",
                                   "3 ds_add_f64 + 2 ds_read_b128 (one step of the kernel)"};
static const int kLdsPerStep[NOPS] = {1, 1, 1, 1, 2, 5};

template <int OP>
__global__ __launch_bounds__(512) void k(double* out, double a, double b) {
  __shared__ double sd[3][512 + 64];
  __shared__ double2 sq[2][512 + 64];
  const int tid = threadIdx.x;
  for (int c = 0; c < 3; c++) sd[c][tid] = 0.0;
  sq[0][tid] = sq[1][tid] = double2{a, b};
  __syncthreads();
  float* sf = reinterpret_cast<float*>(&sd[0][0]);
  double x[UN];
#pragma unroll
  for (int c = 0; c < UN; c++) x[c] = 1.0 + 1e-3 * (tid + c);
  double acc = 0.0;
  for (int it = 0; it < ITERS / UN; it++) {
#pragma unroll
    for (int u = 0; u < UN; u++) {
      if (OP != FMA64) asm volatile("" ::: "memory");  // LDS contents are "unknown" again: no hoisting
      if (OP == FMA64) x[u] = __builtin_fma(x[u], a, b);
      if (OP == ADD_F64) __hip_atomic_fetch_add(&sd[0][tid + u], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == ADD_F32) __hip_atomic_fetch_add(&sf[tid + u], (float)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == READ_B128) {
        const double2 v = sq[0][tid + u];
        acc += v.x + v.y;
      }
      if (OP == RMW_B64) {
        double* p = &sd[0][tid + u];
        *p = *p + a;
      }
      if (OP == ADD_F64_X3_READ_X2) {
        __hip_atomic_fetch_add(&sd[0][tid + u], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sd[1][tid + u], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sd[2][tid + u], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double2 v = sq[0][tid + u];
        const double2 w = sq[1][tid + u];
        acc += (v.x + w.y) + (v.y + w.x);
      }
    }
  }
  __syncthreads();
  double s = acc + sd[0][tid] + sd[1][tid] + sd[2][tid];
#pragma unroll
  for (int c = 0; c < UN; c++) s += x[c];
  if (s == 123.456) out[0] = s;
}

template <int OP>
double run(double* d_out, int threads, int wg_per_cu, int cus) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = cus * wg_per_cu;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001, 1e-9);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e6;  // ns
}

// Does issuing an LDS instruction cost the SIMD VALU issue time?  32 independent v_fma_f64
// per trip (128 cycles of VALU) alone, with 3 ds_add_f64, with 2 ds_read_b128, and with
// 6 v_mov_b32_dpp: the per-trip difference against "alone" is the instruction's cost in
// VALU cycles on a SIMD that has other work (2 waves per SIMD, as in the kernel).
constexpr int MIX_TRIPS = 1 << 18;
template <int OP>
__global__ __launch_bounds__(512) void kmix(double* out, double a, double b) {
  __shared__ double sd[3][512 + 64];
  __shared__ double2 sq[2][512 + 64];
  const int tid = threadIdx.x;
  for (int c = 0; c < 3; c++) sd[c][tid] = 0.0;
  sq[0][tid] = sq[1][tid] = double2{a, b};
  __syncthreads();
  double x[32];
#pragma unroll
  for (int c = 0; c < 32; c++) x[c] = 1.0 + 1e-3 * (tid + c);
  double acc = 0.0;
  int r0 = tid, r1 = tid + 1, r2 = tid + 2, r3 = tid + 3, r4 = tid + 4, r5 = tid + 5;
  for (int it = 0; it < MIX_TRIPS; it++) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 32; c++) x[c] = __builtin_fma(x[c], a, b);
    if (OP == MIX_ADD) {
      __hip_atomic_fetch_add(&sd[0][tid + (it & 7)], x[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sd[1][tid + (it & 7)], x[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sd[2][tid + (it & 7)], x[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (OP == MIX_READ) {
      const double2 v = sq[0][tid + (it & 7)];
      const double2 w = sq[1][tid + (it & 7)];
      acc += v.x;  // 1 extra v_add_f64 per trip
      x[3] = w.y;
    }
    if (OP == MIX_DPP) {
      r0 = __builtin_amdgcn_update_dpp(r0, r0, 0x134, 0xf, 0xf, false);
      r1 = __builtin_amdgcn_update_dpp(r1, r1, 0x134, 0xf, 0xf, false);
      r2 = __builtin_amdgcn_update_dpp(r2, r2, 0x134, 0xf, 0xf, false);
      r3 = __builtin_amdgcn_update_dpp(r3, r3, 0x134, 0xf, 0xf, false);
      r4 = __builtin_amdgcn_update_dpp(r4, r4, 0x134, 0xf, 0xf, false);
      r5 = __builtin_amdgcn_update_dpp(r5, r5, 0x134, 0xf, 0xf, false);
    }
  }
  __syncthreads();
  double s = acc + sd[0][tid] + sd[1][tid] + sd[2][tid] + (r0 + r1 + r2 + r3 + r4 + r5);
#pragma unroll
  for (int c = 0; c < 32; c++) s += x[c];
  if (s == 123.456) out[0] = s;
}

template <int OP>
double run_mix(double* d_out, int cus) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kmix<OP>, dim3(cus), dim3(512), 0, 0, d_out, 1.0000001, 1e-9);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kmix<OP>, dim3(cus), dim3(512), 0, 0, d_out, 1.0000001, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e6;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* d_out; CHECK(hipMalloc(&d_out, 64));
  for (int threads : {256, 512}) {
    const int waves = threads / 64;
    // calibration: each SIMD runs waves/4 waves of ITERS dependent-free FMAs at 4 cycles each
    const double ns_fma = run<FMA64>(d_out, threads, 1, cus);
    const double cyc_ns = (double)ITERS * (waves / 4) * 4.0 / ns_fma;
    printf("%d waves per CU: shader clock under v_fma_f64 = %.2f GHz\n", waves, cyc_ns);
    const double t[NOPS] = {ns_fma,
                            run<ADD_F64>(d_out, threads, 1, cus),
                            run<ADD_F32>(d_out, threads, 1, cus),
                            run<READ_B128>(d_out, threads, 1, cus),
                            run<RMW_B64>(d_out, threads, 1, cus),
                            run<ADD_F64_X3_READ_X2>(d_out, threads, 1, cus)};
    for (int op = 1; op < NOPS; op++) {
      const double instr_per_cu = (double)ITERS * waves * kLdsPerStep[op];
      printf("  %-58s %8.3f ms  %6.2f cycles per wave-instr per CU (at the calibrated clock)\n", kNames[op],
             t[op] * 1e-6, t[op] * cyc_ns / instr_per_cu);
    }
  }
  // 2 waves per SIMD, 32 FMAs = 128 cycles per wave trip = 256 cycles per SIMD trip
  const double t0 = run_mix<MIX_NONE>(d_out, cus);
  const double ghz = (double)MIX_TRIPS * 256.0 / t0;
  printf("VALU cost of j-side instructions next to 32 v_fma_f64 per trip (2 waves per SIMD, clock %.2f GHz):\n", ghz);
  const double ta = run_mix<MIX_ADD>(d_out, cus), tr = run_mix<MIX_READ>(d_out, cus), td = run_mix<MIX_DPP>(d_out, cus);
  printf("  alone %.3f ms; + 3 ds_add_f64: %.3f ms = %.2f cycles per ds_add_f64\n", t0 * 1e-6, ta * 1e-6,
         (ta - t0) * ghz / MIX_TRIPS / 2.0 / 3.0);
  printf("  + 2 ds_read_b128 (+1 v_add_f64): %.3f ms = %.2f cycles per trip (4 of them the add)\n", tr * 1e-6,
         (tr - t0) * ghz / MIX_TRIPS / 2.0);
  printf("  + 6 v_mov_b32_dpp: %.3f ms = %.2f cycles per v_mov_b32_dpp\n", td * 1e-6,
         (td - t0) * ghz / MIX_TRIPS / 2.0 / 6.0);
  return 0;
}
