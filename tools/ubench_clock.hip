// ubench_clock.hip -- the shader clock the chip actually runs at, by what it executes: every wave
// reads s_memtime (shader clock ticks) and the constant 100 MHz clock at its start and end.
// Feeds DESIGN.md's roofline discussion (the 78.6 TFLOP/s FP64 vector peak assumes 2.4 GHz).
// Build: make -C tools.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { SALU_ONLY, FMA32, FMA64, FMA64_RSQ, NKIND };
static const char* kNames[NKIND] = {"scalar loop only (VALU idle)", "v_fma_f32, 8 chains", "v_fma_f64, 8 chains",
                                    "16 v_fma_f64 + v_rsq_f64 (the pair mix)"};

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, double a, double b) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  double x[8];
  float f[8];
#pragma unroll
  for (int c = 0; c < 8; c++) { x[c] = 1.0 + 1e-3 * (threadIdx.x + c); f[c] = static_cast<float>(x[c]); }
  int s = 0;
  for (int it = 0; it < iters; it++) {
    if (KIND == SALU_ONLY) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory"); s += it; }
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (KIND == FMA32) f[c] = __builtin_fmaf(f[c], static_cast<float>(a), static_cast<float>(b));
      if (KIND == FMA64) x[c] = __builtin_fma(x[c], a, b);
      if (KIND == FMA64_RSQ) {
        x[c] = __builtin_fma(x[c], a, b);
        x[c] = __builtin_fma(x[c], a, b);
        if (c == 0) x[0] = __builtin_amdgcn_rsq(x[0]);
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double sum = s;
#pragma unroll
  for (int c = 0; c < 8; c++) sum += x[c] + f[c];
  if (sum == 123.456) out[0] = 1;  // keeps the work live, never true
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = c1 - c0;
    out[2 * w + 1] = w1 - w0;
  }
}

template <int KIND>
void run(unsigned long long* d, int blocks, int iters, const char* what) {
  const int waves = blocks * 4;
  std::vector<unsigned long long> h(2 * waves);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  double us = 0;
  for (int w = 0; w < waves; w++)
    if (h[2 * w + 1] > 0) { ghz.push_back(h[2 * w] / (h[2 * w + 1] * 10.0)); us = std::max(us, h[2 * w + 1] / 100.0); }
  std::sort(ghz.begin(), ghz.end());
  printf("%-42s %-28s %9.1f us   clock min %.3f median %.3f max %.3f GHz\n", kNames[KIND], what, us, ghz.front(),
         ghz[ghz.size() / 2], ghz.back());
}

// a train of launches with no host synchronisation between them: the clock of each launch
template <int KIND>
void train(int blocks, int iters, int launches, int gap_iters) {
  const int waves = blocks * 4;
  unsigned long long* d;
  CHECK(hipMalloc(&d, sizeof(unsigned long long) * 2 * waves * launches + 64));
  CHECK(hipDeviceSynchronize());
  for (int l = 0; l < launches; l++) {
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d + static_cast<size_t>(2) * waves * l, iters, 1.0000001, 1e-9);
    if (gap_iters > 0) hipLaunchKernelGGL(k<SALU_ONLY>, dim3(1), dim3(64), 0, 0, d + static_cast<size_t>(2) * waves * launches, gap_iters, 1.0, 1.0);
  }
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(static_cast<size_t>(2) * waves * launches);
  CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  printf("train of %d launches of %s, %d workgroups, %d trips, %s between launches: median clock of launch", launches, kNames[KIND],
         blocks, iters, gap_iters > 0 ? "a one-wave kernel" : "nothing");
  for (int l = 0; l < launches; l++) {
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) {
      const size_t i = static_cast<size_t>(2) * (static_cast<size_t>(waves) * l + w);
      if (h[i + 1] > 0) ghz.push_back(h[i] / (h[i + 1] * 10.0));
    }
    std::sort(ghz.begin(), ghz.end());
    if (l < 4 || l % (launches / 8) == 0 || l == launches - 1) printf("  #%d %.3f", l, ghz[ghz.size() / 2]);
  }
  printf(" GHz\n");
  CHECK(hipFree(d));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  unsigned long long* d;
  CHECK(hipMalloc(&d, sizeof(unsigned long long) * 2 * 4 * cus * 2 + 64));
  for (int iters : {2000, 200000}) {
    run<SALU_ONLY>(d, 1, iters, "one workgroup");
    run<SALU_ONLY>(d, cus, iters, "one workgroup per CU");
    run<FMA32>(d, 1, iters, "one workgroup");
    run<FMA32>(d, cus, iters, "one workgroup per CU");
    run<FMA32>(d, 2 * cus, iters, "two workgroups per CU");
    run<FMA64>(d, 1, iters, "one workgroup");
    run<FMA64>(d, cus, iters, "one workgroup per CU");
    run<FMA64>(d, 2 * cus, iters, "two workgroups per CU");
    run<FMA64_RSQ>(d, 2 * cus, iters, "two workgroups per CU");
    printf("\n");
  }
  train<FMA64_RSQ>(2 * cus, 400, 64, 0);      // ~35 us launches
  train<FMA64_RSQ>(2 * cus, 1200, 64, 0);     // ~100 us
  train<FMA64_RSQ>(2 * cus, 16000, 64, 0);    // ~1.4 ms
  train<FMA64_RSQ>(2 * cus, 1200, 64, 50);    // ~100 us with a small kernel between (as the finishing kernels are)
  train<FMA64_RSQ>(2 * cus, 200000, 40, 0);   // ~15 ms: 0.6 s in all
  return 0;
}
