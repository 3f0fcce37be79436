// ubench_valu_rates.hip -- issue rates of the VALU instructions the FP32 step is made of, measured:
// cycles per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_rsq_f32, v_fma_f64, v_rsq_f64,
// a DPP row rotation, and mixes of them.  Every SIMD gets 2 waves of independent chains.
// Build: make -C tools ubench_valu_rates ; run: tools/ubench_valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096;
constexpr int kChains = 8;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[kChains];
  float2v p[kChains];
  double d[kChains];
  for (int i = 0; i < kChains; i++) {
    a[i] = seed + i + threadIdx.x;
    p[i] = float2v{a[i], a[i] + 1};
    d[i] = a[i];
  }
  const float m = seed * 0.5f, c = seed * 0.25f;
  const float2v pm = {m, m}, pc = {c, c};
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < kChains; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
      if (KIND == 2) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)m), "v"((double)c));
      if (KIND == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %0 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 6) {  // 3 fma + 1 rsq, as a pair's share goes
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(p[i].x));
      }
      if (KIND == 7) {  // pk_fma with a broadcast low half (op_sel_hi:[1,0,1])
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
      }
      if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
      if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
      if (KIND == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 13) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 14) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 15) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 16) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"((double)m), "v"((double)c));
      if (KIND == 17) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)m));
      if (KIND == 18) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)m));
      if (KIND == 19) asm volatile("v_sub_f32_dpp %0, %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
      if (KIND == 20) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      if (KIND == 22) asm volatile("v_fma_f32 %0, %1, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 23) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 24) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 25) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pm), "v"(pc));
      if (KIND == 26) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(m));
      if (KIND == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
      if (KIND == 28) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      if (KIND == 29) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      if (KIND == 30) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 31) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(it));
      if (KIND == 32) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 33) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 34) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < kChains; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int per_iter, float* out, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU, 256 CUs
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = double(kIters) * kChains * per_iter * waves_per_simd;
  printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name,
         waves_per_simd, ms, ms * 1e6 / instr_per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  for (int w : {2, 4}) {
    run<0>("v_fma_f32", 1, out, w);
    run<1>("v_pk_fma_f32", 1, out, w);
    run<7>("v_pk_fma_f32 op_sel_hi", 1, out, w);
    run<8>("v_pk_mul_f32", 1, out, w);
    run<9>("v_pk_add_f32", 1, out, w);
    run<10>("v_mul_f32", 1, out, w);
    run<11>("v_sub_f32", 1, out, w);
    run<2>("v_rsq_f32", 1, out, w);
    run<3>("v_fma_f64", 1, out, w);
    run<4>("v_rsq_f64", 1, out, w);
    run<5>("v_mov_b32_dpp wave_rol:1", 1, out, w);
    run<6>("3 fma_f32 + rsq_f32", 4, out, w);
    run<12>("v_fmac_f32 (VOP2)", 1, out, w);
    run<13>("v_fma_f32 acc=addend (VOP3)", 1, out, w);
    run<15>("v_fma_f32 a*b - acc", 1, out, w);
    run<22>("v_fma_f32 m*m+c -> acc", 1, out, w);
    run<24>("v_fmac_f32 acc += m*m", 1, out, w);
    run<14>("v_mul_f32_e64", 1, out, w);
    run<16>("v_fmac_f64", 1, out, w);
    run<17>("v_mul_f64", 1, out, w);
    run<18>("v_add_f64", 1, out, w);
    run<19>("v_sub_f32_dpp wave_rol:1", 1, out, w);
    run<20>("v_fmac_f32_dpp wave_rol:1", 1, out, w);
    run<21>("v_mov_b32", 1, out, w);
    run<23>("v_mov_b32_dpp row_shr:1", 1, out, w);
    run<25>("v_pk_fma_f32 acc=addend", 1, out, w);
    run<26>("v_sub_f32 sgpr src", 1, out, w);
    run<27>("v_fmac_f32 sgpr src", 1, out, w);
    run<28>("v_cvt_f32_f64", 1, out, w);
    run<29>("v_cvt_f64_f32", 1, out, w);
    run<30>("v_min3_f32", 1, out, w);
    run<31>("v_ldexp_f64", 1, out, w);
    run<32>("v_rcp_f64", 1, out, w);
    run<33>("v_sqrt_f64", 1, out, w);
    run<34>("v_rcp_f32", 1, out, w);
  }
  hipFree(out);
  return 0;
}
