// ubench_gridsync.hip -- what would a device-wide barrier per small step cost?
// (Question behind it: the saturn ring, N = 500, takes 4.0 us per small step with one kernel
// per small step, 3.2 us of which is the floor of a dependent launch.  A persistent kernel
// that keeps the bodies in registers and exchanges positions through global memory needs one
// device-wide barrier per small step instead.)  G workgroups, K rounds: every workgroup
// publishes 16 doubles, arrives at a monotonic counter (release), spins until all have
// arrived (acquire, bounded: a workgroup that does not see the others within `limit` polls
// gives up and says so), reads every workgroup's 16 doubles.  Timing only: whether the data
// read is the data published is not checked here.
// Result on MI355X (profiles/r01_ubench_gridsync.txt): 1.8 us per round with 8 workgroups,
// 3.4 with 32, 6.0 with 64, 9.7 with 125, 18.7 with 250 -- the arrivals serialise at ~75 ns
// each.  A wave per body needs 125 workgroups at N = 500: the barrier alone would cost more
// than the whole small step does now.  Not pursued.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rounds(double* data, unsigned* counter, int* gave_up, int K, int limit,
                                              double* sink) {
  const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
  __shared__ int stop;
  if (t == 0) stop = 0;
  __syncthreads();
  double acc = 0.0;
  for (int k = 0; k < K; k++) {
    double* buf = data + static_cast<size_t>(k & 1) * G * 16;
    if (t < 16) buf[g * 16 + t] = g + k + acc * 1e-30;
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = static_cast<unsigned>(G) * static_cast<unsigned>(k + 1);
      int polls = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++polls > limit) { stop = 1; atomicExch(gave_up, 1); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (stop || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // every wave reaches this
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (agent scope by default in HIP)
    for (int j = t; j < G * 16; j += 256) acc += buf[j];
  }
  if (t == 0) sink[g] = acc;
}

int main() {
  const int K = 2000, limit = 200000;
  for (int G : {8, 32, 64, 125, 250}) {
    double *data, *sink;
    unsigned* counter;
    int* gave_up;
    CHECK(hipMalloc(&data, sizeof(double) * 2 * G * 16));
    CHECK(hipMalloc(&sink, sizeof(double) * G));
    CHECK(hipMalloc(&counter, sizeof(unsigned)));
    CHECK(hipMalloc(&gave_up, sizeof(int)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipMemset(counter, 0, sizeof(unsigned)));
      CHECK(hipMemset(gave_up, 0, sizeof(int)));
      CHECK(hipMemset(data, 0, sizeof(double) * 2 * G * 16));
      CHECK(hipEventRecord(e0));
      void* args[] = {&data, &counter, &gave_up, (void*)&K, (void*)&limit, &sink};
      CHECK(hipLaunchCooperativeKernel((const void*)rounds, dim3(G), dim3(256), args, 0, nullptr));
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      int gu = 0;
      CHECK(hipMemcpy(&gu, gave_up, sizeof gu, hipMemcpyDeviceToHost));
      if (rep == 1)
        printf("G=%3d workgroups: %.3f us per round (K=%d)%s\n", G, ms * 1e3 / K, K,
               gu ? "  [a workgroup GAVE UP waiting]" : "");
    }
    (void)hipFree(data); (void)hipFree(sink); (void)hipFree(counter); (void)hipFree(gave_up);
  }
  return 0;
}
