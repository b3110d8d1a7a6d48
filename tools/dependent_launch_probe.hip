// What would hiding the kernel boundary buy on a 72-launch dependency chain?  256 workgroups x 1024 threads of MFMA work per launch:
//   (a) all launches on one stream (the boundary between dependent launches is the stream's own barrier);
//   (b) launches alternate between two streams and launch i spins on a completion counter of launch i-1 ("dependent launch by hand").
// Timing only: no data is handed over, so no coherence protocol is needed here.  Both grids fit on the chip at once (2 x 16 waves per CU),
// which is what makes (b) hang-free.   hipcc --offload-arch=gfx950 -O3 tools/dependent_launch_probe.hip -o /tmp/dlp && timeout 30 /tmp/dlp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(1024) void k_work(const unsigned* wait_on, unsigned expected, unsigned* done, int n_mfma, float* sink) {
  if (wait_on != nullptr) {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  floatx16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float a = 0.5f + threadIdx.x * 1e-6f;
  for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 0.25f, acc, 0, 0, 0);
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  __syncthreads();
  if (done != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
  const int chain = 72, grid = 256;
  const int n_mfma = argc > 1 ? atoi(argv[1]) : 64;  // 64 MFMAs x 4 waves per SIMD = 16.4 k cycles = the 512-row contraction's floor
  unsigned* cnt; float* sink;
  hipMalloc(&cnt, chain * sizeof(unsigned)); hipMalloc(&sink, 4096);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemsetAsync(cnt, 0, chain * sizeof(unsigned), s0);
      hipStreamSynchronize(s0); hipStreamSynchronize(s1);
      hipEventRecord(e0, s0);
      if (mode == 1) { hipEventRecord(e1, s0); hipStreamWaitEvent(s1, e1, 0); }
      for (int i = 0; i < chain; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k_work, dim3(grid), dim3(1024), 0, s0, (const unsigned*)nullptr, 0u, (unsigned*)nullptr, n_mfma, sink);
        else hipLaunchKernelGGL(k_work, dim3(grid), dim3(1024), 0, (i & 1) ? s1 : s0, i ? cnt + i - 1 : (const unsigned*)nullptr, (unsigned)grid, cnt + i, n_mfma, sink);
      }
      if (mode == 1) { hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0); }
      hipEventRecord(e1, s0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 3 && ms < best) best = ms;
    }
    printf("%s: %.1f us per launch (chain of %d, %d MFMAs per wave)\n", mode == 0 ? "one stream" : "two streams + completion counters", 1000.f * best / chain, chain, n_mfma);
  }
  return 0;
}
