// Micro-probe: cost of v_mfma_f32_32x32x2_f32 issued as a dependent chain (one accumulator) vs independent
// accumulators, at 1 / 2 / 4 waves per SIMD.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_chain_probe.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k_chain(float* out, unsigned long long* cyc, int n, float a, float b) {
  floatx16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float av = a + threadIdx.x * 1e-6f, bv = b;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u % NACC], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
int run(int waves, int blocks, int n) {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * waves * 64 * 4)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_chain<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, n, 0.5f, 0.25f);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_chain<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, n, 0.5f, 0.25f);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per_simd = (double)c / ((double)n * waves / 4.0);
  printf("acc=%d waves/CU=%2d blocks=%3d: %7.2f us/launch, wave0 %8llu cycles for %d MFMAs -> %.1f cycles per MFMA per SIMD (ideal 64)\n",
         NACC, waves, blocks, 1000.0 * ms / 20, c, n, per_simd);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  for (int blocks : {16, 256})
    for (int waves : {4, 8, 16}) {
      const int n = 4096 * 4 / waves;
      if (run<1>(waves, blocks, n)) return 1;
      if (run<2>(waves, blocks, n)) return 1;
      if (run<4>(waves, blocks, n)) return 1;
    }
  return 0;
}
