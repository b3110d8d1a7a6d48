// What does a dependent launch cost as a function of workgroup size, dynamic LDS and kernel-argument size?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o /tmp/launch_probe && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct BigArgs { float v[72]; };  // 288 bytes, like EntryArgs + FusedGemmArgs
template <int NT> __global__ __launch_bounds__(NT) void k_small(float* out, int n) {
  extern __shared__ float lds[];
  if (n < 0) { lds[threadIdx.x] = 1.f; out[threadIdx.x] = lds[(threadIdx.x + 1) % NT]; }
}
template <int NT> __global__ __launch_bounds__(NT) void k_bigargs(float* out, int n, BigArgs a) {
  extern __shared__ float lds[];
  if (n < 0) { lds[threadIdx.x] = a.v[threadIdx.x % 72]; out[threadIdx.x] = lds[(threadIdx.x + 1) % NT]; }
}
template <class K> float time_it(K launch, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1000.f * ms / iters;
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  BigArgs a{};
  const int lds_sizes[] = {0, 42 * 1024, 81 * 1024, 136 * 1024, 160 * 1024};
  for (int lds : lds_sizes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_small<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_small<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_small<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_bigargs<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float t256 = time_it([&] { hipLaunchKernelGGL(k_small<256>, dim3(256), dim3(256), lds, 0, out, 1); }, 2000);
    float t512 = time_it([&] { hipLaunchKernelGGL(k_small<512>, dim3(256), dim3(512), lds, 0, out, 1); }, 2000);
    float t1024 = time_it([&] { hipLaunchKernelGGL(k_small<1024>, dim3(256), dim3(1024), lds, 0, out, 1); }, 2000);
    float tb = time_it([&] { hipLaunchKernelGGL(k_bigargs<1024>, dim3(256), dim3(1024), lds, 0, out, 1, a); }, 2000);
    printf("256 workgroups, dynamic LDS %3d KB: 256 thr %.2f us  512 thr %.2f us  1024 thr %.2f us  1024 thr + 288 B args %.2f us\n", lds / 1024, t256, t512, t1024, tb);
  }
  return 0;
}
