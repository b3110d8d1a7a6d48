// Instruction-cache reach on gfx950: a straight-line block of N 4-byte VALU instructions executed K times by one wave per workgroup.
// Cycles per instruction by block size, for 1 workgroup and for one workgroup per CU (the I-cache is shared between CUs).
//   hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o tools/bin/icache_probe && tools/bin/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int N>
__global__ void k_block(unsigned long long* out, int iters, int x0) {
  int x = x0 + threadIdx.x, y = 3;
  unsigned long long t0 = 0, t1 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == 1) t0 = __builtin_readcyclecounter();   // (the first pass warms the cache)
    if constexpr (N == 5120) asm volatile(".rept 5120\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 7168) asm volatile(".rept 7168\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 1024) asm volatile(".rept 1024\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 2048) asm volatile(".rept 2048\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 3072) asm volatile(".rept 3072\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 4096) asm volatile(".rept 4096\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 6144) asm volatile(".rept 6144\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
    if constexpr (N == 8192) asm volatile(".rept 8192\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)x; }
}

template <int N>
int run(unsigned long long* d, int grid, int iters) {
  hipLaunchKernelGGL(k_block<N>, dim3(grid), dim3(64), 0, nullptr, d, iters, 1);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  double mx = 0, mn = 1e30;
  for (int b = 0; b < grid; ++b) { mx = mx > (double)h[2 * b] ? mx : (double)h[2 * b]; mn = mn < (double)h[2 * b] ? mn : (double)h[2 * b]; }
  printf("  block %6d instructions = %4d KB, grid %3d: %.2f .. %.2f cycles per instruction\n", N, N * 4 / 1024, grid, mn / ((double)N * (iters - 1)), mx / ((double)N * (iters - 1)));
  return 0;
}

int main() {
  unsigned long long* d; CK(hipMalloc(&d, 16 * 1024));
  for (int grid : {1, 256, 512}) {
    if (run<1024>(d, grid, 20)) return 1;
    if (run<2048>(d, grid, 20)) return 1;
    if (run<3072>(d, grid, 20)) return 1;
    if (run<4096>(d, grid, 20)) return 1;
    if (run<5120>(d, grid, 20)) return 1;
    if (run<6144>(d, grid, 20)) return 1;
    if (run<7168>(d, grid, 20)) return 1;
    if (run<8192>(d, grid, 20)) return 1;
  }
  return 0;
}
