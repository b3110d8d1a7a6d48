// Can the f32 matrix pipe and the f32 vector pipe of gfx950 be driven together?  256 workgroups of 8 waves: waves 0-3 issue
// v_mfma_f32_32x32x2_f32 streams (4 independent accumulators), waves 4-7 issue packed-FMA streams from registers.
//   hipcc --offload-arch=gfx950 -O3 tools/dual_pipe_probe.hip -o /tmp/dual_pipe_probe && /tmp/dual_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k_dual(float* out, int n_mfma, int n_fma, float a, float b) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float av = a + threadIdx.x * 1e-6f;
    for (int i = 0; i < n_mfma / 4; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    floatx2 c[16];
    for (int i = 0; i < 16; ++i) c[i] = floatx2{0.f, (float)i};
    const floatx2 x = {a + threadIdx.x * 1e-6f, b}, y = {b, a};
    for (int i = 0; i < n_fma / 16; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c[j] = __builtin_elementwise_fma(x, y, c[j]);  // v_pk_fma_f32, 16 independent chains
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[i].x + c[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
}

static float run(int n_mfma, int n_fma, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_dual, dim3(256), dim3(512), 0, 0, out, n_mfma, n_fma, 0.5f, 0.25f);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_dual, dim3(256), dim3(512), 0, 0, out, n_mfma, n_fma, 0.5f, 0.25f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1000.f * ms / 50;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int NM = 2048 * 4;      // MFMAs per wave (x 4 waves per CU)
  const int NF = 32768 * 4;     // packed FMAs per wave: same FLOP count per wave as NM MFMAs (4096 FLOP each vs 256)
  for (int rep = 0; rep < 2; ++rep) {
    const float t_m = run(NM, 0, out), t_v = run(0, NF, out), t_b = run(NM, NF, out);
    const double f_m = 256.0 * 4 * NM * 4096.0, f_v = 256.0 * 4 * NF * 256.0;
    printf("MFMA only %.1f us = %.1f TF/s | packed-FMA only %.1f us = %.1f TF/s | both %.1f us = %.1f TF/s total\n", t_m, f_m / t_m / 1e6,
           t_v, f_v / t_v / 1e6, t_b, (f_m + f_v) / t_b / 1e6);
  }
  return 0;
}
