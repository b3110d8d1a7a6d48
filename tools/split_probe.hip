// Numerics probe for the error-compensated f16 split contraction on real hardware:
//   a = hi + lo/2048 with hi = f16(a), lo = f16((a - hi) * 2048)   (same for w)
//   a.w ~= hi_a*hi_w + (hi_a*lo_w + lo_a*hi_w)/2048              (3 v_mfma_f32_32x32x16_f16, fp32 accumulate)
// compared with the exact-f32 path (an fmaf chain = what v_mfma_f32_32x32x2_f32 computes) against an fp64 host reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_f32_chain(const float* A, const float* W, float* C, int M, int N, int K) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], acc);
  C[(size_t)m * N + n] = acc;
}

// one wave per 32x32 output block; planar hi/lo arrays [rows][K] of f16
__global__ __launch_bounds__(64) void k_split(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl,
                                              float* C, int M, int N, int K, int nterms) {
  const int lane = threadIdx.x, m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  floatx16 am, ac, ac2;
  for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; ac2[r] = 0.f; }
  const size_t ra = (size_t)(m0 + (lane & 31)) * K + (lane >> 5) * 8, rw = (size_t)(n0 + (lane & 31)) * K + (lane >> 5) * 8;
  for (int k = 0; k < K; k += 16) {
    const half8 ah = *reinterpret_cast<const half8*>(Ah + ra + k), al = *reinterpret_cast<const half8*>(Al + ra + k);
    const half8 wh = *reinterpret_cast<const half8*>(Wh + rw + k), wl = *reinterpret_cast<const half8*>(Wl + rw + k);
    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, am, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, ac, 0, 0, 0);
    if (nterms == 4) ac2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wl, ac2, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + (lane & 31);
    float v = am[r] + ac[r] * (1.0f / 2048.0f);
    if (nterms == 4) v += ac2[r] * (1.0f / 2048.0f / 2048.0f);
    C[(size_t)row * N + col] = v;
  }
}

int main(int argc, char** argv) {
  const int M = 256, N = 256, K = argc > 1 ? atoi(argv[1]) : 1024;
  const float wscale = argc > 2 ? atof(argv[2]) : 0.03125f;
  std::vector<float> A((size_t)M * K), W((size_t)N * K);
  srand(3);
  for (auto& v : A) { float u = (rand() / (float)RAND_MAX) * 2.f - 1.f; u *= 3.0f; v = u > 0 ? u : 0.01f * u; }   // LeakyReLU-like
  for (auto& v : W) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * wscale;
  std::vector<_Float16> Ah(A.size()), Al(A.size()), Wh(W.size()), Wl(W.size());
  for (size_t i = 0; i < A.size(); ++i) { Ah[i] = (_Float16)A[i]; Al[i] = (_Float16)((A[i] - (float)Ah[i]) * 2048.0f); }
  for (size_t i = 0; i < W.size(); ++i) { Wh[i] = (_Float16)W[i]; Wl[i] = (_Float16)((W[i] - (float)Wh[i]) * 2048.0f); }
  std::vector<double> R((size_t)M * N);
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[(size_t)m*K+k] * (double)W[(size_t)n*K+k]; R[(size_t)m*N+n] = s; }
  float *dA, *dW, *dC; _Float16 *dAh, *dAl, *dWh, *dWl;
  CK(hipMalloc(&dA, A.size()*4)); CK(hipMalloc(&dW, W.size()*4)); CK(hipMalloc(&dC, (size_t)M*N*4));
  CK(hipMalloc(&dAh, A.size()*2)); CK(hipMalloc(&dAl, A.size()*2)); CK(hipMalloc(&dWh, W.size()*2)); CK(hipMalloc(&dWl, W.size()*2));
  CK(hipMemcpy(dA, A.data(), A.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size()*4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dAh, Ah.data(), A.size()*2, hipMemcpyHostToDevice)); CK(hipMemcpy(dAl, Al.data(), A.size()*2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dWh, Wh.data(), W.size()*2, hipMemcpyHostToDevice)); CK(hipMemcpy(dWl, Wl.data(), W.size()*2, hipMemcpyHostToDevice));
  std::vector<float> C((size_t)M * N);
  auto report = [&](const char* name) {
    CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, C.size()*4, hipMemcpyDeviceToHost));
    double mx = 0, ss = 0, ref = 0;
    for (size_t i = 0; i < C.size(); ++i) { double e = fabs((double)C[i] - R[i]); mx = fmax(mx, e); ss += e*e; ref += R[i]*R[i]; }
    printf("%-28s K=%d  max|err| %.3e  rms err %.3e  (rms of result %.3e)\n", name, K, mx, sqrt(ss / C.size()), sqrt(ref / C.size()));
  };
  hipLaunchKernelGGL(k_f32_chain, dim3(N / 256, M), dim3(256), 0, 0, dA, dW, dC, M, N, K); report("f32 fmaf chain (= f32 MFMA)");
  hipLaunchKernelGGL(k_split, dim3(N / 32, M / 32), dim3(64), 0, 0, dAh, dAl, dWh, dWl, dC, M, N, K, 3); report("f16 split, 3 products");
  hipLaunchKernelGGL(k_split, dim3(N / 32, M / 32), dim3(64), 0, 0, dAh, dAl, dWh, dWl, dC, M, N, K, 4); report("f16 split, 4 products");
  // representation error alone (what the inputs lose by being rounded to hi + lo/2048), in fp64
  { double mx = 0, ss = 0; for (int m = 0; m < 64; ++m) for (int n = 0; n < 64; ++n) { double s = 0; for (int k = 0; k < K; ++k) {
      double a = (double)(float)Ah[(size_t)m*K+k] + (double)(float)Al[(size_t)m*K+k] / 2048.0, w = (double)(float)Wh[(size_t)n*K+k] + (double)(float)Wl[(size_t)n*K+k] / 2048.0; s += a * w; }
      double e = fabs(s - R[(size_t)m*N+n]); mx = fmax(mx, e); ss += e*e; } printf("representation-only error (fp64 math on split inputs): max %.3e rms %.3e\n", mx, sqrt(ss / 4096)); }
  return 0;
}
