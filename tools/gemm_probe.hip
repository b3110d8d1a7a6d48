// Standalone probe for the dominant kernel: times every k_gemm_lrelu variant with hipEvents on random data and checks
// each against a naive fp32 reference.  Build & run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip
//   -o gpurun_out/gemm_probe && gpurun_out/gemm_probe [M] [iters]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define IKF_TRACE 1
#define IKF_PROBES 1   // the probes flavour: every measured form, also the rejected ones
#include "../ikflow_amd/csrc/flow_kernels.hip"
#include "../ikflow_amd/csrc/flow_fused.hip"
#include "../ikflow_amd/csrc/flow_split.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_ref(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, float slope) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], acc);
  acc += bias[n];
  C[(size_t)m * N + n] = acc > 0.f ? acc : acc * slope;
}

// pure MFMA stream: 4 independent accumulators, n_mfma instructions per wave, no memory traffic - the matrix-pipe
// ceiling at the clock the chip actually sustains
__global__ __launch_bounds__(512) void k_mfma_only(float* out, int n_mfma, float a, float b) {
  ikf::floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float av = a + threadIdx.x * 1e-6f, bv = b;
  for (int i = 0; i < n_mfma / 4; ++i) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 4096;
  int iters = argc > 2 ? atoi(argv[2]) : 100;
  const int N = 1024; const int K = argc > 4 ? atoi(argv[4]) : 1024;
  const int Mp = (M + 127) / 128 * 128;
  std::vector<float> hA((size_t)Mp * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.03125f * (K > 1024 ? 0.5f : 1.f);
  for (auto& v : hb) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.03125f;
  float *A, *W, *b, *C, *R;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4));
  CK(hipMalloc(&C, (size_t)Mp * N * 4)); CK(hipMalloc(&R, (size_t)Mp * N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ref, dim3(N / 256, M), dim3(256), 0, 0, A, W, b, R, M, N, K, 0.01f);
  CK(hipDeviceSynchronize());
  std::vector<float> hR((size_t)M * N), hC((size_t)M * N);
  CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * M * N * K;
  for (int waves = 4; waves <= 8; waves += 4) {
    const int n_mfma = 2048 * 4 / waves;  // same total MFMA work per CU as one 128x128x1024 tile
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_mfma_only, dim3(256), dim3(waves * 64), 0, 0, C, n_mfma, 0.5f, 0.25f);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_mfma_only, dim3(256), dim3(waves * 64), 0, 0, C, n_mfma, 0.5f, 0.25f);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mfma-only %d waves/CU: %.2f us/launch  %.1f TFLOP/s-equivalent\n", waves, 1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12);
    }
  }
  int vsel = argc > 3 ? atoi(argv[3]) : -1;
  if (argc > 5) ikf::g_split_dma_waves_n = atoi(argv[5]);  // LDS-DMA f16-split kernel: 2 = 4 waves, 4 = 8 waves
  if (vsel == 300) {  // k_subnet_entry<11> timeline: pending coupling (16 slots) + first Linear, rows = M
    const int D = 7, W1 = 1024, IN = 11;
    float *x, *x2, *P, *w1t, *b1, *bl, *h; int* perm;
    CK(hipMalloc(&x, (size_t)M * D * 4)); CK(hipMalloc(&x2, (size_t)M * D * 4)); CK(hipMalloc(&P, (size_t)16 * Mp * 16 * 4));
    CK(hipMalloc(&w1t, (size_t)IN * W1 * 4)); CK(hipMalloc(&b1, W1 * 4)); CK(hipMalloc(&bl, 64)); CK(hipMalloc(&h, (size_t)Mp * W1 * 4)); CK(hipMalloc(&perm, 64));
    CK(hipMemset(x, 0, (size_t)M * D * 4)); CK(hipMemset(P, 0, (size_t)16 * Mp * 16 * 4)); CK(hipMemset(bl, 0, 64));
    CK(hipMemcpy(w1t, hW.data(), (size_t)IN * W1 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b1, hb.data(), W1 * 4, hipMemcpyHostToDevice));
    int hperm[7] = {6, 2, 1, 3, 0, 5, 4}; CK(hipMemcpy(perm, hperm, 28, hipMemcpyHostToDevice));
    ikf::EntryArgs e{}; e.pend.P = P; e.pend.b_last = bl; e.pend.perm_inv = perm; e.pend.slot_stride = (long long)Mp * 16; e.pend.slots = 16; e.pend.which = 2; e.pend.n_out = 6;
    e.x_src = x; e.x_dst = x2; e.M = M; e.D = D; e.L1 = 3; e.clamp = 2.5f; e.x_off = 0; e.n_x = 3;
    e.ps.poses = A; e.ps.idx = nullptr; e.ps.n_mod = M; e.ps.stride = 7; e.ps.softflow = 0.f; e.row0 = 0;
    e.w1t = w1t; e.w1soft = b1; e.b1 = b1; e.width = W1; e.slope = 0.01f; e.h_out = h; e.split_out = 0;
    for (int geom = 0; geom < 6; ++geom) {
      ikf::g_entry_geom_override = geom;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 10; ++i) CK(ikf::launch_subnet_entry(IN - 0, e, 0));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) ikf::launch_subnet_entry(IN, e, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, 1000.0f * ms / iters);
      }
      printf("k_subnet_entry<11> M=%d geometry %d: %.2f us/launch\n", M, geom, best);
    }
    ikf::g_entry_geom_override = argc > 4 ? atoi(argv[4]) : 0;
    if (M <= 512) {  // the one-launch small-batch form (entry + first hidden contraction) against the two launches it replaces
      float *Wf, *Cq; CK(hipMalloc(&Wf, (size_t)N * K * 4)); CK(ikf::launch_wfrag_pack(W, N, K, Wf, 0)); CK(hipMalloc(&Cq, (size_t)Mp * N * 4));
      ikf::FusedGemmArgs g{}; g.A = h; g.W = W; g.Wf = Wf; g.bias = b; g.C = Cq; g.M = M; g.N = N; g.K = K; g.slope = 0.01f;
      const int cfg = M <= 256 ? 6 : 4;
      ikf::g_entry_geom_override = -1;
      for (int form = 0; form < 2; ++form) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          for (int i = 0; i < iters + 10; ++i) {
            if (i == 10) CK(hipEventRecord(e0, 0));
            if (form == 0) { CK(ikf::launch_subnet_entry(IN, e, 0)); CK(ikf::launch_flow_gemm(false, cfg, g, 0)); }
            else CK(ikf::launch_entry_gemm(IN, false, cfg, e, g, 0));
          }
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = fminf(best, 1000.0f * ms / iters);
        }
        printf("%s M=%d cfg %d: %.2f us\n", form == 0 ? "k_subnet_entry + k_flow_gemm_skinny" : "k_entry_gemm_skinny", M, cfg, best);
      }
      // chain-like rotation: 24 subnets, each with its own two weight images (2 x 24 x 4 MB stream through the caches as in a real
      // call) and partial sums rewritten by the preceding launch: [entry (+) contraction<false>] -> contraction<true> -> ...
      {
        const int NSUB = 24;
        std::vector<float*> Wr(2 * NSUB);
        for (auto& pw : Wr) { CK(hipMalloc(&pw, (size_t)N * K * 4)); CK(hipMemcpy(pw, Wf, (size_t)N * K * 4, hipMemcpyDeviceToDevice)); }
        float* wl; CK(hipMalloc(&wl, (size_t)16 * N * 4)); CK(hipMemset(wl, 0, (size_t)16 * N * 4));
        ikf::FusedGemmArgs g2 = g; g2.A = Cq; g2.C = nullptr; g2.w_last = wl; g2.n_out = 6; g2.P_out = P; g2.p_slot_stride = (long long)Mp * 16;
        ikf::EntryArgs e2 = e; e2.pend.slots = cfg == 6 ? 32 : 16;
        for (int form = 0; form < 2; ++form) {
          float best = 1e9f;
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 10; ++it)
              for (int sI = 0; sI < NSUB; ++sI) {
                g.Wf = Wr[2 * sI]; g2.Wf = Wr[2 * sI + 1];
                if (form == 0) { CK(ikf::launch_subnet_entry(IN, e2, 0)); CK(ikf::launch_flow_gemm(false, cfg, g, 0)); }
                else CK(ikf::launch_entry_gemm(IN, false, cfg, e2, g, 0));
                CK(ikf::launch_flow_gemm(true, cfg, g2, 0));
              }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, 1000.0f * ms / (10 * NSUB));
          }
          printf("chain rotation, %s: %.2f us per subnet\n", form == 0 ? "three launches" : "two launches (one-launch form)", best);
        }
        {  // in-rotation timeline of the one-launch kernel (stamps of the last launch survive)
          unsigned long long* tb2; const int nb2 = 4096;
          CK(hipMalloc(&tb2, (size_t)nb2 * 64 * 8)); CK(hipMemset(tb2, 0, (size_t)nb2 * 64 * 8));
          CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb2, sizeof(tb2)));
          for (int sI = 0; sI < NSUB; ++sI) {
            g.Wf = Wr[2 * sI]; g2.Wf = Wr[2 * sI + 1];
            CK(ikf::launch_entry_gemm(IN, false, cfg, e2, g, 0));
            if (sI + 1 < NSUB) CK(ikf::launch_flow_gemm(true, cfg, g2, 0));
          }
          CK(hipDeviceSynchronize());
          std::vector<unsigned long long> ht2((size_t)nb2 * 64); CK(hipMemcpy(ht2.data(), tb2, ht2.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0min = ~0ull, tend = 0;
          for (int bI = 0; bI < 256; ++bI) { if (ht2[bI * 64] && ht2[bI * 64] < t0min) t0min = ht2[bI * 64]; if (ht2[bI * 64 + 41] > tend) tend = ht2[bI * 64 + 41]; }
          printf("in rotation: kernel span first-start..last-end %llu cycles\n", tend - t0min);
          for (int bI : {0, 100, 255}) { const unsigned long long* r = &ht2[(size_t)bI * 64];
            printf("in rotation block %3d: start+%llu pending %llu  publish+first-Linear %llu  K loop %llu  tail %llu  total %llu cycles\n", bI, r[0] - t0min, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[41] - r[3], r[41] - r[0]); }
          unsigned long long z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &z, sizeof(z)));
        }
        g.Wf = Wf;
      }
      unsigned long long* tb; const int nb = 4096;
      CK(hipMalloc(&tb, (size_t)nb * 64 * 8)); CK(hipMemset(tb, 0, (size_t)nb * 64 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb, sizeof(tb)));
      ikf::launch_entry_gemm(IN, false, cfg, e, g, 0); ikf::launch_entry_gemm(IN, false, cfg, e, g, 0); CK(hipDeviceSynchronize());
      std::vector<unsigned long long> ht((size_t)nb * 64); CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
      for (int bI : {0, 100}) { const unsigned long long* r = &ht[(size_t)bI * 64];
        printf("one-launch block %3d: pending %llu  publish+first-Linear %llu  K loop %llu  tail %llu  total %llu cycles\n", bI, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[41] - r[3], r[41] - r[0]); }
      return 0;
    }
    unsigned long long* tb; const int nb = 4096;
    CK(hipMalloc(&tb, (size_t)nb * 64 * 8)); CK(hipMemset(tb, 0, (size_t)nb * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb, sizeof(tb)));
    ikf::launch_subnet_entry(IN, e, 0); ikf::launch_subnet_entry(IN, e, 0); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht((size_t)nb * 64); CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
    for (int bI : {0, 100, 200}) { const unsigned long long* r = &ht[(size_t)bI * 64];
      printf("block %3d: pending %llu  publish+U %llu  first-Linear+store %llu  total %llu cycles\n", bI, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[3] - r[0]); }
    return 0;
  }
  const int scfg = (vsel == 210) ? 10 : (vsel == 200 || vsel == 211) ? 11 : vsel - 200;  // 201..203: smaller split tiles
  if (vsel >= 200 && vsel < 300) {  // the f16-split contraction (k_split_gemm<false>): correctness vs the fp32 reference, timing, timeline
    std::vector<uint16_t> sA((size_t)Mp * K * 2), sW((size_t)N * K * 2);
    ikf::split32_pack_host(hA.data(), Mp, K, sA.data()); ikf::split32_pack_host(hW.data(), N, K, sW.data());
    void *dsA, *dsW, *dsC; CK(hipMalloc(&dsA, sA.size() * 2)); CK(hipMalloc(&dsW, sW.size() * 2)); CK(hipMalloc(&dsC, (size_t)Mp * N * 4));
    CK(hipMemcpy(dsA, sA.data(), sA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dsW, sW.data(), sW.size() * 2, hipMemcpyHostToDevice));
    ikf::SplitGemmArgs g{}; g.A = dsA; g.W = dsW; g.bias = b; g.C = dsC; g.M = M; g.N = N; g.K = K; g.slope = 0.01f;
    CK(ikf::launch_split_gemm(false, scfg, g, 0)); CK(hipDeviceSynchronize());
    std::vector<uint16_t> sC((size_t)Mp * N * 2); CK(hipMemcpy(sC.data(), dsC, sC.size() * 2, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
      _Float16 hi, lo; __builtin_memcpy(&hi, &sC[(size_t)m * N * 2 + (size_t)(n >> 5) * 64 + (n & 31)], 2); __builtin_memcpy(&lo, &sC[(size_t)m * N * 2 + (size_t)(n >> 5) * 64 + 32 + (n & 31)], 2);
      maxerr = fmax(maxerr, fabs((double)(float)hi + (double)(float)lo / 2048.0 - hR[(size_t)m * N + n])); }
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 10; ++i) ikf::launch_split_gemm(false, scfg, g, 0);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) ikf::launch_split_gemm(false, scfg, g, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("k_split_gemm M=%d K=%d: %.2f us/launch  %.1f TFLOP/s (algorithmic)  maxerr vs f32 ref %.2e\n", M, K, 1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12, maxerr);
    }
    unsigned long long* tb; const int nb = 4096;
    CK(hipMalloc(&tb, (size_t)nb * 64 * 8)); CK(hipMemset(tb, 0, (size_t)nb * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb, sizeof(tb)));
    ikf::launch_split_gemm(false, scfg, g, 0); ikf::launch_split_gemm(false, scfg, g, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht((size_t)nb * 64);
    CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
    for (int bI : {0, 100}) {
      const unsigned long long* r = &ht[(size_t)bI * 64];
      printf("block %3d: prologue %llu |", bI, r[1] - r[0]);
      unsigned long long prev = r[1];
      for (int q = 0; q < 32 && r[2 + q]; ++q) { printf(" %llu", r[2 + q] - prev); prev = r[2 + q]; }
      printf(" | tail-tiles %llu  epilogue %llu  total %llu cycles\n", r[40] - prev, r[41] - r[40], r[41] - r[0]);
    }
    return 0;
  }
  if (vsel >= 100) {  // the fused-pipeline contraction (k_flow_gemm<false>, tile config vsel-100): timing + in-kernel timeline
    ikf::FusedGemmArgs g{}; g.A = A; g.W = W; g.bias = b; g.C = C; g.M = M; g.N = N; g.K = K; g.slope = 0.01f;
    const int cfg = vsel - 100;
    float* Wf; CK(hipMalloc(&Wf, (size_t)N * K * 4)); CK(ikf::launch_wfrag_pack(W, N, K, Wf, 0)); g.Wf = Wf;
    CK(hipMemset(C, 0, (size_t)Mp * N * 4));
    CK(ikf::launch_flow_gemm(false, cfg, g, 0)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0; for (size_t i = 0; i < hC.size(); ++i) maxerr = fmax(maxerr, fabs((double)hC[i] - hR[i]));
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 10; ++i) ikf::launch_flow_gemm(false, cfg, g, 0);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) ikf::launch_flow_gemm(false, cfg, g, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("k_flow_gemm cfg %d M=%d K=%d: %.2f us/launch  %.1f TFLOP/s  maxerr %.2e\n", cfg, M, K, 1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12, maxerr);
    }
    if (getenv("IKF_PROBE_PINGPONG")) {  // as in the engine's chain: every launch reads the activation the previous launch wrote, over 48 weight images
      float* C2; CK(hipMalloc(&C2, (size_t)Mp * N * 4)); CK(hipMemcpy(C2, A, (size_t)Mp * K * 4, hipMemcpyDeviceToDevice));
      const int NW = 48;
      float* Wall; CK(hipMalloc(&Wall, (size_t)NW * N * K * 4));
      for (int w = 0; w < NW; ++w) CK(hipMemcpy(Wall + (size_t)w * N * K, W, (size_t)N * K * 4, hipMemcpyDeviceToDevice));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
          ikf::FusedGemmArgs h = g;
          h.A = (i & 1) ? C2 : C; h.C = (i & 1) ? C : C2; h.W = Wall + (size_t)(i % NW) * N * K;
          ikf::launch_flow_gemm(false, cfg, h, 0);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  chain (ping-pong activations, 48 weight images): %.2f us/launch\n", 1000.0 * ms / iters);
      }
    }
    unsigned long long* tb; const int nb = 4096;
    CK(hipMalloc(&tb, (size_t)nb * 64 * 8)); CK(hipMemset(tb, 0, (size_t)nb * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb, sizeof(tb)));
    ikf::launch_flow_gemm(false, cfg, g, 0); ikf::launch_flow_gemm(false, cfg, g, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht((size_t)nb * 64);
    CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
    for (int bI : {0, 100}) {
      const unsigned long long* r = &ht[(size_t)bI * 64];
      printf("block %3d: prologue %llu |", bI, r[1] - r[0]);
      unsigned long long prev = r[1];
      for (int q = 0; q < 32 && r[2 + q]; ++q) { printf(" %llu", r[2 + q] - prev); prev = r[2 + q]; }
      printf(" | tail-tiles %llu  epilogue %llu  total %llu cycles\n", r[40] - prev, r[41] - r[40], r[41] - r[0]);
    }
    unsigned long long z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &z, sizeof(z)));
    return 0;
  }
  if (false) {
    unsigned long long* tb; const int nb = ((M + 127) / 128) * (N / 128);
    CK(hipMalloc(&tb, (size_t)nb * 64 * 8)); CK(hipMemset(tb, 0, (size_t)nb * 64 * 8));
    for (int i = 0; i < 3; ++i) ikf::launch_gemm_lrelu(vsel, A, W, b, C, M, N, K, 0.01f, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &tb, sizeof(tb)));
    ikf::launch_gemm_lrelu(vsel, A, W, b, C, M, N, K, 0.01f, 0);
    ikf::launch_gemm_lrelu(vsel, A, W, b, C, M, N, K, 0.01f, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht((size_t)nb * 64);
    CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0min = ~0ull, tendmax = 0;
    for (int bI = 0; bI < nb; ++bI) { if (ht[bI*64] < t0min) t0min = ht[bI*64]; if (ht[bI*64+41] > tendmax) tendmax = ht[bI*64+41]; }
    printf("trace (cycles of s_memtime clock): kernel span first-start..last-end = %llu\n", tendmax - t0min);
    for (int bI : {0, 100}) {
      if (bI >= nb) continue;
      const unsigned long long* r = &ht[(size_t)bI * 64];
      printf("block %3d: start+%llu  prologue %llu |", bI, r[0] - t0min, r[1] - r[0]);
      unsigned long long prev = r[1];
      for (int q = 0; q < 32 && r[2 + q]; ++q) { printf(" %llu", r[2 + q] - prev); prev = r[2 + q]; }
      printf(" | tail-tiles %llu  epilogue %llu  end+%llu\n", r[40] - prev, r[41] - r[40], tendmax - r[41]);
    }
    unsigned long long z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &z, sizeof(z)));
  }
  for (int v = 0; v < ikf::gemm_variant_count(); ++v) {
    if (vsel >= 0 && v != vsel) continue;
    CK(hipMemset(C, 0, (size_t)Mp * N * 4));
    hipError_t e = ikf::launch_gemm_lrelu(v, A, W, b, C, M, N, K, 0.01f, 0);
    if (e != hipSuccess) { printf("variant %d: launch error %s\n", v, hipGetErrorString(e)); continue; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0; 
    for (size_t i = 0; i < hC.size(); ++i) maxerr = fmax(maxerr, fabs((double)hC[i] - hR[i]));
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 10; ++i) ikf::launch_gemm_lrelu(v, A, W, b, C, M, N, K, 0.01f, 0);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) ikf::launch_gemm_lrelu(v, A, W, b, C, M, N, K, 0.01f, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("variant %d M=%d: %.2f us/launch  %.1f TFLOP/s  maxerr %.2e\n", v, M, 1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12, maxerr);
    }
  }
  return 0;
}
