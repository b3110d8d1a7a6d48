// Conditional-flow inverse pass for gfx950 (MI355X): the three kernels one coupling subnet is made of.
//
// Replaces, per coupling block and subnet, what FrEIA's GLOWCouplingBlock.forward(rev=True) launches through torch
// (call site ikflow/ikflow_solver.py:98; graph ikflow/model.py:300-354; subnet ikflow/model.py:51-96):
//
//   k_first_layer            cat[x_part, cond] -> Linear(in, W) -> LeakyReLU            (in = 10..15: VALU, HBM-write bound)
//   k_gemm_lrelu             Linear(W, W) -> LeakyReLU as a [rows x W] . [W x W]^T contraction on the f32 MFMA
//                            (v_mfma_f32_32x32x2_f32) - 99 % of the FLOPs, the dominant kernel
//   k_last_layer_coupling    Linear(W, 2L) -> split s|t -> s = clamp*0.636*atan(s) -> y = (x - t)*exp(-s)
//                            -> (subnet 2) cat + PermuteRandom^-1 gather -> (last block) FixedLinearTransform^-1,
//                            [:, :ndof], clamp_to_joint_limits   (ikflow_solver.py:99-102)
//
// All arithmetic is fp32 (ikflow/config.py:8); the MFMA used is the exact-f32 one (bitwise an fmaf chain).
#include <type_traits>

#include "ikf_internal.h"

namespace ikf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// first Linear + LeakyReLU
// ---------------------------------------------------------------------------------------------------------------
constexpr int FIRST_ROWS_PER_BLOCK = 16;

template <int IN>
__global__ __launch_bounds__(256) void k_first_layer(const float* __restrict__ w_t, const float* __restrict__ w_soft,
                                                     const float* __restrict__ bias, const float* __restrict__ x_in,
                                                     int D, int x_off, int n_x, PoseSource ps, long long row0,
                                                     long long rows, int width, float slope,
                                                     float* __restrict__ h_out) {
  const long long r_begin = (long long)blockIdx.x * FIRST_ROWS_PER_BLOCK;
  const long long r_end = (r_begin + FIRST_ROWS_PER_BLOCK < rows) ? r_begin + FIRST_ROWS_PER_BLOCK : rows;
  const int n4 = width >> 2;
  for (int c4 = threadIdx.x; c4 < n4; c4 += blockDim.x) {
    float4 w[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) w[k] = reinterpret_cast<const float4*>(w_t + (size_t)k * width)[c4];
    float4 b = reinterpret_cast<const float4*>(bias)[c4];
    if (ps.softflow != 0.0f) {
      const float4 ws = reinterpret_cast<const float4*>(w_soft)[c4];
      b.x = fmaf(ps.softflow, ws.x, b.x);
      b.y = fmaf(ps.softflow, ws.y, b.y);
      b.z = fmaf(ps.softflow, ws.z, b.z);
      b.w = fmaf(ps.softflow, ws.w, b.w);
    }
    for (long long r = r_begin; r < r_end; ++r) {
      // block-uniform addresses: the compiler turns these into scalar loads
      const long long gr = row0 + r;
      const long long pm = gr % ps.n_mod;
      const long long pi = ps.idx ? (long long)ps.idx[pm] : pm;
      const float* pose = ps.poses + pi * ps.stride;
      const float* xr = x_in + (size_t)r * D + x_off;
      float4 acc = b;
#pragma unroll
      for (int k = 0; k < IN; ++k) {
        const float u = (k < n_x) ? xr[k < n_x ? k : 0] : pose[k - n_x];
        acc.x = fmaf(u, w[k].x, acc.x);
        acc.y = fmaf(u, w[k].y, acc.y);
        acc.z = fmaf(u, w[k].z, acc.z);
        acc.w = fmaf(u, w[k].w, acc.w);
      }
      acc.x = acc.x > 0.f ? acc.x : acc.x * slope;
      acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
      acc.z = acc.z > 0.f ? acc.z : acc.z * slope;
      acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
      reinterpret_cast<float4*>(h_out + (size_t)r * width)[c4] = acc;
    }
  }
}

hipError_t launch_first_layer(const SubnetWeights& w, const FlowDims& d, const float* x_in, int x_off,
                              const PoseSource& ps, long long row0, long long rows, float* h_out, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const int in_real = w.n_x + d.n_pose;
  const unsigned grid = (unsigned)((rows + FIRST_ROWS_PER_BLOCK - 1) / FIRST_ROWS_PER_BLOCK);
  const int threads = (d.width / 4 >= 256) ? 256 : ((d.width / 4 + 63) / 64) * 64;
#define IKF_FIRST_CASE(IN)                                                                                       \
  case IN:                                                                                                       \
    hipLaunchKernelGGL((k_first_layer<IN>), dim3(grid), dim3(threads), 0, s, w.w_first_t, w.w_soft, w.b_first,    \
                       x_in, d.D, x_off, w.n_x, ps, row0, rows, d.width, d.slope, h_out);                         \
    break;
  switch (in_real) {
    IKF_FIRST_CASE(8)
    IKF_FIRST_CASE(9)
    IKF_FIRST_CASE(10)
    IKF_FIRST_CASE(11)
    IKF_FIRST_CASE(12)
    IKF_FIRST_CASE(13)
    IKF_FIRST_CASE(14)
    IKF_FIRST_CASE(15)
    default:
      return hipErrorInvalidValue;
  }
#undef IKF_FIRST_CASE
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// hidden Linear + LeakyReLU:  C[m][n] = lrelu( sum_k A[m][k] * W[n][k] + bias[n] )
//
// Both operands are K-contiguous, so A and W tiles are staged identically: global float4 -> ds_write_b128 into a
// [rows][BK+4] LDS image (the +4 pad makes the ds_read_b128 fragment reads conflict-free: row stride 36 dwords),
// double buffered, one barrier per K tile.  Fragments: lane l reads 4 consecutive k of row (l&31) at k-offset
// 4*(l>>5); component c of that float4 feeds MFMA c, i.e. MFMA c contracts k = {k0+c, k0+4+c} - a permutation of
// the k order that A and W share, so the contraction is unchanged.
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void k_gemm_lrelu(const float* __restrict__ A,
                                                                       const float* __restrict__ W,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ C, int M, int N, int K,
                                                                       float slope) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int LDK = BK + 4;
  constexpr int KQ = BK / 4;                 // float4 per tile row
  constexpr int A_F4 = BM * KQ / NT;         // float4 per thread for the A tile
  constexpr int B_F4 = BN * KQ / NT;
  static_assert(BM * KQ % NT == 0 && BN * KQ % NT == 0, "tile/threads mismatch");
  static_assert(BK % 8 == 0, "BK must be a multiple of 8");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][BM][LDK]
  float* sB = smem + 2 * BM * LDK;           // [2][BN][LDK]

  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of logical tiles
  // (consecutive tiles share the A row panel) - speed only, any mapping is correct.
  const int tiles_n = N / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // global staging: thread t owns float4 (row_t + i*RS, kq_t) of each tile, i = 0..F4-1
  constexpr int RS = NT / KQ;  // tile rows covered by one pass of the workgroup
  static_assert(NT % KQ == 0, "threads must cover whole tile rows");
  const int row_t = t / KQ, kq_t = (t % KQ) * 4;
  floatx4 ra[A_F4], rb[B_F4];
  const float* a_base = A + kq_t;
  const float* b_base = W + (size_t)(n0 + row_t) * K + kq_t;
  const int lds_t = row_t * LDK + kq_t;

  const int KT = K / BK;
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    int gr = m0 + row_t + i * RS;
    gr = gr < M ? gr : M - 1;  // rows past M are computed on a clamped row and never stored
    ra[i] = *reinterpret_cast<const floatx4*>(a_base + (size_t)gr * K);
  }
#pragma unroll
  for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const floatx4*>(b_base + (size_t)(i * RS) * K);
#pragma unroll
  for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sA + lds_t + i * RS * LDK) = ra[i];
#pragma unroll
  for (int i = 0; i < B_F4; ++i) *reinterpret_cast<floatx4*>(sB + lds_t + i * RS * LDK) = rb[i];
  __syncthreads();

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < KT);
    if (more) {
      const int koff = (kt + 1) * BK;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        int gr = m0 + row_t + i * RS;
        gr = gr < M ? gr : M - 1;
        ra[i] = *reinterpret_cast<const floatx4*>(a_base + (size_t)gr * K + koff);
      }
#pragma unroll
      for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const floatx4*>(b_base + (size_t)(i * RS) * K + koff);
    }
    const float* cA = sA + cur * BM * LDK + (wm + frag_row) * LDK + frag_k;
    const float* cB = sB + cur * BN * LDK + (wn + frag_row) * LDK + frag_k;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      floatx4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const floatx4*>(cA + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const floatx4*>(cB + j * 32 * LDK + kk * 8);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more) {
      float* nA = sA + (cur ^ 1) * BM * LDK;
      float* nB = sB + (cur ^ 1) * BN * LDK;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(nA + lds_t + i * RS * LDK) = ra[i];
#pragma unroll
      for (int i = 0; i < B_F4; ++i) *reinterpret_cast<floatx4*>(nB + lds_t + i * RS * LDK) = rb[i];
    }
    __syncthreads();
  }

  // epilogue: bias + LeakyReLU, direct unpredicated stores (each half-wave writes 128 contiguous bytes per register).
  // A per-store `row < M` predicate makes hipcc put an s_waitcnt vmcnt(0) in front of EVERY store (stores count in
  // vmcnt on gfx950), serialising 64 stores per wave - so the output buffer is required to be row-padded instead.
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn + j * 32 + col_l;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        float v = acc[i][j][r] + bv;
        v = v > 0.f ? v : v * slope;
        C[(size_t)row * N + col] = v;  // C has its rows padded to a multiple of BM (engine scratch): no predicate
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form of the same contraction: 3 LDS stages, ONE barrier per K tile placed in the middle of the
// tile's MFMA stream, fragments for the next k-group (and for the next tile's first k-group) are read while the
// current group's MFMAs issue.  With one wave per SIMD and 64-cycle MFMAs, everything that is not an MFMA (global
// loads for tile kt+2, ds_write of tile kt+1, the barrier, ds_read of the next fragments) sits in the shadow of >= 16
// queued-up MFMAs, so the matrix pipe never drains inside the K loop.
//   hazards: tile kt+1 is written to stage (kt+1)%3 during iteration kt; that stage last held tile kt-2, whose
//   fragment reads ended in iteration kt-2 and are separated from these writes by iteration kt-1's barrier.  The
//   reads of tile kt+1's first fragments come after iteration kt's barrier, i.e. after every wave's writes.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void k_gemm_lrelu_p3(const float* __restrict__ A,
                                                                          const float* __restrict__ W,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ C, int M, int N, int K,
                                                                          float slope) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int LDK = BK + 4;
  constexpr int KQ = BK / 4;
  constexpr int A_F4 = BM * KQ / NT;
  constexpr int B_F4 = BN * KQ / NT;
  constexpr int RS = NT / KQ;
  constexpr int NKK = BK / 8;  // k-groups per tile (each = one float4 fragment per 32-row block = 4 MFMAs per (i,j))
  constexpr int STAGE = (BM + BN) * LDK;
  static_assert(BM * KQ % NT == 0 && BN * KQ % NT == 0 && NT % KQ == 0, "tile/threads mismatch");
  static_assert(NKK % 2 == 0 && NKK >= 2, "fragment double-buffering needs an even number of k-groups");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][BM + BN][LDK]

  const int tiles_n = N / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int row_t = t / KQ, kq_t = (t % KQ) * 4;
  const float* a_src[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    int gr = m0 + row_t + i * RS;
    gr = gr < M ? gr : M - 1;
    a_src[i] = A + (size_t)gr * K + kq_t;
  }
  const float* b_base = W + (size_t)(n0 + row_t) * K + kq_t;
  const int lds_t = row_t * LDK + kq_t;
  const int fragA = (wm + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int fragB = BM * LDK + (wn + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int KT = K / BK;

  floatx4 ra[A_F4], rb[B_F4];
  floatx4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];

#define IKF_GLOAD(koff)                                                                                           \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const floatx4*>(a_src[i] + (koff)); \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                                              \
        rb[i] = *reinterpret_cast<const floatx4*>(b_base + (size_t)(i * RS) * K + (koff));                        \
  }
#define IKF_LSTORE(stage)                                                                                         \
  {                                                                                                               \
    float* sp_ = smem + (stage) * STAGE + lds_t;                                                                  \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sp_ + i * RS * LDK) = ra[i];      \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                                              \
        *reinterpret_cast<floatx4*>(sp_ + BM * LDK + i * RS * LDK) = rb[i];                                       \
  }
#define IKF_FRAG(FA, FB, stage, kk)                                                                               \
  {                                                                                                               \
    const float* sp_ = smem + (stage) * STAGE + (kk) * 8;                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) FA[i] = *reinterpret_cast<const floatx4*>(sp_ + fragA + i * 32 * LDK); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) FB[j] = *reinterpret_cast<const floatx4*>(sp_ + fragB + j * 32 * LDK); \
  }
#define IKF_MFMA4(FA, FB)                                                                                         \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) {                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].x, FB[j].x, acc[i][j], 0, 0, 0);                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].y, FB[j].y, acc[i][j], 0, 0, 0);                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].z, FB[j].z, acc[i][j], 0, 0, 0);                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].w, FB[j].w, acc[i][j], 0, 0, 0);                     \
    }                                                                                                             \
  }

  // One K tile.  HAS1/HAS2 (tile kt+1 / kt+2 exist) are compile-time so the steady-state body is branch-free: one
  // scheduling region before the barrier and one after it, in which the sched_group_barrier sequence below pins a
  // k-MFMA : 1-memory-op interleave (all ds_write / global_load / ds_read sit in MFMA shadows).
  auto k_tile = [&](auto has1_c, auto has2_c, int kt, int cur, int nxt) {
    constexpr bool HAS1 = decltype(has1_c)::value, HAS2 = decltype(has2_c)::value;
    // ---- first half: k-groups 0 .. NKK/2-1; stage tile kt+1 into LDS; fetch tile kt+2 into registers
#pragma unroll
    for (int kk = 0; kk < NKK / 2; ++kk) {
      if (kk & 1) { IKF_FRAG(fa0, fb0, cur, kk + 1) } else { IKF_FRAG(fa1, fb1, cur, kk + 1) }
      if (kk & 1) { IKF_MFMA4(fa1, fb1) } else { IKF_MFMA4(fa0, fb0) }
      if (kk == 0) {
        if (HAS1) IKF_LSTORE(nxt)
        if (HAS2) IKF_GLOAD((kt + 2) * BK)
      }
    }
    {
      constexpr int n_mfma = (NKK / 2) * MI * NI * 4;
      constexpr int n_mem = (HAS1 ? A_F4 + B_F4 : 0) + (HAS2 ? A_F4 + B_F4 : 0) + (NKK / 2) * (MI + NI);
      constexpr int per = n_mfma / (n_mem > 0 ? n_mem : 1) > 0 ? n_mfma / (n_mem > 0 ? n_mem : 1) : 1;
      // order: first fragment prefetch, LDS writes of tile kt+1, remaining fragment prefetches, then the global
      // loads of tile kt+2 last - so the lgkmcnt(0) in front of the barrier only waits on long-finished LDS ops
#pragma unroll
      for (int i = 0; i < MI + NI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (HAS1) {
#pragma unroll
        for (int i = 0; i < A_F4 + B_F4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < (NKK / 2 - 1) * (MI + NI); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (HAS2) {
#pragma unroll
        for (int i = 0; i < A_F4 + B_F4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    __syncthreads();  // tile kt+1 is now visible to every wave
    // ---- second half: k-groups NKK/2 .. NKK-1; the last one prefetches the next tile's first k-group
#pragma unroll
    for (int kk = NKK / 2; kk < NKK; ++kk) {
      if (kk + 1 < NKK) {
        if (kk & 1) { IKF_FRAG(fa0, fb0, cur, kk + 1) } else { IKF_FRAG(fa1, fb1, cur, kk + 1) }
      } else if (HAS1) {
        IKF_FRAG(fa0, fb0, nxt, 0)
      }
      if (kk & 1) { IKF_MFMA4(fa1, fb1) } else { IKF_MFMA4(fa0, fb0) }
    }
    {
      constexpr int n_mfma = (NKK - NKK / 2) * MI * NI * 4;
      constexpr int n_rd = ((NKK - NKK / 2 - 1) + (HAS1 ? 1 : 0)) * (MI + NI);
      constexpr int per = n_rd > 0 ? (n_mfma / n_rd > 0 ? n_mfma / n_rd : 1) : n_mfma;
#pragma unroll
      for (int i = 0; i < n_rd; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  // prologue: tile 0 -> stage 0, tile 1 -> registers
  IKF_GLOAD(0)
  IKF_LSTORE(0)
  if (KT > 1) IKF_GLOAD(BK)
  __syncthreads();
  IKF_FRAG(fa0, fb0, 0, 0)

  int cur = 0, kt = 0;
  for (; kt + 2 < KT; ++kt) {
    const int nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, T_{}, kt, cur, nxt);
    cur = nxt;
  }
  if (kt + 1 < KT) {
    const int nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, F_{}, kt, cur, nxt);
    cur = nxt;
    ++kt;
  }
  k_tile(F_{}, F_{}, kt, cur, cur);
#undef IKF_GLOAD
#undef IKF_LSTORE
#undef IKF_FRAG
#undef IKF_MFMA4

  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn + j * 32 + col_l;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        float v = acc[i][j][r] + bv;
        v = v > 0.f ? v : v * slope;
        C[(size_t)row * N + col] = v;  // C has its rows padded to a multiple of BM (engine scratch): no predicate
      }
    }
  }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static hipError_t launch_gemm_p3(const float* A, const float* W, const float* bias, float* C, long long M, int N, int K,
                                 float slope, hipStream_t s) {
  if (N % BN != 0 || K % BK != 0) return hipErrorInvalidValue;
  constexpr int LDK = BK + 4;
  constexpr size_t smem = (size_t)3 * (BM + BN) * LDK * sizeof(float);
  auto kern = k_gemm_lrelu_p3<BM, BN, BK, WAVES_M, WAVES_N>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long tiles_m = (M + BM - 1) / BM;
  const long long grid = tiles_m * (N / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES_M * WAVES_N * 64), smem, s, A, W, bias, C, (int)M, N, K,
                     slope);
  return hipGetLastError();
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static hipError_t launch_gemm_t(const float* A, const float* W, const float* bias, float* C, long long M, int N, int K,
                                float slope, hipStream_t s) {
  if (N % BN != 0 || K % BK != 0) return hipErrorInvalidValue;
  constexpr int LDK = BK + 4;
  constexpr size_t smem = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  auto kern = k_gemm_lrelu<BM, BN, BK, WAVES_M, WAVES_N>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long tiles_m = (M + BM - 1) / BM;
  const long long grid = tiles_m * (N / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES_M * WAVES_N * 64), smem, s, A, W, bias, C, (int)M, N, K,
                     slope);
  return hipGetLastError();
}

int gemm_variant_count() { return 9; }
const char* gemm_kernel_name() { return "k_gemm_lrelu"; }

hipError_t launch_gemm_lrelu(int variant, const float* A, const float* W, const float* bias, float* C, long long M,
                             int N, int K, float slope, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  switch (variant) {
    case 0:  // 128x128 tile, 4 waves of 64x64, BK 32: 256 tiles at M=4096,N=1024 = one per CU
      return launch_gemm_t<128, 128, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 1:  // 128x128, 8 waves of 64x32
      return launch_gemm_t<128, 128, 32, 2, 4>(A, W, bias, C, M, N, K, slope, s);
    case 2:  // 128x64, 4 waves of 64x32: 512 tiles, two de-phased workgroups per CU
      return launch_gemm_t<128, 64, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 3:  // 128x128, BK 64
      return launch_gemm_t<128, 128, 64, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 4:  // 64x64 tile, 4 waves of 32x32: small batches (M=512 -> 128 tiles)
      return launch_gemm_t<64, 64, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 5:  // 64x128
      return launch_gemm_t<64, 128, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 6:  // pipelined 128x128, 4 waves
      return launch_gemm_p3<128, 128, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    case 7:  // pipelined 128x128, 8 waves
      return launch_gemm_p3<128, 128, 32, 2, 4>(A, W, bias, C, M, N, K, slope, s);
    case 8:  // pipelined 128x64 (two workgroups per CU)
      return launch_gemm_p3<128, 64, 32, 2, 2>(A, W, bias, C, M, N, K, slope, s);
    default:
      return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// last Linear + affine-coupling inverse (+ permutation, + final rescale/clamp)
// one wave per row; lane l owns k = 4*(64*g + l) .. +3 of the hidden row (the unfused pipeline: n_hidden == 1 and the
// forced gemm variants 0..8 - the fused pipeline reduces the last Linear inside the last contraction instead)
// ---------------------------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(256) void k_last_layer_coupling(const float* __restrict__ w_last,
                                                             const float* __restrict__ b_last,
                                                             const float* __restrict__ h, FlowDims d, CouplingArgs ca,
                                                             long long rows) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const int width = d.width;
  const int G = width >> 8;  // 256 hidden units per pass of the wave (4 per lane)

  const int D = d.D, L1 = d.L1, L2 = d.L2;
  const int nl = (ca.which == 1) ? L2 : L1;  // number of (s,t) pairs this subnet emits

  for (long long row = wave0; row < rows; row += nwaves) {
    float a[OUT];
#pragma unroll
    for (int j = 0; j < OUT; ++j) a[j] = 0.f;
    for (int g = 0; g < G; ++g) {
      const float4 hv = reinterpret_cast<const float4*>(h + (size_t)row * width)[g * 64 + lane];
#pragma unroll
      for (int j = 0; j < OUT; ++j) {
        const float4 w = reinterpret_cast<const float4*>(w_last + (size_t)j * width)[g * 64 + lane];  // L2-resident
        float sacc = a[j];
        sacc = fmaf(hv.x, w.x, sacc);
        sacc = fmaf(hv.y, w.y, sacc);
        sacc = fmaf(hv.z, w.z, sacc);
        sacc = fmaf(hv.w, w.w, sacc);
        a[j] = sacc;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
      for (int j = 0; j < OUT; ++j) a[j] += __shfl_xor(a[j], off, 64);

    // lane j < nl takes (s_j, t_j) = (a[j], a[nl + j]) + bias
    float sv = 0.f, tv = 0.f;
#pragma unroll
    for (int j = 0; j < OUT; ++j) {
      const float aj = a[j] + b_last[j];
      if (j == lane) sv = aj;
      if (j == lane + nl) tv = aj;
    }
    // FrEIA: s = clamp * (0.636 * atan(s));  y = (x - t) * exp(-s)
    const float s_cl = d.clamp * (0.636f * atanf(sv));
    const float e = expf(-s_cl);

    if (ca.which == 1) {
      float xv = 0.f;
      if (lane < D) xv = ca.x_in[(size_t)row * D + lane];
      // lanes L1..D-1 hold x2; the (s,t) for x2[j] sit in lane j -> fetch from lane (lane - L1)
      const int src = (lane >= L1 && lane < D) ? lane - L1 : 0;
      const float t_j = __shfl(tv, src, 64);
      const float e_j = __shfl(e, src, 64);
      float outv = xv;  // x1 is carried through unchanged
      if (lane >= L1 && lane < D) outv = (xv - t_j) * e_j;
      if (lane < D) ca.x_out[(size_t)row * D + lane] = outv;
    } else {
      // state row = [x1 | y2]; y1 = (x1 - t2) * exp(-s2) on lanes < L1
      float xv = 0.f;
      if (lane < D) xv = ca.x_out[(size_t)row * D + lane];
      float cat = xv;
      if (lane < L1) cat = (xv - tv) * e;
      // PermuteRandom rev: out[:, d] = cat[:, perm_inv[d]]
      const int src = (lane < D) ? ca.perm_inv[lane] : 0;
      const float v = __shfl(cat, src, 64);
      if (!ca.is_final) {
        if (lane < D) ca.x_out[(size_t)row * D + lane] = v;
      } else {
        // FixedLinearTransform rev: (x - b).mm(M_inv); then [:, :ndof] and clamp_to_joint_limits
        const float vs = ca.sigmoid ? 1.0f / (1.0f + expf(-v)) : v;  // InvertibleSigmoidFlipped rev
        const float xm = (lane < D) ? vs - ca.b_lin[lane] : 0.f;
        float q = 0.f;
        const int jcol = lane < D ? lane : 0;
        for (int k = 0; k < D; ++k) q = fmaf(__shfl(xm, k, 64), ca.M_inv[k * D + jcol], q);
        if (lane < d.ndof) {
          if (ca.clamp_limits) q = fminf(fmaxf(q, ca.lo[lane]), ca.hi[lane]);
          ca.q_out[(size_t)row * d.ndof + lane] = q;
        }
      }
    }
  }
}

template <int OUT>
static hipError_t launch_last_g(const SubnetWeights& w, const FlowDims& d, const float* h_in, const CouplingArgs& ca,
                                long long rows, hipStream_t s) {
  long long waves = (rows + 3) / 4;  // ~4 rows per wave
  if (waves < 1) waves = 1;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  hipLaunchKernelGGL((k_last_layer_coupling<OUT>), dim3(grid), dim3(256), 0, s, w.w_last, w.b_last, h_in, d, ca, rows);
  return hipGetLastError();
}

hipError_t launch_last_layer_coupling(const SubnetWeights& w, const FlowDims& d, const float* h_in,
                                      const CouplingArgs& ca, long long rows, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (d.width % 256 != 0) return hipErrorInvalidValue;
  switch (w.n_out) {
    case 2: return launch_last_g<2>(w, d, h_in, ca, rows, s);
    case 4: return launch_last_g<4>(w, d, h_in, ca, rows, s);
    case 6: return launch_last_g<6>(w, d, h_in, ca, rows, s);
    case 8: return launch_last_g<8>(w, d, h_in, ca, rows, s);
    case 10: return launch_last_g<10>(w, d, h_in, ca, rows, s);
    case 12: return launch_last_g<12>(w, d, h_in, ca, rows, s);
    case 14: return launch_last_g<14>(w, d, h_in, ca, rows, s);
    case 16: return launch_last_g<16>(w, d, h_in, ca, rows, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ikf
