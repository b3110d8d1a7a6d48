// Hidden Linear + LeakyReLU on the f16 matrix cores with an error-compensated operand split (gfx950).
//
//   a = hi + lo/2048,  hi = f16(a),  lo = f16((a - hi)*2048)    (activations split by the producing kernel's epilogue,
//                                                               weights split once at load time)
//   sum_k a_k w_k ~= sum hi_a hi_w  +  (sum hi_a lo_w + sum lo_a hi_w) / 2048
//
// i.e. three v_mfma_f32_32x32x16_f16 (fp32 accumulate) instead of eight v_mfma_f32_32x32x2_f32 per 16 k: 5.3x less
// matrix-pipe time.  The dropped lo*lo term is 2^-22 relative; on hardware the split result is CLOSER to fp64 than the
// exact-f32 MFMA chain (rms 1.5e-7 vs 4.1e-7 at K = 1024, tools/split_probe.hip) because each MFMA folds 16 products
// into the accumulator with one rounding.  Range: |activation| must stay below 65504 (f16); hidden activations of
// the IKFlow subnets are O(1..100).
//
// Same role, tile and pipeline as k_flow_gemm (flow_fused.hip): 128x128 tile, 4 waves of 64x64, 3 LDS stages of 32 k,
// one barrier per stage placed mid-stage, global loads two stages ahead.  A stage row is the 128-byte split-32 line
// [32 hi | 32 lo] (+16 B pad -> conflict-free ds_read_b128); a fragment read of 16 B = 8 halves = one MFMA operand.
#include <type_traits>

#include "ikf_internal.h"

namespace ikf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifdef IKF_TRACE
extern __device__ unsigned long long* ikf_trace_buf;  // defined in flow_fused.hip (probe build)
#define IKS_TSTAMP(i) if (threadIdx.x == 0 && ikf_trace_buf) ikf_trace_buf[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
#else
#define IKS_TSTAMP(i)
#endif

// tile configurations (chosen by row count so that small batches still spread over the chip):
//   0: 128x128, 4 waves of 64x64   1: 64x128, 4 waves of 32x64   2: 64x64, 4 waves of 32x32   3: 32x64, 2 waves of 32x32
template <int CFG> struct SplitCfg;
template <> struct SplitCfg<0> { static constexpr int BM = 128, BN = 128, WAVES_M = 2, WAVES_N = 2; };
template <> struct SplitCfg<1> { static constexpr int BM = 64, BN = 128, WAVES_M = 2, WAVES_N = 2; };
template <> struct SplitCfg<2> { static constexpr int BM = 64, BN = 64, WAVES_M = 2, WAVES_N = 2; };
template <> struct SplitCfg<3> { static constexpr int BM = 32, BN = 64, WAVES_M = 1, WAVES_N = 2; };
constexpr int kNumSplitCfg = 4;
constexpr int kSplitSkinnyCfg = 4, kSplitSkinny32Cfg = 6;  // k_split_skinny<.., 2> / <.., 1> (same ids as the f32 pipeline)
// tile rows per config: {128, 64, 64, 32}
static const int kSplitBN[kNumSplitCfg] = {128, 128, 64, 64};

template <bool EPI_RED, int CFG>
__global__ __launch_bounds__(SplitCfg<CFG>::WAVES_M* SplitCfg<CFG>::WAVES_N * 64) void k_split_gemm(SplitGemmArgs g) {
  using TC = SplitCfg<CFG>;
  constexpr int BM = TC::BM, BN = TC::BN, SWAVES_M = TC::WAVES_M, SWAVES_N = TC::WAVES_N;
  constexpr int NT = SWAVES_M * SWAVES_N * 64;
  constexpr int WM = BM / SWAVES_M, WN = BN / SWAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int LDK = 36;            // dwords per stage row: 128 B line + 16 B pad
  constexpr int KQ = 8;              // 16-byte slots per line: 0..3 hi (k 0-7, 8-15, 16-23, 24-31), 4..7 lo
  constexpr int A_F4 = BM * KQ / NT;
  constexpr int B_F4 = BN * KQ / NT;
  constexpr int RS = NT / KQ;
  constexpr int STAGE = (BM + BN) * LDK;
  constexpr int LDT = BN + 4;
  static_assert(3 * STAGE >= (BM + 32) * LDT, "epilogue tile must fit in the stage area");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][BM + BN][LDK]

  const int M = g.M, N = g.N, K = g.K;
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
  const int wm = (wave / SWAVES_N) * WM, wn = (wave % SWAVES_N) * WN;

  floatx16 am[MI][NI], ac[MI][NI];  // hi*hi sums, (hi*lo + lo*hi) sums
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { am[i][j][r] = 0.f; ac[i][j][r] = 0.f; }

  const int row_t = t / KQ, kq_t = (t % KQ) * 4;  // dword offset of this thread's 16-B slot in a line
  // buffer loads (SGPR descriptor + loop-invariant 32-bit byte offset + scalar k offset), as in k_flow_gemm
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(g.A)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(g.W)), 0, 0x7fffffff, 0x00020000);
  unsigned a_off[A_F4], b_off[B_F4];  // a split-32 row is K dwords long; K tile kt starts at dword 32*kt
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    int gr = m0 + row_t + i * RS;
    gr = gr < M ? gr : M - 1;
    a_off[i] = ((unsigned)gr * (unsigned)K + kq_t) * 4u;
  }
#pragma unroll
  for (int i = 0; i < B_F4; ++i) b_off[i] = ((unsigned)(n0 + row_t + i * RS) * (unsigned)K + kq_t) * 4u;
#define IKS_BLD(rs, voff, kt_) \
  __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane((kt_) * 128), 0))
  const int lds_t = row_t * LDK + kq_t;
  // fragment of k16-step s, plane p (0 hi, 1 lo): slot p*4 + s*2 + (lane>>5)
  const int fragA = (wm + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int fragB = BM * LDK + (wn + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int KT = K / 32;

  floatx4 ra[2][A_F4], rb[2][B_F4];
  half8 ah0[MI], al0[MI], bh0[NI], bl0[NI], ah1[MI], al1[MI], bh1[NI], bl1[NI];

#define IKS_GLOAD(S, kt_)                                                                                          \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) ra[S][i] = IKS_BLD(rsA, a_off[i], kt_);                       \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i) rb[S][i] = IKS_BLD(rsW, b_off[i], kt_);                       \
  }
#define IKS_LSTORE(S, stage)                                                                                      \
  {                                                                                                               \
    float* sp_ = smem + (stage) * STAGE + lds_t;                                                                  \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sp_ + i * RS * LDK) = ra[S][i];   \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                                              \
        *reinterpret_cast<floatx4*>(sp_ + BM * LDK + i * RS * LDK) = rb[S][i];                                    \
  }
#define IKS_FRAG(AH, AL, BH, BL, stage, s_)                                                                        \
  {                                                                                                               \
    const float* sp_ = smem + (stage) * STAGE + (s_) * 8;                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                              \
      AH[i] = *reinterpret_cast<const half8*>(sp_ + fragA + i * 32 * LDK);                                        \
      AL[i] = *reinterpret_cast<const half8*>(sp_ + fragA + i * 32 * LDK + 16);                                   \
    }                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                              \
      BH[j] = *reinterpret_cast<const half8*>(sp_ + fragB + j * 32 * LDK);                                        \
      BL[j] = *reinterpret_cast<const half8*>(sp_ + fragB + j * 32 * LDK + 16);                                   \
    }                                                                                                             \
  }
#define IKS_MFMA3(AH, AL, BH, BL)                                                                                  \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)                  \
      am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], am[i][j], 0, 0, 0);                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)                  \
      ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], ac[i][j], 0, 0, 0);                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)                  \
      ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], ac[i][j], 0, 0, 0);                         \
  }

  // One stage (32 k = two k16-steps). Step 0 fragments are already in set 0; step 0's MFMAs shadow the LDS write of
  // tile kt+1, the global loads of tile kt+3 and the fragment reads of step 1; after the barrier step 1's MFMAs shadow
  // the fragment reads of the next stage's step 0.
  auto k_tile = [&](auto has1_c, auto has3_c, auto set_c, int kt, int cur, int nxt) {
    constexpr bool HAS1 = decltype(has1_c)::value, HAS3 = decltype(has3_c)::value;
    constexpr int S = decltype(set_c)::value;
    IKS_FRAG(ah1, al1, bh1, bl1, cur, 1)
    IKS_MFMA3(ah0, al0, bh0, bl0)
#if !defined(IKS_ABLATE) || (IKS_ABLATE != 2 && IKS_ABLATE != 4)
    if (HAS1) IKS_LSTORE(S, nxt)
#endif
#if !defined(IKS_ABLATE) || (IKS_ABLATE != 2 && IKS_ABLATE != 5)
    if (HAS3) IKS_GLOAD(S, kt + 3)
#endif
    {
      constexpr int n_mfma = MI * NI * 3;
      constexpr int n_rd = 2 * (MI + NI), n_wr = HAS1 ? A_F4 + B_F4 : 0, n_ld = HAS3 ? A_F4 + B_F4 : 0;
      // 12 MFMAs, 8 reads, 8 writes, 8 loads: reads first (needed right after the barrier), then writes, then loads
#pragma unroll
      for (int i = 0; i < n_mfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int third = n_mfma / 3 > 0 ? n_mfma / 3 : 1;
        if (i < third) __builtin_amdgcn_sched_group_barrier(0x100, (n_rd + third - 1) / third, 0);
        else if (i < 2 * third) { if (n_wr > 0) __builtin_amdgcn_sched_group_barrier(0x200, (n_wr + third - 1) / third, 0); }
        else { if (n_ld > 0) __builtin_amdgcn_sched_group_barrier(0x020, (n_ld + third - 1) / third, 0); }
      }
    }
    __syncthreads();
    if (HAS1) IKS_FRAG(ah0, al0, bh0, bl0, nxt, 0)
    IKS_MFMA3(ah1, al1, bh1, bl1)
    {
      constexpr int n_mfma = MI * NI * 3;
      constexpr int n_rd = HAS1 ? 2 * (MI + NI) : 0;
#pragma unroll
      for (int i = 0; i < n_mfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int span = (2 * n_mfma) / 3 > 0 ? (2 * n_mfma) / 3 : 1;
        if (i < span && n_rd > 0) __builtin_amdgcn_sched_group_barrier(0x100, (n_rd + span - 1) / span, 0);
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;

  IKS_TSTAMP(0)
  {
    floatx4 ra0[A_F4], rb0[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) ra0[i] = IKS_BLD(rsA, a_off[i], 0);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb0[i] = IKS_BLD(rsW, b_off[i], 0);
    if (KT > 1) IKS_GLOAD(1, 1)
    if (KT > 2) IKS_GLOAD(0, 2)
    float* sp0 = smem + lds_t;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sp0 + i * RS * LDK) = ra0[i];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) *reinterpret_cast<floatx4*>(sp0 + BM * LDK + i * RS * LDK) = rb0[i];
  }
  __syncthreads();
  IKS_TSTAMP(1)
  IKS_FRAG(ah0, al0, bh0, bl0, 0, 0)

  // Every stage runs the SAME two code bodies (register-set parity): near the end the prefetch index is clamped to the
  // last tile (a redundant, harmless load + LDS write into a stage nobody reads again) instead of switching to
  // specialised tail code - cold tail instantiations cost ~1000 cycles each in instruction-cache misses.
  int cur = 0;
  for (int kt = 0; kt < KT; kt += 2) {  // KT is even (K % 64 == 0, checked by the launcher)
    int nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, T_{}, S1_{}, (kt + 3 < KT ? kt : KT - 4), cur, nxt);
    cur = nxt;
    nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, T_{}, S0_{}, (kt + 4 < KT ? kt + 1 : KT - 4), cur, nxt);
    cur = nxt;
#ifdef IKF_TRACE
    if ((kt & 3) == 2 && kt < 128) IKS_TSTAMP(2 + (kt >> 2))
#endif
  }
  IKS_TSTAMP(40)
#undef IKS_GLOAD
#undef IKS_BLD
#undef IKS_LSTORE
#undef IKS_FRAG
#undef IKS_MFMA3

  // ---- epilogue: v = hi*hi + corr/2048 + bias, LeakyReLU, into the LDS tile T[BM][LDT] (fp32)
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  __syncthreads();  // every wave is done reading the last stage
  float* T = smem;
  constexpr float inv_scale = 1.0f / IKF_SPLIT_SCALE;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int cl = wn + j * 32 + col_l;
    const float bv = g.bias[n0 + cl];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        float v = fmaf(ac[i][j][r], inv_scale, am[i][j][r]) + bv;
        v = v > 0.f ? v : v * g.slope;
        T[rl * LDT + cl] = v;
      }
    }
  }
  if constexpr (!EPI_RED) {
    // re-split and store the tile as split-32 lines: a thread converts 8 consecutive columns -> 8 hi (16 B) + 8 lo (16 B)
    __syncthreads();
    char* Cb = reinterpret_cast<char*>(g.C);
    constexpr int CH = BN / 8;  // 8-column chunks per row
    unsigned range_max = 0;
    for (int idx = t; idx < BM * CH; idx += NT) {
      const int rl = idx / CH, ch = idx - rl * CH;
      const floatx4 v0 = *reinterpret_cast<const floatx4*>(T + rl * LDT + ch * 8);
      const floatx4 v1 = *reinterpret_cast<const floatx4*>(T + rl * LDT + ch * 8 + 4);
      half8 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hi[q] = (_Float16)v0[q];
        lo[q] = (_Float16)((v0[q] - (float)hi[q]) * IKF_SPLIT_SCALE);
        hi[4 + q] = (_Float16)v1[q];
        lo[4 + q] = (_Float16)((v1[q] - (float)hi[4 + q]) * IKF_SPLIT_SCALE);
      }
      {
        typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
        const uint4_t hb = __builtin_bit_cast(uint4_t, hi);
        range_max = range_track(range_track(range_track(range_track(range_max, hb.x), hb.y), hb.z), hb.w);
      }
      const int col = n0 + ch * 8;  // global column of the chunk; block of 32 columns = one 128-B line
      char* p = Cb + (size_t)(m0 + rl) * N * 4 + (size_t)(col >> 5) * 128 + (col & 31) * 2;
      *reinterpret_cast<half8*>(p) = hi;
      *reinterpret_cast<half8*>(p + 64) = lo;
    }
    if (range_hit(range_max) && g.flag) atomicOr(g.flag, 1);
  } else {
    // last Linear restricted to this tile's columns, in exact f32 MFMA (identical to k_flow_gemm<true>): one slot per 64 columns
    float* Wl = smem + BM * LDT;
    for (int idx = t; idx < 32 * (BN / 4); idx += NT) {
      const int o = idx / (BN / 4), c4 = idx - o * (BN / 4);
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < g.n_out) v = *reinterpret_cast<const floatx4*>(g.w_last + (size_t)o * N + n0 + c4 * 4);
      *reinterpret_cast<floatx4*>(Wl + o * LDT + c4 * 4) = v;
    }
    __syncthreads();
    constexpr int RB = BM / 32, FKH = BN / 64, CW = 64;
    for (int job = wave; job < RB * FKH; job += NT / 64) {  // 8 (row block, column half) jobs over 4 waves
      const int rb = job % RB, kh = job / RB;
      floatx16 pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
      const float* pa = Wl + (lane & 31) * LDT + kh * CW + (lane >> 5) * 4;
      const float* pb = T + (rb * 32 + (lane & 31)) * LDT + kh * CW + (lane >> 5) * 4;
#pragma unroll
      for (int ks = 0; ks < CW / 8; ++ks) {
        const floatx4 a4 = *reinterpret_cast<const floatx4*>(pa + ks * 8);
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(pb + ks * 8);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, pacc, 0, 0, 0);
      }
      float* pout = g.P_out + (size_t)(n0 / CW + kh) * g.p_slot_stride + (size_t)(m0 + rb * 32 + (lane & 31)) * IKF_PSTRIDE;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + row_h;
        pout[o] = pacc[r];
      }
    }
  }
  IKS_TSTAMP(41)
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA form of the 128x128 tile (global_load_lds_dwordx4: 16 B per lane straight into LDS, no VGPR staging, no
// ds_write): 4 stages of 32 KB, loads three tiles ahead, counted vmcnt + raw s_barrier so the DMA stays in flight across
// the barrier.  A DMA wave-instruction writes 1 KB = 8 unpadded 128-B lines, so the conflict-free ds_read_b128 fragment
// pattern comes from a source-side XOR swizzle: physical 16-B slot p of line r holds logical slot p ^ ((r >> 1) & 7).
//   hazards: tile kt+3 is DMA'd into stage (kt+3)%4 (= the stage of tile kt-1) after iteration kt's barrier, i.e. after
//   every wave has finished iteration kt-1 and with it all reads of tile kt-1; tile kt+1 is read (fragment prefetch)
//   only after this wave's `vmcnt` wait for its own tile-(kt+1) DMAs AND the barrier that follows every wave's wait.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define IKD_WN 2
#define IKD_KERNEL k_split_gemm_dma
#include "flow_split_dma.inc"
#undef IKD_WN
#undef IKD_KERNEL
#define IKD_WN 4
#define IKD_KERNEL k_split_gemm_dma8
#include "flow_split_dma.inc"
#undef IKD_WN
#undef IKD_KERNEL

int g_split_dma_waves_n = 4;  // waves along N of the LDS-DMA kernel: 4 = 8 waves (default: -24 % shader cycles, -7..10 % wall time - the chip clocks down under it), 2 = 4 waves; probes may flip it
template <bool EPI_RED, int WN>
static hipError_t launch_sg_dma_w(const SplitGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)4 * 256 * 32 * sizeof(float);
  auto kern = WN == 4 ? k_split_gemm_dma8<EPI_RED> : k_split_gemm_dma<EPI_RED>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + 127) / 128) * (a.N / 128);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(2 * WN * 64), smem, s, a);
  return hipGetLastError();
}
template <bool EPI_RED>
static hipError_t launch_sg_dma(const SplitGemmArgs& a, hipStream_t s) {
  return g_split_dma_waves_n == 4 ? launch_sg_dma_w<EPI_RED, 4>(a, s) : launch_sg_dma_w<EPI_RED, 2>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (<= 512 rows) in the f16-split arithmetic: the structure of k_flow_gemm_skinny (flow_fused.hip) - tile
// 32 x (NH*32), BK = 128 k per stage, 8 k-slices per stage = NH*8 waves, A rows through two LDS stages, W fragments
// straight from a fragment-major image (k_wfrag_pack_split) by buffer loads, one barrier per stage, k-slice partial
// blocks summed through LDS in fixed order.  A wave's slice of a stage is exactly one k16 step: 2 x 16-byte fragment
// reads (hi, lo planes of its A rows), 2 x buffer_load_dwordx4 (hi, lo planes of its W rows), 3 MFMAs.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SKBM = 32, SKBK = 128, SKKS = 8;
template <int NH>
constexpr size_t split_skinny_lds() {
  return sizeof(float) * ((size_t)SKKS * NH * 16 * 64 + (size_t)(SKBM + 32) * (NH * 32 + 4)) > sizeof(float) * 2 * SKBM * (SKBK + 4)
             ? sizeof(float) * ((size_t)SKKS * NH * 16 * 64 + (size_t)(SKBM + 32) * (NH * 32 + 4))
             : sizeof(float) * 2 * SKBM * (SKBK + 4);
}

template <bool EPI_RED, int NH>
__global__ __launch_bounds__(NH * SKKS * 64) void k_split_skinny(SplitGemmArgs g) {
  constexpr int BM = SKBM, BN = NH * 32, BK = SKBK, NT = NH * SKKS * 64, KS = SKKS;
  constexpr int LDK = BK + 4;           // dwords per LDS row: four 128-B split-32 lines + 16 B pad
  constexpr int KQ4 = BK / 4;           // 16-byte slots per tile row
  constexpr int NFA = BM * KQ4 / NT;
  constexpr int STAGE = BM * LDK;
  constexpr int LDT = BN + 4;
  static_assert(BM * KQ4 % NT == 0 && NFA >= 1, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BM][LDK] / reduction scratch

  const int M = g.M, N = g.N, K = g.K;
  const int tiles_n = N / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nh = wave % NH, kq = wave / NH;

  floatx16 am, ac;
#pragma unroll
  for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; }

  unsigned aoff[NFA];
  int ldst[NFA];
#pragma unroll
  for (int i = 0; i < NFA; ++i) {
    const int f = t + i * NT, row = f / KQ4, c4 = f - row * KQ4;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    aoff[i] = ((unsigned)gr * (unsigned)K + c4 * 4) * 4u;  // a split-32 row is K dwords long
    ldst[i] = row * LDK + c4 * 4;
  }
  constexpr int WTILE = KS * 2 * 256;  // dwords per (32-column tile, k tile): 8 k-slices x {hi, lo} x 64 lanes x 16 B
  const int KT = K / BK;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(g.A)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(g.Wf)), 0, 0x7fffffff, 0x00020000);
  const unsigned wtile0 = (unsigned)(tn * NH + nh) * KT;
  const unsigned woff = (unsigned)(kq * (2 * 256) * 4) + lane * 16u;
  // this wave's k16 step of a stage: line kq/2 of the row, step kq%2 -> hi slot 2*(kq%2) + lane/32, lo slot 4 further
  const int fragA = (lane & 31) * LDK + (kq >> 1) * 32 + ((kq & 1) * 2 + (lane >> 5)) * 4;
#define IKQ_LDA(i, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i], __builtin_amdgcn_readfirstlane((kt_) * (BK * 4)), 0))
#define IKQ_LDW(pl, kt_) __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff + (pl) * 1024, __builtin_amdgcn_readfirstlane((wtile0 + (kt_)) * (WTILE * 4)), 0))

  floatx4 rg[NFA];
  half8 w0[2], w1[2];  // {hi, lo} W fragments of two consecutive tiles
#pragma unroll
  for (int i = 0; i < NFA; ++i) rg[i] = IKQ_LDA(i, 0);
  w0[0] = IKQ_LDW(0, 0);
  w0[1] = IKQ_LDW(1, 0);
#pragma unroll
  for (int i = 0; i < NFA; ++i) *reinterpret_cast<floatx4*>(smem + ldst[i]) = rg[i];
  {
    const int k1 = KT > 1 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NFA; ++i) rg[i] = IKQ_LDA(i, k1);
    w1[0] = IKQ_LDW(0, k1);
    w1[1] = IKQ_LDW(1, k1);
  }
  __syncthreads();

#define IKQ_FRAG(stage)                                                               \
  {                                                                                   \
    ah = *reinterpret_cast<const half8*>(smem + (stage) * STAGE + fragA);             \
    al = *reinterpret_cast<const half8*>(smem + (stage) * STAGE + fragA + 16);        \
  }
#define IKQ_PIN __builtin_amdgcn_sched_barrier(0);
  // iteration kt: the fragments of tile kt are already in (ah, al); A tile kt+1 is written into the other LDS stage (its
  // readers finished before the previous barrier), tile kt+2 is requested, three MFMAs, then the barrier and the
  // fragments of tile kt+1.  Branch-free: prefetches past the end re-read the last tile (clamped index).
#define IKQ_ITER(WC, NXT)                                                             \
  {                                                                                   \
    const int k2 = (kt + 2 < KT) ? kt + 2 : KT - 1;                                   \
    _Pragma("unroll") for (int i = 0; i < NFA; ++i) *reinterpret_cast<floatx4*>(smem + (NXT) * STAGE + ldst[i]) = rg[i]; \
    _Pragma("unroll") for (int i = 0; i < NFA; ++i) rg[i] = IKQ_LDA(i, k2);           \
    IKQ_PIN                                                                           \
    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, WC[0], am, 0, 0, 0);              \
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, WC[1], ac, 0, 0, 0);              \
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, WC[0], ac, 0, 0, 0);              \
    WC[0] = IKQ_LDW(0, k2);                                                           \
    WC[1] = IKQ_LDW(1, k2);                                                           \
    IKQ_PIN                                                                           \
    __syncthreads();                                                                  \
    IKQ_FRAG(NXT)                                                                     \
    IKQ_PIN                                                                           \
    ++kt;                                                                             \
  }
  half8 ah, al;
  IKQ_FRAG(0)
  for (int kt = 0; kt < KT;) {  // KT is even (launcher: K % 256 == 0)
    IKQ_ITER(w0, 1)
    IKQ_ITER(w1, 0)
  }
#undef IKQ_ITER
#undef IKQ_PIN
#undef IKQ_FRAG
#undef IKQ_LDA
#undef IKQ_LDW
  __syncthreads();

  // ---- v = hi*hi + corr/2048 per wave, then the k-slice blocks summed in fixed order kq = 0, 1, .. by all waves
  float* red = smem;  // [KS][NH][16][64]
  constexpr float inv_scale = 1.0f / IKF_SPLIT_SCALE;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((kq * NH + nh) * 16 + r) * 64 + lane] = fmaf(ac[r], inv_scale, am[r]);
  __syncthreads();
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  float fin[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * kq + j;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < KS; ++q) v += red[((q * NH + nh) * 16 + r) * 64 + lane];
    v += g.bias[n0 + nh * 32 + col_l];
    fin[j] = v > 0.f ? v : v * g.slope;
  }
  if constexpr (!EPI_RED) {
    // re-split and store: column c of row rl -> hi half at line (c/32), slot c%32; lo half 64 B further
    char* Cb = reinterpret_cast<char*>(g.C);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 2 * kq + j;
      const int row = m0 + (r & 3) + 8 * (r >> 2) + row_h;
      const int col = n0 + nh * 32 + col_l;
      const _Float16 hi = (_Float16)fin[j];
      const _Float16 lo = (_Float16)((fin[j] - (float)hi) * IKF_SPLIT_SCALE);
      char* p = Cb + (size_t)row * N * 4 + (size_t)(col >> 5) * 128 + (col & 31) * 2;  // row-padded buffer: unpredicated
      *reinterpret_cast<_Float16*>(p) = hi;
      *reinterpret_cast<_Float16*>(p + 64) = lo;
    }
    if ((split_out_of_range(fin[0]) || split_out_of_range(fin[1])) && g.flag) atomicOr(g.flag, 1);
  } else {
    float* T = smem + KS * NH * 16 * 64;  // behind red[] (other waves may still be summing)
    float* Wl = T + BM * LDT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 2 * kq + j;
      const int rl = (r & 3) + 8 * (r >> 2) + row_h;
      T[rl * LDT + nh * 32 + col_l] = fin[j];
    }
    for (int idx = t; idx < 32 * (BN / 4); idx += NT) {
      const int o = idx / (BN / 4), c4 = idx - o * (BN / 4);
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < g.n_out) v = *reinterpret_cast<const floatx4*>(g.w_last + (size_t)o * N + n0 + c4 * 4);
      *reinterpret_cast<floatx4*>(Wl + o * LDT + c4 * 4) = v;
    }
    __syncthreads();
    if (wave == 0) {  // last Linear restricted to the tile's columns in exact f32 MFMA: one (half) slot per tile
      floatx16 pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
      const float* pa = Wl + (lane & 31) * LDT + (lane >> 5) * 4;
      const float* pb = T + (lane & 31) * LDT + (lane >> 5) * 4;
#pragma unroll
      for (int ks = 0; ks < BN / 8; ++ks) {
        const floatx4 a4 = *reinterpret_cast<const floatx4*>(pa + ks * 8);
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(pb + ks * 8);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, pacc, 0, 0, 0);
      }
      float* pout = g.P_out + (size_t)(n0 / BN) * g.p_slot_stride + (size_t)(m0 + (lane & 31)) * IKF_PSTRIDE;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + row_h;
        pout[o] = pacc[r];
      }
    }
  }
}

// fragment-major image of a split-32 [N][K] weight for k_split_skinny: 16-byte unit index
//   (((tn32*KT + kt)*8 + kq)*2 + plane)*64 + lane
//      <-  row tn32*32 + lane%32, line kt*4 + kq/2, plane (0 hi / 1 lo), 8 halves of k16 step kq%2, half lane/32
__global__ __launch_bounds__(256) void k_wfrag_pack_split(const char* __restrict__ Ws, char* __restrict__ out, int N, int K) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= (size_t)N * K / 4) return;
  const int KT = K / SKBK;
  const int lane = (int)(f & 63);
  size_t r = f >> 6;
  const int plane = (int)(r & 1); r >>= 1;
  const int kq = (int)(r % SKKS); r /= SKKS;
  const int kt = (int)(r % KT);
  const int tn32 = (int)(r / KT);
  const size_t row = (size_t)tn32 * 32 + (lane & 31);
  const size_t src = row * (size_t)K * 4 + (size_t)(kt * 4 + (kq >> 1)) * 128 + plane * 64 + ((kq & 1) * 2 + (lane >> 5)) * 16;
  *reinterpret_cast<floatx4*>(out + f * 16) = *reinterpret_cast<const floatx4*>(Ws + src);
}
hipError_t launch_wfrag_pack_split(const void* Wsplit, int N, int K, void* out, hipStream_t s) {
  if (N % 32 != 0 || K % SKBK != 0) return hipErrorInvalidValue;
  const size_t n4 = (size_t)N * K / 4;
  hipLaunchKernelGGL(k_wfrag_pack_split, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const char*>(Wsplit), reinterpret_cast<char*>(out), N, K);
  return hipGetLastError();
}

template <bool EPI_RED, int NH>
static hipError_t launch_split_skinny(const SplitGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = split_skinny_lds<NH>();
  auto kern = k_split_skinny<EPI_RED, NH>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + SKBM - 1) / SKBM) * (a.N / (NH * 32));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NH * SKKS * 64), smem, s, a);
  return hipGetLastError();
}

const char* split_kernel_name() { return "k_split_gemm_dma"; }

int split_pick_cfg(long long rows, int width) {
  // from the in-chain sweep (tools/cfg_sweep.py -> profiles/r01_cfg_sweep.jsonl): the small-batch kernels up to 512 rows
  // (32x32 tiles up to 256), 64x64 up to 1024, 64x128 up to 2048, and above that the LDS-DMA 128x128 kernel even with
  // idle CUs (2560 rows: 1.31 ms against 1.72 ms for two rounds of 64x128)
  if (rows <= 512 && width % 64 == 0 && width % (2 * SKBK) == 0) return rows <= 256 ? kSplitSkinny32Cfg : kSplitSkinnyCfg;
  const int want = rows > 2048 ? 0 : rows > 1024 ? 1 : rows > 512 ? 2 : 3;
  for (int c = want; c < kNumSplitCfg; ++c)
    if (width % kSplitBN[c] == 0) return c;
  for (int c = want - 1; c >= 0; --c)
    if (width % kSplitBN[c] == 0) return c;
  return -1;
}
int split_slots(int cfg, int width) { return cfg == kSplitSkinny32Cfg ? width / 32 : width / 64; }
bool split_cfg_needs_frag(int cfg) { return cfg == kSplitSkinnyCfg || cfg == kSplitSkinny32Cfg; }

template <bool EPI_RED, int CFG>
static hipError_t launch_sg(const SplitGemmArgs& a, hipStream_t s) {
  using TC = SplitCfg<CFG>;
  constexpr size_t smem = (size_t)3 * (TC::BM + TC::BN) * 36 * sizeof(float);
  auto kern = k_split_gemm<EPI_RED, CFG>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long tiles_m = ((long long)a.M + TC::BM - 1) / TC::BM;
  const long long grid = tiles_m * (a.N / TC::BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TC::WAVES_M * TC::WAVES_N * 64), smem, s, a);
  return hipGetLastError();
}

hipError_t launch_split_gemm(bool epi_red, int cfg, const SplitGemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (cfg == kSplitSkinnyCfg || cfg == kSplitSkinny32Cfg) {  // small-batch kernels (fragment-major weight image)
    if (a.N % 64 != 0 || a.K % (2 * SKBK) != 0 || a.n_out > 16 || a.Wf == nullptr) return hipErrorInvalidValue;
    if (cfg == kSplitSkinnyCfg) return epi_red ? launch_split_skinny<true, 2>(a, s) : launch_split_skinny<false, 2>(a, s);
    return epi_red ? launch_split_skinny<true, 1>(a, s) : launch_split_skinny<false, 1>(a, s);
  }
  if ((cfg == 10 || cfg == 11) && (a.N % 128 != 0 || a.K % 64 != 0 || a.K < 128 || a.n_out > 16)) return hipErrorInvalidValue;
  if (cfg != 10 && cfg != 11 && (cfg < 0 || cfg >= kNumSplitCfg || a.N % kSplitBN[cfg] != 0 || a.K % 64 != 0 || a.K < 128 || a.n_out > 16)) return hipErrorInvalidValue;
  // config 0 (128x128) runs in its LDS-DMA form; 11 selects the register-staged form of the same tile (probe / A-B)
  if (cfg == 10 || cfg == 0) return epi_red ? launch_sg_dma<true>(a, s) : launch_sg_dma<false>(a, s);
  if (cfg == 11) cfg = 0;
  switch (cfg) {
    case 0: return epi_red ? launch_sg<true, 0>(a, s) : launch_sg<false, 0>(a, s);
    case 1: return epi_red ? launch_sg<true, 1>(a, s) : launch_sg<false, 1>(a, s);
    case 2: return epi_red ? launch_sg<true, 2>(a, s) : launch_sg<false, 2>(a, s);
    default: return epi_red ? launch_sg<true, 3>(a, s) : launch_sg<false, 3>(a, s);
  }
}

// device: fp32 [rows][K] -> split-32 image; one thread converts 8 consecutive k of a row
__global__ __launch_bounds__(256) void k_split32_pack(const float* __restrict__ src, long long n_chunks, int K,
                                                      char* __restrict__ dst, int* __restrict__ flag) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_chunks) return;
  const int cpr = K / 8;  // chunks per row
  const long long row = idx / cpr;
  const int k0 = (int)(idx - row * cpr) * 8;
  const floatx4 v0 = *reinterpret_cast<const floatx4*>(src + row * K + k0);
  const floatx4 v1 = *reinterpret_cast<const floatx4*>(src + row * K + k0 + 4);
  half8 hi, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi[q] = (_Float16)v0[q];
    lo[q] = (_Float16)((v0[q] - (float)hi[q]) * IKF_SPLIT_SCALE);
    hi[4 + q] = (_Float16)v1[q];
    lo[4 + q] = (_Float16)((v1[q] - (float)hi[4 + q]) * IKF_SPLIT_SCALE);
    if ((split_out_of_range(v0[q]) || split_out_of_range(v1[q])) && flag) atomicOr(flag, 1);
  }
  char* p = dst + row * (long long)K * 4 + (long long)(k0 >> 5) * 128 + (k0 & 31) * 2;
  *reinterpret_cast<half8*>(p) = hi;
  *reinterpret_cast<half8*>(p + 64) = lo;
}

hipError_t launch_split32_pack(const float* d_src, long long rows, int K, void* d_dst, int* d_flag, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (K % 32 != 0) return hipErrorInvalidValue;
  const long long n_chunks = rows * (K / 8);
  hipLaunchKernelGGL(k_split32_pack, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, s, d_src, n_chunks, K,
                     reinterpret_cast<char*>(d_dst), d_flag);
  return hipGetLastError();
}

// host: fp32 [rows][K] -> split-32 image (rows * K * 2 uint16 = rows * K * 4 bytes); used by the probes/tests only
void split32_pack_host(const float* src, int rows, int K, uint16_t* dst) {
  for (int r = 0; r < rows; ++r) {
    const float* sr = src + (size_t)r * K;
    uint16_t* dr = dst + (size_t)r * K * 2;
    for (int k = 0; k < K; ++k) {
      const _Float16 hi = (_Float16)sr[k];
      const _Float16 lo = (_Float16)((sr[k] - (float)hi) * IKF_SPLIT_SCALE);
      uint16_t hb, lb;
      __builtin_memcpy(&hb, &hi, 2);
      __builtin_memcpy(&lb, &lo, 2);
      dr[(size_t)(k >> 5) * 64 + (k & 31)] = hb;
      dr[(size_t)(k >> 5) * 64 + 32 + (k & 31)] = lb;
    }
  }
}

}  // namespace ikf
