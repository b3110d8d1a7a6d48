// Three kernels per coupling subnet (gfx950).  The subnet  Linear(in,W)-LReLU-[Linear(W,W)-LReLU]x-Linear(W,2L) + affine
// coupling (FrEIA GLOWCouplingBlock rev, call site ikflow/ikflow_solver.py:98, subnet ikflow/model.py:51-96) runs as
//
//   k_subnet_entry        finish the PREVIOUS subnet (sum its last-Linear partials, s = clamp*0.636*atan(s),
//                         y = (x - t)*exp(-s), PermuteRandom^-1), publish the new flow state, assemble u = [x_part, pose]
//                         in LDS and evaluate the first Linear + LReLU -> h1            (VALU; HBM/L3 write bound)
//   k_flow_gemm<false>    hidden Linear + LReLU,  h -> h                               (f32 MFMA bound)
//   k_flow_gemm<true>     last hidden Linear + LReLU kept on chip (LDS), then the last Linear restricted to this tile's
//                         columns as a second small MFMA contraction -> partial sums P[slot][row][o]; the hidden
//                         activation of the last layer never goes to HBM
//   k_flow_finalize       after the last subnet: "finish the previous subnet" + FixedLinearTransform^-1, [:, :ndof],
//                         clamp_to_joint_limits (ikflow_solver.py:99-102)
//
// The K loop is the 3-stage / one-barrier-per-tile pipeline of k_gemm_lrelu_p3 (flow_kernels.hip).
// (Generating the A operand of the first hidden contraction on the fly was measured and rejected: its ~110 VALU
// instructions per wave per K tile did not stay in the MFMA shadow - 98 us against 65 us + a 5 us entry kernel.)
#include <type_traits>

#include "ikf_internal.h"

namespace ikf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));

// optional in-kernel timeline (tools/gemm_probe.hip builds with -DIKF_TRACE): thread 0 of every workgroup stores the
// shader clock at a few points of k_flow_gemm into ikf_trace_buf[block][64]
#ifdef IKF_TRACE
__device__ unsigned long long* ikf_trace_buf = nullptr;
#define IKF_TSTAMP(i) if (threadIdx.x == 0 && ikf_trace_buf) ikf_trace_buf[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
}  // namespace ikf
// probe builds of the LIBRARY only (IKF_HIPCC_FLAGS=-DIKF_TRACE python -m ikflow_amd.build): point the stamp buffer at caller memory
extern "C" int ikf_debug_set_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(ikf::ikf_trace_buf), &buf, sizeof(buf)); }
namespace ikf {
#else
#define IKF_TSTAMP(i)
#endif

constexpr int ROWBUF = 16;  // floats per row in the small per-row LDS arrays (>= D, >= n_out)

// Tile configurations of the hidden contraction, chosen by the row count so that the launch has >= 256 workgroups
// whenever the batch allows it (one wave of tiles over the 256 CUs):
//   0: 128x128, 8 waves of 64x32   (rows >= 4096)
//   1:  64x128, 8 waves of 32x32   (rows >= 2048)
//   2:  64x64,  4 waves of 32x32   (rows >= 1024)
//   3:  32x64,  2 waves of 32x32   (smaller batches)
constexpr int FBK = 32;
template <int CFG> struct TileCfg;
template <> struct TileCfg<0> { static constexpr int BM = 128, BN = 128, WAVES_M = 2, WAVES_N = 4, BK = FBK; };
template <> struct TileCfg<1> { static constexpr int BM = 64, BN = 128, WAVES_M = 2, WAVES_N = 4, BK = FBK; };
template <> struct TileCfg<2> { static constexpr int BM = 64, BN = 64, WAVES_M = 2, WAVES_N = 2, BK = FBK; };
template <> struct TileCfg<3> { static constexpr int BM = 32, BN = 64, WAVES_M = 1, WAVES_N = 2, BK = FBK; };
template <> struct TileCfg<5> { static constexpr int BM = 128, BN = 128, WAVES_M = 2, WAVES_N = 2, BK = FBK; };  // probe: 4 waves of 64x64
// probe (r03): 128x64, 4 waves of 64x32, K tiles of 16 - three LDS stages are 46 KB, so up to three workgroups share a CU and one's
// prologue / epilogue / barriers can run under the others' K loops
template <> struct TileCfg<7> { static constexpr int BM = 128, BN = 64, WAVES_M = 2, WAVES_N = 2, BK = 16; };
constexpr int kNumTileCfg = 4;
static const int kCfgBM[kNumTileCfg] = {128, 64, 64, 32};
static const int kCfgBN[kNumTileCfg] = {128, 128, 64, 64};

// ---------------------------------------------------------------------------------------------------------------
// Finish a pending coupling for R rows starting at m0.  Phase A: one thread per (row, subnet output o) sums that output's
// partial-sum slots in fixed order (bias, slot 0, slot 1, ..) - all slot loads of a thread (up to 32) are in flight
// together, one memory round trip - and parks the sum in LDS.  Phase B: one thread per (row, state element) applies
// s = clamp*(0.636*atan s),  y = (x - t)*exp(-s)  and leaves the row [y1 | x2] / [x1 | y2] ("cat", BEFORE
// PermuteRandom^-1) in LDS cat[R][ROWBUF].  The permutation is applied by the reader through state_src():
// new_state[d] = cat[state_src(pc, d)].  `sums` is an R*ROWBUF-float LDS scratch.  Ends with a barrier.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int state_src(const PendingCoupling& pc, int d) {
  return (pc.P != nullptr && pc.which == 2) ? pc.perm_inv[d] : d;  // PermuteRandom rev: out[:, d] = cat[:, perm_inv[d]]
}

// The loads of a pending coupling (state element, bias and the first 32 partial-sum slots of each of this thread's items),
// split from their use so that a kernel can issue them BEFORE its other prefetches: VMEM returns in order, and a wait in
// front of the slot sums would otherwise also wait for whatever was queued ahead of them.
template <int ITEMS, int CAP = 32>  // CAP: partial-sum slots held in registers (one memory round trip for up to CAP slots)
struct PendingLoads {
  float xv[ITEMS], bias[ITEMS], a[ITEMS][CAP];
};

// SC1: the partial sums were published inside THIS launch by other workgroups (TailSync): agent-scope loads, which bypass
// this CU's L1 (it may hold the lines of an earlier subnet: the buffer is reused).
template <bool SC1>
__device__ __forceinline__ float load_partial(const float* p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

template <int NT, int R, bool SC1 = false, int CAP = 32>
__device__ __forceinline__ void pending_issue_loads(const PendingCoupling& pc, const float* __restrict__ x_src, int D, int L1,
                                                    int m0, int M, int t, PendingLoads<(R * ROWBUF + NT - 1) / NT, CAP>& pl) {
  constexpr int ITEMS = (R * ROWBUF + NT - 1) / NT;
  const int nl = (pc.which == 1) ? D - L1 : L1;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = t + it * NT, r = idx / ROWBUF, d = idx % ROWBUF;
    int gr = m0 + r;
    gr = gr < M ? gr : M - 1;
    pl.xv[it] = (idx < R * ROWBUF && d < D) ? x_src[(size_t)gr * D + d] : 0.f;
    const bool has_sum = pc.P != nullptr && idx < R * ROWBUF && d < 2 * nl;
    const float* p = pc.P + (size_t)(m0 + r) * IKF_PSTRIDE + d;  // P rows are padded to the tile: no clamp needed
    pl.bias[it] = has_sum ? pc.b_last[d] : 0.f;
#pragma unroll
    for (int q = 0; q < CAP; ++q) pl.a[it][q] = (has_sum && q < pc.slots) ? load_partial<SC1>(p + (size_t)q * pc.slot_stride) : 0.f;
  }
}

// Phase A: one thread per (row, subnet output o) sums that output's slots in fixed order (bias, slot 0, slot 1, ..) and parks
// the sum in LDS.  Phase B: one thread per (row, state element) applies the coupling (see above).  Ends with a barrier.
template <int NT, int R, int CAP = 32>
__device__ __forceinline__ void finish_pending_rows(const PendingCoupling& pc, const PendingLoads<(R * ROWBUF + NT - 1) / NT, CAP>& pl,
                                                    int D, int L1, float clamp, int m0, float* cat, float* sums, int t) {
  constexpr int ITEMS = (R * ROWBUF + NT - 1) / NT;
  const int L2 = D - L1;
  // which == 1: y2 = (x2 - t1) * exp(-s1), x1 untouched.   which == 2: y1 = (x1 - t2) * exp(-s2), x2 (= y2) untouched
  const int nl = (pc.which == 1) ? L2 : L1;
  const int off = (pc.which == 1) ? L1 : 0;
  if (pc.P != nullptr) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = t + it * NT, r = idx / ROWBUF, o = idx % ROWBUF;
      if (idx >= R * ROWBUF || o >= 2 * nl) continue;
      float sv = pl.bias[it];
#pragma unroll
      for (int q = 0; q < CAP; ++q) sv += pl.a[it][q];
      if (pc.slots > CAP) {  // more slots than the registers hold: the remaining ones in chunks of 32 (another round trip each)
        const float* p = pc.P + (size_t)(m0 + r) * IKF_PSTRIDE + o;
        for (int s0 = CAP; s0 < pc.slots; s0 += 32) {
          float a[32];
#pragma unroll
          for (int q = 0; q < 32; ++q) a[q] = (s0 + q < pc.slots) ? p[(size_t)(s0 + q) * pc.slot_stride] : 0.f;
#pragma unroll
          for (int q = 0; q < 32; ++q) sv += a[q];
        }
      }
      sums[r * ROWBUF + o] = sv;
    }
    __syncthreads();
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = t + it * NT, r = idx / ROWBUF, d = idx % ROWBUF;
    if (idx >= R * ROWBUF || d >= D) continue;
    float v = pl.xv[it];
    if (pc.P != nullptr && d >= off && d < off + nl) {
      const int j = d - off;
      const float s_cl = clamp * (0.636f * atanf(sums[r * ROWBUF + j]));
      v = (v - sums[r * ROWBUF + nl + j]) * expf(-s_cl);
    }
    cat[r * ROWBUF + d] = v;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// subnet entry: pending coupling + first Linear + LeakyReLU for ER rows per workgroup
// ---------------------------------------------------------------------------------------------------------------
// Geometry of a launch: NT threads per workgroup, ER rows per workgroup, column split gridDim.y.  More waves per CU hide
// the LDS -> FMA -> store dependency chains of the first-Linear phase and let one workgroup's pending-coupling round trip
// overlap another's stores (tools/gemm_probe.hip 300 sweeps the choices; profiles/r02_entry_sweep.txt).
template <int IN, int NT, int ER>
__global__ __launch_bounds__(NT) void k_subnet_entry(EntryArgs e) {
  constexpr int UI = (ER * ROWBUF + NT - 1) / NT;  // (row, input column) items per thread
  __shared__ __attribute__((aligned(16))) float cat[ER * ROWBUF], U[ER * ROWBUF];  // U doubles as the slot-sum scratch
  const int t = threadIdx.x;
  const int m0 = blockIdx.x * ER;
  const int M = e.M, D = e.D;
  PendingLoads<(ER * ROWBUF + NT - 1) / NT> pl;
  pending_issue_loads<NT, ER>(e.pend, e.x_src, D, e.L1, m0, M, t, pl);  // the critical path's loads go first
  // this thread's slice of the first Linear (first column group) is fetched up front: its L2 latency hides behind the
  // pending-coupling phase below
  // Column split for small batches (gridDim.y workgroups share a row group, each redoing the cheap pending phase): the
  // workgroup owns n4_per float4 column groups; when that is fewer than the NT threads, the threads also split the rows
  // (CPW columns x NT/CPW row groups of RPT rows).
  const int n4 = e.width >> 2;
  const int n4_per = n4 / (int)gridDim.y, c4_base = (int)blockIdx.y * n4_per;
  // (n4_per need not divide the NT threads - width 768 gives 192: the NT % CPW surplus threads sit the column phase out)
  const int CPW = n4_per < NT ? n4_per : NT;
  const int RG = NT / CPW;               // row groups
  const int RPT = (ER + RG - 1) / RG;    // rows per group
  const int tc = t % CPW, rg = t / CPW;
  const int r_first = rg < RG ? rg * RPT : ER;
  const int r_last = r_first + RPT < ER ? r_first + RPT : ER;
  floatx4 w0[IN], b0;
  {
    const int c4 = c4_base + tc;
#pragma unroll
    for (int k = 0; k < IN; ++k) w0[k] = reinterpret_cast<const floatx4*>(e.w1t + (size_t)k * e.width)[c4];
    b0 = reinterpret_cast<const floatx4*>(e.b1)[c4];
  }
  // one item per (row, column) of the 16-wide input row; the pose element is fetched before the pending phase
  float pose_v[UI];
#pragma unroll
  for (int it = 0; it < UI; ++it) {
    const int idx = t + it * NT, ur = idx / ROWBUF, uk = idx % ROWBUF;
    pose_v[it] = 0.f;
    if (idx < ER * ROWBUF && uk >= e.n_x && uk < IN) {
      int gr = m0 + ur;
      gr = gr < M ? gr : M - 1;
      const long long grow = e.row0 + gr;
      // one pose per row (n_mod == rows) and the single-pose broadcast (n_mod == 1) skip the 64-bit modulo
      const long long pm = grow < e.ps.n_mod ? grow : (e.ps.n_mod == 1 ? 0 : grow % e.ps.n_mod);
      const long long pi = e.ps.idx ? (long long)e.ps.idx[pm] : pm;
      pose_v[it] = e.ps.poses[pi * e.ps.stride + (uk - e.n_x)];
    }
  }
  if (e.zero_words != nullptr && blockIdx.x == 0 && blockIdx.y == 0)  // the call's arrival counters (TailSync), visible to the
    for (int i = t; i < e.n_zero; i += NT) e.zero_words[i] = 0u;     // later launches through the kernel boundary
  IKF_TSTAMP(0)
  finish_pending_rows<NT, ER>(e.pend, pl, D, e.L1, e.clamp, m0, cat, U, t);
  IKF_TSTAMP(1)
  // publish the new state and assemble u = [x_part, pose, 0-pad]
#pragma unroll
  for (int it = 0; it < UI; ++it) {
    const int idx = t + it * NT, ur = idx / ROWBUF, uk = idx % ROWBUF;
    if (idx >= ER * ROWBUF) continue;
    if (blockIdx.y == 0 && uk < D && m0 + ur < M) e.x_dst[(size_t)(m0 + ur) * D + uk] = cat[ur * ROWBUF + state_src(e.pend, uk)];
    U[ur * ROWBUF + uk] = uk < e.n_x ? cat[ur * ROWBUF + state_src(e.pend, e.x_off + uk)] : pose_v[it];
  }
  __syncthreads();
  IKF_TSTAMP(2)
  unsigned range_max = 0;  // split_out: running maximum of the |hi| halves written (f16x3 range guard)
  for (int cc = tc; cc < n4_per; cc += CPW) {
    const int c4 = c4_base + cc;
    floatx4 w[IN];
    floatx4 b;
    if (cc == tc) {
#pragma unroll
      for (int k = 0; k < IN; ++k) w[k] = w0[k];
      b = b0;
    } else {
#pragma unroll
      for (int k = 0; k < IN; ++k) w[k] = reinterpret_cast<const floatx4*>(e.w1t + (size_t)k * e.width)[c4];
      b = reinterpret_cast<const floatx4*>(e.b1)[c4];
    }
    if (e.ps.softflow != 0.0f) b += e.ps.softflow * reinterpret_cast<const floatx4*>(e.w1soft)[c4];
#pragma unroll 4
    for (int r = r_first; r < r_last; ++r) {
      // the row's 16 inputs as four LDS broadcast reads (same address in every lane)
      float u[ROWBUF];
#pragma unroll
      for (int q = 0; q < ROWBUF / 4; ++q) {
        const floatx4 uq = *reinterpret_cast<const floatx4*>(U + r * ROWBUF + q * 4);
        u[q * 4 + 0] = uq.x; u[q * 4 + 1] = uq.y; u[q * 4 + 2] = uq.z; u[q * 4 + 3] = uq.w;
      }
      floatx4 acc = b;
#pragma unroll
      for (int k = 0; k < IN; ++k) acc += u[k] * w[k];
      acc.x = acc.x > 0.f ? acc.x : acc.x * e.slope;
      acc.y = acc.y > 0.f ? acc.y : acc.y * e.slope;
      acc.z = acc.z > 0.f ? acc.z : acc.z * e.slope;
      acc.w = acc.w > 0.f ? acc.w : acc.w * e.slope;
      // h rows are padded to a multiple of 128 (engine scratch): unpredicated
      if (!e.split_out) {
        if (e.wt_stores) {
          const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(e.h_out, 0, 0x7fffffff, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, acc), rsH, (unsigned)(((size_t)(m0 + r) * e.width + c4 * 4) * 4), 0, 16);
        } else reinterpret_cast<floatx4*>(e.h_out + (size_t)(m0 + r) * e.width)[c4] = acc;
      } else {
        // split-32 image: 4 consecutive columns -> 4 hi halves (8 B) and 4 lo halves (8 B, 64 B further on)
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        half4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hi[q] = (_Float16)acc[q];
          lo[q] = (_Float16)((acc[q] - (float)hi[q]) * IKF_SPLIT_SCALE);
        }
        {
          typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
          const uint2_t hb = __builtin_bit_cast(uint2_t, hi);
          range_max = range_track(range_track(range_max, hb.x), hb.y);
        }
        char* rowp = reinterpret_cast<char*>(e.h_out) + (size_t)(m0 + r) * e.width * 4 + (size_t)(c4 >> 3) * 128 + (c4 & 7) * 8;
        *reinterpret_cast<half4*>(rowp) = hi;
        *reinterpret_cast<half4*>(rowp + 64) = lo;
      }
    }
  }
  if (e.split_out && e.split_flag && range_hit(range_max)) atomicOr(e.split_flag, 1);
  IKF_TSTAMP(3)
}

// The next subnet's entry phase in the tail of a contraction (TailSync: tail_next_entry) and the one-launch subnet chain for <= 128 rows
// (k_flow_chain16) - priced and rejected in round 3 (CHANGELOG.md; measurement log in git history) - live in flow_fused_probes.inc and exist only in the probes
// library (-DIKF_PROBES).  The FUSE template parameter of the two contraction kernels below is never instantiated true in the product.
[[maybe_unused]] constexpr unsigned kTailSpinLimit = 1u << 21;  // x (s_sleep 8 + one load) ~ a second: only reached when a sibling never runs
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store16_wt(const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off, floatx4 v) {  // write-through
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), rs, byte_off, 0, /*aux: sc1*/ 16);
}
#ifdef IKF_PROBES
#define IKF_PROBES_PART 1
#include "flow_fused_probes.inc"
#undef IKF_PROBES_PART
#else
template <int NT, int R, int BNW>
__device__ __forceinline__ void tail_next_entry(const EntryArgs&, const TailSync&, float*, int, int, int, bool, int) {}
#endif
constexpr size_t tail_lds_floats(int R, int BNW) { return (size_t)3 * R * ROWBUF + (size_t)(ROWBUF + 1) * BNW; }

// ---------------------------------------------------------------------------------------------------------------
// hidden contraction; EPI_RED: reduce the last Linear in the epilogue instead of storing the activation;
// FUSE (with EPI_RED): the next subnet's entry phase runs in the tail (tail_next_entry)
// ---------------------------------------------------------------------------------------------------------------
struct NoTail {};
struct FuseTail {
  EntryArgs e;
  TailSync ts;
};
template <bool EPI_RED, int CFG, bool FUSE = false>
__global__ __launch_bounds__(TileCfg<CFG>::WAVES_M* TileCfg<CFG>::WAVES_N * 64) void k_flow_gemm(
    FusedGemmArgs g, std::conditional_t<FUSE, FuseTail, NoTail> ft) {
  static_assert(!FUSE || EPI_RED, "the fused tail follows the partial-sum epilogue");
  using TC = TileCfg<CFG>;
  constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK, FWAVES_M = TC::WAVES_M, FWAVES_N = TC::WAVES_N;
  constexpr int NT = FWAVES_M * FWAVES_N * 64;
  // partial-sum epilogue: one slot = 64 columns for EVERY tile configuration, so the last Linear is summed in the same
  // order whatever the batch size (results are bit-identical across batch sizes)
  constexpr int FKH = BN / 64;
  static_assert(BN % 64 == 0, "the partial-sum epilogue works in 64-column slots");
  constexpr int WM = BM / FWAVES_M, WN = BN / FWAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int LDK = BK + 4;
  constexpr int KQ = BK / 4;
  constexpr int A_F4 = BM * KQ / NT;
  constexpr int B_F4 = BN * KQ / NT;
  constexpr int RS = NT / KQ;
  constexpr int NKK = BK / 8;
  constexpr int STAGE = (BM + BN) * LDK;
  constexpr int LDT = BN + 4;  // epilogue tile row stride
  static_assert(BM * KQ % NT == 0 && BN * KQ % NT == 0 && NT % KQ == 0, "tile/threads mismatch");
  static_assert(NKK % 2 == 0 && NKK >= 2, "fragment double-buffering needs an even number of k-groups");
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
  const int wm = (wave / FWAVES_N) * WM, wn = (wave % FWAVES_N) * WN;

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int row_t = t / KQ, kq_t = (t % KQ) * 4;
  // buffer loads: SGPR descriptor + loop-invariant 32-bit byte offset per thread + scalar k offset.  (A global_load
  // with a 64-bit VGPR address costs its SIMD roughly three times the issue time and needs vector address arithmetic.)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, 0x7fffffff, 0x00020000);
  unsigned a_off[A_F4], b_off[B_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    int gr = m0 + row_t + i * RS;
    gr = gr < M ? gr : M - 1;
    a_off[i] = ((unsigned)gr * (unsigned)K + kq_t) * 4u;
  }
#pragma unroll
  for (int i = 0; i < B_F4; ++i) b_off[i] = ((unsigned)(n0 + row_t + i * RS) * (unsigned)K + kq_t) * 4u;
#define IKF_BLD(rs, voff, koff) \
  __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane((koff) * 4), 0))
  const int lds_t = row_t * LDK + kq_t;
  const int fragA = (wm + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int fragB = BM * LDK + (wn + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int KT = K / BK;

  floatx4 ra[2][A_F4], rb[2][B_F4];  // two staging sets: global loads run two K tiles ahead of their LDS write
  floatx4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];

#define IKF_GLOAD(S, koff)                                                                                        \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) ra[S][i] = IKF_BLD(rsA, a_off[i], koff);                      \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i) rb[S][i] = IKF_BLD(rsW, b_off[i], koff);                      \
  }
#define IKF_LSTORE(S, stage)                                                                                      \
  {                                                                                                               \
    float* sp_ = smem + (stage) * STAGE + lds_t;                                                                  \
    _Pragma("unroll") for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sp_ + i * RS * LDK) = ra[S][i];   \
    _Pragma("unroll") for (int i = 0; i < B_F4; ++i)                                                              \
        *reinterpret_cast<floatx4*>(sp_ + BM * LDK + i * RS * LDK) = rb[S][i];                                    \
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

  // One K tile; HAS1/HAS2 (tile kt+1 / kt+2 exist) are compile-time so the steady-state body is branch-free and the
  // sched_group_barrier sequences pin a k-MFMA : 1-memory-op interleave on both sides of the barrier.
  // tile kt+1 sits in register set S (loaded two iterations ago) and is written to LDS stage nxt; the freed set is
  // refilled with tile kt+3
  auto k_tile = [&](auto has1_c, auto has3_c, auto set_c, int kt, int cur, int nxt) {
    constexpr bool HAS1 = decltype(has1_c)::value, HAS2 = decltype(has3_c)::value;
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int kk = 0; kk < NKK / 2; ++kk) {
      if (kk & 1) { IKF_FRAG(fa0, fb0, cur, kk + 1) } else { IKF_FRAG(fa1, fb1, cur, kk + 1) }
      if (kk & 1) { IKF_MFMA4(fa1, fb1) } else { IKF_MFMA4(fa0, fb0) }
      if (kk == 0) {
        if (HAS1) IKF_LSTORE(S, nxt)
        if (HAS2) IKF_GLOAD(S, (kt + 3) * BK)
      }
    }
    {
      constexpr int n_mfma = (NKK / 2) * MI * NI * 4;
      constexpr int n_mem = (HAS1 ? A_F4 + B_F4 : 0) + (HAS2 ? A_F4 + B_F4 : 0) + (NKK / 2) * (MI + NI);
      constexpr int per = n_mfma / (n_mem > 0 ? n_mem : 1) > 0 ? n_mfma / (n_mem > 0 ? n_mem : 1) : 1;
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
    __syncthreads();
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

  constexpr int WL_F4 = EPI_RED ? (32 * (BN / 4) + NT - 1) / NT : 1;  // float4 of the last Linear's slice per thread
  float bias_pre[NI];
  floatx4 wl_pre[WL_F4];
#define IKF_EPI_PREFETCH                                                                                                   \
  {                                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) bias_pre[j] = g.bias[n0 + wn + j * 32 + (lane & 31)];                     \
    if constexpr (EPI_RED) {                                                                                                \
      _Pragma("unroll") for (int i = 0; i < WL_F4; ++i) {                                                                    \
        const int idx = t + i * NT, o = idx / (BN / 4), c4 = idx - o * (BN / 4);                                             \
        wl_pre[i] = floatx4{0.f, 0.f, 0.f, 0.f};                                                                             \
        if (idx < 32 * (BN / 4) && o < g.n_out) wl_pre[i] = *reinterpret_cast<const floatx4*>(g.w_last + (size_t)o * N + n0 + c4 * 4); \
      }                                                                                                                     \
    }                                                                                                                       \
  }
  // prologue: the (cold) loads of tiles 0, 1 and 2 are issued back to back so their miss latencies overlap
  IKF_TSTAMP(0)
  {
    floatx4 ra0[A_F4], rb0[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) ra0[i] = IKF_BLD(rsA, a_off[i], 0);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb0[i] = IKF_BLD(rsW, b_off[i], 0);
    if (KT > 1) IKF_GLOAD(1, BK)       // tile 1 -> set 1 (stored by iteration 0)
    if (KT > 2) IKF_GLOAD(0, 2 * BK)   // tile 2 -> set 0 (stored by iteration 1)
    // what the epilogue reads from memory - the bias of this lane's columns and (EPI_RED) this thread's pieces of the last Linear's
    // slice - is requested here, behind the prologue's operand loads: fetched in the epilogue each costs a round trip there
    IKF_EPI_PREFETCH
    float* sp0 = smem + lds_t;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) *reinterpret_cast<floatx4*>(sp0 + i * RS * LDK) = ra0[i];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) *reinterpret_cast<floatx4*>(sp0 + BM * LDK + i * RS * LDK) = rb0[i];
  }
  __syncthreads();
  IKF_TSTAMP(1)
  IKF_FRAG(fa0, fb0, 0, 0)

  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;
  // iteration kt stores tile kt+1 from set (kt+1)&1 and refills that set with tile kt+3.  Every iteration runs the SAME
  // two code bodies: near the end the prefetch index is clamped to the last tile (a redundant, harmless load + LDS
  // write into a stage nobody reads again) - specialised tail instantiations are cold code and cost ~1000 cycles each
  // in instruction-cache misses.
  int cur = 0;
  for (int kt = 0; kt < KT; kt += 2) {  // KT is even (K % 64 == 0, checked by the launcher)
    int nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, T_{}, S1_{}, (kt + 3 < KT ? kt : KT - 4), cur, nxt);
    cur = nxt;
    nxt = (cur == 2) ? 0 : cur + 1;
    k_tile(T_{}, T_{}, S0_{}, (kt + 4 < KT ? kt + 1 : KT - 4), cur, nxt);
    cur = nxt;
#ifdef IKF_TRACE
    if ((kt & 3) == 2 && kt < 128) IKF_TSTAMP(2 + (kt >> 2))
#endif
  }
  IKF_TSTAMP(40)
#undef IKF_EPI_PREFETCH
#undef IKF_GLOAD
#undef IKF_BLD
#undef IKF_LSTORE
#undef IKF_FRAG
#undef IKF_MFMA4

  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  if constexpr (!EPI_RED) {
    // bias + LeakyReLU, unpredicated stores into the row-padded activation buffer
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn + j * 32 + col_l;
      const float bv = bias_pre[j];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          float v = acc[i][j][r] + bv;
          v = v > 0.f ? v : v * g.slope;
#if defined(IKF_EPI_NOSTORE)   // probes only (tools/gemm_probe.hip): what the activation stores + the flush behind them cost
          if (v == 123456.789f) g.C[(size_t)row * N + col] = v;
#elif defined(IKF_EPI_SC1)     // probes only: write-through (agent-scope) stores instead of write-back ones
          __hip_atomic_store(&g.C[(size_t)row * N + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
          if (g.wt_stores) __hip_atomic_store(&g.C[(size_t)row * N + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else g.C[(size_t)row * N + col] = v;
#endif
        }
      }
    }
  } else {
    // ---- last Linear restricted to this tile's columns:  P[row][o] = sum_col h[row][col] * W4[o][col]
    // h tile -> LDS T[BM][LDT]; W4 slice -> LDS Wl[32][LDT] (rows >= n_out zero); wave w owns row block w % 4 and the
    // column half w / 4 (its own partial-sum slot).
    __syncthreads();  // every wave is done reading the last stage
    float* T = smem;
    float* Wl = smem + BM * LDT;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int cl = wn + j * 32 + col_l;
      const float bv = bias_pre[j];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          float v = acc[i][j][r] + bv;
          v = v > 0.f ? v : v * g.slope;
          T[rl * LDT + cl] = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WL_F4; ++i) {
      const int idx = t + i * NT, o = idx / (BN / 4), c4 = idx - o * (BN / 4);
      if (idx < 32 * (BN / 4)) *reinterpret_cast<floatx4*>(Wl + o * LDT + c4 * 4) = wl_pre[i];
    }
    __syncthreads();
    constexpr int RB = BM / 32;
    constexpr int CW = 64;  // columns per slot
    // On v_mfma_f32_16x16x4_f32: n_out <= 16, so the 32x32x2 shape would spend half of its 4096 matrix-pipe cycles per tile on zero rows
    // of W4 (r03, tools/gemm_trace.py: 7.3 k cycles of epilogue behind a K loop that runs at 97 % of the pipe).  "A" = 16 rows of h,
    // "B" = W4 (j = output), four consecutive columns per lane at 4 (lane / 16) of a 16-column group, component c -> MFMA c.
    for (int job = wave; job < RB * FKH; job += NT / 64) {  // wave-uniform: (row block, 64-column slot) jobs over the waves
      const int rb = job % RB, kh = job / RB;
      typedef float floatx4_ __attribute__((ext_vector_type(4)));
      floatx4_ pacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const int gq = lane >> 4, l16 = lane & 15;
      const float* pa = T + (rb * 32 + l16) * LDT + kh * CW + gq * 4;
      const float* pb = Wl + l16 * LDT + kh * CW + gq * 4;
#pragma unroll
      for (int ks = 0; ks < CW / 16; ++ks) {
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(pb + ks * 16);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const floatx4 a4 = *reinterpret_cast<const floatx4*>(pa + sb * 16 * LDT + ks * 16);
          pacc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, pacc[sb], 0, 0, 0);
          pacc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, pacc[sb], 0, 0, 0);
          pacc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, pacc[sb], 0, 0, 0);
          pacc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, pacc[sb], 0, 0, 0);
        }
      }
      // pacc[sb][v] = P[row = 32 rb + 16 sb + 4 (lane / 16) + v][o = lane % 16]: 64-byte runs per row
      float* pout = g.P_out + (size_t)(n0 / CW + kh) * g.p_slot_stride + (size_t)(m0 + rb * 32 + 4 * gq) * IKF_PSTRIDE + l16;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float* dst = pout + (size_t)(16 * sb + v) * IKF_PSTRIDE;
          if constexpr (FUSE) __hip_atomic_store(dst, pacc[sb][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
          else *dst = pacc[sb][v];
        }
    }
    if constexpr (FUSE) {
      static_assert(3 * STAGE >= (int)tail_lds_floats(BM, BN), "the tail's LDS fits in the stage area");
      IKF_TSTAMP(42)
      tail_next_entry<NT, BM, BN>(ft.e, ft.ts, smem, tm, m0, n0, tn == 0, t);
    }
  }
  IKF_TSTAMP(41)
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_finalize(FinalizeArgs f) {
  constexpr int R = 32, NT = 256;
  __shared__ float cat[R * ROWBUF], sums[R * ROWBUF];
  const int t = threadIdx.x;
  const int m0 = blockIdx.x * R;
  PendingLoads<(R * ROWBUF + NT - 1) / NT> pl;
  pending_issue_loads<NT, R>(f.pend, f.x_src, f.D, f.L1, m0, f.M, t, pl);
  finish_pending_rows<NT, R>(f.pend, pl, f.D, f.L1, f.clamp, m0, cat, sums, t);
  // FixedLinearTransform rev: (x - b).mm(M_inv); [:, :ndof]; clamp_to_joint_limits
  const int D = f.D;
  for (int idx = t; idx < R * ROWBUF; idx += NT) {
    const int r = idx / ROWBUF, j = idx % ROWBUF;
    if (j >= f.ndof || m0 + r >= f.M) continue;
    float q = 0.f;
    for (int k = 0; k < D; ++k) {
      float xv = cat[r * ROWBUF + state_src(f.pend, k)];
      if (f.sigmoid) xv = 1.0f / (1.0f + expf(-xv));  // InvertibleSigmoidFlipped rev (ikflow/model.py:124-127)
      q = fmaf(xv - f.b_lin[k], f.M_inv[k * D + j], q);
    }
    if (f.clamp_limits) q = fminf(fmaxf(q, f.lo[j]), f.hi[j]);
    f.q_out[(size_t)(m0 + r) * f.ndof + j] = q;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (rows <= 512): at most 256 tiles of 32x64 exist, one per CU, and the launch is a latency chain, not a
// throughput problem.  This form splits K INSIDE the workgroup: tile 32x64, BK = 128 per stage, 16 waves = 2 column
// halves x 8 k-slices (four waves per SIMD); every wave issues 8 MFMAs per stage.  What the measurements said
// (tools/gemm_probe.hip 104, per-stage cost against the 2048-cycle MFMA floor):
//   * every VMEM wave-instruction issued with 64-bit VGPR addresses costs its SIMD ~70 cycles of matrix-pipe time;
//     buffer loads (SGPR descriptor + loop-invariant VGPR offset + scalar tile offset) cost about a third of that and
//     leave no address arithmetic in the loop: 3000 -> 2450 cycles per stage;
//   * each W element feeds exactly one wave, so W skips LDS and is fetched as fragments from a fragment-major image
//     (k_wfrag_pack; from the row-major weight the same fetch is 32 rows 4 KB apart = one L2 channel: 5200-cycle
//     prologue); only the A rows are staged through LDS (two stages, one barrier per stage);
//   * the loop is branch-free (clamped prefetch index): with conditional loads the compiler waits vmcnt(0).
// The k-slice partial blocks are summed through LDS in fixed order kq = 0..7 by all 16 waves (two accumulator
// registers each).  (The slice split changes the summation order with respect to the large tiles: results agree to
// rounding, not bit for bit, across the 512-row boundary.)
// ---------------------------------------------------------------------------------------------------------------
// NH = column halves per workgroup: 2 -> 32x64 tiles, 16 waves (257..512 rows: <= 256 tiles); 1 -> 32x32 tiles, 8 waves
// (129 .. 256 rows - <= 128 rows run on the 16-row tiles of the 16x16x4 kernels further down - and 513 .. 768 rows, three workgroups
// per CU: still <= 256 tiles up to 256 rows, and half the matrix-pipe time per stage - the per-launch latency drops by a third).
constexpr int KBM = 32, KBN = 64, KBK = 128;
constexpr int KKS = 8;               // k-slices per tile (waves = NH column halves x KKS)
constexpr int KKW = KBK / KKS;       // k per wave per tile (16)
constexpr int KKG = KKW / 8;         // MFMA groups (8 k = 4 MFMAs) per wave per tile (2)
// LDS: two A stages in the loop; afterwards the KKS partial blocks + the epilogue's T / w_last tiles
template <int NH>
constexpr size_t skinny_lds() {
  return sizeof(float) * ((size_t)KKS * NH * 16 * 64 + (size_t)(KBM + 32) * (NH * 32 + 4)) > sizeof(float) * 2 * KBM * (KBK + 4)
             ? sizeof(float) * ((size_t)KKS * NH * 16 * 64 + (size_t)(KBM + 32) * (NH * 32 + 4))
             : sizeof(float) * 2 * KBM * (KBK + 4);
}

// ---------------------------------------------------------------------------------------------------------------
// tail of the small-batch kernels: k-slice reduction + bias / LeakyReLU + store or last-Linear partial sums.
// Call after a barrier that ends every fragment read of the loop (smem is reused from its start).
// ---------------------------------------------------------------------------------------------------------------
template <bool EPI_RED, int NH, bool FUSE = false>
__device__ __forceinline__ void skinny_tail(const FusedGemmArgs& g, const floatx16& acc, float* smem, int m0, int n0, int t,
                                            int lane, int wave, int nh, int kq) {
  constexpr int BM = KBM, BN = NH * 32, NT = NH * KKS * 64, KS = KKS;
  constexpr int LDT = BN + 4;
  const int N = g.N;
  IKF_TSTAMP(40)

  // ---- sum the k-slice partial blocks in fixed order kq = 0, 1, ..: every wave parks its block in LDS, then wave
  // (kq, nh) finishes accumulator registers 2*kq and 2*kq+1 (two output rows per lane half) of column half nh
  float* red = smem;  // [KS][NH][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((kq * NH + nh) * 16 + r) * 64 + lane] = acc[r];
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
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 2 * kq + j;
      const int row = m0 + (r & 3) + 8 * (r >> 2) + row_h;
      // row-padded buffer: unpredicated
      if (g.wt_stores) __hip_atomic_store(&g.C[(size_t)row * N + n0 + nh * 32 + col_l], fin[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else g.C[(size_t)row * N + n0 + nh * 32 + col_l] = fin[j];
    }
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
    // one slot per tile.  Four waves (one per SIMD) contract a quarter of the tile's columns each - 8 (4) MFMAs instead of
    // one wave's 32 (16) - and wave 0 adds the quarters in fixed order 0, 1, 2, 3 through LDS (red[] is dead by now)
    constexpr int NQW = 4, CPQ = BN / NQW;
    static_assert(CPQ % 8 == 0, "a wave's column quarter is a whole number of 8-column MFMA groups");
    float* part = smem;  // [NQW][8][64]
    floatx16 pacc;
    if (wave < NQW) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
      const float* pa = Wl + (lane & 31) * LDT + wave * CPQ + (lane >> 5) * 4;
      const float* pb = T + (lane & 31) * LDT + wave * CPQ + (lane >> 5) * 4;
#pragma unroll
      for (int ks = 0; ks < CPQ / 8; ++ks) {
        const floatx4 a4 = *reinterpret_cast<const floatx4*>(pa + ks * 8);
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(pb + ks * 8);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, pacc, 0, 0, 0);
      }
      // pacc[reg] = P^T[o][row], o = (reg&3) + 8*(reg>>2) + 4*(lane>>5), row = lane&31; o < 16 lives in regs 0..7
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) part[(wave * 8 + r) * 64 + lane] = pacc[r];
      }
    }
    __syncthreads();
    if (wave == 0) {
      float fin8[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = pacc[r];
#pragma unroll
        for (int w = 1; w < NQW; ++w) v += part[(w * 8 + r) * 64 + lane];
        fin8[r] = v;
      }
      // registers 0..3 are outputs row_h .. row_h+3, registers 4..7 outputs 8+row_h .. : two 16-byte stores per lane
      const size_t pidx = (size_t)(n0 / BN) * g.p_slot_stride + (size_t)(m0 + (lane & 31)) * IKF_PSTRIDE + row_h;
      const floatx4 p_lo = {fin8[0], fin8[1], fin8[2], fin8[3]}, p_hi = {fin8[4], fin8[5], fin8[6], fin8[7]};
      if constexpr (FUSE) {  // published inside the launch (TailSync): write-through
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(g.P_out, 0, 0x7fffffff, 0x00020000);
        store16_wt(rsP, (unsigned)(pidx * 4), p_lo);
        store16_wt(rsP, (unsigned)((pidx + 8) * 4), p_hi);
      } else {
        *reinterpret_cast<floatx4*>(g.P_out + pidx) = p_lo;
        *reinterpret_cast<floatx4*>(g.P_out + pidx + 8) = p_hi;
      }
    }
  }
  IKF_TSTAMP(41)
}

// ILV: the first MFMA group's memory instructions (LDS store of the next A tile, its successor's global load) sit BETWEEN the group's
// dependent MFMAs - in the shadow of the wave's own matrix-pipe latency - instead of in front of them.  Pays when several workgroups' or
// 16 waves' worth of other work shares the SIMD (257 .. 768 rows: -1 .. -1.5 % per call), costs with one 8-wave workgroup per CU
// (129 .. 256 rows: +1.6 %), so the launcher picks it by row count.
template <bool EPI_RED, int NH, bool FUSE = false, bool ILV = false>
__global__ __launch_bounds__(NH * KKS * 64) void k_flow_gemm_skinny(FusedGemmArgs g, std::conditional_t<FUSE, FuseTail, NoTail> ft) {
  static_assert(!FUSE || EPI_RED, "the fused tail follows the partial-sum epilogue");
  constexpr int BM = KBM, BN = NH * 32, BK = KBK, NT = NH * KKS * 64, KS = KKS;
  constexpr int LDK = BK + 4;
  constexpr int KQ4 = BK / 4;           // float4 per tile row
  constexpr int NFA = BM * KQ4 / NT;    // float4 of the A tile per thread per stage
  constexpr int STAGE = BM * LDK;       // only A rows are staged through LDS
  static_assert(BM * KQ4 % NT == 0 && NFA >= 1, "tile/threads mismatch");
  static_assert(KKG == 2, "the iteration below is written for two MFMA groups per wave per tile");
  static_assert(skinny_lds<NH>() >= sizeof(float) * 2 * STAGE, "the two A stages must fit");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BM][LDK] / reduction scratch

  const int M = g.M, N = g.N, K = g.K;
  const int tiles_n = N / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);  // wave id as a scalar
  const int nh = wave % NH, kq = wave / NH;

  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // A rows go through LDS (both column halves read them).  Every W element is used by exactly one wave, so the W
  // fragments skip LDS: each lane fetches its own float4 (row n0 + nh*32 + lane%32, k-slice kq, 4 k per MFMA group)
  // straight into registers two tiles ahead, from the fragment-major image (k_wfrag_pack) where a wave's fetch is
  // one contiguous 1 KB burst (32 rows 4 KB apart in the row-major weight all land on one L2 channel).
  // Addresses are (scalar base) + (loop-invariant 32-bit lane offset): the loop carries no vector address arithmetic -
  // every VALU instruction issued between MFMAs delays the matrix pipe of this latency-bound kernel.
  unsigned aoff[NFA];  // byte offset of this thread's float4 in the A operand (tile 0)
  int ldst[NFA];
#pragma unroll
  for (int i = 0; i < NFA; ++i) {
    const int f = t + i * NT, row = f / KQ4, c4 = f - row * KQ4;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    aoff[i] = ((unsigned)gr * (unsigned)K + c4 * 4) * 4u;
    ldst[i] = row * LDK + c4 * 4;
  }
  constexpr int WTILE = KS * KKG * 256;  // floats per (32-column tile, k tile) = 32 x 128
  const int KT = K / BK;
  // buffer loads: (SGPR descriptor) + (loop-invariant 32-bit VGPR offset) + (scalar tile offset)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wf), 0, 0x7fffffff, 0x00020000);
  const unsigned wtile0 = (unsigned)(tn * NH + nh) * KT;                   // (this wave's 32-column tile, k tile 0)
  const unsigned woff = (unsigned)(kq * (KKG * 256) * 4) + lane * 16u;     // wave's k-slice + lane, bytes
  const int fragA = (lane & 31) * LDK + kq * KKW + (lane >> 5) * 4;
#define IKK_LDA(i, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i], __builtin_amdgcn_readfirstlane((kt_) * (BK * 4)), 0))
#define IKK_LDW(kk, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff + (kk) * 1024, __builtin_amdgcn_readfirstlane((wtile0 + (kt_)) * (WTILE * 4)), 0))

  IKF_TSTAMP(0)
  floatx4 rg[NFA], w0[KKG], w1[KKG];
#pragma unroll
  for (int i = 0; i < NFA; ++i) rg[i] = IKK_LDA(i, 0);
#pragma unroll
  for (int kk = 0; kk < KKG; ++kk) w0[kk] = IKK_LDW(kk, 0);
#pragma unroll
  for (int i = 0; i < NFA; ++i) *reinterpret_cast<floatx4*>(smem + ldst[i]) = rg[i];
  {
    const int k1 = KT > 1 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NFA; ++i) rg[i] = IKK_LDA(i, k1);
#pragma unroll
    for (int kk = 0; kk < KKG; ++kk) w1[kk] = IKK_LDW(kk, k1);
  }
  __syncthreads();

#define IKK_FRAG(FA, stage, kk) FA = *reinterpret_cast<const floatx4*>(smem + (stage) * STAGE + fragA + (kk) * 8);
#define IKK_MFMA(FA, FB)                                                                  \
  {                                                                                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.x, FB.x, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.y, FB.y, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.z, FB.z, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.w, FB.w, acc, 0, 0, 0);                 \
  }
  // The loop body is branch-free: prefetches past the last tile re-read the last tile (clamped index) and the extra
  // LDS image is never consumed.  With conditional loads the compiler cannot count outstanding loads across the back
  // edge and falls back to s_waitcnt vmcnt(0) in front of the MFMAs, i.e. a full memory round trip per iteration.
#define IKK_WLOAD(WC, kk) if (!(IKK_ABL & 8)) WC[kk] = IKK_LDW(kk, k2);
  // One barrier per tile, two LDS stages: A tile kt+1 is written into the other stage at the top of iteration kt (all
  // its readers - iteration kt-1 - finished before the barrier of iteration kt-1) and its first fragment is fetched
  // behind the barrier.  WC holds the W fragments of tile kt, WN those of tile kt+1; each WC[kk] is refilled with tile
  // kt+2 as soon as its MFMA group has issued.  The loop is unrolled by two so both the stage index and the register
  // set are compile-time: every LDS address is an immediate offset.  Four waves share a SIMD, so while one of them
  // issues its loads / LDS traffic the other three keep the matrix pipe fed.
#ifndef IKK_ABL   // probes only (tools/gemm_probe.hip): ablations of the loop's memory side - bit 0 no barrier, 1 no LDS store, 2 no A load, 3 no W load
#define IKK_ABL 0
#endif
#define IKK_ITER(WC, WN, CUR, NXT)                                                        \
  {                                                                                       \
    const int k2 = (kt + 2 < KT) ? kt + 2 : KT - 1;                                       \
    IKK_FRAG(fa1, CUR, 1)                                                                 \
    if constexpr (ILV) {                                                                  \
      IKK_PIN                                                                             \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.x, WC[0].x, acc, 0, 0, 0);           \
      IKK_PIN                                                                             \
      if (!(IKK_ABL & 2)) { _Pragma("unroll") for (int i = 0; i < NFA; ++i) *reinterpret_cast<floatx4*>(smem + (NXT) * STAGE + ldst[i]) = rg[i]; } \
      IKK_PIN                                                                             \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.y, WC[0].y, acc, 0, 0, 0);           \
      IKK_PIN                                                                             \
      if (!(IKK_ABL & 4)) { _Pragma("unroll") for (int i = 0; i < NFA; ++i) rg[i] = IKK_LDA(i, k2); }  \
      IKK_PIN                                                                             \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.z, WC[0].z, acc, 0, 0, 0);           \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.w, WC[0].w, acc, 0, 0, 0);           \
    } else {                                                                              \
      if (!(IKK_ABL & 2)) { _Pragma("unroll") for (int i = 0; i < NFA; ++i) *reinterpret_cast<floatx4*>(smem + (NXT) * STAGE + ldst[i]) = rg[i]; } \
      if (!(IKK_ABL & 4)) { _Pragma("unroll") for (int i = 0; i < NFA; ++i) rg[i] = IKK_LDA(i, k2); }  \
      IKK_PIN                                                                             \
      IKK_MFMA(fa0, WC[0])                                                                \
    }                                                                                     \
    IKK_WLOAD(WC, 0)                                                                      \
    IKK_PIN                                                                               \
    if (!(IKK_ABL & 1)) __syncthreads();                                                  \
    IKK_FRAG(fa0, NXT, 0)                                                                 \
    IKK_PIN                                                                               \
    IKK_MFMA(fa1, WC[1])                                                                  \
    IKK_WLOAD(WC, 1)                                                                      \
    IKK_PIN                                                                               \
    IKK_STAMP                                                                             \
    ++kt;                                                                                 \
  }
#define IKK_PIN __builtin_amdgcn_sched_barrier(0);  // keep the hand-placed load / LDS / MFMA order
#if defined(IKF_TRACE) && !defined(IKK_NO_STAGE_STAMP)   // (a stamp per stage costs wave 0 - and behind the barrier everyone - ~240 cycles)
#define IKK_STAMP if (kt < 32) IKF_TSTAMP(2 + kt)
#else
#define IKK_STAMP
#endif
  floatx4 fa0, fa1;
  IKK_FRAG(fa0, 0, 0)
  IKF_TSTAMP(1)
  for (int kt = 0; kt < KT;) {  // KT is even (launcher: K % (2*BK) == 0)
    IKK_ITER(w0, w1, 0, 1)
    IKK_ITER(w1, w0, 1, 0)
  }
#undef IKK_ITER
#undef IKK_PIN
#undef IKK_STAMP
#undef IKK_WLOAD
#undef IKK_FRAG
#undef IKK_MFMA
#undef IKK_LDA
#undef IKK_LDW
  __syncthreads();  // all fragment reads done before the stage area is reused
  skinny_tail<EPI_RED, NH, FUSE>(g, acc, smem, m0, n0, t, lane, wave, nh, kq);
  if constexpr (FUSE) {
    static_assert(skinny_lds<NH>() >= sizeof(float) * tail_lds_floats(BM, BN), "the tail's LDS fits");
    tail_next_entry<NT, BM, BN>(ft.e, ft.ts, smem, tm, m0, n0, tn == 0, t);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches, first hidden contraction of a subnet: k_subnet_entry and k_flow_gemm_skinny in ONE launch.
// A launch at <= 512 rows is a latency chain (73 dependent launches, each ~ one memory round trip + its K loop + a
// kernel boundary); this form removes one launch per subnet, the first-Linear round trip through HBM and every barrier of
// the K loop:
//   * every workgroup of a row tile (32 rows x NH*32 columns) finishes the pending coupling of ITS 32 rows (the cheap
//     phase the column-split entry kernel already repeats per workgroup) and evaluates the whole first Linear + LeakyReLU
//     of those rows - [32 x 16] . [16 x K] - ON THE MATRIX PIPE straight into LDS (A_full[32][K + 4], 131.6 KB at
//     K = 1024): K/32 column blocks x 8 MFMAs, 4096 cycles per SIMD.  (On the VALU the same 360 k FMA per workgroup cost
//     14 k cycles - measured, r02 - because the 16 column-tile workgroups of a row tile all repeat them.)  The accumulator
//     starts from the bias and k runs upward inside and across the MFMAs, so the result is the entry kernel's fmaf chain;
//   * the K loop reads its A fragments from that resident tile: no A loads, no LDS stage writes, no barrier per stage;
//     the W-fragment stream of the first two k tiles and the first-Linear weights are requested before the pending phase;
//   * workgroups with tn == 0 publish the new flow state.
// Needs K <= 1024 (LDS), K/32 divisible by the wave count, and <= 256 tiles (one workgroup per CU).
// ---------------------------------------------------------------------------------------------------------------
constexpr int EG_ROWS = 32;
constexpr int EG_ULD = ROWBUF + 1;  // row stride of the subnet-input tile (17: a column read across 32 rows is conflict-free)
constexpr size_t entry_gemm_lds(int K) { return sizeof(float) * ((size_t)EG_ROWS * (K + 4) + 2 * EG_ROWS * ROWBUF + EG_ROWS * EG_ULD); }

template <bool EPI_RED, int NH>
__global__ __launch_bounds__(NH * KKS * 64) void k_entry_gemm_skinny(EntryArgs e, FusedGemmArgs g, int n_in) {
  constexpr int BN = NH * 32, BK = KBK, NT = NH * KKS * 64, NW = NH * KKS;
  constexpr int R = EG_ROWS;
  constexpr int BPW_MAX = 32 / NW;  // first-Linear column blocks per wave at K = 1024 (2 with 16 waves, 4 with 8)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = g.K, N = g.N;
  const int LDKF = K + 4;
  float* A_full = smem;                    // [R][K + 4]; reused by skinny_tail after the loop
  float* cat = smem + (size_t)R * LDKF;    // [R][ROWBUF]
  float* sums = cat + R * ROWBUF;          // [R][ROWBUF] slot-sum scratch
  float* U = sums + R * ROWBUF;            // [R][EG_ULD] the subnet input rows [x_part, pose, 0...]

  const int tiles_n = N / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * R, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nh = wave % NH, kq = wave / NH;
  const int M = e.M, D = e.D;

  // ---- pending coupling, load half - issued FIRST so that the wait in front of the slot sums does not also wait for the
  // operand prefetches queued behind them (VMEM returns in order)
  PendingLoads<(R * ROWBUF + NT - 1) / NT> pl;
  pending_issue_loads<NT, R>(e.pend, e.x_src, D, e.L1, m0, M, t, pl);

  // ---- W fragments of k tiles 0 and 1 of the contraction (independent of everything below)
  constexpr int WTILE = KKS * KKG * 256;
  const int KT = K / BK;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wf), 0, 0x7fffffff, 0x00020000);
  const unsigned wtile0 = (unsigned)(tn * NH + nh) * KT;
  const unsigned woff = (unsigned)(kq * (KKG * 256) * 4) + lane * 16u;
#define IKE_LDW(kk, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff + (kk) * 1024, __builtin_amdgcn_readfirstlane((wtile0 + (kt_)) * (WTILE * 4)), 0))
  floatx4 w0[KKG], w1[KKG];
#pragma unroll
  for (int kk = 0; kk < KKG; ++kk) w0[kk] = IKE_LDW(kk, 0);
  {
    const int k1 = KT > 1 ? 1 : 0;
#pragma unroll
    for (int kk = 0; kk < KKG; ++kk) w1[kk] = IKE_LDW(kk, k1);
  }

  // ---- first-Linear operands of this wave's column blocks.  The product is taken transposed - MFMA "A" operand = W1^T
  // (lane % 32 = output column of the block), "B" operand = the input rows (lane % 32 = row) - so that a lane's accumulator
  // registers 4q .. 4q+3 are four CONSECUTIVE columns of one row: the tile goes to LDS as 16-byte stores
  const int bpw = (K / 32) / NW;  // column blocks per wave (launcher: divides, <= BPW_MAX)
  const int hs = lane >> 5, cl = lane & 31;
  float wb[BPW_MAX][8];
  floatx4 bias4[BPW_MAX][4];
#pragma unroll
  for (int i = 0; i < BPW_MAX; ++i) {
    const int cb = i < bpw ? wave * bpw + i : 0;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int k = 2 * s8 + hs;
      const float v = e.w1t[(size_t)(k < n_in ? k : 0) * e.width + cb * 32 + cl];
      wb[i][s8] = k < n_in ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // accumulator register 4q + j  <->  column 8q + 4*hs + j of the block
      floatx4 bv = *reinterpret_cast<const floatx4*>(e.b1 + cb * 32 + 8 * q + 4 * hs);
      if (e.ps.softflow != 0.0f) bv += e.ps.softflow * *reinterpret_cast<const floatx4*>(e.w1soft + cb * 32 + 8 * q + 4 * hs);
      bias4[i][q] = bv;
    }
  }

  // ---- pose element of this thread's (row, input column) item, requested before the pending phase
  const int ur = t / ROWBUF, uk = t % ROWBUF;
  float pose_v = 0.f;
  if (t < R * ROWBUF && uk >= e.n_x && uk < n_in) {
    int gr = m0 + ur;
    gr = gr < M ? gr : M - 1;
    const long long grow = e.row0 + gr;
    const long long pm = grow < e.ps.n_mod ? grow : (e.ps.n_mod == 1 ? 0 : grow % e.ps.n_mod);
    const long long pi = e.ps.idx ? (long long)e.ps.idx[pm] : pm;
    pose_v = e.ps.poses[pi * e.ps.stride + (uk - e.n_x)];
  }
  IKF_TSTAMP(0)
  finish_pending_rows<NT, R>(e.pend, pl, D, e.L1, e.clamp, m0, cat, sums, t);
  IKF_TSTAMP(1)
  if (t < R * ROWBUF) {
    if (tn == 0 && uk < D && m0 + ur < M) e.x_dst[(size_t)(m0 + ur) * D + uk] = cat[ur * ROWBUF + state_src(e.pend, uk)];
    U[ur * EG_ULD + uk] = uk < e.n_x ? cat[ur * ROWBUF + state_src(e.pend, e.x_off + uk)] : pose_v;  // 0 beyond n_in
  }
  __syncthreads();

  // ---- first Linear + LeakyReLU of the 32 rows on the matrix pipe -> A_full
  {
    float ua[8];  // "B" fragment of step s: U[row = lane % 32][k = 2s + lane / 32]
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) ua[s8] = U[cl * EG_ULD + 2 * s8 + hs];
#pragma unroll
    for (int i = 0; i < BPW_MAX; ++i) {
      if (i < bpw) {
        floatx16 a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a1[4 * q + 0] = bias4[i][q].x; a1[4 * q + 1] = bias4[i][q].y; a1[4 * q + 2] = bias4[i][q].z; a1[4 * q + 3] = bias4[i][q].w;
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
          if (2 * s8 < n_in) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i][s8], ua[s8], a1, 0, 0, 0);  // steps past n_in add 0 * 0
        float* dst = A_full + (size_t)cl * LDKF + (wave * bpw + i) * 32 + 4 * hs;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          floatx4 v = {a1[4 * q + 0], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]};
          v.x = v.x > 0.f ? v.x : v.x * e.slope;
          v.y = v.y > 0.f ? v.y : v.y * e.slope;
          v.z = v.z > 0.f ? v.z : v.z * e.slope;
          v.w = v.w > 0.f ? v.w : v.w * e.slope;
          *reinterpret_cast<floatx4*>(dst + 8 * q) = v;
        }
      }
    }
  }
  __syncthreads();
  IKF_TSTAMP(2)

  // ---- K loop: A fragments from the resident tile, W fragments streamed two tiles ahead; no barriers
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* fragA = A_full + (size_t)cl * LDKF + kq * KKW + hs * 4;
#define IKE_MFMA(FA, FB)                                                                  \
  {                                                                                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.x, FB.x, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.y, FB.y, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.z, FB.z, acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.w, FB.w, acc, 0, 0, 0);                 \
  }
  // the A fragments of k tile kt + 1 are read from LDS while tile kt's MFMAs run (clamped index: the last read is redundant)
#define IKE_ITER(WC)                                                                      \
  {                                                                                       \
    const int k2 = (kt + 2 < KT) ? kt + 2 : KT - 1;                                       \
    const int k1 = (kt + 1 < KT) ? kt + 1 : KT - 1;                                       \
    const floatx4 fn0 = *reinterpret_cast<const floatx4*>(fragA + k1 * BK);               \
    const floatx4 fn1 = *reinterpret_cast<const floatx4*>(fragA + k1 * BK + 8);           \
    IKE_MFMA(fa0, WC[0])                                                                  \
    WC[0] = IKE_LDW(0, k2);                                                               \
    IKE_MFMA(fa1, WC[1])                                                                  \
    WC[1] = IKE_LDW(1, k2);                                                               \
    fa0 = fn0; fa1 = fn1;                                                                 \
    ++kt;                                                                                 \
  }
  static_assert(KKG == 2, "two MFMA groups per wave per k tile");
  floatx4 fa0 = *reinterpret_cast<const floatx4*>(fragA), fa1 = *reinterpret_cast<const floatx4*>(fragA + 8);
  for (int kt = 0; kt < KT;) {  // KT is even (launcher: K % (2*BK) == 0)
    IKE_ITER(w0)
    IKE_ITER(w1)
  }
#undef IKE_ITER
#undef IKE_MFMA
#undef IKE_LDW
  __syncthreads();  // every wave is done with A_full before the tail reuses the memory
  IKF_TSTAMP(3)
  skinny_tail<EPI_RED, NH>(g, acc, smem, m0, n0, t, lane, wave, nh, kq);
}

// ---------------------------------------------------------------------------------------------------------------
// <= 128 rows: tiles of 16 rows x 32 columns on v_mfma_f32_16x16x4_f32 (r03).  With 32-row tiles a batch of <= 128 rows gives at
// most 128 workgroups, and each of them carries 512 MFMAs (3.4 us on its CU's four SIMDs) while the other half of the chip idles:
// the launch is matrix-pipe bound on too few CUs.  16-row tiles double the workgroups (8 row tiles x 32 column tiles = 256 at 128
// rows) and halve everything a workgroup repeats per row: its K loop, and in the one-launch head its copy of the first Linear.
//   * MFMA shape: D[16 x 16] += A[16 x 4] . B[4 x 16]; lane l feeds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16] and gets
//     D[i = 4 * (l / 16) + v][j = l % 16] in accumulator register v.  A = activation rows, B = weight columns: a lane reads four
//     consecutive k of its row / column at k offset 4 * (l / 16) of the wave's 16-k slice; component c feeds MFMA c (the same k
//     permutation on both sides, as in the 32x32 kernels).  Two column blocks per tile share the A fragment.
//   * the W fragments come from the SAME fragment-major image as the 32x32 kernels' (k_wfrag_pack): element (kk, lane32) of a
//     (32-column tile, k tile, k slice) holds W[col = lane32 % 32][k = 8 kk + 4 (lane32 / 32) ..]; lane l of column block cb wants
//     col = 16 cb + l % 16, k = 4 (l / 16), i.e. element (kk = l / 32, lane32 = 16 cb + l % 16 + 32 ((l / 16) % 2)): a wave's fetch
//     is four 256-byte runs of the 2 KB the 32x32 kernel fetches as one.
//   * partial-sum slots stay 32 columns wide (one per workgroup): the pending coupling reads the same 32 slots at width 1024.
// The tile's summation order differs from the 32x32 kernels' (k slices as before, but the last Linear's 32 columns in one MFMA
// chain): results agree with the other tile shapes to rounding, like every change of tile shape.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S16_ROWS = 16;     // rows of one MFMA row block
constexpr int S16_MAX_RB = 2;    // row blocks per tile: 1 (16-row tiles) or 2 (32-row tiles, 129 .. 256 rows)
constexpr size_t skinny16_tail_lds() {  // the largest case: two row blocks x two column blocks
  return sizeof(float) * ((size_t)KKS * S16_MAX_RB * 2 * 4 * 64 + (size_t)(S16_MAX_RB * S16_ROWS + 16) * (32 + 4));
}
constexpr size_t entry_gemm16_lds() {  // cat, sums, the input rows, and the per-wave slot sums of the pending phase
  return sizeof(float) * ((size_t)2 * S16_MAX_RB * S16_ROWS * ROWBUF + S16_MAX_RB * S16_ROWS * EG_ULD + 16 + (size_t)KKS * S16_MAX_RB * S16_ROWS * ROWBUF);
}
#define IKF_MFMA16(FA, FB, ACC)                                                      \
  {                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(FA.x, FB.x, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(FA.y, FB.y, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(FA.z, FB.z, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(FA.w, FB.w, ACC, 0, 0, 0);            \
  }
// byte offset of lane l's W fragment of column block cb inside a (32-column tile, k tile) of the fragment-major image
__device__ __forceinline__ unsigned wfrag16_off(int kq, int lane, int cb) {
  const int g = lane >> 4;
  return (unsigned)(((kq * KKG + (g >> 1)) * 64 + cb * 16 + (lane & 15) + 32 * (g & 1)) * 16);
}

// tail of the 16x16x4 kernels: k-slice reduction + bias / LeakyReLU + store or last-Linear partial sums.  Call after a barrier that
// ends every LDS access of the caller.  acc[rb][cb][v] = tile[row = 16 rb + 4 (lane / 16) + v][col = 16 cb + lane % 16] over this
// wave's k slice.  NRB x NCB = 16-row x 16-column blocks per tile: 1 x 1 (<= 64 rows), 1 x 2 (<= 128 rows), 2 x 2 (<= 256 rows).
// What the tail reads from memory - the bias of the columns a wave finishes and (EPI_RED) this thread's piece of the last Linear's
// slice - is requested at the START of the kernel (skinny16_prefetch): fetched in the tail each costs a memory round trip there.
template <int NCB, int NRB>
struct Skinny16Pre {
  static constexpr int NREG = NRB * NCB * 4, RPW = (NREG + KKS - 1) / KKS;
  float bias[RPW];
  floatx4 wl;  // w_last[o = t / (BN / 4)][n0 + 4 (t % (BN / 4)) ..] for t < 16 BN / 4 (zero beyond n_out)
};
template <bool EPI_RED, int NCB, int NRB>
__device__ __forceinline__ void skinny16_prefetch(const FusedGemmArgs& g, int n0, int t, int lane, int kq, Skinny16Pre<NCB, NRB>& pre) {
  constexpr int BN = 16 * NCB, NREG = NRB * NCB * 4, RPW = Skinny16Pre<NCB, NRB>::RPW;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int id = RPW * kq + j;
    pre.bias[j] = id < NREG ? g.bias[n0 + 16 * ((id >> 2) % NCB) + (lane & 15)] : 0.f;
  }
  pre.wl = floatx4{0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI_RED) {
    static_assert(16 * (BN / 4) <= KKS * 64, "one float4 of the last Linear's slice per thread");
    const int o = t / (BN / 4), c4 = t - o * (BN / 4);
    if (t < 16 * (BN / 4) && o < g.n_out) pre.wl = *reinterpret_cast<const floatx4*>(g.w_last + (size_t)o * g.N + n0 + c4 * 4);
  }
}
template <bool EPI_RED, int NCB, int NRB>
__device__ __forceinline__ void skinny16_tail(const FusedGemmArgs& g, const floatx4 (&acc)[NRB][NCB], const Skinny16Pre<NCB, NRB>& pre,
                                              float* smem, int m0, int n0, int t, int lane, int kq) {
  constexpr int BN = 16 * NCB, BM = 16 * NRB, LDT = BN + 4;
  constexpr int NREG = NRB * NCB * 4;                 // accumulator registers of a lane
  constexpr int RPW = (NREG + KKS - 1) / KKS;         // registers a wave finishes (2 for 2 x 2 blocks, else 1)
  const int N = g.N;
  float* red = smem;  // [KKS][NREG][64]
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[(kq * NREG + (rb * NCB + cb) * 4 + v) * 64 + lane] = acc[rb][cb][v];
  __syncthreads();
  // wave kq finishes registers RPW kq .. RPW kq + RPW - 1 of every lane: the k slices in fixed order 0, 1, ..
  float fin[RPW];
  int frow[RPW], fcol[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int id = RPW * kq + j;  // (rb * NCB + cb) * 4 + v
    const int v = id & 3, cb = (id >> 2) % NCB, rb = (id >> 2) / NCB;
    frow[j] = 16 * rb + 4 * (lane >> 4) + v;
    fcol[j] = 16 * cb + (lane & 15);
    fin[j] = 0.f;
    if (id < NREG) {
#pragma unroll
      for (int q = 0; q < KKS; ++q) fin[j] += red[(q * NREG + id) * 64 + lane];
      fin[j] += pre.bias[j];
      fin[j] = fin[j] > 0.f ? fin[j] : fin[j] * g.slope;
    }
  }
  if constexpr (!EPI_RED) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      if (RPW * kq + j < NREG) {
        float* dst = g.C + (size_t)(m0 + frow[j]) * N + n0 + fcol[j];  // row-padded buffer: unpredicated
        if (g.wt_stores) __hip_atomic_store(dst, fin[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = fin[j];
      }
    }
  } else {
    float* T = smem + KKS * NREG * 64;  // [BM][LDT], behind red[] (other waves may still be summing)
    float* Wl = T + BM * LDT;           // [16][LDT]: w_last rows (zero beyond n_out), this tile's columns
#pragma unroll
    for (int j = 0; j < RPW; ++j)
      if (RPW * kq + j < NREG) T[frow[j] * LDT + fcol[j]] = fin[j];
    if (t < 16 * (BN / 4)) {
      const int o = t / (BN / 4), c4 = t - o * (BN / 4);
      *reinterpret_cast<floatx4*>(Wl + o * LDT + c4 * 4) = pre.wl;
    }
    __syncthreads();
    if (kq < NRB) {  // wave rb: P[row][o] = sum over the tile's columns of h[row][col] * w_last[o][col] for row block rb, one MFMA chain
      floatx4 pacc = {0.f, 0.f, 0.f, 0.f};
      const int gq = lane >> 4;
#pragma unroll
      for (int h = 0; h < NCB; ++h) {
        const floatx4 a4 = *reinterpret_cast<const floatx4*>(T + (16 * kq + (lane & 15)) * LDT + 16 * h + 4 * gq);
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(Wl + (lane & 15) * LDT + 16 * h + 4 * gq);
        IKF_MFMA16(a4, b4, pacc)
      }
      // pacc[v] = P[row = 16 rb + 4 gq + v][o = lane % 16]
      float* pout = g.P_out + (size_t)(n0 / BN) * g.p_slot_stride + (size_t)(m0 + 16 * kq + 4 * gq) * IKF_PSTRIDE + (lane & 15);
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) pout[vv * IKF_PSTRIDE] = pacc[vv];
    }
  }
}

// hidden contraction on 16x16x4 MFMAs, tiles of 16 NRB rows x 16 NCB columns, eight waves = eight k slices.  Every operand element
// feeds exactly ONE wave, so neither W nor A goes through LDS: each lane fetches its own W fragments from the fragment-major image and
// its own A fragments - row m0 + 16 rb + lane % 16, four k at 16 kq + 4 (lane / 16) of the k tile - straight from the activation
// buffer (64-byte runs per row and wave; the eight waves together read whole 512-byte row segments).  No LDS traffic and no barrier in
// the loop.  (The same on the 32x32x2 kernel, whose fragments are 32-byte runs, loses: 0.520 -> 0.533 ms per call at 256 rows.)
// DEEP (K <= 8 k tiles): the whole operand stream is requested up front, tile by tile (sched_barrier keeps that order: tile 0 must not
// queue behind the rest): a k tile is only 0.1 - 0.4 us of matrix-pipe time per SIMD, two tiles of lead do not cover a round trip.
constexpr int kDeepTiles = 8;
// Hand-over hooks of the bodies below.  NoWait: the operands were complete when the launch started (stand-alone kernels).  k_flow_chain16
// passes an XcdWait (further down): what does not depend on sibling workgroups - weights, biases, poses - is requested BEFORE the wait.
struct NoWait {
  __device__ __forceinline__ bool operator()() const { return true; }
};
// XL ("XCD-local"): the activations / partial sums / state this body reads were written inside this launch by workgroups on the same XCD:
// they sit in the shared L2, and the loads carry sc1 so that they miss this CU's L1 (which may hold the buffers' previous contents).
template <bool EPI_RED, bool DEEP, int NCB, int NRB, bool XL, class Wait>
__device__ __forceinline__ bool gemm16_body(const FusedGemmArgs& g, int tm, int tn, float* smem, const Wait& wait) {
  constexpr int BM = S16_ROWS * NRB, BN = 16 * NCB, BK = KBK;
  constexpr int AUX_A = XL ? 16 : 0;
  const int M = g.M, K = g.K;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, kq = __builtin_amdgcn_readfirstlane(t >> 6);
  IKF_TSTAMP(10)
  Skinny16Pre<NCB, NRB> pre;
  skinny16_prefetch<EPI_RED, NCB, NRB>(g, n0, t, lane, kq, pre);
  floatx4 acc[NRB][NCB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  constexpr int WTILE = KKS * KKG * 256;
  const int KT = K / BK;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wf), 0, 0x7fffffff, 0x00020000);
  const unsigned wtile0 = (unsigned)(n0 >> 5) * KT;  // the 32-column tile of the fragment-major image this tile lies in
  unsigned woffs[NCB], afrag[NRB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) woffs[cb] = wfrag16_off(kq, lane, ((n0 >> 4) & 1) + cb);
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    int far = m0 + 16 * rb + (lane & 15);
    far = far < M ? far : M - 1;
    afrag[rb] = ((unsigned)far * (unsigned)K + kq * KKW + (lane >> 4) * 4) * 4u;
  }
#define IK6_LDA(off, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, __builtin_amdgcn_readfirstlane((kt_) * (BK * 4)), AUX_A))
#define IK6_LDW(off, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsW, off, __builtin_amdgcn_readfirstlane((wtile0 + (kt_)) * (WTILE * 4)), 0))
  if constexpr (DEEP) {
    floatx4 aall[kDeepTiles][NRB], wall[kDeepTiles][NCB];
    if constexpr (XL) {  // the whole W stream first - it does not depend on the siblings - then the wait, then the A stream
#pragma unroll
      for (int kt = 0; kt < kDeepTiles; ++kt) {
        const int kc = kt < KT ? kt : KT - 1;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wall[kt][cb] = IK6_LDW(woffs[cb], kc);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!wait()) return false;
#pragma unroll
      for (int kt = 0; kt < kDeepTiles; ++kt) {
        const int kc = kt < KT ? kt : KT - 1;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) aall[kt][rb] = IK6_LDA(afrag[rb], kc);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < kDeepTiles; ++kt) {  // unconditional loads (clamped index past the last tile): the compiler counts them
        const int kc = kt < KT ? kt : KT - 1;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) aall[kt][rb] = IK6_LDA(afrag[rb], kc);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wall[kt][cb] = IK6_LDW(woffs[cb], kc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < kDeepTiles; ++kt) {
      if (kt < KT) {  // uniform
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) IKF_MFMA16(aall[kt][rb], wall[kt][cb], acc[rb][cb])
      }
    }
  } else {
    // two k tiles ahead; branch-free (clamped prefetch index)
    floatx4 ac[NRB], an[NRB], wc[NCB], wn[NCB];
    const int k1 = KT > 1 ? 1 : 0;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { wc[cb] = IK6_LDW(woffs[cb], 0); wn[cb] = IK6_LDW(woffs[cb], k1); }
    if (!wait()) return false;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) { ac[rb] = IK6_LDA(afrag[rb], 0); an[rb] = IK6_LDA(afrag[rb], k1); }
    for (int kt = 0; kt < KT; ++kt) {
      const int k2 = (kt + 2 < KT) ? kt + 2 : KT - 1;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) IKF_MFMA16(ac[rb], wc[cb], acc[rb][cb])
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) { ac[rb] = an[rb]; an[rb] = IK6_LDA(afrag[rb], k2); }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { wc[cb] = wn[cb]; wn[cb] = IK6_LDW(woffs[cb], k2); }
    }
  }
#undef IK6_LDA
#undef IK6_LDW
  IKF_TSTAMP(11)
  skinny16_tail<EPI_RED, NCB, NRB>(g, acc, pre, smem, m0, n0, t, lane, kq);
  IKF_TSTAMP(12)
  return true;
}
template <bool EPI_RED, bool DEEP, int NCB = 2, int NRB = 1>
__global__ __launch_bounds__(KKS * 64) void k_flow_gemm_skinny16(FusedGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // the tail's scratch only
  const int tiles_n = g.N / (16 * NCB);
  gemm16_body<EPI_RED, DEEP, NCB, NRB, false>(g, blockIdx.x / tiles_n, blockIdx.x % tiles_n, smem, NoWait{});
}

// The pending coupling of R = 16 NRB rows for the 16x16x4 head, by all eight waves.  A memory instruction with 64-bit per-lane addresses
// costs its SIMD ~70 cycles of issue, and the generic form (pending_issue_loads: one dword load per slot and thread) issues 66 of them in
// four waves - tools/small_trace.py put the head's first 9.3 k cycles there.  Here wave w takes slots [w S / 8, (w + 1) S / 8): lane l reads
// 16 bytes - outputs 4 (l % 4) .. + 3 of row l / 4 - per slot through a buffer descriptor (S / 8 <= 8 loads per wave), sums its slots in
// order and parks the partial in LDS; after a barrier thread (row, o) adds the bias and the eight partials in wave order.  (Another fixed
// summation order than slot-by-slot: the 16-row kernels agree with the other forms to rounding, as every tile shape does.)
template <int R>
struct PendingSlots16 {
  static constexpr int PASSES = R / 16;
  floatx4 a[PASSES][8];
  float xv, bias;
};
// SC1: the slots and the state were written inside THIS launch by workgroups of the same XCD (k_flow_chain16): loads that miss the CU's L1.
template <int NT, int R, bool SC1 = false>
__device__ __forceinline__ void pending16_issue(const PendingCoupling& pc, const float* __restrict__ x_src, int D, int L1, int m0, int M,
                                                int t, int lane, int wave, PendingSlots16<R>& ps) {
  const int nl = (pc.which == 1) ? D - L1 : L1;
  const int spw = (pc.slots + 7) >> 3;  // slots per wave (<= 8: the launcher admits at most 64 slots)
  const int r16 = lane >> 2, quad = lane & 3;
  if (pc.P != nullptr) {
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pc.P), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int p = 0; p < PendingSlots16<R>::PASSES; ++p) {
      const unsigned voff = (unsigned)(((m0 + 16 * p + r16) * IKF_PSTRIDE + 4 * quad) * 4);  // P rows are padded to the tile
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int slot = wave * spw + j;
        ps.a[p][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (j < spw && slot < pc.slots && 4 * quad < 2 * nl && m0 + 16 * p + r16 < M)  // (padding rows: nothing to fetch)
          ps.a[p][j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsP, voff, __builtin_amdgcn_readfirstlane((unsigned)slot * (unsigned)pc.slot_stride * 4u), SC1 ? 16 : 0));
      }
    }
  }
  // one (row, column) item per thread for the state and the bias
  const int r = t / ROWBUF, d = t % ROWBUF;
  int gr = m0 + r;
  gr = gr < M ? gr : M - 1;
  ps.xv = (t < R * ROWBUF && d < D) ? load_partial<SC1>(x_src + (size_t)gr * D + d) : 0.f;
  ps.bias = (pc.P != nullptr && t < R * ROWBUF && d < 2 * nl) ? pc.b_last[d] : 0.f;
}
// `part`: KKS * R * ROWBUF floats of LDS scratch; cat / sums as in finish_pending_rows.  Ends with a barrier.
template <int NT, int R>
__device__ __forceinline__ void pending16_finish(const PendingCoupling& pc, const PendingSlots16<R>& ps, int D, int L1, float clamp,
                                                 float* cat, float* sums, float* part, int t, int lane, int wave) {
  static_assert(R * ROWBUF <= NT, "one (row, column) item per thread");
  const int L2 = D - L1;
  const int nl = (pc.which == 1) ? L2 : L1;
  const int off = (pc.which == 1) ? L1 : 0;
  const int r = t / ROWBUF, d = t % ROWBUF;
  if (pc.P != nullptr) {
    const int r16 = lane >> 2, quad = lane & 3;
#pragma unroll
    for (int p = 0; p < PendingSlots16<R>::PASSES; ++p) {
      floatx4 sv = ps.a[p][0];
#pragma unroll
      for (int j = 1; j < 8; ++j) sv += ps.a[p][j];  // slots past this wave's share are zero
      *reinterpret_cast<floatx4*>(part + ((size_t)wave * R + 16 * p + r16) * ROWBUF + 4 * quad) = sv;
    }
    __syncthreads();
    if (t < R * ROWBUF && d < 2 * nl) {
      float sv = ps.bias;
#pragma unroll
      for (int w = 0; w < KKS; ++w) sv += part[((size_t)w * R + r) * ROWBUF + d];
      sums[r * ROWBUF + d] = sv;
    }
    __syncthreads();
  }
  if (t < R * ROWBUF && d < D) {
    float v = ps.xv;
    if (pc.P != nullptr && d >= off && d < off + nl) {
      const int j = d - off;
      const float s_cl = clamp * (0.636f * atanf(sums[r * ROWBUF + j]));
      v = (v - sums[r * ROWBUF + nl + j]) * expf(-s_cl);
    }
    cat[r * ROWBUF + d] = v;
  }
  __syncthreads();
}

// one-launch subnet head on 16x16x4 MFMAs: pending coupling of the tile's 16 NRB rows, the whole first Linear + LeakyReLU of those
// rows on the matrix pipe, K loop.  The hidden activation h1 NEVER LEAVES THE REGISTERS: wave kq evaluates exactly the 16-column blocks
// it will contract - block 8 kt + kq is k slice kq of k tile kt - and the transposed first-Linear product leaves lane l with
// h1[row = 16 rb + l % 16][16 b + 4 (l / 16) + v] in accumulator register v, which is precisely the A fragment (four consecutive k at
// offset 4 (l / 16) of the slice) the K loop's MFMAs want.  No LDS tile, no barrier between the first Linear and the loop.
template <bool EPI_RED, bool DEEP, int NCB, int NRB, bool XL, class Wait>
__device__ __forceinline__ bool entry_gemm16_body(const EntryArgs& e, const FusedGemmArgs& g, int n_in, int tm, int tn, float* smem, const Wait& wait) {
  constexpr int BN = 16 * NCB, BK = KBK, NT = KKS * 64, NW = KKS, R = S16_ROWS * NRB;
  constexpr int BPW_MAX = 8;  // first-Linear 16-column blocks per wave at K = 1024
  const int K = g.K;
  float* cat = smem;               // [R][ROWBUF]   (all four are dead before the tail reuses the memory)
  float* sums = cat + R * ROWBUF;
  float* U = sums + R * ROWBUF;    // [R][EG_ULD]
  float* part = U + R * EG_ULD + 16;  // [KKS][R][ROWBUF] per-wave slot sums (16-byte aligned: R * EG_ULD is a multiple of 16 floats)
  const int m0 = tm * R, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kq = wave;
  const int M = e.M, D = e.D;
  const int gq = lane >> 4, cl = lane & 15;

  IKF_TSTAMP(20)
  PendingSlots16<R> pl;
  if constexpr (!XL) pending16_issue<NT, R>(e.pend, e.x_src, D, e.L1, m0, M, t, lane, wave, pl);  // the critical path's loads go first
  Skinny16Pre<NCB, NRB> pre;
  skinny16_prefetch<EPI_RED, NCB, NRB>(g, n0, t, lane, kq, pre);

  constexpr int WTILE = KKS * KKG * 256;
  const int KT = K / BK;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wf), 0, 0x7fffffff, 0x00020000);
  const unsigned wtile0 = (unsigned)(n0 >> 5) * KT;
  unsigned woffs[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) woffs[cb] = wfrag16_off(kq, lane, ((n0 >> 4) & 1) + cb);
#define IK6_LDW(off, kt_) __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsW, off, __builtin_amdgcn_readfirstlane((wtile0 + (kt_)) * (WTILE * 4)), 0))
  floatx4 wc[NCB], wn[NCB];
  floatx4 wall[DEEP ? kDeepTiles : 1][NCB];
  if constexpr (DEEP) {  // the whole W-fragment stream now: it has the pending and first-Linear phases to arrive in
#pragma unroll
    for (int kt = 0; kt < kDeepTiles; ++kt) {
      const int kc = kt < KT ? kt : KT - 1;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) wall[kt][cb] = IK6_LDW(woffs[cb], kc);
    }
  } else {
    const int k1 = KT > 1 ? 1 : 0;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      wc[cb] = IK6_LDW(woffs[cb], 0);
      wn[cb] = IK6_LDW(woffs[cb], k1);
    }
  }

  // first-Linear operands of this wave's 16-column blocks: MFMA "A" = W1^T (i = column of the block), "B" = the input rows
  // (j = row), so that a lane's four accumulator registers are four CONSECUTIVE columns of one row
  const int bpw = (K / 16) / NW;  // = K / 128 = the k tiles (launcher: <= BPW_MAX); block i of this wave is k slice kq of k tile i
  float wb[BPW_MAX][4];
  floatx4 bias4[BPW_MAX];
  {
    // buffer loads (descriptor + 32-bit lane offset + scalar block offset): a third of the issue cost of 64-bit-address loads
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.w1t), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.b1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.w1soft), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < BPW_MAX; ++i) {
      const int cb = i < bpw ? i * NW + wave : 0;
      const unsigned cboff = __builtin_amdgcn_readfirstlane((unsigned)(cb * 16 * 4));
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 4 * s4 + gq;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW1, (unsigned)(((k < n_in ? k : 0) * e.width + cl) * 4), cboff, 0));
        wb[i][s4] = k < n_in ? v : 0.f;
      }
      floatx4 bv = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB1, (unsigned)(4 * gq * 4), cboff, 0));
      if (e.ps.softflow != 0.0f)
        bv += e.ps.softflow * __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsS1, (unsigned)(4 * gq * 4), cboff, 0));
      bias4[i] = bv;
    }
  }
  // one (row, input column) item per thread: R * ROWBUF <= NT
  static_assert(R * ROWBUF <= NT, "one input element per thread");
  const int ur = t / ROWBUF, uk = t % ROWBUF;
  float pose_v = 0.f;
  if (t < R * ROWBUF && uk >= e.n_x && uk < n_in) {
    int gr = m0 + ur;
    gr = gr < M ? gr : M - 1;
    const long long grow = e.row0 + gr;
    const long long pm = grow < e.ps.n_mod ? grow : (e.ps.n_mod == 1 ? 0 : grow % e.ps.n_mod);
    const long long pi = e.ps.idx ? (long long)e.ps.idx[pm] : pm;
    pose_v = e.ps.poses[pi * e.ps.stride + (uk - e.n_x)];
  }
  if constexpr (XL) {  // everything above is independent of the sibling workgroups; their partial sums and the state come now
    if (!wait()) return false;
    pending16_issue<NT, R, true>(e.pend, e.x_src, D, e.L1, m0, M, t, lane, wave, pl);
  }
  IKF_TSTAMP(21)
  pending16_finish<NT, R>(e.pend, pl, D, e.L1, e.clamp, cat, sums, part, t, lane, wave);
  IKF_TSTAMP(22)
  if (t < R * ROWBUF) {
    if (tn == 0 && uk < D && m0 + ur < M) e.x_dst[(size_t)(m0 + ur) * D + uk] = cat[ur * ROWBUF + state_src(e.pend, uk)];
    U[ur * EG_ULD + uk] = uk < e.n_x ? cat[ur * ROWBUF + state_src(e.pend, e.x_off + uk)] : pose_v;  // 0 beyond n_in
  }
  __syncthreads();
  floatx4 afrag[BPW_MAX][NRB];  // afrag[kt][rb][v] = h1[row = 16 rb + lane % 16][k = 128 kt + 16 kq + 4 (lane / 16) + v]: the K loop's A fragments
  {
    float ua[NRB][4];  // "B" fragment of step s: U[row = 16 rb + lane % 16][k = 4 s + lane / 16]
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) ua[rb][s4] = U[(16 * rb + cl) * EG_ULD + 4 * s4 + gq];
#pragma unroll
    for (int i = 0; i < BPW_MAX; ++i) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        floatx4 a1 = bias4[i];
        if (i < bpw) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            if (4 * s4 < n_in) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][s4], ua[rb][s4], a1, 0, 0, 0);  // k past n_in: 0 * 0
          a1.x = a1.x > 0.f ? a1.x : a1.x * e.slope;
          a1.y = a1.y > 0.f ? a1.y : a1.y * e.slope;
          a1.z = a1.z > 0.f ? a1.z : a1.z * e.slope;
          a1.w = a1.w > 0.f ? a1.w : a1.w * e.slope;
        }
        afrag[i][rb] = a1;
      }
    }
  }
  IKF_TSTAMP(23)
  floatx4 acc[NRB][NCB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  static_assert(BPW_MAX == kDeepTiles, "one first-Linear block per k tile");
  if constexpr (DEEP) {
#pragma unroll
    for (int kt = 0; kt < kDeepTiles; ++kt) {
      if (kt < KT) {  // uniform; no LDS, no barrier
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) IKF_MFMA16(afrag[kt][rb], wall[kt][cb], acc[rb][cb])
      }
    }
  } else {
#pragma unroll
    for (int kt = 0; kt < kDeepTiles; ++kt) {  // W fragments two tiles ahead (unconditional loads, clamped index)
      const int k2 = (kt + 2 < KT) ? kt + 2 : KT - 1;
      if (kt < KT) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) IKF_MFMA16(afrag[kt][rb], wc[cb], acc[rb][cb])
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        wc[cb] = wn[cb];
        wn[cb] = IK6_LDW(woffs[cb], k2);
      }
    }
  }
#undef IK6_LDW
  IKF_TSTAMP(24)
  __syncthreads();  // every wave is done with the input rows in LDS before the tail reuses the memory
  skinny16_tail<EPI_RED, NCB, NRB>(g, acc, pre, smem, m0, n0, t, lane, kq);
  IKF_TSTAMP(25)
  return true;
}
template <bool EPI_RED, bool DEEP, int NCB = 2, int NRB = 1>
__global__ __launch_bounds__(KKS * 64) void k_entry_gemm_skinny16(EntryArgs e, FusedGemmArgs g, int n_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tiles_n = g.N / (16 * NCB);
  entry_gemm16_body<EPI_RED, DEEP, NCB, NRB, false>(e, g, n_in, blockIdx.x / tiles_n, blockIdx.x % tiles_n, smem, NoWait{});
}

// ---------------------------------------------------------------------------------------------------------------
// <= 128 rows, the whole subnet chain in ONE launch (r03): 256 persistent workgroups, one per CU; the hand-over between two layers is
// an arrival counter in the XCD's own L2.
//   * A kernel boundary costs a dependent chain ~2.4 us plus a cold operand fetch (~1.2 us) - 49 times per call.  A hand-over between
//     workgroups anywhere on the chip costs more than that (TailSync above: 19 us per round in tools/xcd_sync_probe.cpp's form), but
//     between workgroups of ONE XCD it is three L2 round trips: stores acknowledged by the shared L2, one atomic add, polls that hit the
//     L2, operand loads that miss the L1 (sc1) and hit the L2 - 1.9 us per round for 32 workgroups exchanging 8 KB, no fence, no
//     write-through.  And the next layer's weights are requested BEFORE the wait.
//   * Decomposition: row tile (16 rows) <-> XCD, column tile (32 columns) <-> one of the XCD's 32 workgroups - exactly the 8 x 32 tiles
//     of the 16-row kernels at 128 rows, whose bodies run here unchanged (entry_gemm16_body, gemm16_body; XL = sibling-written data is
//     loaded with sc1).  All exchange (activations, partial sums, the flow state) stays inside a row tile, i.e. inside an XCD.
//   * Nothing is assumed about placement: a workgroup reads its XCC_ID and takes a ticket from that XCD's counter; (XCC_ID, ticket) is
//     its (row tile, column tile).  The launcher checks once per handle that the dispatcher hands 32 workgroups to each of 8 XCDs; a
//     33rd ticket, or a wait that runs out (~1 s), sets the host-visible give-up word and the abort word that ends every other wait.
//   * The last workgroup to leave zeroes the control words for the next call.
// ---------------------------------------------------------------------------------------------------------------
#ifdef IKF_PROBES
#define IKF_PROBES_PART 2
#include "flow_fused_probes.inc"
#undef IKF_PROBES_PART
#endif
__global__ __launch_bounds__(256) void k_wfrag_pack(const float* __restrict__ W, float* __restrict__ out, int N, int K) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= (size_t)N * K / 4) return;
  const int KT = K / KBK;
  const int lane = (int)(f & 63);
  size_t r = f >> 6;
  const int kk = (int)(r % KKG); r /= KKG;
  const int kq = (int)(r % KKS); r /= KKS;
  const int kt = (int)(r % KT);
  const int tn32 = (int)(r / KT);
  const size_t row = (size_t)tn32 * 32 + (lane & 31);
  const int k = kt * KBK + kq * KKW + kk * 8 + (lane >> 5) * 4;
  reinterpret_cast<floatx4*>(out)[f] = *reinterpret_cast<const floatx4*>(W + row * K + k);
}
hipError_t launch_wfrag_pack(const float* W, int N, int K, float* out, hipStream_t s) {
  if (N % KBN != 0 || K % KBK != 0) return hipErrorInvalidValue;
  const size_t n4 = (size_t)N * K / 4;
  hipLaunchKernelGGL(k_wfrag_pack, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, W, out, N, K);
  return hipGetLastError();
}

template <bool EPI_RED, int NH, bool ILV>
static hipError_t launch_skinny_t(const FusedGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = skinny_lds<NH>();
  auto kern = k_flow_gemm_skinny<EPI_RED, NH, false, ILV>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + KBM - 1) / KBM) * (a.N / (NH * 32));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NH * KKS * 64), smem, s, a, NoTail{});
  return hipGetLastError();
}
template <bool EPI_RED, int NH>
static hipError_t launch_skinny(const FusedGemmArgs& a, hipStream_t s) {
  // (the instruction order inside a stage does not change the arithmetic: same bits either way)
  const long long tiles = (((long long)a.M + KBM - 1) / KBM) * (a.N / (NH * 32));
  if (NH == 2 || tiles > 256) return launch_skinny_t<EPI_RED, NH, true>(a, s);
  return launch_skinny_t<EPI_RED, NH, false>(a, s);
}
#ifdef IKF_PROBES
static hipError_t launch_skinny_tail(const FusedGemmArgs& a, const FuseTail& ft, hipStream_t s) {
  constexpr int NH = 2;
  constexpr size_t smem = skinny_lds<NH>();
  auto kern = k_flow_gemm_skinny<true, NH, true>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + KBM - 1) / KBM) * (a.N / (NH * 32));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NH * KKS * 64), smem, s, a, ft);
  return hipGetLastError();
}
#endif  // IKF_PROBES

template <bool EPI_RED, int NH>
static hipError_t launch_entry_gemm_t(const EntryArgs& e, const FusedGemmArgs& a, int n_in, hipStream_t s) {
  const size_t tail = skinny_lds<NH>();
  size_t smem = entry_gemm_lds(a.K);
  if (smem < tail) smem = tail;
  auto kern = k_entry_gemm_skinny<EPI_RED, NH>;
  static bool lds_ok[64] = {};
  if (hipError_t err = ensure_dynamic_lds(kern, (size_t)160 * 1024, lds_ok); err != hipSuccess) return err;
  const long long grid = (((long long)a.M + EG_ROWS - 1) / EG_ROWS) * (a.N / (NH * 32));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NH * KKS * 64), smem, s, e, a, n_in);
  return hipGetLastError();
}

template <bool EPI_RED, bool DEEP, int NCB, int NRB = 1>
static hipError_t launch_skinny16(const FusedGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = skinny16_tail_lds();
  auto kern = k_flow_gemm_skinny16<EPI_RED, DEEP, NCB, NRB>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + S16_ROWS * NRB - 1) / (S16_ROWS * NRB)) * (a.N / (16 * NCB));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KKS * 64), smem, s, a);
  return hipGetLastError();
}
template <bool EPI_RED, bool DEEP, int NCB, int NRB = 1>
static hipError_t launch_entry_gemm16(const EntryArgs& e, const FusedGemmArgs& a, int n_in, hipStream_t s) {
  size_t smem = entry_gemm16_lds();
  if (smem < skinny16_tail_lds()) smem = skinny16_tail_lds();
  auto kern = k_entry_gemm_skinny16<EPI_RED, DEEP, NCB, NRB>;
  static bool lds_ok[64] = {};
  if (hipError_t err = ensure_dynamic_lds(kern, smem, lds_ok); err != hipSuccess) return err;
  const long long grid = (((long long)a.M + S16_ROWS * NRB - 1) / (S16_ROWS * NRB)) * (a.N / (16 * NCB));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KKS * 64), smem, s, e, a, n_in);
  return hipGetLastError();
}

// ---- the priced alternatives of rounds 2 - 3 (CHANGELOG.md; measurement log in git history) are compiled only into the probes library
// (-DIKF_PROBES, ikflow_amd/lib/libikflow_amd_probes.so): the one-launch subnet chain for <= 128 rows (k_flow_chain16), the next subnet's
// entry phase in the tail of a contraction (TailSync), tile configurations 5 / 7 / 11.  The shipped library answers "not supported".
#ifndef IKF_PROBES
bool flow_chain16_ok(long long, int, int, int, int) { return false; }
hipError_t launch_xcd_census(unsigned*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_flow_chain16(const ChainSubnet*, int, const ChainCall&, const ChainSync&, int, hipStream_t) { return hipErrorNotSupported; }
#else
constexpr size_t kChainLds = 84 * 1024;  // more than half a CU's LDS: one chain workgroup per CU
bool flow_chain16_ok(long long rows, int width, int D, int n_out, int n_hidden) {
  return rows >= 1 && rows <= (long long)S16_ROWS * IKF_CHAIN_XCDS && n_hidden == 3 && width == 32 * IKF_CHAIN_PER_XCD &&
         entry_gemm_ok(fused_skinny16_cfg(), rows, width, D, n_out);
}
__global__ __launch_bounds__(KKS * 64) void k_xcd_census(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
hipError_t launch_xcd_census(unsigned* d_out, hipStream_t s) {
  static bool lds_ok[64] = {};
  if (hipError_t err = ensure_dynamic_lds(k_xcd_census, kChainLds, lds_ok); err != hipSuccess) return err;
  hipLaunchKernelGGL(k_xcd_census, dim3(IKF_CHAIN_XCDS * IKF_CHAIN_PER_XCD), dim3(KKS * 64), kChainLds, s, d_out);
  return hipGetLastError();
}
hipError_t launch_flow_chain16(const ChainSubnet* d_tab, int n_sub, const ChainCall& call, const ChainSync& cs, int K, hipStream_t s) {
  if (call.M <= 0) return hipSuccess;
  static_assert(kChainLds >= entry_gemm16_lds() && kChainLds >= skinny16_tail_lds(), "the chain's bodies fit");
  static bool lds_ok[64] = {};
  const dim3 grid(IKF_CHAIN_XCDS * IKF_CHAIN_PER_XCD), block(KKS * 64);
  if (K <= kDeepTiles * KBK) {
    if (hipError_t err = ensure_dynamic_lds(k_flow_chain16<true>, kChainLds, lds_ok); err != hipSuccess) return err;
    hipLaunchKernelGGL(k_flow_chain16<true>, grid, block, kChainLds, s, d_tab, n_sub, call, cs);
  } else {
    if (hipError_t err = ensure_dynamic_lds(k_flow_chain16<false>, kChainLds, lds_ok); err != hipSuccess) return err;
    hipLaunchKernelGGL(k_flow_chain16<false>, grid, block, kChainLds, s, d_tab, n_sub, call, cs);
  }
  return hipGetLastError();
}
#endif  // IKF_PROBES

// true when the first hidden contraction of a subnet can run as k_entry_gemm_skinny for this batch
bool entry_gemm_ok(int cfg, long long rows, int width, int D, int n_out) {
  if (cfg == 9 || cfg == 10 || cfg == 11) {  // kSkinny16Cfg: 16 x 32 tiles; kSkinny16x16Cfg: 16 x 16; kSkinny32x32v2Cfg: 32 x 32 on 16x16x4
    const int br = cfg == 11 ? 32 : 16;
    const long long tiles = ((rows + br - 1) / br) * (width / (cfg == 10 ? 16 : 32));
    return tiles <= 512 && width / (cfg == 10 ? 16 : 32) <= 64 &&  // <= 64 partial-sum slots: eight per wave in pending16_issue
           width % (2 * KBK) == 0 && (width / 16) % KKS == 0 && (width / 16) / KKS <= 8 && D <= ROWBUF && n_out <= ROWBUF;  // (its 69 KB of LDS at width 1024 lets two workgroups share a CU)
  }
  if (cfg != 4 && cfg != 6) return false;  // kSkinnyCfg / kSkinny32Cfg
  const int NH = cfg == 4 ? 2 : 1;
  const int NW = NH * KKS;
  const long long tiles = ((rows + EG_ROWS - 1) / EG_ROWS) * (width / (NH * 32));
  return tiles <= 256 && width <= 1024 && width % (2 * KBK) == 0 && (width / 32) % NW == 0 && D <= ROWBUF && n_out <= ROWBUF;
}

hipError_t launch_entry_gemm(int n_in, bool epi_red, int cfg, const EntryArgs& e, const FusedGemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (!entry_gemm_ok(cfg, a.M, a.N, e.D, e.pend.n_out) || a.K != a.N || a.Wf == nullptr || a.n_out > 16 || n_in > ROWBUF - 1 ||
      e.width != a.K)
    return hipErrorInvalidValue;
  if (cfg == 9) {
    if ((a.tune & IKF_TUNE_DEEP16) != 0 && a.K <= kDeepTiles * KBK)
      return epi_red ? launch_entry_gemm16<true, true, 2>(e, a, n_in, s) : launch_entry_gemm16<false, true, 2>(e, a, n_in, s);
    return epi_red ? launch_entry_gemm16<true, false, 2>(e, a, n_in, s) : launch_entry_gemm16<false, false, 2>(e, a, n_in, s);
  }
  if (cfg == 10) {
    if ((a.tune & IKF_TUNE_DEEP16) != 0 && a.K <= kDeepTiles * KBK)
      return epi_red ? launch_entry_gemm16<true, true, 1>(e, a, n_in, s) : launch_entry_gemm16<false, true, 1>(e, a, n_in, s);
    return epi_red ? launch_entry_gemm16<true, false, 1>(e, a, n_in, s) : launch_entry_gemm16<false, false, 1>(e, a, n_in, s);
  }
  if (cfg == 11) {
#ifdef IKF_PROBES
    if ((a.tune & IKF_TUNE_DEEP16) != 0 && a.K <= kDeepTiles * KBK)
      return epi_red ? launch_entry_gemm16<true, true, 2, 2>(e, a, n_in, s) : launch_entry_gemm16<false, true, 2, 2>(e, a, n_in, s);
    return epi_red ? launch_entry_gemm16<true, false, 2, 2>(e, a, n_in, s) : launch_entry_gemm16<false, false, 2, 2>(e, a, n_in, s);
#else
    return hipErrorNotSupported;
#endif
  }
  if (cfg == 4) return epi_red ? launch_entry_gemm_t<true, 2>(e, a, n_in, s) : launch_entry_gemm_t<false, 2>(e, a, n_in, s);
  return epi_red ? launch_entry_gemm_t<true, 1>(e, a, n_in, s) : launch_entry_gemm_t<false, 1>(e, a, n_in, s);
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int kSkinnyCfg = 4;    // k_flow_gemm_skinny<.., 2>: 32x64 tiles
constexpr int kSkinny32Cfg = 6;  // k_flow_gemm_skinny<.., 1>: 32x32 tiles (5 is the 4-wave probe of the large tile)
constexpr int kSkinny16Cfg = 9;     // k_flow_gemm_skinny16<.., 2>: 16x32 tiles on v_mfma_f32_16x16x4_f32 (<= 128 rows)
constexpr int kSkinny16x16Cfg = 10;  // k_flow_gemm_skinny16<.., 1>: 16x16 tiles (<= 64 rows)
constexpr int kSkinny32v2Cfg = 11;   // k_flow_gemm_skinny16<.., 2, 2>: 32x32 tiles built from 16x16x4 MFMAs (129 .. 256 rows)
                                     // opt-in, IKF_TUNE_ROWS32_V2 (r03: 0.519 - 0.543 against 0.499 - 0.519 ms per call on the 32x32x2 kernels)
int fused_skinny_cfg() { return kSkinnyCfg; }
int fused_skinny32_cfg() { return kSkinny32Cfg; }
int fused_skinny16_cfg() { return kSkinny16Cfg; }
int fused_skinny16x16_cfg() { return kSkinny16x16Cfg; }
int fused_skinny32v2_cfg() { return kSkinny32v2Cfg; }
int fused_pick_cfg(long long rows, int width, int tune) {
  if (width % KBN == 0 && width % (2 * KBK) == 0) {
    const bool rows16 = (tune & IKF_TUNE_ROWS16) != 0;
    if (rows <= 64 && rows16 && (tune & IKF_TUNE_TILES16) != 0 && ((rows + S16_ROWS - 1) / S16_ROWS) * (width / 16) <= 256) return kSkinny16x16Cfg;
    if (rows <= 128 && rows16 && ((rows + S16_ROWS - 1) / S16_ROWS) * (width / 32) <= 256) return kSkinny16Cfg;
    if (rows <= 256 && rows16 && (tune & IKF_TUNE_ROWS32_V2) != 0 && ((rows + 31) / 32) * (width / 32) <= 256 && width <= kDeepTiles * KBK) return kSkinny32v2Cfg;
    if (rows <= 256) return kSkinny32Cfg;
    if (rows <= 512) return kSkinnyCfg;
    if (rows <= 768) return kSkinny32Cfg;  // three co-resident 32x32 workgroups per CU: 1.00 ms against 1.06 (64x64 tiles)
  }
  // Cost model fitted to the in-chain sweep (tools/cfg_sweep.py -> profiles/r01_cfg_sweep.jsonl; us per contraction at
  // K = 1024, only the ratios matter): the 128x128 and 64x128 tiles run one workgroup per CU, so a launch costs whole
  // rounds of 256 tiles (64.2 / 34 us); two 64x64 workgroups share a CU: 16.5 us per 256 tiles, 19.8 when alone.
  // E.g. 3072 rows: 128x128 = 1 round = 64 us, 64x128 = 2 rounds = 68 us, 64x64 = 3 x 16.5 = 50 us.
  int best = -1;
  double best_t = 0.0;
  for (int c = 0; c < 3; ++c) {
    if (width % kCfgBN[c] != 0) continue;
    const long long tiles = ((rows + kCfgBM[c] - 1) / kCfgBM[c]) * (width / kCfgBN[c]);
    const double rounds = (double)((tiles + 255) / 256);
    const double t = c == 0 ? 64.2 * rounds : c == 1 ? 34.0 * rounds : (tiles <= 256 ? 19.8 : 16.5 * rounds);
    if (best < 0 || t < best_t - 1e-9) { best = c; best_t = t; }
  }
  if (best >= 0) return best;
  for (int c = kNumTileCfg - 1; c >= 0; --c)
    if (width % kCfgBN[c] == 0) return c;
  return -1;
}
// partial-sum slots of the last Linear: one per 64 columns, except the 32-column small-batch tiles (half slots)
int fused_slots(int cfg, int width) {
  return cfg == kSkinny16x16Cfg ? width / 16 : (cfg == kSkinny32Cfg || cfg == kSkinny16Cfg || cfg == kSkinny32v2Cfg) ? width / 32 : width / 64;
}
int fused_max_slots(int width) { return width / 16; }
const char* fused_kernel_name() { return "k_flow_gemm"; }

template <bool EPI_RED, int CFG>
static hipError_t launch_fg(const FusedGemmArgs& a, hipStream_t s) {
  using TC = TileCfg<CFG>;
  constexpr int NT = TC::WAVES_M * TC::WAVES_N * 64;
  constexpr size_t smem = (size_t)3 * (TC::BM + TC::BN) * (TC::BK + 4) * sizeof(float);
  auto kern = k_flow_gemm<EPI_RED, CFG, false>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long tiles_m = ((long long)a.M + TC::BM - 1) / TC::BM;
  const long long grid = tiles_m * (a.N / TC::BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), smem, s, a, NoTail{});
  return hipGetLastError();
}
#ifdef IKF_PROBES
static hipError_t launch_fg_tail(const FusedGemmArgs& a, const FuseTail& ft, hipStream_t s) {
  using TC = TileCfg<0>;
  constexpr int NT = TC::WAVES_M * TC::WAVES_N * 64;
  constexpr size_t smem = (size_t)3 * (TC::BM + TC::BN) * (FBK + 4) * sizeof(float);
  auto kern = k_flow_gemm<true, 0, true>;
  static bool lds_ok[64] = {};
  if (hipError_t e = ensure_dynamic_lds(kern, smem, lds_ok); e != hipSuccess) return e;
  const long long grid = (((long long)a.M + TC::BM - 1) / TC::BM) * (a.N / TC::BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), smem, s, a, ft);
  return hipGetLastError();
}
#endif  // IKF_PROBES

// The in-launch hand-over needs every workgroup of the launch resident at once (a workgroup waits for its row tile's other
// column tiles): both kernels that carry it run one workgroup per CU (LDS), so the grid may have at most 256 tiles.
[[maybe_unused]] constexpr int kResidentTiles = 256;
int fused_tail_col_tiles(int cfg, int width) { return cfg == 0 ? width / TileCfg<0>::BN : width / KBN; }
bool fused_tail_ok(int cfg, long long rows, int width, int D, int n_out) {
#ifndef IKF_PROBES
  return false;   // (the in-launch entry phase exists in the probes library only)
#else
  if (cfg != 0 && cfg != kSkinnyCfg) return false;
  const int bm = cfg == 0 ? TileCfg<0>::BM : KBM, bn = cfg == 0 ? TileCfg<0>::BN : KBN;
  if (width % bn != 0 || width / 64 > 32) return false;  // (the sc1 slot loads cover the first 32 slots)
  if (cfg == kSkinnyCfg && width % (2 * KBK) != 0) return false;
  const long long tiles = ((rows + bm - 1) / bm) * (width / bn);
  return tiles <= kResidentTiles && D <= ROWBUF && n_out <= ROWBUF;
#endif
}
hipError_t launch_flow_gemm_tail(int cfg, const FusedGemmArgs& a, const EntryArgs& e, const TailSync& ts, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (!fused_tail_ok(cfg, a.M, a.N, e.D, e.pend.n_out) || a.K != a.N || a.n_out > 16 || ts.n_in > ROWBUF - 1 || e.width != a.N ||
      e.split_out || ts.arrive == nullptr || ts.give_up == nullptr || (cfg == kSkinnyCfg && a.Wf == nullptr) || a.K % 64 != 0 || a.K < 128)
    return hipErrorInvalidValue;
#ifdef IKF_PROBES
  FuseTail ft{e, ts};
  return cfg == 0 ? launch_fg_tail(a, ft, s) : launch_skinny_tail(a, ft, s);
#else
  return hipErrorNotSupported;
#endif
}

hipError_t launch_flow_gemm(bool epi_red, int cfg, const FusedGemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
#ifdef IKF_PROBES
  if (cfg == 5) return epi_red ? launch_fg<true, 5>(a, s) : launch_fg<false, 5>(a, s);
  if (cfg == 7) return epi_red ? launch_fg<true, 7>(a, s) : launch_fg<false, 7>(a, s);
#else
  if (cfg == 5 || cfg == 7 || cfg == kSkinny32v2Cfg) return hipErrorNotSupported;
#endif
  if (cfg == kSkinny16Cfg || cfg == kSkinny16x16Cfg || cfg == kSkinny32v2Cfg) {
    if (a.N % KBN != 0 || a.K % (2 * KBK) != 0 || a.n_out > 16 || a.Wf == nullptr) return hipErrorInvalidValue;
    const bool deep = (a.tune & IKF_TUNE_DEEP16) != 0 && a.K <= kDeepTiles * KBK;
#ifdef IKF_PROBES
    if (cfg == kSkinny32v2Cfg) {
      if (deep) return epi_red ? launch_skinny16<true, true, 2, 2>(a, s) : launch_skinny16<false, true, 2, 2>(a, s);
      return epi_red ? launch_skinny16<true, false, 2, 2>(a, s) : launch_skinny16<false, false, 2, 2>(a, s);
    }
#endif
    if (cfg == kSkinny16Cfg) {
      if (deep) return epi_red ? launch_skinny16<true, true, 2>(a, s) : launch_skinny16<false, true, 2>(a, s);
      return epi_red ? launch_skinny16<true, false, 2>(a, s) : launch_skinny16<false, false, 2>(a, s);
    }
    if (deep) return epi_red ? launch_skinny16<true, true, 1>(a, s) : launch_skinny16<false, true, 1>(a, s);
    return epi_red ? launch_skinny16<true, false, 1>(a, s) : launch_skinny16<false, false, 1>(a, s);
  }
  if (cfg == kSkinnyCfg || cfg == kSkinny32Cfg) {
    if (a.N % KBN != 0 || a.K % (2 * KBK) != 0 || a.n_out > 16 || a.Wf == nullptr) return hipErrorInvalidValue;
    if (cfg == kSkinnyCfg) return epi_red ? launch_skinny<true, 2>(a, s) : launch_skinny<false, 2>(a, s);
    return epi_red ? launch_skinny<true, 1>(a, s) : launch_skinny<false, 1>(a, s);
  }
  if (cfg < 0 || cfg >= kNumTileCfg || a.N % kCfgBN[cfg] != 0 || a.K % 64 != 0 || a.K < 128 || a.n_out > 16) return hipErrorInvalidValue;
  switch (cfg) {
    case 0: return epi_red ? launch_fg<true, 0>(a, s) : launch_fg<false, 0>(a, s);
    case 1: return epi_red ? launch_fg<true, 1>(a, s) : launch_fg<false, 1>(a, s);
    case 2: return epi_red ? launch_fg<true, 2>(a, s) : launch_fg<false, 2>(a, s);
    default: return epi_red ? launch_fg<true, 3>(a, s) : launch_fg<false, 3>(a, s);
  }
}

// entry-kernel geometry: {threads, rows per workgroup}
int g_entry_geom_override = -1;  // probes: force a geometry (index into the table below)
template <int IN>
static hipError_t launch_entry_in(const EntryArgs& e, int geom, unsigned cs, hipStream_t s) {
#define IKF_ENTRY_GEOM(G, NT_, ER_) \
  case G: hipLaunchKernelGGL((k_subnet_entry<IN, NT_, ER_>), dim3((unsigned)((e.M + ER_ - 1) / ER_), cs), dim3(NT_), 0, s, e); break;
  switch (geom) {
    IKF_ENTRY_GEOM(0, 256, 16)
    IKF_ENTRY_GEOM(1, 512, 16)
    IKF_ENTRY_GEOM(2, 256, 8)
    IKF_ENTRY_GEOM(3, 512, 32)
    IKF_ENTRY_GEOM(4, 1024, 32)
    IKF_ENTRY_GEOM(5, 512, 8)
    default: return hipErrorInvalidValue;
  }
#undef IKF_ENTRY_GEOM
  return hipGetLastError();
}

hipError_t launch_subnet_entry(int n_in, const EntryArgs& e, hipStream_t s) {
  if (e.M <= 0) return hipSuccess;
  if (e.D > ROWBUF || e.pend.n_out > ROWBUF || e.width % 4 != 0) return hipErrorInvalidValue;
  // up to 1024 rows (<= 64 row groups) four workgroups share a row group's columns: the launch is a latency chain
  // there and the first Linear of 16 rows x `width` is its longest link
  const int n4 = e.width / 4;
  unsigned cs = 1;
  if (e.M <= 1024 && n4 % 4 == 0 && n4 / 4 >= 64) cs = 4;
  else if (e.M <= 2048 && n4 % 2 == 0 && n4 / 2 >= 64) cs = 2;
  // 512 threads: two waves per SIMD overlap one's LDS -> FMA -> store chains with the other's; 8-row workgroups at small
  // batches double the workgroups in flight (tools/gemm_probe.hip 300: 8.6 -> 7.7 us at 4096 rows, 5.2 -> 4.9 at 512)
  const int geom = g_entry_geom_override >= 0 ? g_entry_geom_override : (e.M <= 768 ? 5 : 1);
  switch (n_in) {
    case 8: return launch_entry_in<8>(e, geom, cs, s);
    case 9: return launch_entry_in<9>(e, geom, cs, s);
    case 10: return launch_entry_in<10>(e, geom, cs, s);
    case 11: return launch_entry_in<11>(e, geom, cs, s);
    case 12: return launch_entry_in<12>(e, geom, cs, s);
    case 13: return launch_entry_in<13>(e, geom, cs, s);
    case 14: return launch_entry_in<14>(e, geom, cs, s);
    case 15: return launch_entry_in<15>(e, geom, cs, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_flow_finalize(const FinalizeArgs& f, hipStream_t s) {
  if (f.M <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_flow_finalize, dim3((unsigned)((f.M + 31) / 32)), dim3(256), 0, s, f);
  return hipGetLastError();
}

}  // namespace ikf
