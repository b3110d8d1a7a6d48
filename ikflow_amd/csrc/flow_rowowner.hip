// Row-owner flow kernel (gfx950): ONE launch per call; workgroup w keeps rows [16 w, 16 w + 16) on chip for the whole inverse pass
// (ikflow/ikflow_solver.py:98 -> GraphINN rev over ikflow/model.py:336-352: every GLOWCouplingBlock's two subnets, PermuteRandom^-1,
// then FixedLinearTransform^-1, [:, :ndof], clamp - ikflow_solver.py:99-102).
//
//   * state [16][D], the conditional and the 16 x 1024 hidden activation tile live in LDS (two tiles, ping-pong: one barrier per layer);
//   * the hidden contractions run transposed on v_mfma_f32_16x16x4_f32 (A operand = 16 output columns of W, B operand = the 16 rows):
//     wave v owns columns [128 v, 128 v + 128) = 8 accumulator blocks of 4 registers, whose lane layout (row = lane % 16, four
//     consecutive columns 4 (lane / 16) + r) goes back to the tile as one ds_write_b128 per block;
//   * every parameter of a subnet reaches the wave through ONE linear stream of 8 KB groups (rowowner_pack below) consumed in order
//     through a ring of register slots, requested PF groups ahead of use and never drained - not at barriers, not between subnets:
//       group 0          first Linear  [16 k slots x the wave's 128 columns]; slot k = 4 q + c feeds MFMA c: c = 0, 1 hold what does not
//                        depend on the flow state (7 pose entries, bias as an input that is always 1), c = 2, 3 the x inputs and the
//                        softflow column (ro_input_slot) - the cluster form evaluates the first half while it waits for its peers
//       group 1 / 66     bias of hidden Linear 2 / 3 in accumulator layout (the accumulators START from it)
//       group 2..65      hidden Linear 2, 16 k per group            group 67..130  hidden Linear 3
//       group 131        last Linear, the wave's 128-k slice x 16 outputs (zero rows beyond n_out)
//     132 groups per subnet = a multiple of the ring length, so every group sits in a slot known at compile time;
//   * rows are independent: no inter-workgroup synchronisation, no activation ever goes to HBM, no entry / finalize launches.
// The price is the weight stream: every CU reads every weight (32 B/clk/CU from its XCD's L2 while the matrix pipe runs flat out).
// Numerics: exact f32 fma chains on the matrix pipe like the per-layer kernels; another summation order (k ascending inside 16-k
// groups permuted as k = 16 g + 4 j + c -> (c, j); even and odd groups in two chains that are added at the end), same tolerance
// against the oracle.
#include "ikf_internal.h"

#include <vector>

namespace ikf {

typedef float ro_f4 __attribute__((ext_vector_type(4)));
typedef unsigned ro_u4 __attribute__((ext_vector_type(4)));

constexpr int RO_W = 1024;                       // hidden width this kernel is built for
constexpr int RO_ROWS = 16;
constexpr int RO_LDA = RO_W + 4;                 // tile row stride in floats
constexpr int RO_WAVES = 8;
constexpr int RO_NCB = 8;                        // 16-column accumulator blocks per wave
constexpr int RO_KG = RO_W / 16;                 // 16-k groups per hidden layer (64)
constexpr int RO_SUB_GROUPS = 2 + 2 * (1 + RO_KG);   // 132
constexpr unsigned RO_WAVE_GROUP_BYTES = RO_NCB * 64 * 16;          // 8 KB: one ring slot of one wave
constexpr unsigned RO_GROUP_BYTES = RO_WAVES * RO_WAVE_GROUP_BYTES;  // 64 KB
constexpr int RO_MAX_SUB = 32;
constexpr int RO_US = 20, RO_RS = 20;            // row strides of the small LDS arrays
constexpr int RO_SMALL_WORDS = 36;
// LDS map (floats)
constexpr int RO_OFF_TILE0 = 0;
constexpr int RO_OFF_TILE1 = RO_ROWS * RO_LDA;
constexpr int RO_OFF_XS = 2 * RO_ROWS * RO_LDA;            // [2][16][16] state, ping-pong
constexpr int RO_OFF_US = RO_OFF_XS + 2 * 256;             // [16][20] first-Linear input rows
constexpr int RO_OFF_RED = RO_OFF_US + RO_ROWS * RO_US;    // [8][16][20] last-Linear partial sums per wave
constexpr int RO_OFF_COND = RO_OFF_RED + RO_WAVES * RO_ROWS * RO_RS;  // [16][8] pose + softflow
constexpr int RO_OFF_SMALL = RO_OFF_COND + RO_ROWS * 8;    // [n_sub][36] b_last, perm_inv, which / n_x / x_off / n_half
constexpr int RO_LDS_FLOATS = RO_OFF_SMALL + RO_MAX_SUB * RO_SMALL_WORDS;
constexpr size_t RO_LDS_BYTES = sizeof(float) * RO_LDS_FLOATS;

static_assert(sizeof(RoSubnet) == RO_SMALL_WORDS * 4, "RoSubnet layout");

// First-Linear input slot k = 4 q + c (component c of lane-quarter q feeds MFMA c): what sits there.
//   >= 0: pose entry (0..6);  -1: the constant 1 (bias);  -2 - e: x input e (0 .. n_x - 1);  -100: the softflow entry;  -200: nothing (zero)
__host__ __device__ __forceinline__ int ro_input_slot(int k, int n_x) {
  const int c = k & 3, q = k >> 2;
  if (c < 2) {
    const int s = c * 4 + q;
    return s < 7 ? s : -1;
  }
  const int e = (c - 2) * 4 + q;
  return e < n_x ? -2 - e : (e == n_x ? -100 : -200);
}

__device__ __forceinline__ void ro_barrier() {
  // LDS traffic of this wave done, then the workgroup barrier; the weight-stream loads stay in flight
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// value of new-state element d of row `row` after the pending coupling of subnet `sm` (or the plain state when sm == null)
__device__ __forceinline__ float ro_new_state(const float* __restrict__ xs_old, const float* __restrict__ red, const float* __restrict__ sm,
                                              int row, int d, int L1, float clamp) {
  if (sm == nullptr) return xs_old[row * 16 + d];
  const int* smi = reinterpret_cast<const int*>(sm);
  const int which = smi[32], nl = smi[35];
  const int src = which == 2 ? smi[16 + d] : d;
  const int off = which == 1 ? L1 : 0;
  float v = xs_old[row * 16 + src];
  if (src >= off && src < off + nl) {
    const int j = src - off;
    float s = sm[j], tt = sm[nl + j];
#pragma unroll
    for (int w = 0; w < RO_WAVES; ++w) {
      s += red[w * (RO_ROWS * RO_RS) + row * RO_RS + j];
      tt += red[w * (RO_ROWS * RO_RS) + row * RO_RS + nl + j];
    }
    const float s_cl = clamp * (0.636f * atanf(s));
    v = (v - tt) * expf(-s_cl);
  }
  return v;
}

#define RO_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NBUF>
__global__ __launch_bounds__(RO_WAVES * 64) void k_flow_rowowner(RoArgs a) {
  static_assert(RO_SUB_GROUPS % NBUF == 0 && RO_KG % NBUF == 0, "ring length must divide the subnet's group count");
  constexpr int PF = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const tile0 = smem + RO_OFF_TILE0;
  float* const tile1 = smem + RO_OFF_TILE1;
  float* const xs = smem + RO_OFF_XS;
  float* const us = smem + RO_OFF_US;
  float* const red = smem + RO_OFF_RED;
  float* const cond = smem + RO_OFF_COND;
  float* const small = smem + RO_OFF_SMALL;
  if (a.run_if != nullptr && __hip_atomic_load(a.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // (uniform)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * RO_ROWS;
  const int lrow = lane & 15, lq = lane >> 4;
#define RO_STAMP(i) if (a.trace != nullptr && t == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
  RO_STAMP(0)
  const unsigned long long ro_t0_wall = a.trace != nullptr ? wall_clock64() : 0ull;

  // ---- the weight stream: ring of NBUF slots x 8 float4 per lane
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.stream), 0, a.stream_bytes, 0x00020000);
  const unsigned voff = lane * 16;
  unsigned g_issue = wave * RO_WAVE_GROUP_BYTES;   // byte offset of the next group to request (wave-uniform)
  ro_f4 wb[NBUF][RO_NCB];
#define RO_ISSUE(slot)                                                                                                             \
  {                                                                                                                                \
    const unsigned so_ = __builtin_amdgcn_readfirstlane(g_issue);                                                                  \
    _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_)                                                                       \
        wb[slot][cb_] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + cb_ * 1024, so_, 0));           \
    g_issue += RO_GROUP_BYTES;                                                                                                     \
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) RO_ISSUE(s)

  // ---- rows: state, conditional, per-subnet small parameters -> LDS
  if (t < 256) {
    const int row = t >> 4, d = t & 15;
    int gr = m0 + row;
    gr = gr < a.M ? gr : a.M - 1;
    xs[row * 16 + d] = d < a.D ? a.x0[(size_t)gr * a.D + d] : 0.f;
  } else if (t < 256 + 128) {
    const int row = (t - 256) >> 3, k = t & 7;
    int gr = m0 + row;
    gr = gr < a.M ? gr : a.M - 1;
    const long long grow = a.row0 + gr;
    const long long pm = grow < a.ps.n_mod ? grow : (a.ps.n_mod == 1 ? 0 : grow % a.ps.n_mod);
    const long long pi = a.ps.idx ? (long long)a.ps.idx[pm] : pm;
    cond[row * 8 + k] = k < 7 ? a.ps.poses[pi * a.ps.stride + k] : a.ps.softflow;
  }
  for (int i = t; i < a.n_sub * RO_SMALL_WORDS; i += RO_WAVES * 64) small[i] = reinterpret_cast<const float*>(a.sub)[i];
  ro_barrier();

  // input rows of subnet `nxt` from the state after the pending coupling of subnet `pend` (null: none) - threads 256..511;
  // the new state itself - threads 0..255
  auto advance = [&](const float* pend_sm, const float* nxt_sm, const float* xs_old, float* xs_new) {
    if (t < 256) {
      const int row = t >> 4, d = t & 15;
      if (d < a.D) xs_new[row * 16 + d] = ro_new_state(xs_old, red, pend_sm, row, d, a.L1, a.clamp);
    } else if (nxt_sm != nullptr) {
      const int row = (t - 256) >> 4, k = t & 15;
      const int* ni = reinterpret_cast<const int*>(nxt_sm);
      const int n_x = ni[33], x_off = ni[34];
      const int what = ro_input_slot(k, n_x);
      float v = 0.f;
      if (what >= 0) v = cond[row * 8 + what];
      else if (what == -1) v = 1.0f;
      else if (what == -100) v = cond[row * 8 + 7];
      else if (what > -100) v = ro_new_state(xs_old, red, pend_sm, row, x_off + (-2 - what), a.L1, a.clamp);
      us[row * RO_US + k] = v;
    }
  };
  advance(nullptr, small, xs, xs + 256);
  ro_barrier();
  int xcur = 1;   // xs + 256 * xcur holds the current state

  ro_f4 acc[RO_NCB];
  // Odd k groups accumulate into a second set: two half-length f32 chains per output, added at the end, instead of one chain of 1024
  // products (+ bias).  Rounding error grows with the chain length: measured over every row of the baseline batches the distance to the
  // oracle drops from 4.05e-6 to 2.86e-6 rad (Panda, 4096 rows) and from 8.1e-6 to 3.4e-6 relative with O(1) coupling coefficients (last
  // Linear x 2.5) - at the same speed (28 more registers, the same MFMAs).
  ro_f4 accb[RO_NCB];
  ro_f4 af[2];
#define RO_EPILOGUE(tile_out)                                                                                            \
  _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) {                                                             \
    ro_f4 v_ = acc[cb_];                                                                                                 \
    v_ = __builtin_elementwise_max(v_, v_ * a.slope); /* LeakyReLU for 0 <= slope <= 1 (the launcher checks) */         \
    *reinterpret_cast<ro_f4*>((tile_out) + lrow * RO_LDA + wave * 128 + cb_ * 16 + 4 * lq) = v_;                         \
  }
#define RO_AFRAG(tile_in, kg) *reinterpret_cast<const ro_f4*>((tile_in) + lrow * RO_LDA + (kg) * 16 + 4 * lq)
  // one 16-k group: request the group PF ahead into the slot consumed last, read the next A fragment, 32 MFMAs
#define RO_ACC(par) ((par) ? accb : acc)
#define RO_KGSTEP(slot, tile_in, kg, par)                                                                                \
  {                                                                                                                      \
    RO_ISSUE(((slot) + PF) % NBUF)                                                                                       \
    af[(par) ^ 1] = RO_AFRAG(tile_in, (kg) + 1);                                                                         \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                                     \
        _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) RO_ACC(par)[cb_] = RO_MFMA(wb[slot][cb_][c_], af[par][c_], RO_ACC(par)[cb_]);  \
    /* pinned order: the A fragment read, then one weight request per four MFMAs */                                      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < RO_NCB; ++i_) {                                                              \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                                 \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                 \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
#define RO_CHAIN_INIT _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) accb[cb_] = ro_f4{0.f, 0.f, 0.f, 0.f};
#define RO_CHAIN_JOIN _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) acc[cb_] += accb[cb_];
  // a hidden layer: accumulators start from the bias group (slot SB), then 64 groups starting in slot (SB + 1) % NBUF
#define RO_LAYER(SB, tile_in, tile_out)                                                                                  \
  {                                                                                                                      \
    RO_ISSUE(((SB) + PF) % NBUF)                                                                                         \
    af[0] = RO_AFRAG(tile_in, 0);                                                                                        \
    _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) acc[cb_] = wb[SB][cb_];                                     \
    RO_CHAIN_INIT                                                                                                        \
    for (int kg = 0; kg < RO_KG; kg += NBUF) {                                                                           \
      _Pragma("unroll") for (int u_ = 0; u_ < NBUF; ++u_) RO_KGSTEP(((SB) + 1 + u_) % NBUF, tile_in, kg + u_, u_ & 1)    \
    }                                                                                                                    \
    RO_CHAIN_JOIN                                                                                                        \
    RO_EPILOGUE(tile_out)                                                                                                \
    ro_barrier();                                                                                                        \
  }
  static_assert(NBUF % 2 == 0 || NBUF == 0, "the A-fragment parity is tied to the ring position: NBUF must be even");

  for (int s = 0; s < a.n_sub; ++s) {
    const float* sm = small + s * RO_SMALL_WORDS;
    RO_STAMP(1 + (s < 31 ? s : 31))
#define RO_PHASE(i) if (s == 2) { RO_STAMP(40 + (i)) }
    RO_PHASE(0)
    // ---- group 0 (slot 0): first Linear + LeakyReLU -> tile0
    {
      RO_ISSUE(PF % NBUF)
      const ro_f4 uf = *reinterpret_cast<const ro_f4*>(us + lrow * RO_US + 4 * lq);
#pragma unroll
      for (int cb = 0; cb < RO_NCB; ++cb) acc[cb] = ro_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int cb = 0; cb < RO_NCB; ++cb) acc[cb] = RO_MFMA(wb[0][cb][c], uf[c], acc[cb]);
      RO_EPILOGUE(tile0)
      ro_barrier();
    }
    RO_PHASE(1)
    // ---- groups 1..65: hidden Linear 2 (tile0 -> tile1); groups 66..130: hidden Linear 3 (tile1 -> tile0)
    RO_LAYER(1 % NBUF, tile0, tile1)
    RO_PHASE(2)
    RO_LAYER(66 % NBUF, tile1, tile0)
    RO_PHASE(3)
    // ---- group 131: last Linear, this wave's 128-k slice; partial sums [row][16 outputs] -> red[wave]
    {
      RO_ISSUE((131 + PF) % NBUF)
      constexpr int SL = 131 % NBUF;
      ro_f4 hf[RO_NCB];
#pragma unroll
      for (int j = 0; j < RO_NCB; ++j) hf[j] = *reinterpret_cast<const ro_f4*>(tile0 + lrow * RO_LDA + wave * 128 + j * 16 + 4 * lq);
      ro_f4 p4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) p4[c] = ro_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < RO_NCB; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) p4[c] = RO_MFMA(wb[SL][j][c], hf[j][c], p4[c]);
      const ro_f4 p = (p4[0] + p4[1]) + (p4[2] + p4[3]);
      *reinterpret_cast<ro_f4*>(red + wave * (RO_ROWS * RO_RS) + lrow * RO_RS + 4 * lq) = p;
      ro_barrier();
    }
    RO_PHASE(4)
    // ---- coupling of this subnet -> new state; input rows of the next subnet
    advance(sm, s + 1 < a.n_sub ? sm + RO_SMALL_WORDS : nullptr, xs + 256 * xcur, xs + 256 * (xcur ^ 1));
    xcur ^= 1;
    ro_barrier();
    RO_PHASE(5)
  }
  RO_STAMP(33)
  // ---- FixedLinearTransform rev: (x - b).mm(M_inv); [:, :ndof]; clamp_to_joint_limits
  if (t < 256) {
    const int row = t >> 4, j = t & 15;
    if (j < a.ndof && m0 + row < a.M) {
      const float* x = xs + 256 * xcur + row * 16;
      float q = 0.f;
      for (int k = 0; k < a.D; ++k) {
        float xv = x[k];
        if (a.sigmoid) xv = 1.0f / (1.0f + expf(-xv));
        q = fmaf(xv - a.b_lin[k], a.M_inv[k * a.D + j], q);
      }
      if (a.clamp_limits) q = fminf(fmaxf(q, a.lo[j]), a.hi[j]);
      a.q_out[(size_t)(m0 + row) * a.ndof + j] = q;
    }
  }
  RO_STAMP(34)
  if (a.trace != nullptr && t == 0) {  // probes: the constant 100 MHz clock beside the shader clock (effective GHz of this launch)
    a.trace[(size_t)blockIdx.x * 64 + 36] = wall_clock64();
    a.trace[(size_t)blockIdx.x * 64 + 35] = ro_t0_wall;
  }
#undef RO_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// Cluster form for batches that cannot fill the chip with one workgroup per 16 rows (<= 2048 rows): the 16 rows of a row tile are owned by
// G workgroups (G = 2, 4, 8, 16, 32 for <= 2048 / 1024 / 512 / 256 / 128 rows; workgroup b = member b % G of row tile b / G - with the observed placement b -> XCD b % 8 the members of a
// tile sit on different XCDs, so every XCD's L2 pulls only its members' column slices of W).  Member j computes columns [j 1024/G,
// (j+1) 1024/G) of every hidden layer from the FULL 16 x 1024 input tile in its LDS, streaming only its slice of W (same image, same
// register ring - never drained by a barrier or an exchange).  The first Linear (13 inputs) is evaluated in full by every member - cheaper
// than exchanging it.  Per subnet the members exchange two things through global memory:
//   the h2 (hidden 2) slices -> all-gather into every member's input tile of hidden 3;
//   the last Linear's [16 x 16] partial sums over the member's own h3 columns -> all-gather, summed in member order (h3 never leaves).
// Hand-over of h2 (cdna_hip_programming.md Guideline 16, form R1): payload as 16-byte write-through (sc1) stores, every storing wave
// drains, barrier, ONE lane stores the member's epoch word (agent scope); consumers poll the G - 1 epoch words from one wave (relaxed
// agent-scope loads + s_sleep, bounded), barrier, read the payload with sc1 loads (L1-bypassing; the producer stored sc1, so no acquire
// fence).  The partial sums (1 KB per member) go the same way with 4-byte sc1 stores and loads.  (Priced and dropped: the partial sums as
// 8-byte {epoch, value} granules that four waves re-read until every tag matches - 9.0 k cycles per subnet against 5.1 k with the epoch
// word at 512 rows: behind the CU's own weight stream a re-read pass costs a full round trip, and the first pass usually comes too early.)
// A hidden layer starts on its OWN slice of the k range (it is already in LDS) and waits for the peers only then: k runs in the rotated
// order j, j+1, .. (mod G) - a fixed order per output column, so results are reproducible bit for bit, and equal to the row-owner form's
// to rounding.  Placement-independent; needs all workgroups resident (grid <= CUs, one per CU: the launcher checks); a wait that runs out
// sets the abort word (every other wait ends) and the host-visible give-up word.
// G = 16 / 32: a member has fewer 16-column blocks (4 / 2) than waves, so KS = 2 / 4 waves split the k range of a block and their
// partial accumulators are summed through LDS in share order in the epilogue; these layers wait for the peers at their start (only the
// first share's groups are the member's own).  (G = 64 was measured and dropped: 0.336 ms at 64 rows against 0.286 with G = 32 on half
// the chip - its partial-sum exchange reads 63 peers per thread.)
// ---------------------------------------------------------------------------------------------------------------
// How long a member waits for a peer before it gives the launch up: about 5 ms, as a number of failed polls per kind of wait (a poll = s_sleep 1 +
// one round of L2-missing loads; measured with a workgroup left out, tools/cluster_wait_probe.py: re-read passes of a tagged h2 gather, polls of
// the tagged partial sums, polls of an epoch word).  A peer that IS resident answers within one subnet (< 50 us at any G: a handful of polls); one
// that is not (another process holds its CU) costs the call this wait plus the repair launch - not the ~0.1 s of r05's 2^16 / 2^18 polls.
// (Tried and dropped, same-box A/B with tools/lib_ab.py: the 100 MHz wall clock instead of counts - read in front of every wait + 4 % at 512 rows,
// every 16th failed poll + 1.5 %, every 256th still + 2 % at 256 rows: not the reads, the registers they hold in kernels that have none to spare.)
constexpr unsigned kGatherWaitPasses = 1u << 13;
constexpr unsigned kSumWaitPolls = 1u << 14;
constexpr unsigned kEpochWaitPolls = 1u << 14;

__device__ __forceinline__ unsigned ro_xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID

// LOCAL (G = 4 / 8 / 16): every member of a row tile on ONE XCD - workgroup b is member (b / 8) % G of tile b % 8 + 8 ((b / 8) / G); with
// the observed placement b -> XCD b % 8 the tile lives on XCD b % 8 - and the hand-over goes through that XCD's L2: plain stores (they stay
// in the L2 every member shares; the drain waits for the L2, not for memory), same epoch words, same sc1 loads (L1-bypassing).  2.5 % / 4 % /
// 4.5 % off a 1024- / 512- / 256-row call; every L2 then pulls every member's weight slice (8 x the fabric traffic) - which is why G = 32
// stays spread (0.34 against 0.28 ms at 128 rows) and G = 2 gains nothing.  NOTHING is assumed: a census at load (cluster_placement_census),
// and in every launch a member publishes its XCC_ID in the top byte of every epoch word; a consumer that meets another XCD's id gives up
// (abort word, host word = 2) BEFORE it reads a payload, the repair launch recomputes the rows, and the handle goes back to the spread form.
// (TAG form: no epoch words - every member writes (launch number, XCC_ID) into a word of its own at the start of every launch, one 128-byte line
// per row tile, and nothing in the launch reads them.  A peer on another XCD keeps its plain payload stores in ITS L2: the consumer sees the old
// parity, re-reads, and its wait runs out (code 1) - never a wrong value.  The HOST, when it folds a give-up of such a launch, reads the words
// (ikf_api.hip cluster_fold_give_up): members of one row tile that started that launch on different XCDs make it a placement failure - the
// handle goes back to the spread form - instead of "a peer was not resident" (pause).  The kernel has no register to spare for this: the same
// classification inside the failure path made the G = 4 / 8 kernels spill (+ 21 % at 512 rows), and checking the words up front, in front of
// the first payload read, cost 2.3 % (same-box A/B, tools/lib_ab.py) - for a case the census at load has never let happen.)
// TAG (r05): the hand-over without a drain, an epoch word or a poll of one - every exchanged float carries the subnet's parity in the LEAST
// significant bit of its mantissa (<= 1 ulp; every member works with the same tagged values, so the forms still agree member for member).
// The producer stores and goes on; a consumer reads the payload itself and re-reads what still shows the other parity: stale data of the
// previous subnet always does (a member can never be two subnets ahead of a reader, see the buffer-reuse argument above), and between
// calls every float of the exchange buffers has parity 1 (n_sub is even; the buffers are created as 0xff bytes and re-created after an
// aborted launch).  Needs nothing but the atomicity of a 4-byte store.  Measured (tools/rowowner_probe, modes 4 / 5 against 2 / 0): 512 rows
// G = 8 47.3 k -> 45.6 k cycles per subnet, 1024 rows G = 4 79.8 -> 77.5 k, 256 rows G = 16 32.4 -> 31.2 k, 2048 rows G = 2 149.9 -> 145.4 k.
// Requesting the peers' slices EARLY (under the member's own k groups) returns nothing: a load reads L2 when it is REQUESTED, not when its data
// comes back behind the weight ring, so a request made before the peers' stores have landed reads the old parity and the re-read costs what the
// late request costs (45.5 k against 45.6 k cycles per subnet at 512 rows); requested half way through the own k groups: - 0.5 k of 77 k at G = 4,
// - 0.5 k of 145 k at G = 2.
// NOT for G = 32: a consumer's first read comes before its 31 peers are done and every re-read moves 64 KB per CU - 0.30 - 0.33 ms at 128
// rows against 0.284 with epoch words, whatever delay precedes the first read.
__device__ __forceinline__ ro_f4 ro_tag(ro_f4 v, unsigned par) {
  ro_u4 b = __builtin_bit_cast(ro_u4, v);
  b = (b & ~1u) | par;
  return __builtin_bit_cast(ro_f4, b);
}
__device__ __forceinline__ float ro_tag1(float v, unsigned par) { return __uint_as_float((__float_as_uint(v) & ~1u) | par); }
__device__ __forceinline__ unsigned ro_tag_bad(ro_f4 v, unsigned par) {
  const ro_u4 b = __builtin_bit_cast(ro_u4, v);
  return ((b[0] ^ par) | (b[1] ^ par) | (b[2] ^ par) | (b[3] ^ par)) & 1u;
}
template <int G, bool LOCAL, bool TAG = false>
__global__ __launch_bounds__(RO_WAVES * 64) void k_flow_cluster(RcArgs c) {
  constexpr int NBM = RO_KG / G;                     // 16-column blocks of one member's slice (= 16-k groups of its k range)
  constexpr int KS = NBM >= RO_WAVES ? 1 : RO_WAVES / NBM;   // G >= 16: fewer blocks than waves - KS waves split the k range of a block
  constexpr int NCB = NBM >= RO_WAVES ? NBM / RO_WAVES : 1;  // 16-column blocks per wave
  constexpr int KGW = RO_KG / KS;                    // k groups a wave runs per layer
  constexpr int NBUF = KS > 1 ? (KGW < 16 ? KGW : 16) : 2 * G;   // ring slots of NCB float4: 16 float4 per lane (8 at G = 64)
  constexpr int PF = NBUF - 1;
  constexpr int N1 = NBM;                            // 16-k groups of one member's slice
  constexpr int CS = RO_W / G;                       // columns per member
  static_assert(KGW % NBUF == 0 && (KGW & (KGW - 1)) == 0, "ring length must divide a wave's share of a layer");
  const RoArgs& a = c.ro;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const tile0 = smem + RO_OFF_TILE0;
  float* const tile1 = smem + RO_OFF_TILE1;
  float* const xs = smem + RO_OFF_XS;
  float* const us = smem + RO_OFF_US;
  float* const red = smem + RO_OFF_RED;      // [8][16][20] per-wave partial sums; afterwards [16][20] the subnet's summed outputs
  float* const cond = smem + RO_OFF_COND;
  float* const small = smem + RO_OFF_SMALL;
  __shared__ unsigned s_ok;
  __shared__ float s_sum[RO_ROWS * RO_RS];   // the pending subnet's summed last-Linear outputs (bias included)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = LOCAL ? (int)(blockIdx.x / 8) % G : (int)blockIdx.x % G;
  const int rt = LOCAL ? (int)(blockIdx.x % 8) + 8 * ((int)(blockIdx.x / 8) / G) : (int)blockIdx.x / G;
  if (LOCAL && rt >= c.n_rt) return;   // (the grid is padded to whole groups of 8 row tiles; uniform per workgroup, nobody waits for these)
  // (TAG) an earlier launch on these buffers gave up: they may hold the wrong parity, and nothing re-creates them until the host has seen it -
  // this launch leaves everything to the repair launch queued behind it (the abort word is still set)
  if (TAG && __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  constexpr int ST_AUX = LOCAL ? 0 : 16;   // payload stores: write-back into the shared L2 / write-through (sc1)
  const unsigned my_xcc = LOCAL ? ro_xcc_id() : 0u;
  const unsigned pub_xcc = my_xcc ^ ((LOCAL && c.test_far != 0 && blockIdx.x == 0) ? 1u : 0u);   // (tests: workgroup 0 claims to sit elsewhere)
  if (t == 0) s_ok = 1;
  // (TAG + LOCAL) where this member sits, for its peers' placement check (agent-scope store: visible outside this XCD's L2)
  if constexpr (TAG && LOCAL) {
    if (t == 0) __hip_atomic_store(c.xcc_words + (size_t)rt * 32 + j, (c.launch_seq << 8) | pub_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (tests, variant 191) workgroup 0 also BEHAVES like a member on another XCD - its payload never reaches its peers: it stops here (its rows
  // are recomputed with its tile's by the repair launch)
  if (TAG && LOCAL && c.test_far != 0 && blockIdx.x == 0) return;
  const int m0 = rt * RO_ROWS;
  const int lrow = lane & 15, lq = lane >> 4;
#define RC_STAMP(i) if (a.trace != nullptr && t == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
  RC_STAMP(0)
  const unsigned long long t0_wall = a.trace != nullptr ? wall_clock64() : 0ull;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.stream), 0, a.stream_bytes, 0x00020000);
  const int bw = KS > 1 ? wave % NBM : wave * NCB;   // this wave's first block inside the member's slice
  const int ks = KS > 1 ? wave / NBM : 0;            // ... and its share of the k range: physical groups [ks KGW, (ks + 1) KGW)
  const unsigned cbg0 = (unsigned)j * NBM + (unsigned)bw;   // first 16-column (16-k) block of this wave, true index
  const unsigned voff = cbg0 * 1024 + lane * 16;
  const int n_hl = 2 * a.n_sub;             // hidden layers of the call
  // byte offset of the stream group that holds k group i (in this member's rotated order) of hidden layer hl
  auto grp = [&](int hl, int i) -> unsigned {
    hl = hl < n_hl ? hl : n_hl - 1;
    const unsigned g = (unsigned)(hl >> 1) * RO_SUB_GROUPS + ((hl & 1) ? 3 + RO_KG : 2) + (unsigned)((j * N1 + ks * KGW + i) & (RO_KG - 1));
    return g * RO_GROUP_BYTES;
  };
  ro_f4 wb[NBUF][NCB];
#define RC_ISSUE(slot, hl_, i_)                                                                                                    \
  {                                                                                                                                \
    const unsigned so_ = __builtin_amdgcn_readfirstlane(grp((hl_), (i_)));                                                         \
    _Pragma("unroll") for (int cb_ = 0; cb_ < NCB; ++cb_)                                                                          \
        wb[slot][cb_] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + cb_ * 1024, so_, 0));           \
  }
  // a subnet's small groups (first Linear, the two hidden biases, last Linear): this wave's NCB blocks of each
  auto small_load = [&](int sub, int g, ro_f4 (&dst)[NCB]) {
    sub = sub < a.n_sub ? sub : a.n_sub - 1;
    const unsigned so = __builtin_amdgcn_readfirstlane(((unsigned)sub * RO_SUB_GROUPS + (unsigned)g) * RO_GROUP_BYTES);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dst[cb] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + cb * 1024, so, 0));
  };
  // the first Linear in full: this wave's 128 TRUE columns [128 wave, 128 wave + 128) of group 0 (the row-owner form's operand)
  ro_f4 w1[RO_NCB];
  auto w1_load = [&](int sub) {
    sub = sub < a.n_sub ? sub : a.n_sub - 1;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)sub * RO_SUB_GROUPS * RO_GROUP_BYTES + (unsigned)wave * RO_WAVE_GROUP_BYTES);
#pragma unroll
    for (int cb = 0; cb < RO_NCB; ++cb) w1[cb] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16 + cb * 1024), so, 0));
  };
  ro_f4 b2[NCB], b3[NCB], wl[NCB];
  w1_load(0);
#pragma unroll
  for (int s = 0; s < PF; ++s) RC_ISSUE(s, 0, s)

  // exchange buffers of this row tile
  // (h2 of subnet s + 1 may overwrite h2 of subnet s: a member publishes it only after it has seen every peer's partial sums of subnet s,
  // which a peer publishes after it has read the h2 tile of subnet s)
  float* const X1 = c.xbuf + (size_t)rt * RO_ROWS * RO_W;                                // h2 [16][1024], true column order
  float* const P = c.pbuf + (size_t)rt * G * 256;                                        // [G][16][16]
  unsigned* const flags = c.flags + (size_t)rt * G * 32;
  const __amdgpu_buffer_rsrc_t rsX1 = __builtin_amdgcn_make_buffer_rsrc(X1, 0, RO_ROWS * RO_W * 4, 0x00020000);

  // ---- rows: state, conditional, per-subnet small parameters -> LDS (every member holds the whole state)
  if (t < 256) {
    const int row = t >> 4, d = t & 15;
    int gr = m0 + row;
    gr = gr < a.M ? gr : a.M - 1;
    xs[row * 16 + d] = d < a.D ? a.x0[(size_t)gr * a.D + d] : 0.f;
  } else if (t < 256 + 128) {
    const int row = (t - 256) >> 3, k = t & 7;
    int gr = m0 + row;
    gr = gr < a.M ? gr : a.M - 1;
    const long long grow = a.row0 + gr;
    const long long pm = grow < a.ps.n_mod ? grow : (a.ps.n_mod == 1 ? 0 : grow % a.ps.n_mod);
    const long long pi = a.ps.idx ? (long long)a.ps.idx[pm] : pm;
    cond[row * 8 + k] = k < 7 ? a.ps.poses[pi * a.ps.stride + k] : a.ps.softflow;
  }
  for (int i = t; i < a.n_sub * RO_SMALL_WORDS; i += RO_WAVES * 64) small[i] = reinterpret_cast<const float*>(a.sub)[i];
  ro_barrier();

  // new state (threads 0..255) and the next subnet's input rows (threads 256..511) from the pending subnet's summed outputs in s_sum
  auto new_state = [&](const float* sm, const float* xs_old, int row, int d) -> float {
    if (sm == nullptr) return xs_old[row * 16 + d];
    const int* smi = reinterpret_cast<const int*>(sm);
    const int which = smi[32], nl = smi[35];
    const int src = which == 2 ? smi[16 + d] : d;
    const int off = which == 1 ? a.L1 : 0;
    float v = xs_old[row * 16 + src];
    if (src >= off && src < off + nl) {
      const int jj = src - off;
      const float s_cl = a.clamp * (0.636f * atanf(s_sum[row * RO_RS + jj]));
      v = (v - s_sum[row * RO_RS + nl + jj]) * expf(-s_cl);
    }
    return v;
  };
  auto advance = [&](const float* pend_sm, const float* nxt_sm, const float* xs_old, float* xs_new) {
    if (t < 256) {
      const int row = t >> 4, d = t & 15;
      if (d < a.D) xs_new[row * 16 + d] = new_state(pend_sm, xs_old, row, d);
    } else if (nxt_sm != nullptr) {
      const int row = (t - 256) >> 4, k = t & 15;
      const int* ni = reinterpret_cast<const int*>(nxt_sm);
      const int n_x = ni[33], x_off = ni[34];
      const int what = ro_input_slot(k, n_x);
      float v = 0.f;
      if (what >= 0) v = cond[row * 8 + what];
      else if (what == -1) v = 1.0f;
      else if (what == -100) v = cond[row * 8 + 7];
      else if (what > -100) v = new_state(pend_sm, xs_old, row, x_off + (-2 - what));
      us[row * RO_US + k] = v;
    }
  };
  advance(nullptr, small, xs, xs + 256);
  ro_barrier();
  int xcur = 1;
  // the state-independent half of a first Linear (the current w1): pose entries and the bias input never change during a call
  ro_f4 a1[RO_NCB];
  const ro_f4 uf_static = *reinterpret_cast<const ro_f4*>(us + lrow * RO_US + 4 * lq);   // (.x, .y are used)
#define RC_FIRST_STATIC                                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) a1[cb_] = RO_MFMA(w1[cb_][0], uf_static[0], (ro_f4{0.f, 0.f, 0.f, 0.f})); \
    _Pragma("unroll") for (int cb_ = 0; cb_ < RO_NCB; ++cb_) a1[cb_] = RO_MFMA(w1[cb_][1], uf_static[1], a1[cb_]);                     \
  }
  RC_FIRST_STATIC

  // ---- hand-over primitives
  // this workgroup's payload stores are issued: drain them (every storing wave), then ONE lane publishes the epoch
#define RC_PUBLISH(e_)                                                                                     \
  {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
    ro_barrier();                                                                                          \
    if (t == 0) __hip_atomic_store(flags + j * 32, (unsigned)(e_) | (pub_xcc << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
  }
  // every peer has published epoch e_ (one wave polls, lane p watches member p); false = a wait ran out / somebody aborted / (LOCAL) a
  // peer sits on another XCD - found before any of its payload is read
  auto wait_peers = [&](unsigned e) -> bool {
    if (wave == 0) {
      unsigned ok = 1, far = 0, late = 0;
      if (lane < G && lane != j) {
        unsigned n = 0, w;
        while (((w = __hip_atomic_load(flags + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffffu) < e) {
          __builtin_amdgcn_s_sleep(1);
          if ((++n & 63u) == 0) {
            if (n > kEpochWaitPolls) late = 1;
            if (late || __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
              ok = 0;
              break;
            }
          }
        }
        if (LOCAL && ok && (w >> 24) != my_xcc) far = 1;
      }
      far = __any(far != 0) ? 1u : 0u;
      late = __any(late != 0) ? 1u : 0u;
      ok = (__all(ok != 0) && !far) ? 1u : 0u;
      if (lane == 0) {
        if (!ok) {
          __hip_atomic_store(c.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // the host word is written by whoever FOUND the reason (plain stores to host memory: no PCIe atomics needed); a workgroup that
          // merely saw the abort word leaves it alone.  2: placement (the XCD-local form), 1: a peer never arrived
          if (far) __hip_atomic_store(c.give_up, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else if (late) __hip_atomic_store(c.give_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s_ok = ok;
      }
    }
    ro_barrier();
    return s_ok != 0;
  };
  // the peers' column slices of an exchanged activation -> this workgroup's LDS tile (member-relative column order: peer j + m at columns
  // [m CS, (m + 1) CS)).  One address register for the loads and one for the LDS writes; the rest is a scalar offset per load and an
  // immediate per write (a per-load index computation is loop-invariant, gets hoisted out of the subnet loop and costs two registers per
  // load for the whole kernel).  All loads of a thread (<= 8) are in flight together.
  // (TAG) the same gather as below, re-read until every float shows parity `par`; false: the re-reads ran out / somebody aborted
  auto gather_tagged = [&](const __amdgpu_buffer_rsrc_t& rsX, float* tile, unsigned par) -> bool {
    constexpr int TH = RO_WAVES * 64;
    constexpr int PEER4 = RO_ROWS * CS / 4;
    constexpr int C4 = CS / 4;
    constexpr int PPR = PEER4 >= TH ? 1 : TH / PEER4;
    constexpr int RPP = PEER4 >= TH ? PEER4 / TH : 1;
    constexpr int ROWS_PR = TH / PPR / C4;
    constexpr int NLD = PPR == 1 ? (G - 1) * RPP : G / PPR;
    static_assert(NLD <= 8, "all loads of a gather in flight together");
    const int hw = PPR == 1 ? 0 : wave / (RO_WAVES / PPR);
    const int tt = PPR == 1 ? t : (t & (TH / PPR - 1));
    const int row_t = tt / C4, c4 = tt % C4;
    const unsigned voffx = (unsigned)((row_t * RO_W + c4 * 4) * 4);
    float* const tdst = tile + row_t * RO_LDA + c4 * 4 + hw * (NLD * CS);
    ro_f4 v[NLD];
    unsigned tries = 0, ok = 1;
    for (;;) {
      unsigned bad = 0;
#pragma unroll
      for (int r = 0; r < NLD; ++r) {
        const int mq = 1 + r / RPP, rr = r % RPP;
        const int m = mq + hw * NLD;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((rr * ROWS_PR * RO_W + ((j + m) & (G - 1)) * CS) * 4);
        if (m < G) v[r] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voffx, so, /*sc1*/ 16));
      }
#pragma unroll
      for (int r = 0; r < NLD; ++r)
        if (1 + r / RPP + hw * NLD < G) bad |= ro_tag_bad(v[r], par);
      if (!__any(bad != 0)) break;
      if ((++tries & 15u) == 0 && (tries > kGatherWaitPasses || __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#ifdef IKF_RC_FINE_STAMPS
    if (a.trace != nullptr && t == 0) a.trace[(size_t)blockIdx.x * 64 + 50] += tries;   // failed passes, summed over the call's gathers
#endif
    if (!ok && lane == 0) {
      __hip_atomic_store(c.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tries > kGatherWaitPasses) __hip_atomic_store(c.give_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s_ok = 0;
    }
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int mq = 1 + r / RPP, rr = r % RPP;
      if (mq + hw * NLD < G) *reinterpret_cast<ro_f4*>(tdst + rr * ROWS_PR * RO_LDA + mq * CS) = v[r];
    }
    ro_barrier();
    return s_ok != 0;
  };
  auto gather = [&](const __amdgpu_buffer_rsrc_t& rsX, float* tile) {
    constexpr int TH = RO_WAVES * 64;
    constexpr int PEER4 = RO_ROWS * CS / 4;                     // float4 of one peer's slice: 2048 / 1024 / 512 / 256 / 128
    constexpr int C4 = CS / 4;
    constexpr int PPR = PEER4 >= TH ? 1 : TH / PEER4;           // peers per round of 512 loads: 1 / 1 / 1 / 2 / 4 (by groups of waves)
    constexpr int RPP = PEER4 >= TH ? PEER4 / TH : 1;           // rounds per peer: 4 / 2 / 1 / 1 / 1
    constexpr int ROWS_PR = TH / PPR / C4;                      // rows a round covers: 4 / 8 / 16 / 16 / 16
    constexpr int NLD = PPR == 1 ? (G - 1) * RPP : G / PPR;     // rounds = loads per thread: 4 / 6 / 7 / 8 / 8
    static_assert(NLD <= 8, "all loads of a gather in flight together");
    const int hw = PPR == 1 ? 0 : wave / (RO_WAVES / PPR);      // (wave-uniform) which of the round's peers
    const int tt = PPR == 1 ? t : (t & (TH / PPR - 1));
    const int row_t = tt / C4, c4 = tt % C4;
    const unsigned voffx = (unsigned)((row_t * RO_W + c4 * 4) * 4);
    float* const tdst = tile + row_t * RO_LDA + c4 * 4 + hw * (NLD * CS);
    ro_f4 v[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int mq = 1 + r / RPP, rr = r % RPP;                  // (compile-time) peer of the round's first wave group, row group
      const int m = mq + hw * NLD;
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((rr * ROWS_PR * RO_W + ((j + m) & (G - 1)) * CS) * 4);
      if (m < G) v[r] = __builtin_bit_cast(ro_f4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voffx, so, /*sc1*/ 16));
    }
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int mq = 1 + r / RPP, rr = r % RPP;
      if (mq + hw * NLD < G) *reinterpret_cast<ro_f4*>(tdst + rr * ROWS_PR * RO_LDA + mq * CS) = v[r];
    }
    ro_barrier();
  };

  ro_f4 acc[NCB];
  // as in the row-owner launch: odd k groups accumulate into a second set (two half-length f32 chains per output) wherever a wave runs the
  // whole k range (the k-split forms have shorter chains already)
  constexpr bool TWO = KS == 1;
  ro_f4 accb[TWO ? NCB : 1];
#define RC_ACC(i_) ((TWO && ((i_) & 1)) ? accb : acc)
  ro_f4 af[2];
  // LeakyReLU of the accumulators -> own columns of the LDS tile, and (PUB) write-through to the row tile's exchange buffer
#define RC_EPILOGUE(tile_out, PUB, rsX)                                                                                  \
  if constexpr (KS == 1) {                                                                                               \
    _Pragma("unroll") for (int cb_ = 0; cb_ < NCB; ++cb_) {                                                              \
      ro_f4 v_ = acc[cb_];                                                                                               \
      if constexpr (TWO) v_ += accb[cb_];                                                                                \
      v_ = __builtin_elementwise_max(v_, v_ * a.slope);                                                                  \
      if (TAG && (PUB)) v_ = ro_tag(v_, tag_par);                                                                        \
      const int col_ = (int)(cbg0 + cb_) * 16 + 4 * lq;                                                                  \
      *reinterpret_cast<ro_f4*>((tile_out) + lrow * RO_LDA + (bw + cb_) * 16 + 4 * lq) = v_;                             \
      if (PUB) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ro_u4, v_), rsX, (unsigned)((lrow * RO_W + col_) * 4), 0, ST_AUX); \
    }                                                                                                                    \
  } else {  /* the KS waves of a block hold partial sums over their k shares: through LDS, summed in share order */       \
    *reinterpret_cast<ro_f4*>(red + wave * (RO_ROWS * RO_RS) + lrow * RO_RS + 4 * lq) = acc[0];                          \
    ro_barrier();                                                                                                        \
    if (t < NBM * 64) {                                                                                                  \
      const int b_ = t >> 6, l_ = t & 63, r_ = l_ & 15, q_ = l_ >> 4;                                                    \
      ro_f4 v_ = *reinterpret_cast<const ro_f4*>(red + b_ * (RO_ROWS * RO_RS) + r_ * RO_RS + 4 * q_);                    \
      _Pragma("unroll") for (int k_ = 1; k_ < KS; ++k_)                                                                  \
          v_ += *reinterpret_cast<const ro_f4*>(red + (k_ * NBM + b_) * (RO_ROWS * RO_RS) + r_ * RO_RS + 4 * q_);        \
      v_ = __builtin_elementwise_max(v_, v_ * a.slope);                                                                  \
      if (TAG && (PUB)) v_ = ro_tag(v_, tag_par);                                                                        \
      *reinterpret_cast<ro_f4*>((tile_out) + r_ * RO_LDA + b_ * 16 + 4 * q_) = v_;                                       \
      if (PUB) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ro_u4, v_), rsX, (unsigned)((r_ * RO_W + (j * NBM + b_) * 16 + 4 * q_) * 4), 0, ST_AUX); \
    }                                                                                                                    \
  }
  // (the LDS tiles hold the columns in MEMBER-RELATIVE order - own slice first, then members j+1, j+2, .. - so the rotated k order is the
  // ascending physical order and every LDS offset of the layer is a compile-time constant)
#define RC_AFRAG(tile_in, i_) *reinterpret_cast<const ro_f4*>((tile_in) + lrow * RO_LDA + (ks * KGW + (i_)) * 16 + 4 * lq)
  // a hidden layer, fully unrolled: k groups in the member's rotated order - its own slice first, the peers' after the hand-over
  // (k split, G >= 16: only the first share's own groups are there before the hand-over - every wave waits at the start of the layer)
  constexpr int WAIT_AT = KS > 1 ? 0 : N1;
#define RC_LAYER(hl_, bias_, tile_in, rsXin, e_in, WAIT)                                                                 \
  {                                                                                                                      \
    _Pragma("unroll") for (int cb_ = 0; cb_ < NCB; ++cb_) acc[cb_] = ks == 0 ? bias_[cb_] : ro_f4{0.f, 0.f, 0.f, 0.f};   \
    if constexpr (TWO) { _Pragma("unroll") for (int cb_ = 0; cb_ < NCB; ++cb_) accb[cb_] = ro_f4{0.f, 0.f, 0.f, 0.f}; }  \
    if (!(WAIT && WAIT_AT == 0)) af[0] = RC_AFRAG(tile_in, 0);                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < KGW; ++i_) {                                                                 \
      if (WAIT && i_ == WAIT_AT) {                                                                                       \
        if constexpr (TAG) {                                                                                             \
          RC_FINE(46)                                                                                                    \
          if (!gather_tagged(rsXin, tile_in, tag_par)) return;                                                           \
          RC_FINE(47)                                                                                                    \
        } else {                                                                                                         \
          if (!wait_peers(e_in)) return;                                                                                 \
          gather(rsXin, tile_in);                                                                                        \
        }                                                                                                                \
        af[i_ & 1] = RC_AFRAG(tile_in, i_);                                                                              \
      }                                                                                                                  \
      RC_ISSUE((i_ + PF) % NBUF, (hl_) + ((i_ + PF) / KGW), (i_ + PF) % KGW)                                             \
      if (i_ + 1 < KGW && !(WAIT && i_ + 1 == WAIT_AT)) af[(i_ + 1) & 1] = RC_AFRAG(tile_in, i_ + 1);                    \
      _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                                   \
          _Pragma("unroll") for (int cb_ = 0; cb_ < NCB; ++cb_)                                                          \
              RC_ACC(i_)[cb_] = RO_MFMA(wb[i_ % NBUF][cb_][c_], af[i_ & 1][c_], RC_ACC(i_)[cb_]);                        \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                 \
      _Pragma("unroll") for (int q_ = 0; q_ < NCB; ++q_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                               \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
  }

  for (int s = 0; s < a.n_sub; ++s) {
    const float* sm = small + s * RO_SMALL_WORDS;
    const unsigned e0 = (unsigned)s;   // epochs of this subnet's two exchanges: 2 s + 1 (h2), 2 s + 2 (partial sums)
    const unsigned tag_par = (unsigned)s & 1u;   // (TAG) parity carried by everything this subnet exchanges
    RC_STAMP(1 + (s < 31 ? s : 31))
#define RC_PHASE(i) if (s == 2) { RC_STAMP(40 + (i)) }
#ifdef IKF_RC_FINE_STAMPS   /* probes: the h2 gather of subnet 2 on its own (tools/rowowner_probe.hip) */
#define RC_FINE(i) if (s == 2) { RC_STAMP(i) }
#else
#define RC_FINE(i)
#endif
    RC_PHASE(0)
    small_load(s, 1, b2);
    small_load(s, 2 + RO_KG, b3);
    small_load(s, RO_SUB_GROUPS - 1, wl);
    // ---- first Linear + LeakyReLU, ALL 1024 columns (every member repeats it: cheaper than an exchange) -> tile0, stored in
    //      member-relative column order.  Its state-independent half (pose entries, bias: MFMA 0, 1) is already in a1 - evaluated while
    //      the previous subnet's partial sums were in flight; what is left is the x / softflow half (MFMA 2, 3)
    {
      const ro_f4 uf = *reinterpret_cast<const ro_f4*>(us + lrow * RO_US + 4 * lq);
#pragma unroll
      for (int cc = 2; cc < 4; ++cc)
#pragma unroll
        for (int cb = 0; cb < RO_NCB; ++cb) a1[cb] = RO_MFMA(w1[cb][cc], uf[cc], a1[cb]);
#pragma unroll
      for (int cb = 0; cb < RO_NCB; ++cb) {
        ro_f4 v = a1[cb];
        v = __builtin_elementwise_max(v, v * a.slope);
        const int pb = (wave * RO_NCB + cb - j * (RO_KG / G)) & (RO_KG - 1);   // physical 16-column block of true block 8 wave + cb
        *reinterpret_cast<ro_f4*>(tile0 + lrow * RO_LDA + pb * 16 + 4 * lq) = v;
      }
      w1_load(s + 1);   // (consumed a whole subnet later)
      ro_barrier();
    }
    RC_PHASE(1)
    // ---- hidden 2: tile0 (h1, complete) -> own columns of tile1 and X, epoch s + 1
    RC_LAYER(2 * s, b2, tile0, rsX1, 0u, false)
    RC_EPILOGUE(tile1, true, rsX1)
    if constexpr (TAG) ro_barrier();   // (own slice in LDS for every wave; nobody waits for the stores)
    else RC_PUBLISH(2 * e0 + 1)
    RC_PHASE(2)
    // ---- hidden 3: tile1 (h2) -> own columns of tile0 (h3 stays here)
    RC_LAYER(2 * s + 1, b3, tile1, rsX1, 2 * e0 + 1, true)
    // h3 feeds nothing but this member's last Linear, and a wave that ran the whole k range of its blocks (KS == 1) holds exactly the last Linear's
    // B operand of those blocks in its accumulators (row = lane % 16, four consecutive k): it stays in registers - no LDS round trip, no barrier
    [[maybe_unused]] ro_f4 h3v[NCB];
    if constexpr (KS == 1) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        ro_f4 v = acc[cb];
        if constexpr (TWO) v += accb[cb];
        h3v[cb] = __builtin_elementwise_max(v, v * a.slope);
      }
    } else {
      RC_EPILOGUE(tile0, false, rsX1)
      ro_barrier();
    }
    RC_PHASE(3)
    // ---- last Linear over the member's own h3 columns: per-wave partial sums -> red[wave] -> the member's partial -> P[j], epoch 3 s + 3
    {
      ro_f4 p4[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) p4[cc] = ro_f4{0.f, 0.f, 0.f, 0.f};
      if (ks == 0) {   // (k split: one wave per block of the member's h3 columns)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          ro_f4 hf;
          if constexpr (KS == 1) hf = h3v[cb];
          else hf = *reinterpret_cast<const ro_f4*>(tile0 + lrow * RO_LDA + (bw + cb) * 16 + 4 * lq);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) p4[cc] = RO_MFMA(wl[cb][cc], hf[cc], p4[cc]);
        }
      }
      const ro_f4 p = (p4[0] + p4[1]) + (p4[2] + p4[3]);
      *reinterpret_cast<ro_f4*>(red + wave * (RO_ROWS * RO_RS) + lrow * RO_RS + 4 * lq) = p;
      ro_barrier();
      // the member's partial -> P[j] (write-through), epoch word, then every member sums the G partials in member order
      float mine = 0.f;
      if (t < 256) {
        const int row = t >> 4, o = t & 15;
#pragma unroll
        for (int w = 0; w < RO_WAVES; ++w) mine += red[w * (RO_ROWS * RO_RS) + row * RO_RS + o];
        if constexpr (TAG) mine = ro_tag1(mine, tag_par);
        if constexpr (LOCAL) P[j * 256 + t] = mine;
        else __hip_atomic_store(P + j * 256 + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if constexpr (!TAG) RC_PUBLISH(2 * e0 + 2)
      RC_FIRST_STATIC   // the next subnet's first Linear, state-independent half, under the exchange's latency (w1 = its weights by now)
      if constexpr (TAG) {
        // every member's partial, re-read until all of them show this subnet's parity (the own one is in `mine` already)
        if (t < 256) {
          const int row = t >> 4, o = t & 15;
          constexpr int CH = G < 16 ? G : 16;   // loads in flight together
          float sum = sm[o];   // b_last (zero beyond n_out)
          unsigned tries = 0, ok = 1;
#pragma unroll
          for (int mb = 0; mb < G; mb += CH) {
            float part[CH];
            for (;;) {
              unsigned bad = 0;
#pragma unroll
              for (int q = 0; q < CH; ++q) part[q] = __hip_atomic_load(P + (mb + q) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
              for (int q = 0; q < CH; ++q) bad |= (mb + q == j) ? 0u : ((__float_as_uint(part[q]) ^ tag_par) & 1u);
              if (!__any(bad != 0) || !ok) break;
              if ((++tries & 15u) == 0 && (tries > kSumWaitPolls || __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                ok = 0;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) sum += (mb + q == j) ? mine : part[q];   // (member order)
          }
          if (!ok && lane == 0) {
            __hip_atomic_store(c.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tries > kSumWaitPolls) __hip_atomic_store(c.give_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s_ok = 0;
          }
          s_sum[row * RO_RS + o] = sum;
        }
        ro_barrier();
        if (s_ok == 0) return;
      } else {
      if (!wait_peers(2 * e0 + 2)) return;
      if (t < 256) {
        const int row = t >> 4, o = t & 15;
        // every member's partial (the own one included: its write-through store was drained above), 16 loads in flight at a time, added in
        // member order
        float sum = sm[o];   // b_last (zero beyond n_out)
        constexpr int CH = G < 16 ? G : 16;
#pragma unroll
        for (int mb = 0; mb < G; mb += CH) {
          float part[CH];
#pragma unroll
          for (int q = 0; q < CH; ++q) part[q] = __hip_atomic_load(P + (mb + q) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int q = 0; q < CH; ++q) sum += part[q];
        }
        s_sum[row * RO_RS + o] = sum;
      }
      ro_barrier();
      }
    }
    RC_PHASE(4)
    advance(sm, s + 1 < a.n_sub ? sm + RO_SMALL_WORDS : nullptr, xs + 256 * xcur, xs + 256 * (xcur ^ 1));
    xcur ^= 1;
    ro_barrier();
    RC_PHASE(5)
  }
  RC_STAMP(33)
  if (j == 0 && t < 256) {
    const int row = t >> 4, jj = t & 15;
    if (jj < a.ndof && m0 + row < a.M) {
      const float* x = xs + 256 * xcur + row * 16;
      float q = 0.f;
      for (int k = 0; k < a.D; ++k) {
        float xv = x[k];
        if (a.sigmoid) xv = 1.0f / (1.0f + expf(-xv));
        q = fmaf(xv - a.b_lin[k], a.M_inv[k * a.D + jj], q);
      }
      if (a.clamp_limits) q = fminf(fmaxf(q, a.lo[jj]), a.hi[jj]);
      a.q_out[(size_t)(m0 + row) * a.ndof + jj] = q;
    }
  }
  RC_STAMP(34)
  if (a.trace != nullptr && t == 0) {
    a.trace[(size_t)blockIdx.x * 64 + 36] = wall_clock64();
    a.trace[(size_t)blockIdx.x * 64 + 35] = t0_wall;
  }
#undef RC_STAMP
}

// ---- the stream image of one subnet (see the header of this file); one thread per float4
struct RoPackArgs {
  SubnetWeights w;
  float* out;        // 132 groups x 64 KB
};
__global__ __launch_bounds__(256) void k_rowowner_pack(RoPackArgs p) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 index inside the subnet image
  const int lane = (int)(i4 & 63), cb = (int)((i4 >> 6) & 7), wave = (int)((i4 >> 9) & 7);
  const int g = (int)(i4 >> 12);
  if (g >= RO_SUB_GROUPS) return;
  const int lrow = lane & 15, lq = lane >> 4;
  const SubnetWeights& w = p.w;
  ro_f4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = 0.f;
    if (g == 0) {
      const int col = wave * 128 + cb * 16 + lrow, what = ro_input_slot(4 * lq + c, w.n_x);
      if (what >= 0) x = w.w_first_t[(size_t)(w.n_x + what) * RO_W + col];   // (rows of w_first_t: the x inputs, then the 7 pose entries)
      else if (what == -1) x = w.b_first[col];
      else if (what == -100) x = w.w_soft[col];
      else if (what > -100) x = w.w_first_t[(size_t)(-2 - what) * RO_W + col];
    } else if (g == 1 || g == 2 + RO_KG) {
      x = w.b_mid[g == 1 ? 0 : 1][wave * 128 + cb * 16 + 4 * lq + c];
    } else if (g < RO_SUB_GROUPS - 1) {
      const int l = g < 2 + RO_KG ? 0 : 1;
      const int kg = g - (l == 0 ? 2 : 3 + RO_KG);
      const int col = wave * 128 + cb * 16 + lrow, k = kg * 16 + 4 * lq + c;
      x = w.w_mid[l][(size_t)col * RO_W + k];
    } else {
      const int o = lrow, k = wave * 128 + cb * 16 + 4 * lq + c;
      if (o < w.n_out) x = w.w_last[(size_t)o * RO_W + k];
    }
    v[c] = x;
  }
  reinterpret_cast<ro_f4*>(p.out)[i4] = v;
}

size_t rowowner_subnet_floats() { return (size_t)RO_SUB_GROUPS * (RO_GROUP_BYTES / 4); }
const char* rowowner_kernel_name() { return "k_flow_rowowner"; }
size_t rowowner_stream_floats(int n_sub) { return ((size_t)n_sub * RO_SUB_GROUPS + 4) * (RO_GROUP_BYTES / 4); }  // + ring lead padding
bool rowowner_shape_ok(const FlowDims& d, int n_sub) {
  const int nmax = d.L1 > d.L2 ? d.L1 : d.L2;
  return d.width == RO_W && d.n_hidden == 3 && d.D <= 16 && nmax + 1 <= 8 && 2 * nmax <= 16 && n_sub <= RO_MAX_SUB && d.ndof <= 16;   // (x inputs + softflow: 8 slots)
}
hipError_t launch_rowowner_pack(const SubnetWeights& w, float* out, hipStream_t s) {
  RoPackArgs p{w, out};
  const size_t n4 = (size_t)RO_SUB_GROUPS * (RO_GROUP_BYTES / 16);
  hipLaunchKernelGGL(k_rowowner_pack, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_flow_rowowner(const RoArgs& a, int nbuf, hipStream_t s) {
  static bool done4[64] = {}, done2[64] = {};
  const unsigned grid = (unsigned)((a.M + RO_ROWS - 1) / RO_ROWS);
  hipError_t e;
  if (nbuf == 2) {
    e = ensure_dynamic_lds(k_flow_rowowner<2>, RO_LDS_BYTES, done2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_flow_rowowner<2>, dim3(grid), dim3(RO_WAVES * 64), RO_LDS_BYTES, s, a);
  } else {
    e = ensure_dynamic_lds(k_flow_rowowner<4>, RO_LDS_BYTES, done4);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_flow_rowowner<4>, dim3(grid), dim3(RO_WAVES * 64), RO_LDS_BYTES, s, a);
  }
  return hipGetLastError();
}

size_t cluster_xbuf_floats(int n_rt) { return (size_t)n_rt * RO_ROWS * RO_W; }
size_t cluster_sync_bytes(int n_rt, int G) { return (size_t)n_rt * G * (256 * 4 + 32 * 4) + 128; }   // partial sums, epoch words, abort word
template <int G, bool LOCAL>
static hipError_t launch_cluster_g(const RcArgs& c, unsigned grid, hipStream_t s) {
  static bool done[64] = {};
  hipError_t e = ensure_dynamic_lds(k_flow_cluster<G, LOCAL>, RO_LDS_BYTES, done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_flow_cluster<G, LOCAL>), dim3(grid), dim3(RO_WAVES * 64), RO_LDS_BYTES, s, c);
  return hipGetLastError();
}
// Where the dispatcher puts workgroup b of a grid: the XCD-local form needs workgroups b and b + 8 k on the same XCD.  One launch of
// n_cu single-wave workgroups, each reporting its XCC_ID; true when id(b) == id(b mod 8) for every b (8 XCDs round-robin, or any number
// of XCDs that divides 8, or one).  Asked once per handle at load; every launch of the form still checks its own members.
__global__ void k_xcc_census(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = ro_xcc_id();
}
hipError_t cluster_placement_census(int n_cu, bool* groups_of_8_share_an_xcd) {
  *groups_of_8_share_an_xcd = false;
  if (n_cu <= 0) return hipSuccess;
  unsigned* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(unsigned) * n_cu);
  if (e != hipSuccess) return e;
  std::vector<unsigned> h(n_cu, 0xffffffffu);
  hipLaunchKernelGGL(k_xcc_census, dim3(n_cu), dim3(64), 0, nullptr, d);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(h.data(), d, sizeof(unsigned) * n_cu, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return e;
  bool ok = true;
  for (int b = 0; b < n_cu; ++b) ok = ok && h[b] <= 0xfu && h[b] == h[b % 8];
  *groups_of_8_share_an_xcd = ok;
  return hipSuccess;
}
// TAG variant: nothing is zeroed per launch - the exchange buffers are 0xff bytes when created (cluster_tagged_init) and hold parity 1
// everywhere at the end of every call; the abort word stays 0 until a launch gives up (the engine then re-creates the buffers)
template <int G, bool LOCAL>
static hipError_t launch_cluster_tagged_g(const RcArgs& c, unsigned grid, hipStream_t s) {
  static bool done[64] = {};
  hipError_t e = ensure_dynamic_lds(k_flow_cluster<G, LOCAL, true>, RO_LDS_BYTES, done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_flow_cluster<G, LOCAL, true>), dim3(grid), dim3(RO_WAVES * 64), RO_LDS_BYTES, s, c);
  return hipGetLastError();
}
hipError_t cluster_tagged_init(float* xbuf, size_t xbuf_floats, float* sync, size_t sync_bytes, unsigned* abort_word, hipStream_t s) {
  hipError_t e = hipMemsetAsync(xbuf, 0xff, xbuf_floats * sizeof(float), s);
  if (e == hipSuccess) e = hipMemsetAsync(sync, 0xff, sync_bytes, s);
  if (e == hipSuccess) e = hipMemsetAsync(abort_word, 0, sizeof(unsigned), s);
  return e;
}
hipError_t launch_flow_cluster_tagged(const RcArgs& c, int G, hipStream_t s, int drop_workgroups, bool local) {
  local = local && (G == 4 || G == 8 || G == 16) && drop_workgroups <= 0;
  const unsigned grid = (local ? (unsigned)((c.n_rt + 7) / 8 * 8 * G) : (unsigned)c.n_rt * (unsigned)G) - (unsigned)(drop_workgroups > 0 ? 1 : 0);
  if (G == 2) return launch_cluster_tagged_g<2, false>(c, grid, s);
  if (G == 4) return local ? launch_cluster_tagged_g<4, true>(c, grid, s) : launch_cluster_tagged_g<4, false>(c, grid, s);
  if (G == 8) return local ? launch_cluster_tagged_g<8, true>(c, grid, s) : launch_cluster_tagged_g<8, false>(c, grid, s);
  if (G == 16) return local ? launch_cluster_tagged_g<16, true>(c, grid, s) : launch_cluster_tagged_g<16, false>(c, grid, s);
  if (G == 32) return launch_cluster_tagged_g<32, false>(c, grid, s);
  return hipErrorInvalidValue;
}
bool cluster_local_form(int G) { return G == 4 || G == 8 || G == 16; }   // (G = 2 gains nothing, G = 32 is bound by its weight stream: they stay spread)
unsigned cluster_grid(int n_rt, int G, bool local) { return local ? (unsigned)((n_rt + 7) / 8 * 8 * G) : (unsigned)n_rt * (unsigned)G; }
hipError_t launch_flow_cluster(const RcArgs& c, int G, hipStream_t s, int drop_workgroups, bool local) {
  // (drop_workgroups > 0: tests of the repair path - the last workgroup of the spread form is not launched, its row tile's members wait in vain)
  local = local && cluster_local_form(G) && drop_workgroups <= 0;
  const unsigned grid = cluster_grid(c.n_rt, G, local) - (unsigned)(drop_workgroups > 0 ? 1 : 0);
  // epochs count from 1 inside a call: everything polled is zeroed in front of it (one memset: pbuf and flags are one block)
  hipError_t e = hipMemsetAsync(c.pbuf, 0, cluster_sync_bytes(c.n_rt, G), s);
  if (e != hipSuccess) return e;
  if (G == 2) return launch_cluster_g<2, false>(c, grid, s);
  if (G == 4) return local ? launch_cluster_g<4, true>(c, grid, s) : launch_cluster_g<4, false>(c, grid, s);
  if (G == 8) return local ? launch_cluster_g<8, true>(c, grid, s) : launch_cluster_g<8, false>(c, grid, s);
  if (G == 16) return local ? launch_cluster_g<16, true>(c, grid, s) : launch_cluster_g<16, false>(c, grid, s);
  if (G == 32) return launch_cluster_g<32, false>(c, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace ikf
