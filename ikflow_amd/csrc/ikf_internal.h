// Internal declarations shared by the HIP translation units of libikflow_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ikflow_amd.h"
#include "../../include/ikflow_amd_debug.h"
#include "kin_math.h"

namespace ikf {

// ---- packed weights of one coupling subnet (device pointers into the model's weight arena) -----------------
struct SubnetWeights {
  const float* w_first_t;  // [in_real][width]  first Linear, transposed, softflow column dropped into w_soft
  const float* w_soft;     // [width]           first-Linear column that multiplies the softflow scale (cond[7])
  const float* b_first;    // [width]
  const float* w_mid[3];   // [width][width]    hidden Linear layers (row = output feature, K contiguous)
  const float* b_mid[3];   // [width]
  const float* w_last;     // [out][width]      last Linear
  const float* b_last;     // [out]
  int n_x;                 // number of x inputs feeding the subnet (L1 for subnet1, L2 for subnet2)
  int n_out;               // 2*L2 for subnet1, 2*L1 for subnet2
};

// ---- where the conditional (target pose) of a flow row comes from ------------------------------------------
// pose of global row r = poses + stride * (idx ? idx[r % n_mod] : r % n_mod)
//   batch form      : idx = null, n_mod = n,  stride = 7      (ikflow_solver.py:338)
//   single-pose form: idx = null, n_mod = 1,  stride = 0..7   (ikflow_solver.py:333-336, y.expand)
//   exact-IK tiling : idx = active-pose list, n_mod = n_active (ikflow_solver.py:182-186, cond.repeat((R,1)))
struct PoseSource {
  const float* poses;
  const int* idx;
  long long n_mod;
  int stride;
  float softflow;
};

struct FlowDims {
  int D, L1, L2, width, n_hidden, ndof, n_pose;  // n_pose = 7 real pose entries of the conditional
  float clamp, slope;
};

// flow_kernels.hip

// The > 64 KB dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is per kernel and per device: `done` is
// the caller's static per-kernel flag array, indexed by the current device.
template <class Kern>
inline hipError_t ensure_dynamic_lds(Kern kern, size_t bytes, bool (&done)[64]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || done[dev]) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[dev] = true;
  return e;
}

hipError_t launch_first_layer(const SubnetWeights& w, const FlowDims& d, const float* x_in, int x_off,
                              const PoseSource& ps, long long row0, long long rows, float* h_out, hipStream_t s);
hipError_t launch_gemm_lrelu(int variant, const float* A, const float* W, const float* bias, float* C, long long M,
                             int N, int K, float slope, hipStream_t s);
struct CouplingArgs {
  const float* x_in;   // [rows][D] state before this block (latent for the first executed block)
  float* x_out;        // [rows][D] state buffer owned by the engine
  const int* perm_inv; // [D]   (subnet 2 only)
  const float* M_inv;  // [D][D] (final only)
  const float* b_lin;  // [D]    (final only)
  const float* lo;     // [ndof] (final only)
  const float* hi;     // [ndof]
  float* q_out;        // [rows][ndof] (final only)
  int which;           // 1 or 2
  int is_final;        // last executed block, subnet 2: apply FixedLinearTransform^-1, slice, clamp
  int clamp_limits;
  int sigmoid;         // sigmoid_on_output graph: 1/(1+exp(-x)) before the linear transform
};
hipError_t launch_last_layer_coupling(const SubnetWeights& w, const FlowDims& d, const float* h_in,
                                      const CouplingArgs& ca, long long rows, hipStream_t s);
int gemm_variant_count();
const char* gemm_kernel_name();

// flow_fused.hip - the three-kernel-per-subnet form: the last Linear is reduced to per-tile partial sums in the epilogue
// of the last hidden contraction, and the affine-coupling update runs at the head of the NEXT subnet's entry kernel.
constexpr int IKF_PSTRIDE = 16;  // floats per row in one partial-sum slot
struct PendingCoupling {         // coupling of the previous subnet, whose last Linear exists only as partial sums
  const float* P;                // [slots][rows_pad][IKF_PSTRIDE], or null: no pending (state passes through)
  const float* b_last;           // [n_out]
  const int* perm_inv;           // [D], used when which == 2
  long long slot_stride;         // rows_pad * IKF_PSTRIDE
  int slots, which, n_out;
};
struct EntryArgs {      // k_subnet_entry: pending coupling + first Linear of the next subnet
  PendingCoupling pend;
  const float* x_src;   // [M][D] state before the pending coupling
  float* x_dst;         // [M][D] state after it
  int M, D, L1;
  float clamp;
  int x_off, n_x;       // slice of the new state that feeds this subnet
  PoseSource ps;
  long long row0;
  const float* w1t;     // [IN][width]
  const float* w1soft;  // [width]
  const float* b1;      // [width]
  int width;
  float slope;
  float* h_out;         // [rows_pad][width]  fp32, or the f16 hi/lo split image when split_out != 0
  int split_out;
  int* split_flag;      // split_out: OR'ed with 1 when an activation is non-finite or beyond the f16 range (65504)
  int wt_stores;        // activation stores write-through (sc1): nothing is left dirty in the L2s for the launch's end to flush
  unsigned* zero_words; // first entry kernel of a call whose later launches hand over inside a launch: words it zeroes
  int n_zero;           //   (the row tiles' arrival counters, see TailSync), else null / 0
};
// Row-tile-local hand-over inside a launch (k_flow_gemm<true, .., FUSE> / k_flow_gemm_skinny<true, 2, FUSE>): the column-tile
// workgroups of one row tile publish their last-Linear partial sums write-through, arrive on the row tile's counter, wait
// for their siblings and then run the NEXT subnet's entry phase (pending coupling + their column slice of its first
// Linear) themselves - the k_subnet_entry launch between two subnets disappears.  Placement-independent: payload stores
// and loads are agent-scope (sc1), the counter is an agent-scope atomic; needs every workgroup of the launch resident
// (the launcher only uses it when the grid is at most one workgroup per CU).
struct TailSync {
  unsigned* arrive;     // [row tiles] arrival counters, zeroed by the call's first entry kernel
  unsigned target;      // arrive[tm] once every column tile of this subnet has published: (fused subnets so far + 1) * column tiles
  int* give_up;         // host-visible word, set non-zero by a workgroup whose wait ran out (a sibling was not resident)
  int n_in;             // inputs of the next subnet's first Linear (n_x + 7 pose entries)
};
struct FusedGemmArgs {
  // contraction C = lrelu(A . W^T + bias), K = N = width
  const float* A;     // [rows_pad][K]
  const float* W;     // [N][K]
  const float* Wf;    // fragment-major image of W for k_flow_gemm_skinny (launch_wfrag_pack), or null
  const float* bias;  // [N]
  float* C;           // [rows_pad][N]  (unused when the epilogue reduces to partials)
  int M, N, K;
  float slope;
  int wt_stores;      // activation stores write-through (sc1), see EntryArgs
  int tune;           // IKF_TUNE_* bits the launcher looks at (IKF_TUNE_DEEP16)
  // partial-sum epilogue: P_out[slot][row][o] = sum over the tile's columns of lrelu(...)[row][col] * w_last[o][col]
  const float* w_last;  // [n_out][N]
  int n_out;
  float* P_out;         // [slots][rows_pad][IKF_PSTRIDE]
  long long p_slot_stride;
};
// The whole subnet chain of a call in one launch for <= 128 rows (k_flow_chain16, flow_fused.hip): 256 persistent workgroups, row tile <->
// XCD, hand-over between layers through an arrival counter in the XCD's L2.
constexpr int IKF_CHAIN_XCDS = 8, IKF_CHAIN_PER_XCD = 32;     // the placement the launcher verifies (launch_xcd_census)
constexpr int IKF_CHAIN_TICKET = 0;                           // ctl[TICKET + 32 x]: workgroups of XCD x so far (one 128-byte line per XCD)
constexpr int IKF_CHAIN_ARRIVE = 32 * IKF_CHAIN_XCDS;         // ctl[ARRIVE + 32 x]: phase arrivals of row tile x
constexpr int IKF_CHAIN_ABORT = 64 * IKF_CHAIN_XCDS;          // non-zero: a wait ran out somewhere, every wait ends
constexpr int IKF_CHAIN_DONE = IKF_CHAIN_ABORT + 1;           // workgroups that left; the last one zeroes ctl[]
constexpr int IKF_CHAIN_CTL_WORDS = IKF_CHAIN_DONE + 1;
constexpr unsigned kChainSpinLimit = 1u << 22;                // polls (s_sleep 1 + one L2 load each): about a second
struct ChainSubnet {       // device table, execution order; built once per (weights, scratch) - nothing in it depends on the call
  EntryArgs e;             // pend.P == nullptr for the first subnet; ps / row0 / M / (first subnet) x_src come from ChainCall
  FusedGemmArgs g[2];      // the two hidden contractions (g[0] runs in the head; g[1] ends in the last Linear's partial sums)
  int n_in;
};
struct ChainCall {         // what changes from call to call
  PoseSource ps;
  const float* x0;         // state rows of the first subnet (the caller's latent, already offset to the chunk)
  long long row0;
  int M;
};
struct ChainSync {
  unsigned* ctl;           // IKF_CHAIN_CTL_WORDS words, zero between calls
  int* give_up;            // host-visible word (see TailSync)
  int row_tiles;           // XCDs x >= row_tiles have nothing to do
};
bool flow_chain16_ok(long long rows, int width, int D, int n_out, int n_hidden);
hipError_t launch_flow_chain16(const ChainSubnet* d_tab, int n_sub, const ChainCall& call, const ChainSync& cs, int K, hipStream_t s);
hipError_t launch_xcd_census(unsigned* d_out /* [256] */, hipStream_t s);  // out[b] = XCC_ID of workgroup b of a chain-shaped launch
struct FinalizeArgs {
  PendingCoupling pend;
  const float* x_src;  // [M][D]
  int M, D, L1, ndof;
  float clamp;
  const float* M_inv;  // [D][D]
  const float* b_lin;  // [D]
  const float* lo;
  const float* hi;
  int clamp_limits;
  int sigmoid;         // apply 1/(1+exp(-x)) before the linear transform (sigmoid_on_output graph)
  float* q_out;        // [M][ndof]
};
// per-handle tuning switches of the small-batch kernels (ikf_set_gemm_variant 150 .. 163); every combination computes the same function
enum : int {
  IKF_TUNE_ROWS16 = 1,     // batches of <= 128 rows on 16-row tiles (v_mfma_f32_16x16x4_f32)
  IKF_TUNE_DEEP16 = 2,     // those kernels request their whole operand stream up front
  IKF_TUNE_TILES16 = 4,    // batches of <= 64 rows on 16 x 16 tiles
  IKF_TUNE_ROWS32_V2 = 8,  // 129 .. 256 rows on 32 x 32 tiles built from 16x16x4 MFMAs (off by default: measured slower)
  IKF_TUNE_DEFAULT = IKF_TUNE_ROWS16 | IKF_TUNE_DEEP16 | IKF_TUNE_TILES16,
};
int fused_pick_cfg(long long rows, int width, int tune = IKF_TUNE_DEFAULT);  // tile configuration for a batch (-1: width not supported)
int fused_slots(int cfg, int width);            // partial-sum slots that configuration produces per row
int fused_max_slots(int width);
hipError_t launch_subnet_entry(int n_in, const EntryArgs& e, hipStream_t s);
hipError_t launch_flow_gemm(bool epi_red, int cfg, const FusedGemmArgs& a, hipStream_t s);
// the last hidden contraction of a subnet with the next subnet's entry phase in its tail (e: the entry arguments of the
// NEXT subnet, its pending coupling = this launch's partial sums)
bool fused_tail_ok(int cfg, long long rows, int width, int D, int n_out);
int fused_tail_col_tiles(int cfg, int width);
hipError_t launch_flow_gemm_tail(int cfg, const FusedGemmArgs& a, const EntryArgs& e, const TailSync& ts, hipStream_t s);
// small batches: k_subnet_entry + the first hidden contraction in one launch (k_entry_gemm_skinny)
bool entry_gemm_ok(int cfg, long long rows, int width, int D, int n_out);
hipError_t launch_entry_gemm(int n_in, bool epi_red, int cfg, const EntryArgs& e, const FusedGemmArgs& a, hipStream_t s);
extern int g_entry_geom_override;  // probes: force an entry-kernel geometry
hipError_t launch_flow_finalize(const FinalizeArgs& a, hipStream_t s);
const char* fused_kernel_name();

// flow_rowowner.hip - the whole inverse pass in ONE launch: a workgroup keeps its 16 rows on chip through every subnet and streams
// the weights past them (width 1024, coeff_fn_config 3).  No scratch, no inter-workgroup synchronisation.
struct RoSubnet {          // one executed subnet (execution order): what is not in the stream
  float b_last[16];
  int perm_inv[16];        // which == 2: new_state[d] = cat[perm_inv[d]] (PermuteRandom rev); identity otherwise
  int which, n_x, x_off, n_half;   // n_half = n_out / 2 state elements rewritten, starting at (which == 1 ? L1 : 0)
};

struct RoArgs {
  const float* stream;     // [n_sub][132][8 waves][8 blocks][64 lanes][4], + PF groups of padding
  unsigned stream_bytes;
  const RoSubnet* sub;
  int n_sub;
  const float* x0;         // [M][D] latent rows of this chunk
  PoseSource ps;
  long long row0;
  int M, D, L1, ndof;
  float clamp, slope;
  const float* M_inv;      // [D][D]
  const float* b_lin;      // [D]
  const float* lo;
  const float* hi;
  int clamp_limits, sigmoid;
  float* q_out;            // [M][ndof]
  unsigned long long* trace;   // probes: [grid][64] shader-clock stamps, or null
  const unsigned* run_if;  // null, or a device word: the launch does nothing unless it is non-zero (the cluster form's repair launch)
};

struct RcArgs {            // cluster form (k_flow_cluster<G>): G workgroups per row tile split the hidden columns
  RoArgs ro;
  int n_rt;                // row tiles = ceil(M / 16); grid = n_rt * G workgroups, all resident
  float* xbuf;             // [n_rt][16][1024] exchanged h2 tiles
  float* pbuf;             // [n_rt][G][16][16] last-Linear partial sums per member
  unsigned* flags;         // [n_rt][G][32] epoch published by each member (one 128-byte line each) = pbuf + n_rt * G * 256 (one memset)
  unsigned* abort_word;    // device word behind the flags (same memset): set when a wait ran out - every other wait ends, and the
                           // row-owner launch queued behind this one (run_if = abort_word) recomputes the chunk
  int test_far;            // tests: workgroup 0 of the XCD-local form publishes a wrong XCC_ID (its peers must give up with code 2); in the tagged
                           // form it also drops its payload stores, as a member whose stores stay in another XCD's L2 would
  unsigned* xcc_words;     // tagged + XCD-local form: [n_rt][32] words (launch_seq << 8 | XCC_ID), one per member, one 128-byte line per row tile,
                           // written at the start of every launch; read only when a wait has run out (was it placement, or a peer not resident?)
  unsigned launch_seq;     // ... 24 bits, never 0xffffff (what the buffers are created as), another value in every launch
  int* give_up;            // host-visible twin: 1 = a wait ran out (the engine stops using the cluster form on this handle), 2 = a member
                           // of the XCD-local form met a peer on another XCD (the engine goes back to the spread form)
};
size_t cluster_xbuf_floats(int n_rt);
size_t cluster_sync_bytes(int n_rt, int G);   // pbuf + flags + the abort word
hipError_t launch_flow_cluster(const RcArgs& c, int G, hipStream_t s, int drop_workgroups = 0, bool local = false);
// the drain-free hand-over (every exchanged float carries the subnet's parity in its mantissa's last bit): buffers initialised once
hipError_t cluster_tagged_init(float* xbuf, size_t xbuf_floats, float* sync, size_t sync_bytes, unsigned* abort_word, hipStream_t s);
hipError_t launch_flow_cluster_tagged(const RcArgs& c, int G, hipStream_t s, int drop_workgroups = 0, bool local = false);
hipError_t cluster_placement_census(int n_cu, bool* groups_of_8_share_an_xcd);   // one tiny launch: where workgroup b of a grid lands
bool cluster_local_form(int G);                          // G = 4 / 8 / 16: a form with every member of a row tile on one XCD exists
unsigned cluster_grid(int n_rt, int G, bool local);      // workgroups of a launch (the local form pads to whole groups of 8 row tiles)
constexpr int IKF_RO_ROWS = 16;                       // rows per workgroup
size_t rowowner_subnet_floats();                        // floats of one subnet's stream image
size_t rowowner_stream_floats(int n_sub);               // whole image incl. the ring's lead padding
bool rowowner_shape_ok(const FlowDims& d, int n_sub);
hipError_t launch_rowowner_pack(const SubnetWeights& w, float* out, hipStream_t s);
hipError_t launch_flow_rowowner(const RoArgs& a, int nbuf, hipStream_t s);
const char* rowowner_kernel_name();

// flow_split.hip - the hidden contraction on the f16 matrix cores with an error-compensated operand split:
//   a = hi + lo/2048,  hi = f16(a),  lo = f16((a - hi) * 2048)        (same for the weights, split once at load)
//   a.w ~= hi_a*hi_w + (hi_a*lo_w + lo_a*hi_w)/2048                    (3 v_mfma_f32_32x32x16_f16, fp32 accumulate)
// Measured on MI355X against fp64 (tools/split_probe.hip): rms error 1.5e-7 vs 4.1e-7 for the exact-f32 MFMA at K=1024.
// Operands live in HBM as the "split-32" image: per row, per block of 32 k: 32 hi halves (64 B) then 32 lo halves (64 B)
// - 4 bytes per element like fp32, one full 128-B line per (row, K tile).
constexpr float IKF_SPLIT_SCALE = 2048.0f;
struct SplitGemmArgs {
  const void* A;      // [rows_pad][K] split-32 image
  const void* W;      // [N][K] split-32 image
  const void* Wf;     // fragment-major image of W for k_split_skinny (launch_wfrag_pack_split), or null
  const float* bias;  // [N]
  void* C;            // [rows_pad][N] split-32 image (unused when the epilogue reduces to partials)
  int M, N, K;
  float slope;
  const float* w_last;  // [n_out][N] fp32
  int n_out;
  float* P_out;
  long long p_slot_stride;
  int* flag;            // OR'ed with 1 when an activation written to C is non-finite or beyond the f16 range
};
// true when a does not fit the f16 hi part of the split (|a| > 65504, inf, NaN)
__device__ __forceinline__ bool split_out_of_range(float a) { return !(__builtin_fabsf(a) <= 65504.0f); }
// The same test on converted operands, two at a time: a value beyond the f16 range (or non-finite) converts to an f16 with all
// exponent bits set, i.e. |half| >= 0x7C00.  range_track() keeps the running maximum of the packed |hi| halves (one v_and and
// one v_pk_max_u16 per two elements - the float test costs a compare and a scalar OR per element); range_hit() reads it once.
typedef unsigned short ikf_ushort2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned range_track(unsigned running, unsigned packed_hi_halves) {
#ifdef IKF_NO_RANGE_FLAG  // probes only: A/B of what the detection costs
  return running;
#else
  const ikf_ushort2 a = __builtin_bit_cast(ikf_ushort2, running);
  const ikf_ushort2 b = __builtin_bit_cast(ikf_ushort2, packed_hi_halves & 0x7FFF7FFFu);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, b));
#endif
}
__device__ __forceinline__ bool range_hit(unsigned running) { return (running & 0xFFFFu) >= 0x7C00u || (running >> 16) >= 0x7C00u; }
hipError_t launch_split_gemm(bool epi_red, int cfg, const SplitGemmArgs& a, hipStream_t s);
int split_pick_cfg(long long rows, int width);
int split_slots(int cfg, int width);
bool split_cfg_needs_frag(int cfg);
hipError_t launch_wfrag_pack_split(const void* Wsplit, int N, int K, void* out, hipStream_t s);
void split32_pack_host(const float* src, int rows, int K, uint16_t* dst);  // host-side packer (probe tool)
int fused_skinny_cfg();
int fused_skinny32_cfg();
int fused_skinny16_cfg();
int fused_skinny16x16_cfg();
int fused_skinny32v2_cfg();
hipError_t launch_wfrag_pack(const float* W, int N, int K, float* out, hipStream_t s);
hipError_t launch_split32_pack(const float* d_src, long long rows, int K, void* d_dst, int* d_flag, hipStream_t s);
const char* split_kernel_name();

// kin_kernels.hip (struct Chain and the per-row arithmetic: kin_math.h)
// IKF_MAX_CAPSULES (24) comes from include/ikflow_amd.h
constexpr int IKF_MAX_CAPSULE_PAIRS = IKF_MAX_CAPSULES * (IKF_MAX_CAPSULES - 1) / 2;
struct CollisionModel {
  int n_caps, n_pairs;
  int frame[IKF_MAX_CAPSULES];      // 0 = base, j + 1 = the frame that follows actuated joint j
  float p0[IKF_MAX_CAPSULES][3], p1[IKF_MAX_CAPSULES][3], radius[IKF_MAX_CAPSULES];
  uint8_t pair_a[IKF_MAX_CAPSULE_PAIRS], pair_b[IKF_MAX_CAPSULE_PAIRS];
};
hipError_t launch_self_collision(const Chain* d_chain, const CollisionModel* d_cm, int ndof, const float* q, long long n,
                                 float* min_dist, uint8_t* colliding, hipStream_t s);
hipError_t launch_fk(const Chain* d_chain, int ndof, const float* q, long long n, float* poses, hipStream_t s);
hipError_t launch_pose_error(const Chain* d_chain, int ndof, const float* q, const float* targets, long long n,
                             float* pos_err, float* rot_err, hipStream_t s);
// lm_precision: 1 = fp64 inside (default), 0 = the reference's fp32 arithmetic with an LU / partial-pivoting solve
hipError_t launch_lm_step(const Chain* d_chain, int ndof, const float* targets, const float* q, long long n,
                          float* q_out, int lm_precision, hipStream_t s);
hipError_t launch_jacobian(const Chain* d_chain, int ndof, const float* q, long long n, float* jac, hipStream_t s);
hipError_t launch_clamp(const Chain* d_chain, int ndof, const float* q, long long n, float* q_out, hipStream_t s);
hipError_t launch_limits_exceeded(const Chain* d_chain, int ndof, const float* q, long long n, uint8_t* out,
                                  hipStream_t s);
// exact-IK round kernels
constexpr int IKF_MAX_LIMIT_COLS = 32;
hipError_t launch_pose_distance(const float* a, const float* b, long long n, float acos_eps, float* pe, float* re,
                                hipStream_t s);
hipError_t launch_limits_exceeded_table(const float* lo, const float* hi, int ncols, const float* q, long long n,
                                        uint8_t* out, hipStream_t s);
// exact IK: all LM iterations of a round in one launch (row_valid_iter[row] = iteration, 1-based, at which the row first met the
// thresholds, 0 = never; q keeps that iteration's value) and the per-pose selection (earliest iteration, highest repeat within it)
hipError_t launch_exact_lm_iters(const Chain* d_chain, int ndof, const float* poses, const int* pose_idx, int n_active, int repeat,
                                 int n_steps, const float* q_in /* seeds; may be q */, float* q, uint8_t* row_valid_iter,
                                 unsigned* pose_first /* [n_active] scratch, or null: no early exit */, float pos_thr, float rot_thr,
                                 int lm_precision, hipStream_t s);
hipError_t launch_exact_select_first(int ndof, const int* pose_idx, int n_active, int repeat, const float* q,
                                     const uint8_t* row_valid_iter, float* q_out, uint8_t* valid_out, int init, hipStream_t s);
hipError_t launch_all_active(long long n, int* idx_out, int* count_out, hipStream_t s);  // idx = 0 .. n-1, count = n (round 0)
// ordered list of the indices with valid[i] == 0 and their count; block_scratch: 2 * compact_blocks(n) ints
long long compact_blocks(long long n);
hipError_t launch_compact_invalid(const uint8_t* valid, long long n, int* idx_out, int* count_out, int* block_scratch,
                                  hipStream_t s);

}  // namespace ikf
