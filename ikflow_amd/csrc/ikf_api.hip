// C-ABI of libikflow_amd.so: handle lifetime, weight packing, and the host-side orchestration of the approximate and
// exact IK paths (the launches that replace IKFlowSolver._run_inference / _generate_exact_ik_solutions,
// ikflow/ikflow_solver.py:85-247, 345-411).  See include/ikflow_amd.h for the contract.
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "ikf_internal.h"

using namespace ikf;

static thread_local std::string g_last_error;

static ikf_status fail(ikf_status code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
#define IKF_HIP(call)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      return fail(IKF_ERR_HIP, std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                   std::to_string(__LINE__) + ")");                                            \
  } while (0)

// Every entry point runs on the handle's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    else if (err == hipSuccess) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define IKF_ON_DEVICE(m)                                                                                       \
  DeviceGuard dev_guard_((m)->device);                                                                         \
  if (dev_guard_.err != hipSuccess)                                                                            \
    return fail(IKF_ERR_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(dev_guard_.err));

struct ikf_model {
  int device = 0;
  // the per-handle scratch is shared by all calls: a call that arrives on a different stream than the previous one
  // first waits for the event recorded behind the previous call's work
  hipEvent_t tail_event = nullptr;
  hipStream_t tail_stream = nullptr;
  bool tail_valid = false;
  ikf_model_desc desc{};
  FlowDims dims{};
  bool loaded = false;
  int gemm_variant = -1;  // -1 = choose by batch size
  int tile_cfg = -1;      // fused pipeline: -1 = choose by batch size, 0..3 forced (variant 100..103)
  int fuse_entry = 1;     // small batches: entry kernel + first hidden contraction as one launch (0: always two launches)
  // the next subnet's entry phase in the tail of the last hidden contraction (TailSync).  OFF by default: measured slower than the
  // k_subnet_entry launch it replaces (r03: +3.6 % per call at 4096 rows, +7.4 % at 512) - kept as a tested,
  // bit-identical opt-in (ikf_set_gemm_variant 121) because it is the priced answer to "hand over inside the launch"
  int tune = IKF_TUNE_DEFAULT;  // IKF_TUNE_* switches of the small-batch kernels (ikf_set_gemm_variant 150 .. 163)
  int fuse_tail = 0;
  // Activation stores write-through (sc1): the lines go to memory as they are written instead of sitting dirty in the XCD L2s until the
  // launch's end flushes them (r03, tools/variant_ab.py: 3.261 -> 3.229 ms per call at 4096 rows with the contractions' stores
  // write-through, 1.831 -> 1.796 at 2048 and 0.721 -> 0.706 at 512 with the entry kernel's too; the entry kernel's 16-byte stores do
  // not pay at 4096 rows).  bit 0 contractions, bit 1 entry kernel; -1 = by batch size (contractions always, entry kernel <= 2048 rows)
  int wt_stores = -1;
  // <= 128 rows: the whole subnet chain in one launch, hand-over inside each XCD (k_flow_chain16, flow_fused.hip).  OFF by default:
  // bit-identical to the per-layer launches but slower (r03: 0.422 against 0.365 ms per call at 128 rows, 0.362 against 0.271 at 1 row) -
  // a hand-over inside an XCD needs the ROWS partitioned over the XCDs, so every XCD's L2 pulls the whole 4.2 MB weight matrix of every
  // layer (8 x the traffic of the per-layer launches, whose column tiles are spread over the XCDs), and small batches are bound by
  // exactly that stream.  Kept as the tested, priced answer to "XCD-local synchronisation".
  //   chain_mode: 0 off, 1 on (ikf_set_gemm_variant 170 / 171); chain_census: -1 not yet taken, 0 the dispatcher does not hand 32
  //   workgroups to each of 8 XCDs on this device (the chain is never used), 1 verified
  int chain_mode = 0, chain_census = -1;
  ChainSubnet* d_chain_tab = nullptr;  // [2 nb_nodes] per-subnet arguments, rebuilt when the weights or the scratch change
  bool chain_tab_valid = false;
  unsigned* d_chain_ctl = nullptr;     // [IKF_CHAIN_CTL_WORDS], zero between calls (the launch's last workgroup re-zeroes it)
  unsigned* d_arrive = nullptr;  // [kArriveWords] row-tile arrival counters of the fused tail (zeroed by every call's first entry kernel)
  int* h_give_up = nullptr;      // pinned, device-visible: set by a workgroup whose in-launch wait ran out
  int precision = 0;      // 0: hidden contractions on the exact-f32 MFMA; 1: error-compensated 3x f16 MFMA split
  int lm_precision = 1;   // LM step: 1 fp64 inside (Cholesky), 0 the reference's fp32 arithmetic (LU, partial pivoting) - ikf_set_lm_precision
  uint16_t* split_arena = nullptr;  // split-32 images of the hidden Linear weights
  std::vector<const void*> w_mid_split;  // [subnet][layer] -> device pointer (flattened: subnet*3 + layer)
  float* split_frag_arena = nullptr;     // fragment-major copies of the split-32 images (small-batch f16-split kernel)
  std::vector<const void*> w_mid_split_frag;
  // fragment-major images of the hidden Linear weights (small-batch per-layer kernels, <= 512 rows): built by the first chunk that needs
  // them, or ahead of time by ikf_reserve - a handle whose small batches run the cluster form never pays the 201 MB / the pack launches
  float* wfrag_arena = nullptr;
  bool wfrag_built = false;
  double load_ms = 0.0, frag_ms = 0.0;   // host wall time of the last ikf_load_weights (device work included) / of building these images
  std::vector<const float*> w_mid_frag;  // [subnet][layer], same flattening; null when the width does not fit

  // Row-owner form (flow_rowowner.hip): the whole inverse pass of a batch in ONE launch, a workgroup per 16 rows, weights streamed past
  // them from `ro_stream` (the subnets' parameters in execution and consumption order, +203 MB for Panda).  Taken for the full rounds of
  // n_cu x 16 rows of a batch and for a last partial round of at least ro_min_tail rows; the rest runs on the per-layer kernels.
  //   ro_mode: -1 by batch size, 0 never, 1 always (ikf_set_gemm_variant 180 / 181 / 182)
  float* ro_stream = nullptr;
  RoSubnet* d_ro_sub = nullptr;
  int ro_mode = -1, ro_nbuf = 4;
  int n_cu = 256;
  long long ro_min_tail = -1;  // -1: the last partial round goes to whatever plan_tail finds cheapest; >= 0 (probes): to the row-owner launch from that many rows on
  // Cluster form (k_flow_cluster<G>, flow_rowowner.hip) for what is left below a round: G = 8 / 4 / 2 workgroups per 16-row tile split the
  // hidden columns and exchange activations inside the launch (<= 512 / 1024 / 2048 rows).  Every cluster launch is followed by a
  // predicated row-owner launch of the same rows that runs only if a wait ran out (cl_abort set): results are valid either way, and the
  // handle stops using the form (cl_give_up).   cl_mode: -1 by batch size, 0 never, 1 whenever the grid fits (ikf_set_gemm_variant 185 / 186 / 187)
  int cl_mode = -1;
  long long cl_rows = 0;          // row capacity of the exchange buffers
  float* cl_xbuf = nullptr;       // [tiles][16][1024]
  float* cl_sync = nullptr;       // partial sums, epoch words, abort word (one memset per launch)
  // The drain-free hand-over (r05, flow_rowowner.hip TAG): every exchanged float carries its subnet's parity in the last mantissa bit, so a
  // producer neither drains its stores nor publishes an epoch and a consumer validates what it reads.  Taken for G = 2 .. 16 (- 3 .. 4.5 %
  // per call, and no memset in front of the launch; with 32 members the early re-reads of whole slices cost more than the epoch words).
  // Its buffers are its own: every float in them has parity 1 between calls (created as 0xff bytes; n_sub is even), which an epoch-word
  // launch's zeroing memset or untagged payload would break.  cl_tag_dirty: a tagged launch gave up - the buffers are re-created in front of
  // the next one (until then every tagged launch returns at once and its repair launch does the work: the abort word stays set).
  //   cl_tagged: 1 (default) / 0 (ikf_set_gemm_variant 193 / 192)
  int cl_tagged = 1;
  float* cl_xbuf_t = nullptr;
  float* cl_sync_t = nullptr;
  size_t cl_sync_t_bytes = 0;
  bool cl_tag_dirty = false;
  int* h_cl_give_up = nullptr;    // pinned, device-visible
  int cl_drop_next = 0;           // tests: the next cluster launch runs one workgroup short (ikf_set_gemm_variant 188): its tile's waits run out
  long long cl_repairs = 0;       // give-ups seen so far (ikf_cluster_repairs)
  // A wait that ran out means a peer was not resident - another process's kernel held CUs just then.  That tenant may be gone a second
  // later, so the form is not switched off for good: it sits out cl_pause plans (ikf_generate_* calls), 16 after the first give-up and
  // twice as many after every further one (at most 65536); kClusterCleanStreak calls of the form in a row without a give-up forget the
  // history.  ikf_cluster_backoff reports what is left of the pause.
  long long cl_pause = 0;         // plans the form still sits out
  long long cl_backoff = 0;       // length of the last pause (0: no give-up on record)
  int cl_clean = 0;               // cluster calls since the last give-up
  bool cl_used_last = false;      // the previous plan contained a cluster launch
  int cl_census_ok = -1;          // the placement census at load: workgroups b and b + 8 k share an XCD (1) or not (0); -1 not asked
  int cl_far_next = 0;            // tests (ikf_set_gemm_variant 191): the next XCD-local launch's workgroup 0 publishes a wrong XCC_ID
  unsigned cl_launch_seq = 0;     // tagged + XCD-local launches carry a 24-bit sequence number in their placement words (RcArgs::launch_seq)
  int cl_tl_nrt = 0, cl_tl_G = 0; // row tiles / members of the most recent tagged + XCD-local launch (whose placement words a give-up makes the host read)
  unsigned* cl_tl_words = nullptr;
  int cl_local = 1;               // G = 4 / 8 / 16: the form with a row tile's members on one XCD (hand-over through its L2); 0 after a member met a
                                  // peer on another XCD (placement is verified in the launch, never assumed) or by ikf_set_gemm_variant 189

  // packed weights (one arena)
  float* arena = nullptr;
  size_t arena_floats = 0;
  std::vector<SubnetWeights> subnets;  // [2*block + (which-1)]
  int* d_perm_inv = nullptr;           // [nb_nodes][D]
  float* d_Minv = nullptr;             // [D][D]
  float* d_blin = nullptr;             // [D]
  Chain* d_chain = nullptr;            // robot chain + limits
  CollisionModel* d_collision = nullptr;  // capsules + pairs (ikf_set_collision_model), or null

  // scratch
  long long chunk_rows = 0;  // capacity of the per-chunk flow scratch
  float* xbuf = nullptr;     // [chunk][D]
  float* xbuf2 = nullptr;    // [chunk][D]   second state buffer (fused path ping-pongs the state)
  float* pbuf = nullptr;     // [slots][chunk][IKF_PSTRIDE] last-Linear partial sums (fused path)
  float* pbuf_alt = nullptr; // second set for odd subnets when a subnet has ONE hidden contraction (see ensure_scratch); else == pbuf
  float* hA = nullptr;       // [chunk][width]
  float* hB = nullptr;
  // exact-IK scratch
  long long exact_rows = 0, exact_poses = 0;
  long long exact_upfront_rows = 32LL << 20;  // ikf_set_exact_upfront_rows
  float* ex_q = nullptr;          // [rows][ndof]
  uint8_t* ex_row_valid = nullptr;  // [rows]
  unsigned* ex_pose_first = nullptr;  // [poses] earliest valid iteration over a pose's repeats in the running round (early-exit hint)
  int* ex_pose_idx = nullptr;     // [poses]
  int* ex_block_scratch = nullptr;  // [2 * compact_blocks(poses)] per-block counts / offsets of the ordered compaction
  int* ex_count = nullptr;        // device
  int* h_count = nullptr;         // pinned host
  // f16x3 range guard
  int* d_split_flag = nullptr;    // device word OR'ed by every kernel that produces a split operand out of the f16 range
  int* h_split_flag = nullptr;    // pinned host
  int split_guard = 1;
  long long split_fallbacks = 0;

  // optional per-launch HIP-event timing of the dominant kernel (ikf_profile_begin/_end)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;  // pairs
  size_t prof_used = 0;
  double last_event_overhead_ms = 0.0;
};

static const int kArriveWords = 256;  // >= row tiles of any launch that hands over inside the launch (<= 256 tiles)
static const size_t kProfMaxPairs = 8192;
static hipError_t prof_mark(ikf_model* m, hipStream_t s) {
  if (!m->prof_on || m->prof_used >= 2 * kProfMaxPairs) return hipSuccess;
  // the pool grows by whole pairs in front of a pair's FIRST record only: a hipEventCreate between a launch and its closing record
  // would delay that record on the host - behind a 3 ms launch the first creations were seen to add 0.3 ms to the measured pair
  if (m->prof_used >= m->prof_ev.size() && (m->prof_used & 1) == 0) {
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e;
      hipError_t r = hipEventCreate(&e);
      if (r != hipSuccess) return r;
      m->prof_ev.push_back(e);
    }
  }
  if (m->prof_used >= m->prof_ev.size()) return hipErrorInvalidValue;
  return hipEventRecord(m->prof_ev[m->prof_used++], s);
}

static hipError_t stream_enter(ikf_model* m, hipStream_t s) {
  if (m->tail_valid && s != m->tail_stream) return hipStreamWaitEvent(s, m->tail_event, 0);
  return hipSuccess;
}
static hipError_t stream_leave(ikf_model* m, hipStream_t s);
// Records the handle's tail event behind whatever a call has enqueued, on EVERY exit path (an error return in the middle of
// a call leaves kernels in flight that still use the shared scratch; the next call on another stream must wait for them).
struct StreamScope {
  ikf_model* m;
  hipStream_t s;
  bool armed = false;
  StreamScope(ikf_model* m_, hipStream_t s_) : m(m_), s(s_) {}
  hipError_t enter() {
    hipError_t e = stream_enter(m, s);
    armed = (e == hipSuccess);
    return e;
  }
  hipError_t leave() {  // the success path: reports the record's own status
    armed = false;
    return stream_leave(m, s);
  }
  ~StreamScope() {
    if (armed) (void)stream_leave(m, s);
  }
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
};
static hipError_t stream_leave(ikf_model* m, hipStream_t s) {
  if (!m->tail_event) {
    hipError_t e = hipEventCreateWithFlags(&m->tail_event, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  m->tail_stream = s;
  m->tail_valid = true;
  return hipEventRecord(m->tail_event, s);
}

static const long long kMaxChunkRows = 16384;  // keeps the [chunk x width] activations (64 MB each at width 1024) inside the 256 MB L3
// The kernels tile the hidden width in units of 256.  Any other coeff_fn_internal_size (ikflow/model.py:51-96 accepts any)
// is run at the next multiple of 256 with zero weights and biases in the padding: a padded unit outputs lrelu(0) = 0 and
// feeds 0 * 0 into every later sum, so the results are those of the unpadded network exactly.
static const int kWidthUnit = 256;
static const int kMaxWidth = 4096;
static long long chunk_cap(const ikf_model* m);

extern "C" const char* ikf_last_error(void) { return g_last_error.c_str(); }
extern "C" int ikf_abi_version(void) { return IKF_ABI_VERSION; }
extern "C" const char* ikf_dominant_kernel_name(void) { return fused_kernel_name(); }
extern "C" const char* ikf_split_kernel_name(void) { return split_kernel_name(); }

static void free_scratch(ikf_model* m) {
  if (m->xbuf) (void)hipFree(m->xbuf);
  if (m->hA) (void)hipFree(m->hA);
  if (m->hB) (void)hipFree(m->hB);
  if (m->xbuf2) (void)hipFree(m->xbuf2);
  if (m->pbuf) (void)hipFree(m->pbuf);
  m->xbuf = m->hA = m->hB = m->xbuf2 = m->pbuf = m->pbuf_alt = nullptr;
  m->chunk_rows = 0;
  m->chain_tab_valid = false;  // (the table holds these pointers)
}
static void free_exact(ikf_model* m) {
  if (m->ex_q) (void)hipFree(m->ex_q);
  if (m->ex_row_valid) (void)hipFree(m->ex_row_valid);
  if (m->ex_pose_idx) (void)hipFree(m->ex_pose_idx);
  if (m->ex_block_scratch) (void)hipFree(m->ex_block_scratch);
  if (m->ex_pose_first) (void)hipFree(m->ex_pose_first);
  m->ex_q = nullptr; m->ex_row_valid = nullptr; m->ex_pose_idx = nullptr;
  m->ex_block_scratch = nullptr; m->ex_pose_first = nullptr;
  m->exact_rows = m->exact_poses = 0;
}

extern "C" ikf_status ikf_create(const ikf_model_desc* desc, int device, ikf_model** out) {
  if (!desc || !out) return fail(IKF_ERR_NULL_POINTER, "ikf_create: null argument");
  *out = nullptr;
  if (desc->abi_version != IKF_ABI_VERSION)
    return fail(IKF_ERR_BAD_ARGUMENT, "ikf_create: ABI version mismatch (header " + std::to_string(IKF_ABI_VERSION) +
                                          ", caller " + std::to_string(desc->abi_version) + ")");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(IKF_ERR_NO_DEVICE, "ikf_create: no HIP device visible (this engine has no CPU path)");
  if (device < 0 || device >= ndev) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_create: device index out of range");
  const int D = desc->dim;
  if (desc->nb_nodes < 1 || D < 2 || D > IKF_MAX_DIM) return fail(IKF_ERR_BAD_SHAPE, "ikf_create: nb_nodes/dim out of range (2 <= D <= 16)");
  if (desc->dim_cond != 7 && desc->dim_cond != 8) return fail(IKF_ERR_BAD_SHAPE, "ikf_create: dim_cond must be 7 or 8");
  if (desc->sigmoid_on_output && desc->dim_cond != 7)
    return fail(IKF_ERR_BAD_ARGUMENT, "sigmoid_on_output and softflow are incompatible, disable one or the other");
  if (desc->n_hidden < 1 || desc->n_hidden > 4) return fail(IKF_ERR_BAD_SHAPE, "ikf_create: Number of layers `n_layers` must be in [1, ..., 4]");
  if (desc->width < 1 || desc->width > kMaxWidth)
    return fail(IKF_ERR_BAD_SHAPE, "ikf_create: coeff_fn_internal_size must be in [1, 4096]");
  if (desc->ndof < 4 || desc->ndof > IKF_MAX_DOF || desc->ndof > D)
    return fail(IKF_ERR_BAD_SHAPE, "ikf_create: ndof must be in [4, 8] and <= dim");
  DeviceGuard dev_guard_(device);
  if (dev_guard_.err != hipSuccess) return fail(IKF_ERR_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(dev_guard_.err));

  ikf_model* m = new ikf_model();
  m->device = device;
  {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) m->n_cu = n_cu;
  }
  m->desc = *desc;
  m->dims.D = D;
  m->dims.L1 = D / 2;  // ikflow/model.py:336 (old FrEIA rule)
  m->dims.L2 = D - D / 2;
  m->dims.width = (desc->width + kWidthUnit - 1) / kWidthUnit * kWidthUnit;
  m->dims.n_hidden = desc->n_hidden;
  m->dims.ndof = desc->ndof;
  m->dims.n_pose = 7;
  m->dims.clamp = desc->clamp;
  m->dims.slope = desc->leaky_slope;

  Chain ch{};
  ch.ndof = desc->ndof;
  for (int j = 0; j < desc->ndof; ++j) {
    ch.joints[j] = desc->chain[j];
    ch.lo[j] = desc->joint_lo[j];
    ch.hi[j] = desc->joint_hi[j];
    if (ch.joints[j].kind != 1 && ch.joints[j].kind != 2) {
      delete m;
      return fail(IKF_ERR_BAD_ARGUMENT, "ikf_create: chain joint kind must be 1 (revolute) or 2 (prismatic)");
    }
  }
  memcpy(ch.tool, desc->tool, sizeof(ch.tool));
  hipError_t e = hipMalloc(&m->d_chain, sizeof(Chain));
  if (e == hipSuccess) e = hipMemcpy(m->d_chain, &ch, sizeof(Chain), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&m->ex_count, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(&m->h_count, sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&m->d_split_flag, sizeof(int));
  if (e == hipSuccess) e = hipMemset(m->d_split_flag, 0, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(&m->h_split_flag, sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&m->d_arrive, sizeof(unsigned) * kArriveWords);
  if (e == hipSuccess) e = hipMemset(m->d_arrive, 0, sizeof(unsigned) * kArriveWords);
  if (e == hipSuccess) e = hipHostMalloc(&m->h_give_up, sizeof(int), hipHostMallocMapped);
  if (e == hipSuccess) *m->h_give_up = 0;
  if (e == hipSuccess) e = hipHostMalloc(&m->h_cl_give_up, sizeof(int), hipHostMallocMapped);
  if (e == hipSuccess) *m->h_cl_give_up = 0;
  if (e == hipSuccess) e = hipMalloc(&m->d_chain_ctl, sizeof(unsigned) * IKF_CHAIN_CTL_WORDS);
  if (e == hipSuccess) e = hipMemset(m->d_chain_ctl, 0, sizeof(unsigned) * IKF_CHAIN_CTL_WORDS);
  if (e == hipSuccess) e = hipMalloc(&m->d_chain_tab, sizeof(ChainSubnet) * 2 * (size_t)desc->nb_nodes);
  if (e != hipSuccess) {
    ikf_destroy(m);
    return fail(IKF_ERR_HIP, std::string("ikf_create: allocation failed: ") + hipGetErrorString(e));
  }
  *out = m;
  return IKF_OK;
}

extern "C" void ikf_destroy(ikf_model* m) {
  if (!m) return;
  DeviceGuard dev_guard_(m->device);
  free_scratch(m);
  free_exact(m);
  if (m->arena) (void)hipFree(m->arena);
  if (m->split_arena) (void)hipFree(m->split_arena);
  if (m->split_frag_arena) (void)hipFree(m->split_frag_arena);
  if (m->wfrag_arena) (void)hipFree(m->wfrag_arena);
  if (m->ro_stream) (void)hipFree(m->ro_stream);
  if (m->d_ro_sub) (void)hipFree(m->d_ro_sub);
  if (m->cl_xbuf) (void)hipFree(m->cl_xbuf);
  if (m->cl_sync) (void)hipFree(m->cl_sync);
  if (m->cl_xbuf_t) (void)hipFree(m->cl_xbuf_t);
  if (m->cl_sync_t) (void)hipFree(m->cl_sync_t);
  if (m->h_cl_give_up) (void)hipHostFree(m->h_cl_give_up);
  if (m->d_perm_inv) (void)hipFree(m->d_perm_inv);
  if (m->d_Minv) (void)hipFree(m->d_Minv);
  if (m->d_blin) (void)hipFree(m->d_blin);
  if (m->d_chain) (void)hipFree(m->d_chain);
  if (m->d_collision) (void)hipFree(m->d_collision);
  if (m->ex_count) (void)hipFree(m->ex_count);
  if (m->h_count) (void)hipHostFree(m->h_count);
  if (m->d_split_flag) (void)hipFree(m->d_split_flag);
  if (m->h_split_flag) (void)hipHostFree(m->h_split_flag);
  if (m->d_arrive) (void)hipFree(m->d_arrive);
  if (m->h_give_up) (void)hipHostFree(m->h_give_up);
  if (m->d_chain_ctl) (void)hipFree(m->d_chain_ctl);
  if (m->d_chain_tab) (void)hipFree(m->d_chain_tab);
  for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
  if (m->tail_event) (void)hipEventDestroy(m->tail_event);
  delete m;
}

extern "C" int ikf_weights_loaded(const ikf_model* m) { return (m && m->loaded) ? 1 : 0; }

// ---------------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------------
static const ikf_tensor* find_tensor(const std::unordered_map<std::string, const ikf_tensor*>& idx, const std::string& k) {
  auto it = idx.find(k);
  return it == idx.end() ? nullptr : it->second;
}

static ikf_status need(const std::unordered_map<std::string, const ikf_tensor*>& idx, const std::string& key, int dtype,
                       std::initializer_list<int64_t> shape, const ikf_tensor** out) {
  const ikf_tensor* t = find_tensor(idx, key);
  if (!t) return fail(IKF_ERR_MISSING_TENSOR, "Missing key(s) in state_dict: \"" + key + "\"");
  if (!t->h_data) return fail(IKF_ERR_NULL_POINTER, "state_dict tensor \"" + key + "\" has a null data pointer");
  if (t->dtype != dtype) return fail(IKF_ERR_MISSING_TENSOR, "state_dict tensor \"" + key + "\" has the wrong dtype");
  bool ok = (t->ndim == (int)shape.size());
  int i = 0;
  if (ok)
    for (int64_t s : shape) ok = ok && (t->shape[i++] == s);
  if (!ok) {
    std::string got = "(", want = "(";
    for (int k = 0; k < t->ndim; ++k) got += std::to_string(t->shape[k]) + (k + 1 < t->ndim ? ", " : "");
    i = 0;
    for (int64_t s : shape) want += std::to_string(s) + (++i < (int)shape.size() ? ", " : "");
    return fail(IKF_ERR_MISSING_TENSOR, "size mismatch for " + key + ": copying a param with shape " + got +
                                            ") from checkpoint, the shape in current model is " + want + ").");
  }
  *out = t;
  return IKF_OK;
}

static size_t align64(size_t n) { return (n + 63) & ~size_t(63); }  // 64 floats = 256 B

// split-32 images (same bytes as fp32) of every hidden Linear weight, converted on the device from the fp32 arena
static ikf_status build_split_weights(ikf_model* m) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes, W = d.width;
  if (m->split_arena || d.n_hidden < 2 || W % 128 != 0 || !m->loaded) return IKF_OK;
  const size_t per = (size_t)W * W * 2;  // uint16 elements per layer
  const size_t n_layers = (size_t)2 * NB * (d.n_hidden - 1);
  // the range word is shared with the activation guard: take what is pending out of it so that only the weight packs below
  // can set it, and put the pending bits back afterwards (they belong to ikf_split_overflow_pending)
  int pending_flag = 0;
  IKF_HIP(hipDeviceSynchronize());
  IKF_HIP(hipMemcpy(&pending_flag, m->d_split_flag, sizeof(int), hipMemcpyDeviceToHost));
  IKF_HIP(hipMemset(m->d_split_flag, 0, sizeof(int)));
  IKF_HIP(hipMalloc(&m->split_arena, sizeof(uint16_t) * per * n_layers));
  m->w_mid_split.assign((size_t)2 * NB * 3, nullptr);
  size_t li = 0;
  for (int si = 0; si < 2 * NB; ++si)
    for (int l = 0; l < d.n_hidden - 1; ++l, ++li) {
      uint16_t* dst = m->split_arena + li * per;
      IKF_HIP(launch_split32_pack(m->subnets[si].w_mid[l], W, W, dst, m->d_split_flag, nullptr));
      m->w_mid_split[(size_t)si * 3 + l] = dst;
    }
  // fragment-major copies for the small-batch kernel (same bytes again)
  m->w_mid_split_frag.assign((size_t)2 * NB * 3, nullptr);
  if (split_cfg_needs_frag(split_pick_cfg(1, W))) {
    const size_t per_f = (size_t)W * W;  // dwords
    IKF_HIP(hipMalloc(&m->split_frag_arena, sizeof(float) * per_f * n_layers));
    li = 0;
    for (int si = 0; si < 2 * NB; ++si)
      for (int l = 0; l < d.n_hidden - 1; ++l, ++li) {
        float* dstf = m->split_frag_arena + li * per_f;
        IKF_HIP(launch_wfrag_pack_split(m->w_mid_split[(size_t)si * 3 + l], W, W, dstf, nullptr));
        m->w_mid_split_frag[(size_t)si * 3 + l] = dstf;
      }
  }
  IKF_HIP(hipDeviceSynchronize());
  // a weight beyond the f16 range cannot be split: the mode is refused (the f32 path is unaffected)
  int wflag = 0;
  IKF_HIP(hipMemcpy(&wflag, m->d_split_flag, sizeof(int), hipMemcpyDeviceToHost));
  IKF_HIP(hipMemcpy(m->d_split_flag, &pending_flag, sizeof(int), hipMemcpyHostToDevice));
  if (wflag != 0) {
    (void)hipFree(m->split_arena); m->split_arena = nullptr;
    if (m->split_frag_arena) { (void)hipFree(m->split_frag_arena); m->split_frag_arena = nullptr; }
    m->w_mid_split.assign((size_t)2 * NB * 3, nullptr);
    m->w_mid_split_frag.assign((size_t)2 * NB * 3, nullptr);
    m->precision = 0;
    return fail(IKF_ERR_BAD_ARGUMENT, "f16x3 precision refused: a hidden Linear weight is non-finite or exceeds the f16 range (65504); staying on f32");
  }
  return IKF_OK;
}

// fragment-major images (k_wfrag_pack) of every hidden Linear weight for k_flow_gemm_skinny (rows <= 512): the second
// copy costs width^2 * 4 B per layer (201 MB for the Panda model) of the 288 GB
static void drop_frag_weights(ikf_model* m) {
  if (m->wfrag_arena) { (void)hipFree(m->wfrag_arena); m->wfrag_arena = nullptr; }
  m->w_mid_frag.assign((size_t)2 * m->desc.nb_nodes * 3, nullptr);
  m->wfrag_built = false;
}
static ikf_status build_frag_weights(ikf_model* m) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes, W = d.width;
  if (m->wfrag_built) return IKF_OK;
  const auto t0 = std::chrono::steady_clock::now();
  drop_frag_weights(m);
  m->wfrag_built = true;   // (also when the shape has no such image: nothing to build)
  if (d.n_hidden < 2 || fused_pick_cfg(512, W) != fused_skinny_cfg()) return IKF_OK;
  const size_t per = (size_t)W * W;
  const size_t n_layers = (size_t)2 * NB * (d.n_hidden - 1);
  IKF_HIP(hipMalloc(&m->wfrag_arena, sizeof(float) * per * n_layers));
  size_t li = 0;
  for (int si = 0; si < 2 * NB; ++si)
    for (int l = 0; l < d.n_hidden - 1; ++l, ++li) {
      float* dst = m->wfrag_arena + li * per;
      IKF_HIP(launch_wfrag_pack(m->subnets[si].w_mid[l], W, W, dst, nullptr));
      m->w_mid_frag[(size_t)si * 3 + l] = dst;
    }
  IKF_HIP(hipDeviceSynchronize());
  m->chain_tab_valid = false;  // (the chain's argument table carries these pointers)
  m->frag_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return IKF_OK;
}

// the row-owner kernel's parameter stream: every subnet's weights in execution order (block NB-1 .. 0, s1 then s2) and, inside a subnet,
// in the order the kernel consumes them (k_rowowner_pack), plus the small per-subnet table (last-Linear bias, perm_inv, split)
static ikf_status ensure_cluster_scratch(ikf_model* m, long long rows);
static void drop_rowowner_stream(ikf_model* m) {
  if (m->ro_stream) { (void)hipFree(m->ro_stream); m->ro_stream = nullptr; }
  if (m->d_ro_sub) { (void)hipFree(m->d_ro_sub); m->d_ro_sub = nullptr; }
}
// Nothing here is needed by the per-layer kernels: whatever fails (the second 203 MB, the census launch, the exchange buffers) leaves the
// handle WITHOUT the resident-row forms - rowowner_allowed / cluster_allowed test ro_stream - and ikf_load_weights still succeeds; the
// reason is kept for ikf_last_error.
static hipError_t build_rowowner_stream_hip(ikf_model* m, const std::vector<int>& perm_host) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes, n_sub = 2 * NB;
  const size_t floats = rowowner_stream_floats(n_sub);
  hipError_t e = hipMalloc(&m->ro_stream, sizeof(float) * floats);
  if (e != hipSuccess) return e;
  if ((e = hipMemset(m->ro_stream, 0, sizeof(float) * floats)) != hipSuccess) return e;
  if ((e = hipMalloc(&m->d_ro_sub, sizeof(RoSubnet) * n_sub)) != hipSuccess) return e;
  std::vector<RoSubnet> tab(n_sub);
  for (int sidx = 0; sidx < n_sub; ++sidx) {
    const int b = NB - 1 - sidx / 2, which = 1 + (sidx & 1);
    const SubnetWeights& w = m->subnets[2 * b + which - 1];
    if ((e = launch_rowowner_pack(w, m->ro_stream + (size_t)sidx * rowowner_subnet_floats(), nullptr)) != hipSuccess) return e;
    RoSubnet& r = tab[sidx];
    memset(&r, 0, sizeof(r));
    if ((e = hipMemcpy(r.b_last, w.b_last, sizeof(float) * w.n_out, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    for (int k = 0; k < 16; ++k) r.perm_inv[k] = k < d.D ? perm_host[(size_t)b * d.D + k] : k;
    r.which = which; r.n_x = w.n_x; r.x_off = which == 1 ? 0 : d.L1; r.n_half = w.n_out / 2;
  }
  if ((e = hipMemcpy(m->d_ro_sub, tab.data(), sizeof(RoSubnet) * n_sub, hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  // the XCD-local hand-over of the cluster form (G = 4 / 8 / 16) needs workgroups b and b + 8 k of a grid on one XCD: asked of the device once
  // (and checked again by every such launch among its own members)
  bool grouped = false;
  if ((e = cluster_placement_census(m->n_cu, &grouped)) != hipSuccess) return e;
  m->cl_census_ok = grouped ? 1 : 0;
  if (!grouped) m->cl_local = 0;
  return hipSuccess;
}
static ikf_status build_rowowner_stream(ikf_model* m, const std::vector<int>& perm_host) {
  const FlowDims& d = m->dims;
  const int n_sub = 2 * m->desc.nb_nodes;
  drop_rowowner_stream(m);
  if (!rowowner_shape_ok(d, n_sub) || d.slope < 0.f || d.slope > 1.f) return IKF_OK;
  if (rowowner_stream_floats(n_sub) * 4 >= (size_t)1 << 32) return IKF_OK;  // (one 32-bit buffer descriptor)
  hipError_t e = build_rowowner_stream_hip(m, perm_host);
  // the cluster form's exchange buffers have one size (8 MB + 1.2 MB): reserved here, so that no call ever allocates for them
  if (e == hipSuccess && ensure_cluster_scratch(m, 1) != IKF_OK) e = hipErrorOutOfMemory;
  if (e != hipSuccess) {
    (void)hipGetLastError();   // (a refused allocation is sticky only until it is read)
    drop_rowowner_stream(m);
    (void)fail(IKF_ERR_HIP, std::string("ikf_load_weights: the resident-row forms are not available on this handle (") + hipGetErrorString(e) +
                                "); every batch size runs the per-layer kernels");
  }
  return IKF_OK;
}

extern "C" ikf_status ikf_load_weights(ikf_model* m, const ikf_tensor* tensors, int n_tensors) {
  if (!m || !tensors) return fail(IKF_ERR_NULL_POINTER, "ikf_load_weights: null argument");
  IKF_ON_DEVICE(m)
  const auto t_load0 = std::chrono::steady_clock::now();
  std::unordered_map<std::string, const ikf_tensor*> idx;
  for (int i = 0; i < n_tensors; ++i)
    if (tensors[i].name) idx[tensors[i].name] = &tensors[i];

  const FlowDims& d = m->dims;
  const int D = d.D, W = d.width, NB = m->desc.nb_nodes, C = m->desc.dim_cond;
  const int Wu = m->desc.width;  // width of the tensors in the file; W >= Wu is the padded width the kernels run at
  const int n_lin = d.n_hidden + 1;

  // pass 1: sizes
  size_t total = 0;
  for (int b = 0; b < NB; ++b)
    for (int which = 1; which <= 2; ++which) {
      const int n_x = (which == 1) ? d.L1 : d.L2;
      const int n_out = 2 * ((which == 1) ? d.L2 : d.L1);
      total += align64((size_t)(n_x + 7) * W) + 2 * align64(W);           // first (transposed), soft column, bias
      total += (size_t)(d.n_hidden - 1) * (align64((size_t)W * W) + align64(W));
      total += align64((size_t)n_out * W) + align64(n_out);
    }
  std::vector<float> host(total, 0.f);
  std::vector<SubnetWeights> subs(2 * NB);
  std::vector<size_t> off_first(2 * NB), off_soft(2 * NB), off_bfirst(2 * NB), off_last(2 * NB), off_blast(2 * NB);
  std::vector<std::vector<size_t>> off_mid(2 * NB), off_bmid(2 * NB);
  std::vector<int> perm_host((size_t)NB * D);

  size_t cur = 0;
  for (int b = 0; b < NB; ++b) {
    const int moff = m->desc.sigmoid_on_output ? 1 : 0;
    const std::string pkey = "module_list." + std::to_string(2 * b + 1 + moff) + ".perm_inv";
    const ikf_tensor* tp = nullptr;
    ikf_status st = need(idx, pkey, 1, {D}, &tp);
    if (st != IKF_OK) return st;
    std::vector<char> seen(D, 0);
    for (int k = 0; k < D; ++k) {
      const int64_t v = static_cast<const int64_t*>(tp->h_data)[k];
      if (v < 0 || v >= D || seen[v]) return fail(IKF_ERR_MISSING_TENSOR, pkey + " is not a permutation of range(D)");
      seen[v] = 1;
      perm_host[(size_t)b * D + k] = (int)v;
    }
    for (int which = 1; which <= 2; ++which) {
      const int si = 2 * b + which - 1;
      const int n_x = (which == 1) ? d.L1 : d.L2;
      const int n_out = 2 * ((which == 1) ? d.L2 : d.L1);
      const std::string base = "module_list." + std::to_string(2 * b + 2 + moff) + ".subnet" + std::to_string(which) + ".";
      // first Linear: weight [W][n_x + C] -> transposed [n_x + 7][W] (+ softflow column apart)
      const ikf_tensor *tw = nullptr, *tb = nullptr;
      st = need(idx, base + "0.weight", 0, {Wu, n_x + C}, &tw);
      if (st != IKF_OK) return st;
      st = need(idx, base + "0.bias", 0, {Wu}, &tb);
      if (st != IKF_OK) return st;
      const float* w0 = static_cast<const float*>(tw->h_data);
      off_first[si] = cur;
      for (int k = 0; k < n_x + 7; ++k)
        for (int c = 0; c < Wu; ++c) host[cur + (size_t)k * W + c] = w0[(size_t)c * (n_x + C) + k];
      cur += align64((size_t)(n_x + 7) * W);
      off_soft[si] = cur;
      if (C == 8)
        for (int c = 0; c < Wu; ++c) host[cur + c] = w0[(size_t)c * (n_x + C) + n_x + 7];
      cur += align64(W);
      off_bfirst[si] = cur;
      memcpy(&host[cur], tb->h_data, sizeof(float) * Wu);
      cur += align64(W);
      for (int l = 1; l < d.n_hidden; ++l) {
        st = need(idx, base + std::to_string(2 * l) + ".weight", 0, {Wu, Wu}, &tw);
        if (st != IKF_OK) return st;
        st = need(idx, base + std::to_string(2 * l) + ".bias", 0, {Wu}, &tb);
        if (st != IKF_OK) return st;
        off_mid[si].push_back(cur);
        for (int r = 0; r < Wu; ++r)
          memcpy(&host[cur + (size_t)r * W], static_cast<const float*>(tw->h_data) + (size_t)r * Wu, sizeof(float) * Wu);
        cur += align64((size_t)W * W);
        off_bmid[si].push_back(cur);
        memcpy(&host[cur], tb->h_data, sizeof(float) * Wu);
        cur += align64(W);
      }
      st = need(idx, base + std::to_string(2 * (n_lin - 1)) + ".weight", 0, {n_out, Wu}, &tw);
      if (st != IKF_OK) return st;
      st = need(idx, base + std::to_string(2 * (n_lin - 1)) + ".bias", 0, {n_out}, &tb);
      if (st != IKF_OK) return st;
      off_last[si] = cur;
      for (int r = 0; r < n_out; ++r)
        memcpy(&host[cur + (size_t)r * W], static_cast<const float*>(tw->h_data) + (size_t)r * Wu, sizeof(float) * Wu);
      cur += align64((size_t)n_out * W);
      off_blast[si] = cur;
      memcpy(&host[cur], tb->h_data, sizeof(float) * n_out);
      cur += align64(n_out);
      subs[si].n_x = n_x;
      subs[si].n_out = n_out;
    }
  }
  if (cur != total) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_load_weights: internal packing size mismatch");

  const ikf_tensor* tM = nullptr;
  ikf_status st = need(idx, "module_list.0.M_inv", 0, {D, D}, &tM);
  if (st != IKF_OK) return st;
  std::vector<float> blin(D, 0.f);
  if (m->desc.sigmoid_on_output && !find_tensor(idx, "module_list.0.b"))  // the scaling node's offset is never zero
    return fail(IKF_ERR_MISSING_TENSOR, "Missing key(s) in state_dict: \"module_list.0.b\"");
  if (const ikf_tensor* tb = find_tensor(idx, "module_list.0.b")) {
    if (tb->dtype != 0 || !tb->h_data) return fail(IKF_ERR_MISSING_TENSOR, "module_list.0.b has the wrong dtype");
    int64_t numel = 1;
    for (int k = 0; k < tb->ndim; ++k) numel *= tb->shape[k];
    if (numel != D) return fail(IKF_ERR_MISSING_TENSOR, "size mismatch for module_list.0.b");
    memcpy(blin.data(), tb->h_data, sizeof(float) * D);
  }

  // upload
  if (m->arena) { (void)hipFree(m->arena); m->arena = nullptr; }
  if (!m->d_perm_inv) IKF_HIP(hipMalloc(&m->d_perm_inv, sizeof(int) * (size_t)NB * D));
  if (!m->d_Minv) IKF_HIP(hipMalloc(&m->d_Minv, sizeof(float) * D * D));
  if (!m->d_blin) IKF_HIP(hipMalloc(&m->d_blin, sizeof(float) * D));
  IKF_HIP(hipMalloc(&m->arena, sizeof(float) * total));
  m->arena_floats = total;
  IKF_HIP(hipMemcpy(m->arena, host.data(), sizeof(float) * total, hipMemcpyHostToDevice));
  IKF_HIP(hipMemcpy(m->d_perm_inv, perm_host.data(), sizeof(int) * (size_t)NB * D, hipMemcpyHostToDevice));
  IKF_HIP(hipMemcpy(m->d_Minv, tM->h_data, sizeof(float) * D * D, hipMemcpyHostToDevice));
  IKF_HIP(hipMemcpy(m->d_blin, blin.data(), sizeof(float) * D, hipMemcpyHostToDevice));
  for (int si = 0; si < 2 * NB; ++si) {
    SubnetWeights& s = subs[si];
    s.w_first_t = m->arena + off_first[si];
    s.w_soft = m->arena + off_soft[si];
    s.b_first = m->arena + off_bfirst[si];
    for (int l = 0; l < 3; ++l) {
      s.w_mid[l] = (l < (int)off_mid[si].size()) ? m->arena + off_mid[si][l] : nullptr;
      s.b_mid[l] = (l < (int)off_bmid[si].size()) ? m->arena + off_bmid[si][l] : nullptr;
    }
    s.w_last = m->arena + off_last[si];
    s.b_last = m->arena + off_blast[si];
  }
  m->subnets = subs;
  // the split-32 weight images of the f16-split contraction are built on the device when that mode is selected
  if (m->split_arena) { (void)hipFree(m->split_arena); m->split_arena = nullptr; }
  if (m->split_frag_arena) { (void)hipFree(m->split_frag_arena); m->split_frag_arena = nullptr; }
  m->w_mid_split.assign((size_t)2 * NB * 3, nullptr);
  m->w_mid_split_frag.assign((size_t)2 * NB * 3, nullptr);
  // The f32 images first and unconditionally: whatever happens to the f16x3 images below, every batch size of the f32 path
  // must see the NEW weights (the <= 512-row kernels read the fragment-major copy).
  m->loaded = false;
  m->chain_tab_valid = false;  // (the chain's argument table points into the weight arenas)
  drop_frag_weights(m);        // (rebuilt from the new arena by the first chunk that needs them, or by ikf_reserve)
  ikf_status fst = build_rowowner_stream(m, perm_host);
  if (fst != IKF_OK) return fst;
  m->loaded = true;
  m->cl_pause = m->cl_backoff = 0;
  m->cl_clean = 0;
  if (m->precision == 1) {
    ikf_status sst = build_split_weights(m);  // refusal: precision falls back to f32, the handle stays usable
    if (sst != IKF_OK) return sst;
  }
  IKF_HIP(hipDeviceSynchronize());
  m->load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load0).count();
  return IKF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// scratch
// ---------------------------------------------------------------------------------------------------------------
static long long chunk_cap(const ikf_model* m) {  // rows per chunk: 16384 up to width 1024, fewer for wider subnets
  const long long w = m->dims.width > 1024 ? m->dims.width : 1024;
  return kMaxChunkRows * 1024 / w / 128 * 128;
}

static ikf_status ensure_scratch(ikf_model* m, long long rows) {
  const long long cap = chunk_cap(m);
  long long want = rows < cap ? rows : cap;
  want = (want + 127) / 128 * 128;  // the contraction kernels store whole 128-row tiles (no row predicate)
  if (want <= m->chunk_rows) return IKF_OK;
  free_scratch(m);
  IKF_HIP(hipMalloc(&m->xbuf, sizeof(float) * (size_t)want * m->dims.D));
  IKF_HIP(hipMalloc(&m->hA, sizeof(float) * (size_t)want * m->dims.width));
  IKF_HIP(hipMalloc(&m->hB, sizeof(float) * (size_t)want * m->dims.width));
  IKF_HIP(hipMalloc(&m->xbuf2, sizeof(float) * (size_t)want * m->dims.D));
  const size_t slots = (size_t)(fused_max_slots(m->dims.width) > 0 ? fused_max_slots(m->dims.width) : 1);
  // With ONE hidden contraction per subnet (coeff_fn_config 2, e.g. TINY_MODEL_PARAMS) the one-launch subnet head (k_entry_gemm_skinny*) is
  // also the subnet's LAST contraction: the same launch reads the previous subnet's partial sums (pending coupling, every slot of its row
  // tile) and writes its own.  In one buffer that is a write-after-read hazard between the workgroups of a row tile - harmless only while all
  // of them start together; when another process holds CUs a late workgroup read slots a finished sibling had already overwritten (r06: the
  // red two-ranks-on-one-GPU test of round 5, tools/two_tenant_determinism.py).  Such shapes alternate between two sets by subnet parity.
  const size_t pset = slots * (size_t)want * IKF_PSTRIDE;
  const bool two_sets = m->dims.n_hidden == 2;
  IKF_HIP(hipMalloc(&m->pbuf, sizeof(float) * pset * (two_sets ? 2 : 1)));
  m->pbuf_alt = two_sets ? m->pbuf + pset : m->pbuf;
  m->chunk_rows = want;
  return IKF_OK;
}

// exact-IK state: per-pose buffers (active list, solved flags, compaction scratch) and per-row buffers (q, row validity) grow
// independently - the row buffers carry nothing from one retry round to the next, so they may be regrown between rounds
// (right after the round's count has been read, i.e. with the stream idle) without touching the active-pose list.
static ikf_status ensure_exact_poses(ikf_model* m, long long poses) {
  if (poses <= m->exact_poses) return IKF_OK;
  if (m->ex_pose_idx) (void)hipFree(m->ex_pose_idx);
  if (m->ex_block_scratch) (void)hipFree(m->ex_block_scratch);
  if (m->ex_pose_first) (void)hipFree(m->ex_pose_first);
  m->ex_pose_idx = nullptr; m->ex_block_scratch = nullptr; m->ex_pose_first = nullptr;
  m->exact_poses = 0;
  IKF_HIP(hipMalloc(&m->ex_pose_idx, sizeof(int) * (size_t)poses));
  IKF_HIP(hipMalloc(&m->ex_pose_first, sizeof(unsigned) * (size_t)poses));
  IKF_HIP(hipMalloc(&m->ex_block_scratch, sizeof(int) * 2 * (size_t)(compact_blocks(poses) + 1)));
  m->exact_poses = poses;
  return IKF_OK;
}
static ikf_status ensure_exact_rows(ikf_model* m, long long rows) {
  if (rows <= m->exact_rows) return IKF_OK;
  if (m->ex_q) (void)hipFree(m->ex_q);
  if (m->ex_row_valid) (void)hipFree(m->ex_row_valid);
  m->ex_q = nullptr; m->ex_row_valid = nullptr;
  m->exact_rows = 0;
  IKF_HIP(hipMalloc(&m->ex_q, sizeof(float) * (size_t)rows * m->dims.ndof));
  IKF_HIP(hipMalloc(&m->ex_row_valid, (size_t)rows));
  m->exact_rows = rows;
  return IKF_OK;
}
static ikf_status ensure_exact(ikf_model* m, long long poses, long long rows) {
  ikf_status st = ensure_exact_poses(m, poses);
  return st != IKF_OK ? st : ensure_exact_rows(m, rows);
}
// Worst-case row state (every pose unsolved in the round with the largest repeat count) is reserved up front only while it is
// small (ikf_set_exact_upfront_rows, default 32 Mi rows); beyond that a call starts with round 0's rows and grows per round from the measured survivor count, so a
// large n with a big last-round repeat but few survivors neither allocates nor is rejected for the worst case.

static bool rowowner_allowed_fwd(const ikf_model* m);
static bool cluster_allowed_now(const ikf_model* m);
extern "C" ikf_status ikf_reserve(ikf_model* m, int64_t max_rows) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_reserve: null model");
  if (max_rows < 1) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_reserve: max_rows must be positive");
  IKF_ON_DEVICE(m)
  ikf_status st = ensure_scratch(m, max_rows);
  // the small-batch per-layer kernels' weight image (201 MB at the released shape): also on a handle whose small batches normally take the
  // cluster form - during a back-off pause (another process held CUs) they run these kernels, and that is the worst moment for a hipMalloc,
  // 48 pack launches and a device-wide synchronisation inside a call
  if (st == IKF_OK && m->loaded && !m->wfrag_built) st = build_frag_weights(m);
  return st;
}

extern "C" int ikf_probes_build(void) {
#ifdef IKF_PROBES
  return 1;
#else
  return 0;
#endif
}

extern "C" ikf_status ikf_set_gemm_variant(ikf_model* m, int variant) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_gemm_variant: null model");
#ifndef IKF_PROBES
  if (variant == 121 || variant == 163 || variant == 164 || variant == 171 || variant == 106 || variant == 108)
    return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_gemm_variant(" + std::to_string(variant) + "): a priced-and-rejected form of rounds 2 - 3 (in-launch entry phase, "
                "one-launch chain for <= 128 rows, tile configurations 5 / 7 / 11) - compiled only into the probes library "
                "(ikflow_amd/lib/libikflow_amd_probes.so, python -m ikflow_amd.build --probes)");
#endif
  if (variant >= 110 && variant <= 112) {  // small-batch one-launch form (entry + first contraction): off / auto / forced
    m->fuse_entry = variant - 110;
    return IKF_OK;
  }
  if (variant == 152 || variant == 153) {  // 16-row kernels: whole operand stream up front off / on
    m->tune = variant == 153 ? (m->tune | IKF_TUNE_DEEP16) : (m->tune & ~IKF_TUNE_DEEP16);
    return IKF_OK;
  }
  if (variant == 150 || variant == 151) {  // <= 128 rows on 16x32 tiles (v_mfma_f32_16x16x4_f32): off / on
    m->tune = variant == 151 ? (m->tune | IKF_TUNE_ROWS16) : (m->tune & ~IKF_TUNE_ROWS16);
    return IKF_OK;
  }
  if (variant >= 130 && variant <= 134) {  // write-through activation stores: none / contractions / entry kernel / both / by batch size
    m->wt_stores = variant == 134 ? -1 : variant - 130;
    return IKF_OK;
  }
  if (variant == 192 || variant == 193) {  // cluster form, G = 2 .. 16: hand-over by epoch words / by parity-tagged payload (default)
    m->cl_tagged = variant - 192;
    return IKF_OK;
  }
  if (variant == 191) {  // tests of the placement check: the next XCD-local cluster launch is told that workgroup 0 sits on another XCD
    m->cl_far_next = 1;
    return IKF_OK;
  }
  if (variant == 189 || variant == 190) {  // cluster form, G = 4 / 8 / 16: a row tile's members spread over the XCDs / on one XCD (default)
    if (variant == 190 && m->cl_census_ok == 0)
      return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_gemm_variant(190): on this device workgroups b and b + 8 k of a grid do not share an XCD");
    m->cl_local = variant - 189;
    return IKF_OK;
  }
  if (variant == 188) {  // tests of the repair path: the next cluster launch is one workgroup short
    m->cl_drop_next = 1;
    return IKF_OK;
  }
  if (variant >= 185 && variant <= 187) {  // cluster form for the rows below a round: never / by batch size / whenever the grid fits
    m->cl_mode = variant == 185 ? 0 : (variant == 186 ? -1 : 1);
    return IKF_OK;
  }
  if (variant >= 180 && variant <= 182) {  // row-owner form (one launch per call, rows resident on chip): never / by batch size / always
    if (variant == 182 && m->loaded && m->ro_stream == nullptr)
      return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_gemm_variant(182): the row-owner kernel needs coeff_fn_internal_size 1024 and coeff_fn_config 3");
    m->ro_mode = variant == 180 ? 0 : (variant == 181 ? -1 : 1);
    return IKF_OK;
  }
  if (variant == 170 || variant == 171) {  // <= 128 rows: the whole subnet chain in one launch (XCD-local hand-over): off / on
    m->chain_mode = variant - 170;
    return IKF_OK;
  }
  if (variant == 120 || variant == 121) {  // next subnet's entry phase in the tail of the last hidden contraction: off / on
    m->fuse_tail = variant - 120;
    return IKF_OK;
  }
  if (variant == 162 || variant == 163) {  // 129 .. 256 rows on 32x32 tiles built from 16x16x4 MFMAs: off (default) / on
    m->tune = variant == 163 ? (m->tune | IKF_TUNE_ROWS32_V2) : (m->tune & ~IKF_TUNE_ROWS32_V2);
    return IKF_OK;
  }
  if (variant == 164) {  // ... forced (tile config 11)
    m->gemm_variant = 100;
    m->tile_cfg = 11;
    return IKF_OK;
  }
  if (variant == 160 || variant == 161) {  // fused pipeline with the 16x32 / 16x16 small-batch tiles forced (tile config 9 / 10)
    m->gemm_variant = 100;
    m->tile_cfg = variant - 151;
    return IKF_OK;
  }
  if (variant == 158 || variant == 159) {  // <= 64 rows on 16x16 tiles: off / on
    m->tune = variant == 159 ? (m->tune | IKF_TUNE_TILES16) : (m->tune & ~IKF_TUNE_TILES16);
    return IKF_OK;
  }
  if (variant >= 100 && variant <= 108) {  // fused pipeline; 100 = tile by batch size, 101..108 = tile config 0..7
    m->gemm_variant = 100;
    m->tile_cfg = variant - 101;
    return IKF_OK;
  }
  if (variant < -1 || variant >= gemm_variant_count())
    return fail(IKF_ERR_BAD_ARGUMENT, "unknown gemm variant (-1 auto, 0..N-1 unfused tile shapes, 100 fused by batch size, 101..108 fused with tile configuration 0..7, 110 / 111 / 112 one-launch small-batch form off / auto / forced, 120 / 121 in-launch entry phase off / on, 130..134 write-through activation stores none / contractions / entry / both / by batch size, 150 / 151 16-row tiles for <= 128 rows off / on, 152 / 153 their whole-stream prefetch off / on, 158 / 159 16 x 16 tiles for <= 64 rows off / on, 160 / 161 / 164 small-batch tile configurations 9 / 10 / 11 forced, 162 / 163 configuration 11 for 129..256 rows off / on, 170 / 171 one-launch subnet chain for <= 128 rows off / on, 180 / 181 / 182 row-owner launch never / by plan / always, 185 / 186 / 187 cluster form never / by plan / whenever the grid fits, 188 / 191 tests of its repair paths, 189 / 190 its members spread / on one XCD, 192 / 193 its hand-over by epoch words / tagged payload; see include/ikflow_amd_debug.h)");
  m->gemm_variant = variant;
  m->tile_cfg = -1;
  return IKF_OK;
}

static int pick_variant(const ikf_model* m, long long rows) {
  if (m->gemm_variant >= 0) return m->gemm_variant;
  const int W = m->dims.width;
  // fill the 256 CUs: 128x128 tiles when that already gives >= 256 tiles, smaller tiles for smaller batches
  const long long t128 = ((rows + 127) / 128) * (W / 128);
  if (t128 >= 256) return 0;
  const long long t12864 = ((rows + 127) / 128) * (W / 64);
  if (t12864 >= 256) return 2;
  return 4;
}

// ---------------------------------------------------------------------------------------------------------------
// flow inverse pass over `rows` rows (chunked); replaces nn_model(latent, c=cond, rev=True) + slice + clamp
// ---------------------------------------------------------------------------------------------------------------
static const float* chain_lo(const ikf_model* m) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(m->d_chain) + offsetof(Chain, lo));
}
static const float* chain_hi(const ikf_model* m) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(m->d_chain) + offsetof(Chain, hi));
}

// the small-batch tile configurations take their W fragments straight from the fragment-major image (narrow models pick them for
// larger batches too: the choice goes by tile count, not by rows)
static bool cfg_reads_frag_image(int cfg) {
  return cfg == fused_skinny_cfg() || cfg == fused_skinny32_cfg() || cfg == fused_skinny16_cfg() || cfg == fused_skinny16x16_cfg() ||
         cfg == fused_skinny32v2_cfg();
}
// fragment-major image of hidden layer l of subnet si, or null (not built for this width / not loaded)
static const float* frag_image(const ikf_model* m, int si, int l) {
  const size_t i = (size_t)si * 3 + l;
  return i < m->w_mid_frag.size() ? m->w_mid_frag[i] : nullptr;
}

static bool fused_ok(const ikf_model* m) {
  const FlowDims& d = m->dims;
  if (m->gemm_variant >= 0 && m->gemm_variant != 100) return false;
  return d.n_hidden >= 2 && fused_pick_cfg(128, d.width) >= 0 && d.D <= 16 && 2 * d.L2 <= 16 &&
         d.L1 + d.n_pose >= 8 && d.L2 + d.n_pose <= 15;
}

// three kernels per subnet (flow_fused.hip): entry (pending coupling + first Linear), hidden contraction(s), the last
// of which reduces the last Linear to partial sums; one finalize kernel after the last subnet
// ---- <= 128 rows: the subnet chain in one launch (k_flow_chain16) + the finalize kernel
// One-time check per handle that a chain-shaped launch gets 32 workgroups on each of 8 XCDs (the kernel would notice and give up;
// this keeps a device in another partition mode, or with masked CUs, from ever trying).  Synchronous: first use only.
static ikf_status chain_census(ikf_model* m, hipStream_t s) {
  if (m->chain_census >= 0) return IKF_OK;
  m->chain_census = 0;
  hipDeviceProp_t prop{};
  IKF_HIP(hipGetDeviceProperties(&prop, m->device));
  if (prop.multiProcessorCount != IKF_CHAIN_XCDS * IKF_CHAIN_PER_XCD) return IKF_OK;
  unsigned* d_out = nullptr;
  IKF_HIP(hipMalloc(&d_out, sizeof(unsigned) * 256));
  hipError_t e = launch_xcd_census(d_out, s);
  unsigned h[256];
  if (e == hipSuccess) e = hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d_out);
  IKF_HIP(e);
  int per[16] = {};
  for (int b = 0; b < 256; ++b) per[h[b] & 15]++;
  bool ok = true;
  for (int x = 0; x < 16; ++x) ok = ok && per[x] == (x < IKF_CHAIN_XCDS ? IKF_CHAIN_PER_XCD : 0);
  m->chain_census = ok ? 1 : 0;
  return IKF_OK;
}
static bool chain_usable(const ikf_model* m, long long nr) {
  const FlowDims& d = m->dims;
  if (m->chain_mode == 0 || m->chain_census == 0 || m->tile_cfg >= 0 || m->fuse_entry != 1 || m->fuse_tail != 0 || m->prof_on) return false;
  if ((m->tune & (IKF_TUNE_ROWS16 | IKF_TUNE_DEEP16)) != (IKF_TUNE_ROWS16 | IKF_TUNE_DEEP16)) return false;
  if (!flow_chain16_ok(nr, d.width, d.D, 2 * (d.L1 > d.L2 ? d.L1 : d.L2), d.n_hidden)) return false;
  for (int si = 0; si < 2 * m->desc.nb_nodes; ++si)
    if (m->subnets[si].n_x + d.n_pose > 15) return false;
  return true;
}
static ikf_status chain_table(ikf_model* m) {
  if (m->chain_tab_valid) return IKF_OK;
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes;
  const long long rows_pad = m->chunk_rows;
  std::vector<ChainSubnet> tab((size_t)2 * NB);
  float* xb[2] = {m->xbuf, m->xbuf2};
  PendingCoupling pend{};
  const float* x_src = nullptr;
  for (int sidx = 0; sidx < 2 * NB; ++sidx) {
    const int b = NB - 1 - sidx / 2, which = 1 + (sidx & 1);
    const int si = 2 * b + which - 1;
    const SubnetWeights& w = m->subnets[si];
    if (frag_image(m, si, 0) == nullptr || frag_image(m, si, 1) == nullptr) return fail(IKF_ERR_HIP, "chain_table: no fragment image");
    ChainSubnet& c = tab[sidx];
    memset(&c, 0, sizeof(c));
    EntryArgs& e = c.e;
    e.pend = pend;
    e.x_src = x_src; e.x_dst = xb[sidx & 1];
    e.D = d.D; e.L1 = d.L1; e.clamp = d.clamp;
    e.x_off = (which == 1) ? 0 : d.L1; e.n_x = w.n_x;
    e.w1t = w.w_first_t; e.w1soft = w.w_soft; e.b1 = w.b_first;
    e.width = d.width; e.slope = d.slope; e.h_out = m->hA;
    c.n_in = w.n_x + d.n_pose;
    for (int l = 0; l < 2; ++l) {
      FusedGemmArgs& g = c.g[l];
      g.N = d.width; g.K = d.width; g.slope = d.slope; g.tune = m->tune;
      g.w_last = w.w_last; g.n_out = w.n_out; g.P_out = m->pbuf; g.p_slot_stride = rows_pad * IKF_PSTRIDE;
      g.W = w.w_mid[l]; g.bias = w.b_mid[l]; g.Wf = frag_image(m, si, l);
    }
    c.g[0].A = m->hA; c.g[0].C = m->hB;   // (the head keeps the first Linear's output in registers: A is unused)
    c.g[1].A = m->hB; c.g[1].C = nullptr;
    pend = PendingCoupling{};
    pend.P = m->pbuf;
    pend.b_last = w.b_last;
    pend.perm_inv = m->d_perm_inv + (size_t)b * d.D;
    pend.slot_stride = rows_pad * IKF_PSTRIDE;
    pend.slots = fused_slots(fused_skinny16_cfg(), d.width);
    pend.which = which;
    pend.n_out = w.n_out;
    x_src = e.x_dst;
  }
  IKF_HIP(hipMemcpy(m->d_chain_tab, tab.data(), sizeof(ChainSubnet) * tab.size(), hipMemcpyHostToDevice));
  m->chain_tab_valid = true;
  return IKF_OK;
}
static ikf_status run_flow_chunk_chain(ikf_model* m, const PoseSource& ps, const float* d_latent, long long r0, long long nr,
                                       int clamp_limits, float* d_q_out, hipStream_t s) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes;
  ikf_status st = chain_table(m);
  if (st != IKF_OK) return st;
  ChainCall call{};
  call.ps = ps; call.x0 = d_latent + (size_t)r0 * d.D; call.row0 = r0; call.M = (int)nr;
  ChainSync cs{};
  cs.ctl = m->d_chain_ctl; cs.give_up = m->h_give_up; cs.row_tiles = (int)((nr + 15) / 16);
  IKF_HIP(launch_flow_chain16(m->d_chain_tab, 2 * NB, call, cs, d.width, s));
  const SubnetWeights& w = m->subnets[1];  // the last subnet in execution order: block 0, s2
  PendingCoupling pend{};
  pend.P = m->pbuf; pend.b_last = w.b_last; pend.perm_inv = m->d_perm_inv;
  pend.slot_stride = m->chunk_rows * IKF_PSTRIDE; pend.slots = fused_slots(fused_skinny16_cfg(), d.width);
  pend.which = 2; pend.n_out = w.n_out;
  float* xb[2] = {m->xbuf, m->xbuf2};
  FinalizeArgs f{};
  f.pend = pend; f.x_src = xb[(2 * NB - 1) & 1]; f.M = (int)nr; f.D = d.D; f.L1 = d.L1; f.ndof = d.ndof; f.clamp = d.clamp;
  f.M_inv = m->d_Minv; f.b_lin = m->d_blin; f.lo = chain_lo(m); f.hi = chain_hi(m);
  f.clamp_limits = clamp_limits; f.sigmoid = m->desc.sigmoid_on_output ? 1 : 0; f.q_out = d_q_out + (size_t)r0 * d.ndof;
  IKF_HIP(launch_flow_finalize(f, s));
  return IKF_OK;
}

static ikf_status run_flow_chunk_fused(ikf_model* m, const PoseSource& ps, const float* d_latent, long long r0,
                                       long long nr, int clamp_limits, float* d_q_out, hipStream_t s) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes;
  const long long rows_pad = m->chunk_rows;
  if (!m->wfrag_built && (chain_usable(m, nr) || cfg_reads_frag_image((m->tile_cfg >= 0) ? m->tile_cfg : fused_pick_cfg(nr, d.width, m->tune)))) {
    // the first chunk on this path that reads the fragment-major image since the weights were loaded (or ikf_reserve built it already)
    ikf_status fst = build_frag_weights(m);
    if (fst != IKF_OK) return fst;
  }
  if (chain_usable(m, nr)) {
    if (m->chain_census < 0) {
      ikf_status cst = chain_census(m, s);
      if (cst != IKF_OK) return cst;
    }
    if (m->chain_census == 1) return run_flow_chunk_chain(m, ps, d_latent, r0, nr, clamp_limits, d_q_out, s);
  }
  const int cfg = (m->tile_cfg >= 0) ? m->tile_cfg : fused_pick_cfg(nr, d.width, m->tune);
  // f16x3 mode: its own tile choice; the partial-sum slots follow the kernel that writes them
  // (f16x3 mode, batches that pick the 16-row f32 tiles - <= 128 rows: the exact-f32 kernels are the faster ones there since round 3,
  // 0.43 against 0.46 ms per call, so the mode steps aside; a forced tile configuration keeps the split kernels)
  // (also when such a tile configuration is FORCED - ikf_set_gemm_variant 160 / 161 / 164: the split kernels number their tiles differently)
  const bool split = (m->precision == 1) && m->split_arena != nullptr && !(cfg == fused_skinny16_cfg() || cfg == fused_skinny16x16_cfg() || cfg == fused_skinny32v2_cfg());
  int scfg = -1;
  if (split) {
    scfg = (m->tile_cfg >= 0) ? m->tile_cfg : split_pick_cfg(nr, d.width);
    if (split_cfg_needs_frag(scfg) && m->split_frag_arena == nullptr) scfg = 3;
  }
  const int slots = split ? split_slots(scfg, d.width) : fused_slots(cfg, d.width);
  // In-launch hand-over (TailSync): the last hidden contraction of subnet s also runs subnet s+1's entry phase, so only the
  // first subnet of the call has an entry launch.  Taken when every workgroup of that contraction is resident at once.
  const bool tail = !split && m->fuse_tail != 0 && d.n_hidden >= 2 && NB * 2 > 1 &&
                    fused_tail_ok(cfg, nr, d.width, d.D, 2 * (d.L1 > d.L2 ? d.L1 : d.L2)) &&
                    (cfg != fused_skinny_cfg() || frag_image(m, 0, d.n_hidden - 2) != nullptr);
  int tails_done = 0;
  bool entry_done = false;  // this subnet's entry phase already ran in the previous subnet's last launch
  PendingCoupling pend{};
  pend.P = nullptr;
  const float* x_src = d_latent + (size_t)r0 * d.D;
  float* xb[2] = {m->xbuf, m->xbuf2};
  float* const pb[2] = {m->pbuf, m->pbuf_alt};   // partial sums of even / odd subnets (the same buffer unless a launch both reads and writes them)
  auto entry_args = [&](int sidx, const PendingCoupling& pc, const float* xs) {
    const int b = NB - 1 - sidx / 2, which = 1 + (sidx & 1);
    const SubnetWeights& w = m->subnets[2 * b + which - 1];
    EntryArgs e{};
    e.pend = pc;
    e.x_src = xs; e.x_dst = xb[sidx & 1];
    e.M = (int)nr; e.D = d.D; e.L1 = d.L1; e.clamp = d.clamp;
    e.x_off = (which == 1) ? 0 : d.L1; e.n_x = w.n_x;
    e.ps = ps; e.row0 = r0;
    e.w1t = w.w_first_t; e.w1soft = w.w_soft; e.b1 = w.b_first;
    e.width = d.width; e.slope = d.slope; e.h_out = m->hA; e.split_out = split ? 1 : 0;
    e.split_flag = split ? m->d_split_flag : nullptr;
    e.wt_stores = m->wt_stores < 0 ? (nr <= 2048 ? 1 : 0) : (m->wt_stores >> 1) & 1;
    return e;
  };
  for (int sidx = 0; sidx < 2 * NB; ++sidx) {
    const int b = NB - 1 - sidx / 2, which = 1 + (sidx & 1);
    const SubnetWeights& w = m->subnets[2 * b + which - 1];
    EntryArgs e = entry_args(sidx, pend, x_src);
    if (tail && sidx == 0) { e.zero_words = m->d_arrive; e.n_zero = kArriveWords; }
    FusedGemmArgs g{};
    g.M = (int)nr; g.N = d.width; g.K = d.width; g.slope = d.slope;
    g.wt_stores = m->wt_stores < 0 ? 1 : (m->wt_stores & 1);
    g.tune = m->tune;
    g.w_last = w.w_last; g.n_out = w.n_out; g.P_out = pb[sidx & 1]; g.p_slot_stride = rows_pad * IKF_PSTRIDE;
    const int n_mid = d.n_hidden - 1;
    // small batches: the entry kernel and the first hidden contraction run as one launch (k_entry_gemm_skinny).  In the
    // chain it pays with the 32x32 tiles (129 .. 256 rows: 0.56 -> 0.53 ms per call) and the 16-row tiles (<= 128 rows, where it
    // exists only in this form worth having: r03); with the 32x64 tiles (257..512 rows) the
    // one launch takes as long as the two it replaces (18.4 us against 5.5 + 13.0), so those keep the two-launch form
    // unless it is forced (fuse_entry == 2, ikf_set_gemm_variant 112)
    const bool one_launch = !tail && !split &&
                            (m->fuse_entry == 2 || (m->fuse_entry == 1 && (cfg == fused_skinny32_cfg() || cfg == fused_skinny16_cfg() || cfg == fused_skinny16x16_cfg() || cfg == fused_skinny32v2_cfg()))) &&
                            entry_gemm_ok(cfg, nr, d.width, d.D, pend.P ? pend.n_out : 0) &&
                            frag_image(m, 2 * b + which - 1, 0) != nullptr;
    if (!one_launch && !entry_done) IKF_HIP(launch_subnet_entry(w.n_x + d.n_pose, e, s));
    entry_done = false;
    // the pending coupling this subnet leaves behind (its last Linear exists only as partial sums)
    PendingCoupling mine{};
    mine.P = pb[sidx & 1];
    mine.b_last = w.b_last;
    mine.perm_inv = m->d_perm_inv + (size_t)b * d.D;
    mine.slot_stride = rows_pad * IKF_PSTRIDE;
    mine.slots = slots;
    mine.which = which;
    mine.n_out = w.n_out;
    float* cur = m->hA;
    float* nxt = m->hB;
    for (int l = 0; l < n_mid; ++l) {
      const bool last = (l == n_mid - 1);
      IKF_HIP(prof_mark(m, s));
      if (split) {
        SplitGemmArgs sg{};
        sg.A = cur; sg.C = last ? nullptr : nxt; sg.W = m->w_mid_split[(size_t)(2 * b + which - 1) * 3 + l];
        sg.Wf = m->w_mid_split_frag.empty() ? nullptr : m->w_mid_split_frag[(size_t)(2 * b + which - 1) * 3 + l];
        sg.bias = w.b_mid[l]; sg.M = (int)nr; sg.N = d.width; sg.K = d.width; sg.slope = d.slope;
        sg.w_last = w.w_last; sg.n_out = w.n_out; sg.P_out = pb[sidx & 1]; sg.p_slot_stride = rows_pad * IKF_PSTRIDE;
        sg.flag = m->d_split_flag;
        IKF_HIP(launch_split_gemm(last, scfg, sg, s));
      } else {
        g.A = cur; g.C = last ? nullptr : nxt; g.W = w.w_mid[l]; g.bias = w.b_mid[l];
        g.Wf = frag_image(m, 2 * b + which - 1, l);
        if (l == 0 && one_launch) IKF_HIP(launch_entry_gemm(w.n_x + d.n_pose, last, cfg, e, g, s));
        else if (last && tail && sidx + 1 < 2 * NB) {
          // subnet sidx+1's entry phase rides in this launch's tail: its pending coupling is what this launch produces, its
          // state source is the state this subnet's entry phase published
          const int b2 = NB - 1 - (sidx + 1) / 2, which2 = 1 + ((sidx + 1) & 1);
          const EntryArgs e2 = entry_args(sidx + 1, mine, e.x_dst);
          TailSync ts{};
          ts.arrive = m->d_arrive;
          ts.target = (unsigned)(tails_done + 1) * (unsigned)fused_tail_col_tiles(cfg, d.width);
          ts.give_up = m->h_give_up;
          ts.n_in = m->subnets[2 * b2 + which2 - 1].n_x + d.n_pose;
          IKF_HIP(launch_flow_gemm_tail(cfg, g, e2, ts, s));
          ++tails_done;
          entry_done = true;
        } else IKF_HIP(launch_flow_gemm(last, cfg, g, s));
      }
      IKF_HIP(prof_mark(m, s));
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
    pend = mine;
    x_src = e.x_dst;
  }
  FinalizeArgs f{};
  f.pend = pend; f.x_src = x_src; f.M = (int)nr; f.D = d.D; f.L1 = d.L1; f.ndof = d.ndof; f.clamp = d.clamp;
  f.M_inv = m->d_Minv; f.b_lin = m->d_blin; f.lo = chain_lo(m); f.hi = chain_hi(m);
  f.clamp_limits = clamp_limits; f.sigmoid = m->desc.sigmoid_on_output ? 1 : 0; f.q_out = d_q_out + (size_t)r0 * d.ndof;
  IKF_HIP(launch_flow_finalize(f, s));
  return IKF_OK;
}

// four-plus kernels per subnet (flow_kernels.hip): first Linear, hidden contractions, last Linear + coupling
static ikf_status run_flow_chunk_unfused(ikf_model* m, const PoseSource& ps, const float* d_latent, long long r0,
                                         long long nr, int clamp_limits, float* d_q_out, hipStream_t s) {
  const FlowDims& d = m->dims;
  const int NB = m->desc.nb_nodes;
  const int variant = pick_variant(m, nr);
  for (int b = NB - 1; b >= 0; --b) {
    const float* x_in = (b == NB - 1) ? d_latent + (size_t)r0 * d.D : m->xbuf;
    for (int which = 1; which <= 2; ++which) {
      const SubnetWeights& w = m->subnets[2 * b + which - 1];
      const float* x_src = (which == 1) ? x_in : m->xbuf;
      IKF_HIP(launch_first_layer(w, d, x_src, which == 1 ? 0 : d.L1, ps, r0, nr, m->hA, s));
      float* cur = m->hA;
      float* nxt = m->hB;
      for (int l = 0; l < d.n_hidden - 1; ++l) {
        IKF_HIP(prof_mark(m, s));
        IKF_HIP(launch_gemm_lrelu(variant, cur, w.w_mid[l], w.b_mid[l], nxt, nr, d.width, d.width, d.slope, s));
        IKF_HIP(prof_mark(m, s));
        float* tmp = cur; cur = nxt; nxt = tmp;
      }
      CouplingArgs ca{};
      ca.x_in = x_in;
      ca.x_out = m->xbuf;
      ca.perm_inv = m->d_perm_inv + (size_t)b * d.D;
      ca.M_inv = m->d_Minv;
      ca.b_lin = m->d_blin;
      ca.lo = chain_lo(m);
      ca.hi = chain_hi(m);
      ca.q_out = d_q_out + (size_t)r0 * d.ndof;
      ca.which = which;
      ca.is_final = (b == 0 && which == 2) ? 1 : 0;
      ca.clamp_limits = clamp_limits;
      ca.sigmoid = m->desc.sigmoid_on_output ? 1 : 0;
      IKF_HIP(launch_last_layer_coupling(w, d, cur, ca, nr, s));
    }
  }
  return IKF_OK;
}

// ---- which form runs which rows of a batch -------------------------------------------------------------------------------------------
// Three forms compute the same function: the per-layer kernels (any shape), the row-owner launch (a round of CUs x 16 rows costs the same
// whatever part of it is used) and the cluster form (G = 8 / 4 / 2: <= 512 / 1024 / 2048 rows at a fixed cost each).  A batch is cut into
// consecutive chunks by the cheapest plan under the measured costs of the released 12-block shape on 256 CUs (ms per launch,
// tools/rowowner_ab.py, profiles/r04_rowowner_ab.jsonl) - the ratios, not the absolute values, decide, and they hold for any depth:
//   row-owner round 2.82;  cluster 0.285 / 0.335 / 0.485 / 0.82 / 1.51 for G = 32 / 16 / 8 / 4 / 2 (<= 128 / 256 / 512 / 1024 / 2048 rows; r05);
//   per-layer 0.272 / 0.305 / 0.316 / 0.367 / 0.52 / 0.71 / 1.04 / 1.75 / 2.56 / 2.62 / 3.20 up to 1 / 16 / 64 / 128 / 256 / 512 / 1024 /
//   2048 / 2560 / 3072 / 4096 rows (+ 0.10 beside the resident-row forms: another weight image, see plan_tail);  + 0.01 per extra chunk.
// e.g. 1 .. 128 -> cluster 32; 200 -> cluster 16; 512 -> cluster 8; 600 -> cluster 4; 1536 -> cluster 4 (1024) + cluster 8 (512); 2304 -> cluster 2 (2048) + per-layer (256);
// 3400 -> one row-owner round; 4096 k + r -> k rounds in one row-owner launch + the plan of r.
struct FlowChunk {
  int form;         // 0 per-layer, 1 row-owner, 2 / 4 / 8 cluster members
  long long rows;
};
static bool rowowner_allowed(const ikf_model* m) {
  return m->ro_stream != nullptr && m->ro_mode != 0 && m->precision == 0 && m->loaded &&
         (m->ro_mode == 1 || (m->gemm_variant < 0 && m->tile_cfg < 0 && m->fuse_tail == 0));
}
// A wait of a tagged + XCD-local launch ran out: was it placement?  Every member of that launch wrote (launch number << 8 | XCC_ID) into its word
// at its start (flow_rowowner.hip); launches queued behind the one that gave up returned before they wrote anything, so the largest launch number
// found is the failed launch's.  Members of one row tile that ran it on different XCDs = the XCD-local form was used where it must not be.
// Rare path (a 5 ms stall has just happened): one device synchronisation and a copy of a few KB.
static bool cluster_tagged_local_misplaced(ikf_model* m) {
  DeviceGuard guard(m->device);
  if (guard.err != hipSuccess || m->cl_tl_words == nullptr) return false;
  const int n_rt = m->cl_tl_nrt, G = m->cl_tl_G;
  std::vector<unsigned> w((size_t)n_rt * 32, 0xffffffffu);
  if (hipDeviceSynchronize() != hipSuccess) return false;
  if (hipMemcpy(w.data(), m->cl_tl_words, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return false;
  unsigned seq = 0;
  bool any = false;
  for (int rt = 0; rt < n_rt; ++rt)
    for (int j = 0; j < G; ++j) {
      const unsigned v = w[(size_t)rt * 32 + j];
      if (v == 0xffffffffu) continue;
      if (!any || (((v >> 8) - seq) & 0xffffffu) < 0x800000u) seq = v >> 8;   // (24-bit sequence numbers wrap)
      any = true;
    }
  if (!any) return false;
  for (int rt = 0; rt < n_rt; ++rt) {
    int first = -1;
    for (int j = 0; j < G; ++j) {
      const unsigned v = w[(size_t)rt * 32 + j];
      if (v == 0xffffffffu || (v >> 8) != seq) continue;
      if (first < 0) first = (int)(v & 0xffu);
      else if ((int)(v & 0xffu) != first) return true;
    }
  }
  return false;
}
static const long long kClusterFirstPause = 16, kClusterMaxPause = 65536;
static const int kClusterCleanStreak = 64;
// folds a pending give-up word into the handle's state (no side effect otherwise)
static void cluster_fold_give_up(ikf_model* m) {
  if (!m->h_cl_give_up || *m->h_cl_give_up == 0) return;
  // an earlier call's cluster launch gave up (its rows were recomputed by the repair launch)
  int why = *m->h_cl_give_up;
  *m->h_cl_give_up = 0;
  if (why == 1 && m->cl_local != 0 && m->cl_tl_nrt > 0 && cluster_tagged_local_misplaced(m)) why = 2;
  ++m->cl_repairs;
  m->cl_clean = 0;
  m->cl_tag_dirty = true;                        // (whichever hand-over it was: the tagged buffers are re-created before their next use)
  if (why == 2) m->cl_local = 0;                 // a member of the XCD-local form met a peer on another XCD: back to the spread form
  else {                                         // a wait ran out: somebody else held CUs - sit out, twice as long as the last time
    m->cl_backoff = m->cl_backoff == 0 ? kClusterFirstPause : (m->cl_backoff * 2 < kClusterMaxPause ? m->cl_backoff * 2 : kClusterMaxPause);
    m->cl_pause = m->cl_backoff;
  }
}
static bool cluster_allowed_now(const ikf_model* m) {
  if (m->ro_stream == nullptr || m->cl_mode == 0 || m->precision != 0 || !m->loaded || m->cl_pause > 0) return false;
  return m->cl_mode == 1 || (m->gemm_variant < 0 && m->tile_cfg < 0 && m->fuse_tail == 0 && m->ro_mode != 0);
}
// the planner's question, asked once per plan: counts the pause down and the clean streak up
static bool cluster_allowed(ikf_model* m) {
  cluster_fold_give_up(m);
  if (m->cl_used_last && m->cl_pause == 0 && m->cl_backoff != 0 && ++m->cl_clean >= kClusterCleanStreak) m->cl_backoff = 0;
  m->cl_used_last = false;
  if (m->cl_pause > 0) {
    --m->cl_pause;
    return false;
  }
  return cluster_allowed_now(m);
}
static bool rowowner_allowed_fwd(const ikf_model* m) { return rowowner_allowed(m); }
static double per_layer_cost(long long rows_on_256) {
  static const struct { long long rows; double ms; } t[] = {{1, 0.272}, {16, 0.305}, {64, 0.316}, {128, 0.367}, {256, 0.52}, {512, 0.71}, {1024, 1.04},
                                                            {2048, 1.75}, {2560, 2.56}, {3072, 2.62}, {4096, 3.20}};
  for (const auto& e : t)
    if (rows_on_256 <= e.rows) return e.ms;
  return 3.20 * (double)rows_on_256 / 4096.0;
}
// cheapest plan for `rows` rows below one row-owner round; returns its cost, appends its chunks.  Only the forms of G <= 8 (>= 512 rows
// per launch) are taken as a full launch in front of a rest - the small forms only for a whole (rest of a) tail - and results are memoised
// by the remaining row count: a handful of states, a few microseconds per call (an unbounded search over 128-row pieces is exponential).
struct TailPlan {
  double cost;
  std::vector<FlowChunk> chunks;
};
// The per-layer kernels read their own weight images (2 x 203 MB beside the row-owner stream's 203 MB; the Infinity Cache holds 256 MB): next
// to a chunk of another form (mixed), or on a handle whose other calls use the cluster form (cl), they find the cache holding the other
// image and leave it holding theirs - measured + 0.07 ... 0.1 ms on a 513- / 1025- / 2049-row call whose last row went to them.  They are
// charged for it, so that where the resident-row forms are allowed EVERY size reads one image (1 row alone: 0.273 against cluster32's 0.277;
// an exact-IK call on one pose runs rounds of 1, 3, 10 rows); they remain what a handle without those forms, or another shape, runs.
static const TailPlan& plan_tail(long long rows, long long round, bool ro, bool cl, bool mixed, std::unordered_map<long long, TailPlan>& memo) {
  auto it = memo.find(rows);
  if (it != memo.end()) return it->second;
  TailPlan best{0.0, {}};
  if (rows > 0) {
    const long long on256 = rows * 4096 / round;   // the cost tables are in rows of a 256-CU chip
    best = TailPlan{per_layer_cost(on256) + ((mixed || cl) ? 0.10 : 0.0), {{0, rows}}};
    if (ro && 2.82 < best.cost) best = TailPlan{2.82, {{1, rows}}};
    if (cl) {
      static const struct { int G; double ms; } forms[] = {{32, 0.285}, {16, 0.335}, {8, 0.485}, {4, 0.82}, {2, 1.51}};   // (r05: tagged hand-over for G <= 16)
      for (const auto& f : forms) {
        const long long cap = (round / IKF_RO_ROWS) / f.G * IKF_RO_ROWS;   // rows of a full grid of this form: whole tiles, at most one workgroup per CU
        if (cap <= 0) continue;
        if (rows <= cap) {                   // the whole tail in one launch of this form
          if (f.ms < best.cost) best = TailPlan{f.ms, {{f.G, rows}}};
        } else if (f.G <= 8) {               // a full launch of this form, then the plan of what is left
          const TailPlan rest = plan_tail(rows - cap, round, ro, cl, true, memo);   // (by value: the map may rehash)
          const double c = f.ms + 0.01 + rest.cost;
          if (c < best.cost) {
            best = TailPlan{c, {{f.G, cap}}};
            best.chunks.insert(best.chunks.end(), rest.chunks.begin(), rest.chunks.end());
          }
        }
      }
    }
  }
  return memo.emplace(rows, std::move(best)).first->second;
}
// (host logic only - no device, no handle: ikf_plan_describe_for runs it in the CPU tests.  ro / cl: the form may be used at all;
// ro_mode / cl_mode 1: forced; ro_min_tail >= 0: the probes' explicit threshold for the last partial round)
static std::vector<FlowChunk> plan_rows(long long rows, int n_cu, bool ro, bool cl, int ro_mode, int cl_mode, long long ro_min_tail) {
  std::vector<FlowChunk> plan;
  if (rows <= 0 || n_cu <= 0) return plan;
  const long long round = (long long)n_cu * IKF_RO_ROWS;
  if (ro && ro_mode == 1) return {{1, rows}};
  if (cl && cl_mode == 1 && rows <= round / 2) {  // forced: one launch of the widest form whose grid fits
    for (int g = 32; g >= 2; g /= 2)
      if (rows <= (long long)(n_cu / g) * IKF_RO_ROWS) return {{g, rows}};
  }
  long long full = ro ? rows / round * round : 0;
  if (!ro && cl && rows > round) {
    // no row-owner launch to take the full rounds: whole 2-member cluster launches (the cheapest form per row) are peeled off here, one
    // chunk each, and plan_tail only sees what is left below a round (its recursion is one level per full launch)
    const long long cap2 = (round / IKF_RO_ROWS) / 2 * IKF_RO_ROWS;
    while (cap2 > 0 && rows - full > round) {
      plan.push_back({2, cap2});
      full += cap2;
    }
  }
  const bool peeled = !ro && full > 0;
  std::vector<FlowChunk> tail;
  if (rows - full > 0) {
    if (ro && ro_min_tail >= 0) {
      if (rows - full >= ro_min_tail) tail = {{1, rows - full}};
      else tail = {{0, rows - full}};
    } else {
      std::unordered_map<long long, TailPlan> memo;
      tail = plan_tail(rows - full, round, ro, cl, full > 0, memo).chunks;
    }
  }
  if (full > 0 && !peeled) plan.push_back({1, full});
  for (const FlowChunk& c : tail) {
    if (!plan.empty() && plan.back().form == 1 && c.form == 1) plan.back().rows += c.rows;   // the partial round rides in the same launch
    else plan.push_back(c);
  }
  return plan;
}
// `consume`: this plan is about to run (it counts against a pause of the cluster form); the describing entry points pass false
static std::vector<FlowChunk> plan_flow(ikf_model* m, long long rows, bool consume = false) {
  if (rows <= 0) return {};
  cluster_fold_give_up(m);
  const bool ro = rowowner_allowed(m), cl = consume ? cluster_allowed(m) : cluster_allowed_now(m);
  std::vector<FlowChunk> plan = plan_rows(rows, m->n_cu, ro, cl, m->ro_mode, m->cl_mode, m->ro_min_tail);
  if (consume)
    for (const FlowChunk& c : plan) m->cl_used_last = m->cl_used_last || c.form >= 2;
  return plan;
}
static std::string plan_text(const std::vector<FlowChunk>& plan) {
  std::string out;
  for (const FlowChunk& c : plan) {
    if (!out.empty()) out += " ";
    out += (c.form == 0 ? std::string("perlayer") : c.form == 1 ? std::string("rowowner") : "cluster" + std::to_string(c.form)) + ":" + std::to_string(c.rows);
  }
  return out;
}
static RoArgs rowowner_args(ikf_model* m, const PoseSource& ps, const float* d_latent, long long r0, long long nr, int clamp_limits, float* d_q_out) {
  const FlowDims& d = m->dims;
  RoArgs a{};
  a.stream = m->ro_stream;
  a.stream_bytes = (unsigned)(rowowner_stream_floats(2 * m->desc.nb_nodes) * sizeof(float));
  a.sub = m->d_ro_sub; a.n_sub = 2 * m->desc.nb_nodes;
  a.x0 = d_latent + (size_t)r0 * d.D;
  a.ps = ps; a.row0 = r0; a.M = (int)nr; a.D = d.D; a.L1 = d.L1; a.ndof = d.ndof; a.clamp = d.clamp; a.slope = d.slope;
  a.M_inv = m->d_Minv; a.b_lin = m->d_blin; a.lo = chain_lo(m); a.hi = chain_hi(m);
  a.clamp_limits = clamp_limits; a.sigmoid = m->desc.sigmoid_on_output ? 1 : 0;
  a.q_out = d_q_out + (size_t)r0 * d.ndof;
  return a;
}
static ikf_status run_flow_rowowner(ikf_model* m, const PoseSource& ps, const float* d_latent, long long r_base, long long rows,
                                    int clamp_limits, float* d_q_out, hipStream_t s) {
  const long long kMaxLaunchRows = 1LL << 24;
  for (long long r0 = r_base; r0 < r_base + rows; r0 += kMaxLaunchRows) {
    const long long nr = r_base + rows - r0 < kMaxLaunchRows ? r_base + rows - r0 : kMaxLaunchRows;
    const RoArgs a = rowowner_args(m, ps, d_latent, r0, nr, clamp_limits, d_q_out);
    IKF_HIP(prof_mark(m, s));
    IKF_HIP(launch_flow_rowowner(a, m->ro_nbuf, s));
    IKF_HIP(prof_mark(m, s));
  }
  return IKF_OK;
}

static ikf_status ensure_cluster_scratch(ikf_model* m, long long rows) {
  if (rows <= m->cl_rows) return IKF_OK;
  if (m->cl_xbuf) (void)hipFree(m->cl_xbuf);
  if (m->cl_sync) (void)hipFree(m->cl_sync);
  if (m->cl_xbuf_t) (void)hipFree(m->cl_xbuf_t);
  if (m->cl_sync_t) (void)hipFree(m->cl_sync_t);
  m->cl_xbuf = nullptr; m->cl_sync = nullptr; m->cl_xbuf_t = nullptr; m->cl_sync_t = nullptr; m->cl_rows = 0;
  const long long cap = (long long)m->n_cu / 2 * IKF_RO_ROWS;   // the largest chunk the form takes (G = 2)
  const int tiles = (int)((cap + IKF_RO_ROWS - 1) / IKF_RO_ROWS);
  IKF_HIP(hipMalloc(&m->cl_xbuf, sizeof(float) * cluster_xbuf_floats(tiles)));
  IKF_HIP(hipMalloc(&m->cl_sync, cluster_sync_bytes(tiles, 8)));   // (sized for G = 8 on every tile: 4.6 KB per tile)
  // the tagged hand-over's own pair (same sizes; its abort word is the block's LAST word, wherever a launch's partial sums end)
  m->cl_sync_t_bytes = cluster_sync_bytes(tiles, 8);
  IKF_HIP(hipMalloc(&m->cl_xbuf_t, sizeof(float) * cluster_xbuf_floats(tiles)));
  IKF_HIP(hipMalloc(&m->cl_sync_t, m->cl_sync_t_bytes));
  m->cl_tag_dirty = true;   // (created by the first launch that uses them, on its stream)
  m->cl_rows = cap;
  return IKF_OK;
}
static ikf_status run_flow_cluster(ikf_model* m, int G, const PoseSource& ps, const float* d_latent, long long r0, long long nr,
                                   int clamp_limits, float* d_q_out, hipStream_t s) {
  ikf_status st = ensure_cluster_scratch(m, nr);
  if (st != IKF_OK) return st;
  RcArgs c{};
  c.ro = rowowner_args(m, ps, d_latent, r0, nr, clamp_limits, d_q_out);
  c.n_rt = (int)((nr + IKF_RO_ROWS - 1) / IKF_RO_ROWS);
  const bool tagged = m->cl_tagged != 0 && G <= 16;   // (the parity argument needs an even number of subnets: 2 per coupling block)
  const bool local = m->cl_local != 0 && cluster_local_form(G) && cluster_grid(c.n_rt, G, true) <= (unsigned)m->n_cu;
  c.give_up = m->h_cl_give_up;
  if (tagged) {
    unsigned* const abort_t = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(m->cl_sync_t) + m->cl_sync_t_bytes - 128);
    if (m->cl_tag_dirty) {
      IKF_HIP(cluster_tagged_init(m->cl_xbuf_t, cluster_xbuf_floats((int)(m->cl_rows / IKF_RO_ROWS)), m->cl_sync_t, m->cl_sync_t_bytes - 128, abort_t, s));
      m->cl_tag_dirty = false;
    }
    c.xbuf = m->cl_xbuf_t;
    c.pbuf = m->cl_sync_t;
    c.flags = nullptr;
    c.abort_word = abort_t;
    c.test_far = local ? m->cl_far_next : 0;   // (variant 191: honoured by the tagged form's placement check too)
    if (local) m->cl_far_next = 0;
    // the members' placement words sit behind the partial sums of the largest launch (the epoch-word form's flag area, unused here)
    c.xcc_words = reinterpret_cast<unsigned*>(m->cl_sync_t) + (size_t)(m->cl_rows / IKF_RO_ROWS) * 8 * 256;
    m->cl_launch_seq = (m->cl_launch_seq + 1) & 0xffffffu;
    if (m->cl_launch_seq == 0xffffffu) m->cl_launch_seq = 0;
    c.launch_seq = m->cl_launch_seq;
    if (local) { m->cl_tl_nrt = c.n_rt; m->cl_tl_G = G; m->cl_tl_words = c.xcc_words; }
    IKF_HIP(prof_mark(m, s));
    IKF_HIP(launch_flow_cluster_tagged(c, G, s, m->cl_drop_next, local));
  } else {
    c.xbuf = m->cl_xbuf;
    c.pbuf = m->cl_sync;
    c.flags = reinterpret_cast<unsigned*>(m->cl_sync) + (size_t)c.n_rt * G * 256;
    c.abort_word = c.flags + (size_t)c.n_rt * G * 32;
    c.test_far = local ? m->cl_far_next : 0;
    if (local) m->cl_far_next = 0;
    IKF_HIP(prof_mark(m, s));
    IKF_HIP(launch_flow_cluster(c, G, s, m->cl_drop_next, local));
  }
  m->cl_drop_next = 0;
  IKF_HIP(prof_mark(m, s));
  // the repair launch: the same rows through the row-owner kernel, which returns at once unless a wait of the cluster launch ran out
  RoArgs rep = c.ro;
  rep.run_if = c.abort_word;
  IKF_HIP(launch_flow_rowowner(rep, m->ro_nbuf, s));
  return IKF_OK;
}

// "rowowner:4096 cluster16:200" - the chunks run_flow would cut a call of `rows` rows into (tests / tools)
extern "C" ikf_status ikf_plan_describe(ikf_model* m, int64_t rows, char* buf, int buf_len) {
  if (!m || !buf || buf_len < 1) return fail(IKF_ERR_NULL_POINTER, "ikf_plan_describe: null argument");
  const std::string out = plan_text(plan_flow(m, rows));
  if ((int)out.size() + 1 > buf_len) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_plan_describe: buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return IKF_OK;
}
// the same decision without a handle or a device: a chip of n_cu CUs, the resident-row forms allowed or not (released shape, f32)
extern "C" ikf_status ikf_plan_describe_for(int n_cu, int64_t rows, int rowowner_allowed, int cluster_allowed, char* buf, int buf_len) {
  if (!buf || buf_len < 1) return fail(IKF_ERR_NULL_POINTER, "ikf_plan_describe_for: null argument");
  if (n_cu <= 0) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_plan_describe_for: n_cu must be positive");
  const std::string out = plan_text(plan_rows(rows, n_cu, rowowner_allowed != 0, cluster_allowed != 0, -1, -1, -1));
  if ((int)out.size() + 1 > buf_len) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_plan_describe_for: buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return IKF_OK;
}
// 1: cluster launches with 4 / 8 / 16 members hand over through one XCD's L2 (the load-time placement census agreed and no launch has met a
// member elsewhere since); 0: through memory
extern "C" int ikf_cluster_local(ikf_model* m) {
  if (!m) return 0;
  cluster_fold_give_up(m);
  return (m->cl_local != 0 && m->cl_mode != 0 && m->ro_stream != nullptr) ? 1 : 0;
}
extern "C" int64_t ikf_cluster_repairs(ikf_model* m) {
  if (!m) return 0;
  cluster_fold_give_up(m);
  return (int64_t)m->cl_repairs;
}
// calls the cluster form still sits out after a wait of one of its launches ran out (0: in use / never paused)
extern "C" int64_t ikf_cluster_backoff(ikf_model* m) {
  if (!m) return 0;
  cluster_fold_give_up(m);
  return (int64_t)m->cl_pause;
}
// host wall time of the last ikf_load_weights (pack launches and the device-side images included) and of building the small-batch
// per-layer kernels' weight image (0 until a chunk or ikf_reserve needed it)
extern "C" double ikf_load_time_ms(const ikf_model* m) { return m ? m->load_ms : 0.0; }
extern "C" double ikf_frag_image_time_ms(const ikf_model* m) { return m ? m->frag_ms : 0.0; }
extern "C" const char* ikf_dominant_kernel_for(const ikf_model* m, int64_t rows) {
  if (m && rows > 0) {
    long long by_form[3] = {0, 0, 0};
    for (const FlowChunk& c : plan_flow(const_cast<ikf_model*>(m), rows)) by_form[c.form >= 2 ? 2 : c.form] += c.rows;
    if (by_form[1] >= by_form[0] && by_form[1] >= by_form[2] && by_form[1] > 0) return rowowner_kernel_name();
    if (by_form[2] >= by_form[0] && by_form[2] > 0) return "k_flow_cluster";
  }
  return (m && m->precision == 1 && m->split_arena) ? split_kernel_name() : fused_kernel_name();
}
static ikf_status run_flow(ikf_model* m, PoseSource ps, const float* d_latent, long long rows, int clamp_limits,
                           float* d_q_out, hipStream_t s) {
  const std::vector<FlowChunk> plan = plan_flow(m, rows, /*consume=*/true);
  const bool fused = fused_ok(m);
  long long r_base = 0;
  for (const FlowChunk& c : plan) {
    ikf_status st = IKF_OK;
    if (c.form == 1) st = run_flow_rowowner(m, ps, d_latent, r_base, c.rows, clamp_limits, d_q_out, s);
    else if (c.form >= 2 && cluster_grid((int)((c.rows + IKF_RO_ROWS - 1) / IKF_RO_ROWS), c.form, false) <= (unsigned)m->n_cu)   // (every workgroup of a cluster launch must be resident)
      st = run_flow_cluster(m, c.form, ps, d_latent, r_base, c.rows, clamp_limits, d_q_out, s);
    else {
      st = ensure_scratch(m, c.rows);
      for (long long r0 = r_base; st == IKF_OK && r0 < r_base + c.rows; r0 += m->chunk_rows) {
        const long long nr = (r_base + c.rows - r0 < m->chunk_rows) ? r_base + c.rows - r0 : m->chunk_rows;
        st = fused ? run_flow_chunk_fused(m, ps, d_latent, r0, nr, clamp_limits, d_q_out, s)
                   : run_flow_chunk_unfused(m, ps, d_latent, r0, nr, clamp_limits, d_q_out, s);
      }
    }
    if (st != IKF_OK) return st;
    r_base += c.rows;
  }
  return IKF_OK;
}

static ikf_status run_flow_guarded(ikf_model* m, PoseSource ps, const float* d_latent, long long rows, int clamp_limits,
                                   float* d_q_out, hipStream_t s);

static ikf_status check_ready(ikf_model* m, const char* fn) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, std::string(fn) + ": null model");
  if (m->h_give_up && *m->h_give_up != 0) {
    // a workgroup of an EARLIER call gave up waiting for a sibling inside a launch (it was never resident: the device is
    // shared or partitioned in a way the launcher did not expect).  That call's results are invalid; the hand-over is
    // switched off for this handle and every later call uses the plain launch boundary.
    *m->h_give_up = 0;
    m->fuse_tail = 0;
    m->chain_mode = 0;
    if (m->d_chain_ctl) (void)hipMemset(m->d_chain_ctl, 0, sizeof(unsigned) * IKF_CHAIN_CTL_WORDS);
    return fail(IKF_ERR_HIP, std::string(fn) + ": an in-launch hand-over of a PREVIOUS call timed out - that call's results are invalid; "
                                               "the in-launch hand-over is now disabled for this handle, repeat the call");
  }
  if (!m->loaded)
    return fail(IKF_ERR_NOT_LOADED, "Model weights have not been loaded. Call load_state_dict(...)");
  return IKF_OK;
}

extern "C" ikf_status ikf_generate_approx(ikf_model* m, const float* d_poses, int pose_broadcast, const float* d_latent,
                                          int64_t n, int clamp_to_limits, float softflow_scale, float* d_q_out,
                                          void* stream) {
  ikf_status st = check_ready(m, "ikf_generate_approx");
  if (st != IKF_OK) return st;
  if (n < 0) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_generate_approx: n must be >= 0");
  if (n == 0) return IKF_OK;
  if (!d_poses || !d_latent || !d_q_out) return fail(IKF_ERR_NULL_POINTER, "ikf_generate_approx: null device pointer");
  IKF_ON_DEVICE(m)
  PoseSource ps{d_poses, nullptr, pose_broadcast ? 1 : (long long)n, 7, softflow_scale};
  hipStream_t s = static_cast<hipStream_t>(stream);
  StreamScope scope(m, s);
  IKF_HIP(scope.enter());
  st = run_flow_guarded(m, ps, d_latent, n, clamp_to_limits, d_q_out, s);
  if (st != IKF_OK) return st;
  IKF_HIP(scope.leave());
  return IKF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// kinematics
// ---------------------------------------------------------------------------------------------------------------
#define IKF_KIN_PROLOGUE(fn)                                                         \
  if (!m) return fail(IKF_ERR_NULL_POINTER, fn ": null model");                      \
  if (n < 0) return fail(IKF_ERR_BAD_ARGUMENT, fn ": n must be >= 0");               \
  if (n == 0) return IKF_OK;                                                         \
  IKF_ON_DEVICE(m)                                                                   \
  hipStream_t s = static_cast<hipStream_t>(stream);

extern "C" ikf_status ikf_forward_kinematics(ikf_model* m, const float* d_q, int64_t n, float* d_poses_out, void* stream) {
  IKF_KIN_PROLOGUE("ikf_forward_kinematics")
  if (!d_q || !d_poses_out) return fail(IKF_ERR_NULL_POINTER, "ikf_forward_kinematics: null device pointer");
  IKF_HIP(launch_fk(m->d_chain, m->dims.ndof, d_q, n, d_poses_out, s));
  return IKF_OK;
}
extern "C" ikf_status ikf_pose_error(ikf_model* m, const float* d_q, const float* d_target_poses, int64_t n,
                                     float* d_pos_err, float* d_rot_err, void* stream) {
  IKF_KIN_PROLOGUE("ikf_pose_error")
  if (!d_q || !d_target_poses || !d_pos_err || !d_rot_err) return fail(IKF_ERR_NULL_POINTER, "ikf_pose_error: null device pointer");
  IKF_HIP(launch_pose_error(m->d_chain, m->dims.ndof, d_q, d_target_poses, n, d_pos_err, d_rot_err, s));
  return IKF_OK;
}
extern "C" ikf_status ikf_lm_step(ikf_model* m, const float* d_target_poses, const float* d_q, int64_t n, float* d_q_out,
                                  void* stream) {
  IKF_KIN_PROLOGUE("ikf_lm_step")
  if (!d_q || !d_target_poses || !d_q_out) return fail(IKF_ERR_NULL_POINTER, "ikf_lm_step: null device pointer");
  IKF_HIP(launch_lm_step(m->d_chain, m->dims.ndof, d_target_poses, d_q, n, d_q_out, m->lm_precision, s));
  return IKF_OK;
}
extern "C" ikf_status ikf_jacobian(ikf_model* m, const float* d_q, int64_t n, float* d_jac_out, void* stream) {
  IKF_KIN_PROLOGUE("ikf_jacobian")
  if (!d_q || !d_jac_out) return fail(IKF_ERR_NULL_POINTER, "ikf_jacobian: null device pointer");
  IKF_HIP(launch_jacobian(m->d_chain, m->dims.ndof, d_q, n, d_jac_out, s));
  return IKF_OK;
}
extern "C" ikf_status ikf_clamp_to_joint_limits(ikf_model* m, const float* d_q, int64_t n, float* d_q_out, void* stream) {
  IKF_KIN_PROLOGUE("ikf_clamp_to_joint_limits")
  if (!d_q || !d_q_out) return fail(IKF_ERR_NULL_POINTER, "ikf_clamp_to_joint_limits: null device pointer");
  IKF_HIP(launch_clamp(m->d_chain, m->dims.ndof, d_q, n, d_q_out, s));
  return IKF_OK;
}
extern "C" ikf_status ikf_joint_limits_exceeded(ikf_model* m, const float* d_q, int64_t n, uint8_t* d_exceeded_out,
                                                void* stream) {
  IKF_KIN_PROLOGUE("ikf_joint_limits_exceeded")
  if (!d_q || !d_exceeded_out) return fail(IKF_ERR_NULL_POINTER, "ikf_joint_limits_exceeded: null device pointer");
  IKF_HIP(launch_limits_exceeded(m->d_chain, m->dims.ndof, d_q, n, d_exceeded_out, s));
  return IKF_OK;
}

extern "C" ikf_status ikf_set_collision_model(ikf_model* m, const ikf_capsule* h_capsules, int n_capsules,
                                              const int32_t* h_pairs, int n_pairs) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_collision_model: null model");
  if (n_capsules < 0 || n_capsules > IKF_MAX_CAPSULES || n_pairs < 0 || n_pairs > IKF_MAX_CAPSULE_PAIRS)
    return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_collision_model: at most 24 capsules and 276 pairs");
  if ((n_capsules > 0 && !h_capsules) || (n_pairs > 0 && !h_pairs))
    return fail(IKF_ERR_NULL_POINTER, "ikf_set_collision_model: null table");
  CollisionModel cm{};
  cm.n_caps = n_capsules;
  cm.n_pairs = n_pairs;
  for (int c = 0; c < n_capsules; ++c) {
    if (h_capsules[c].frame < 0 || h_capsules[c].frame > m->dims.ndof || !(h_capsules[c].radius >= 0.f))
      return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_collision_model: capsule frame must be in [0, ndof] and radius >= 0");
    cm.frame[c] = h_capsules[c].frame;
    cm.radius[c] = h_capsules[c].radius;
    for (int k = 0; k < 3; ++k) { cm.p0[c][k] = h_capsules[c].p0[k]; cm.p1[c][k] = h_capsules[c].p1[k]; }
  }
  for (int k = 0; k < n_pairs; ++k) {
    const int a = h_pairs[2 * k], b = h_pairs[2 * k + 1];
    if (a < 0 || a >= n_capsules || b < 0 || b >= n_capsules || a == b)
      return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_collision_model: pair index out of range");
    cm.pair_a[k] = (uint8_t)a;
    cm.pair_b[k] = (uint8_t)b;
  }
  IKF_ON_DEVICE(m)
  if (!m->d_collision) IKF_HIP(hipMalloc(&m->d_collision, sizeof(CollisionModel)));
  IKF_HIP(hipMemcpy(m->d_collision, &cm, sizeof(CollisionModel), hipMemcpyHostToDevice));
  return IKF_OK;
}

extern "C" ikf_status ikf_self_collision(ikf_model* m, const float* d_q, int64_t n, float* d_min_dist_out,
                                         uint8_t* d_colliding_out, void* stream) {
  IKF_KIN_PROLOGUE("ikf_self_collision")
  if (!m->d_collision) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_self_collision: no collision model has been set");
  if (!d_q || (!d_min_dist_out && !d_colliding_out)) return fail(IKF_ERR_NULL_POINTER, "ikf_self_collision: null device pointer");
  IKF_HIP(launch_self_collision(m->d_chain, m->d_collision, m->dims.ndof, d_q, n, d_min_dist_out, d_colliding_out, s));
  return IKF_OK;
}

// model-free evaluation helpers (current device; evaluation_utils.py:37-51, :100-112)
extern "C" ikf_status ikf_pose_distance(const float* d_poses_a, const float* d_poses_b, int64_t n, float acos_epsilon,
                                        float* d_pos_err, float* d_rot_err, void* stream) {
  if (n < 0) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_pose_distance: n must be >= 0");
  if (n == 0) return IKF_OK;
  if (!d_poses_a || !d_poses_b || !d_pos_err || !d_rot_err)
    return fail(IKF_ERR_NULL_POINTER, "ikf_pose_distance: null device pointer");
  IKF_HIP(launch_pose_distance(d_poses_a, d_poses_b, n, acos_epsilon, d_pos_err, d_rot_err,
                               static_cast<hipStream_t>(stream)));
  return IKF_OK;
}

extern "C" ikf_status ikf_limits_exceeded(const float* d_q, int64_t n, int n_cols, const float* h_lower,
                                          const float* h_upper, uint8_t* d_exceeded_out, void* stream) {
  if (n < 0) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_limits_exceeded: n must be >= 0");
  if (n_cols < 1 || n_cols > IKF_MAX_LIMIT_COLS)
    return fail(IKF_ERR_BAD_ARGUMENT, "ikf_limits_exceeded: n_cols must be in [1, 32]");
  if (n == 0) return IKF_OK;
  if (!d_q || !h_lower || !h_upper || !d_exceeded_out)
    return fail(IKF_ERR_NULL_POINTER, "ikf_limits_exceeded: null pointer");
  IKF_HIP(launch_limits_exceeded_table(h_lower, h_upper, n_cols, d_q, n, d_exceeded_out,
                                       static_cast<hipStream_t>(stream)));
  return IKF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// exact IK
// ---------------------------------------------------------------------------------------------------------------
// f16x3 range guard: read the overflow word after the flow of a call / round; true -> the caller re-runs on the f32 path
static ikf_status split_overflowed(ikf_model* m, hipStream_t s, bool* out) {
  *out = false;
  if (m->precision != 1 || !m->split_arena || !m->d_split_flag) return IKF_OK;
  IKF_HIP(hipMemcpyAsync(m->h_split_flag, m->d_split_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  IKF_HIP(hipStreamSynchronize(s));
  if (*m->h_split_flag != 0) {
    *out = true;
    IKF_HIP(hipMemsetAsync(m->d_split_flag, 0, sizeof(int), s));
  }
  return IKF_OK;
}

// flow with the range guard applied (guard on + f16x3 mode: one flag read per call; out of range -> f32 re-run)
static ikf_status run_flow_guarded(ikf_model* m, PoseSource ps, const float* d_latent, long long rows, int clamp_limits,
                                   float* d_q_out, hipStream_t s) {
  ikf_status st = run_flow(m, ps, d_latent, rows, clamp_limits, d_q_out, s);
  if (st != IKF_OK || m->precision != 1 || !m->split_guard) return st;
  bool bad = false;
  st = split_overflowed(m, s, &bad);
  if (st != IKF_OK || !bad) return st;
  m->precision = 0;
  st = run_flow(m, ps, d_latent, rows, clamp_limits, d_q_out, s);
  m->precision = 1;
  ++m->split_fallbacks;
  return st;
}

// The retry schedule of generate_exact_ik_solutions (:345-411) over rounds of _generate_exact_ik_solutions (:119-247).
// Seeds of a round come from the flow (latent_fn) or, for parity runs, from the caller (seed_fn).
static ikf_status run_exact(ikf_model* m, const float* d_target_poses, int64_t n, const int32_t* repeat_counts,
                            int n_rounds, int n_lm_steps, float pos_thr, float rot_thr, ikf_latent_fn latent_fn,
                            ikf_seed_fn seed_fn, void* user, float* d_q_out, uint8_t* d_valid_out, int64_t* h_stats,
                            hipStream_t s, const char* fn) {
  const std::string who(fn);
  if (n < 0) return fail(IKF_ERR_BAD_ARGUMENT, who + ": n must be >= 0");
  if (!repeat_counts || n_rounds < 1 || n_rounds > IKF_MAX_ROUNDS)
    return fail(IKF_ERR_BAD_ARGUMENT, who + ": repeat_counts must hold 1..8 rounds");
  if (n_lm_steps < 1 || n_lm_steps > 255) return fail(IKF_ERR_BAD_ARGUMENT, who + ": n_lm_steps must be in 1..255");
  int max_repeat = 0;
  for (int r = 0; r < n_rounds; ++r) {
    if (repeat_counts[r] < 1) return fail(IKF_ERR_BAD_ARGUMENT, who + ": repeat counts must be >= 1");
    max_repeat = repeat_counts[r] > max_repeat ? repeat_counts[r] : max_repeat;
  }
  if (h_stats) memset(h_stats, 0, sizeof(int64_t) * 4 * n_rounds);
  if (n == 0) return IKF_OK;
  if (!d_target_poses || !d_q_out || !d_valid_out) return fail(IKF_ERR_NULL_POINTER, who + ": null device pointer");
  if (n > 0x7fffffffLL / 64 || n * (long long)repeat_counts[0] > 0x7fffffffLL) return fail(IKF_ERR_BAD_ARGUMENT, who + ": n too large");
  const int ndof = m->dims.ndof;
  // Row state: when the worst case of the schedule (all n poses still unsolved in the round with the largest repeat count) is
  // small it is sized once, before any work is enqueued, so no allocation (= device-wide synchronisation) happens between the
  // rounds; a larger worst case - or whatever ikf_reserve_exact has already provided - is not allocated for: the call
  // starts with round 0's rows and each later round grows to its measured n_active * R if it has to.
  const long long worst_rows = n * (long long)max_repeat;
  ikf_status st = ensure_exact(m, n, worst_rows <= m->exact_upfront_rows ? worst_rows : n * (long long)repeat_counts[0]);
  if (st != IKF_OK) return st;
  StreamScope scope(m, s);
  IKF_HIP(scope.enter());

  // (no memset of the outputs: round 0's selection writes every pose - its solution, or zeros and valid = 0 (:197))
  long long n_active = n;
  for (int r = 0; r < n_rounds; ++r) {
    const int R = repeat_counts[r];
    // active pose list = ordered indices of still-invalid poses (every pose in round 0)
    if (r == 0) IKF_HIP(launch_all_active(n, m->ex_pose_idx, m->ex_count, s));
    else IKF_HIP(launch_compact_invalid(d_valid_out, n, m->ex_pose_idx, m->ex_count, m->ex_block_scratch, s));
    if (r > 0) {
      IKF_HIP(hipMemcpyAsync(m->h_count, m->ex_count, sizeof(int), hipMemcpyDeviceToHost, s));
      IKF_HIP(hipStreamSynchronize(s));
      n_active = *m->h_count;
      if (h_stats) h_stats[4 * (r - 1) + 3] = h_stats[4 * (r - 1) + 0] - n_active;
      if (n_active == 0) break;  // everything converged (:383-385, :402-408)
    }
    const long long rows = n_active * R;
    if (rows > 0x7fffffffLL) return fail(IKF_ERR_BAD_ARGUMENT, who + ": a retry round has more than 2^31 - 1 rows");
    if (rows > m->exact_rows) {  // r > 0 only (round 0 was sized above); the stream is idle: the count was just read
      st = ensure_exact_rows(m, rows);
      if (st != IKF_OK) return st;
    }
    const float* d_q_seed = m->ex_q;
    if (seed_fn) {
      d_q_seed = seed_fn(user, r, n_active, R, m->ex_pose_idx, ndof);  // read in place by the LM kernel (no copy)
      if (!d_q_seed) return fail(IKF_ERR_NULL_POINTER, who + ": seed_fn returned null");
    } else {
      const float* d_latent = latent_fn(user, r, rows, m->dims.D);
      if (!d_latent) return fail(IKF_ERR_NULL_POINTER, who + ": latent_fn returned null");
      PoseSource ps{d_target_poses, m->ex_pose_idx, n_active, 7, 0.0f};
      st = run_flow_guarded(m, ps, d_latent, rows, /*clamp=*/1, m->ex_q, s);  // seeds (:188)
      if (st != IKF_OK) return st;
    }
    // all LM iterations of the round in one launch + one selection (kin_kernels.hip: k_exact_lm_iters)
    IKF_HIP(launch_exact_lm_iters(m->d_chain, ndof, d_target_poses, m->ex_pose_idx, (int)n_active, R, n_lm_steps, d_q_seed, m->ex_q,
                                  m->ex_row_valid, m->ex_pose_first, pos_thr, rot_thr, m->lm_precision, s));
    IKF_HIP(launch_exact_select_first(ndof, m->ex_pose_idx, (int)n_active, R, m->ex_q, m->ex_row_valid, d_q_out, d_valid_out,
                                      r == 0 ? 1 : 0, s));
    if (h_stats) {
      h_stats[4 * r + 0] = n_active;
      h_stats[4 * r + 1] = rows;
      h_stats[4 * r + 2] = rows * n_lm_steps;  // upper bound: a row stops at its first valid iteration, or once a sibling repeat was valid earlier
    }
  }
  if (h_stats && n_active > 0) {
    IKF_HIP(launch_compact_invalid(d_valid_out, n, m->ex_pose_idx, m->ex_count, m->ex_block_scratch, s));
    IKF_HIP(hipMemcpyAsync(m->h_count, m->ex_count, sizeof(int), hipMemcpyDeviceToHost, s));
    IKF_HIP(hipStreamSynchronize(s));
    h_stats[4 * (n_rounds - 1) + 3] = h_stats[4 * (n_rounds - 1) + 0] - *m->h_count;
  }
  IKF_HIP(scope.leave());
  return IKF_OK;
}

extern "C" ikf_status ikf_generate_exact(ikf_model* m, const float* d_target_poses, int64_t n,
                                         const int32_t* repeat_counts, int n_rounds, int n_lm_steps,
                                         float pos_thr, float rot_thr, ikf_latent_fn latent_fn, void* latent_user,
                                         float* d_q_out, uint8_t* d_valid_out, int64_t* h_stats, void* stream) {
  ikf_status st = check_ready(m, "ikf_generate_exact");
  if (st != IKF_OK) return st;
  if (!latent_fn) return fail(IKF_ERR_NULL_POINTER, "ikf_generate_exact: latent_fn is required");
  IKF_ON_DEVICE(m)
  return run_exact(m, d_target_poses, n, repeat_counts, n_rounds, n_lm_steps, pos_thr, rot_thr, latent_fn, nullptr,
                   latent_user, d_q_out, d_valid_out, h_stats, static_cast<hipStream_t>(stream), "ikf_generate_exact");
}

extern "C" ikf_status ikf_generate_exact_seeded(ikf_model* m, const float* d_target_poses, int64_t n,
                                                const int32_t* repeat_counts, int n_rounds, int n_lm_steps,
                                                float pos_thr, float rot_thr, ikf_seed_fn seed_fn, void* seed_user,
                                                float* d_q_out, uint8_t* d_valid_out, int64_t* h_stats, void* stream) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_generate_exact_seeded: null model");
  if (!seed_fn) return fail(IKF_ERR_NULL_POINTER, "ikf_generate_exact_seeded: seed_fn is required");
  IKF_ON_DEVICE(m)
  return run_exact(m, d_target_poses, n, repeat_counts, n_rounds, n_lm_steps, pos_thr, rot_thr, nullptr, seed_fn,
                   seed_user, d_q_out, d_valid_out, h_stats, static_cast<hipStream_t>(stream),
                   "ikf_generate_exact_seeded");
}

static const float* fixed_seeds(void* user, int, int64_t, int, const int32_t*, int) { return static_cast<const float*>(user); }

extern "C" ikf_status ikf_refine_exact(ikf_model* m, const float* d_target_poses, int64_t n, int repeat,
                                       const float* d_seeds_q, int n_lm_steps, float pos_thr, float rot_thr,
                                       float* d_q_out, uint8_t* d_valid_out, void* stream) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_refine_exact: null model");
  if (n > 0 && !d_seeds_q) return fail(IKF_ERR_NULL_POINTER, "ikf_refine_exact: null device pointer");
  IKF_ON_DEVICE(m)
  const int32_t rc[1] = {repeat};
  return run_exact(m, d_target_poses, n, rc, 1, n_lm_steps, pos_thr, rot_thr, nullptr, fixed_seeds,
                   const_cast<float*>(d_seeds_q), d_q_out, d_valid_out, nullptr, static_cast<hipStream_t>(stream),
                   "ikf_refine_exact");
}

extern "C" ikf_status ikf_reserve_exact(ikf_model* m, int64_t max_poses, int max_repeat) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_reserve_exact: null model");
  if (max_poses < 1 || max_repeat < 1 || max_poses * (long long)max_repeat > 0x7fffffffLL)
    return fail(IKF_ERR_BAD_ARGUMENT, "ikf_reserve_exact: max_poses and max_repeat must be positive (product < 2^31)");
  IKF_ON_DEVICE(m)
  return ensure_exact(m, max_poses, max_poses * (long long)max_repeat);
}

extern "C" ikf_status ikf_set_exact_upfront_rows(ikf_model* m, int64_t max_rows) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_exact_upfront_rows: null model");
  if (max_rows < 0) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_exact_upfront_rows: max_rows must be >= 0");
  m->exact_upfront_rows = max_rows;
  return IKF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// measurement hook
// ---------------------------------------------------------------------------------------------------------------
extern "C" ikf_status ikf_time_gemm(ikf_model* m, int64_t rows, int iters, float* ms_out, void* stream) {
  ikf_status st = check_ready(m, "ikf_time_gemm");
  if (st != IKF_OK) return st;
  if (!ms_out || rows < 1 || iters < 1) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_time_gemm: bad argument");
  if (m->dims.n_hidden < 2) return fail(IKF_ERR_BAD_SHAPE, "ikf_time_gemm: model has no width x width layer");
  IKF_ON_DEVICE(m)
  hipStream_t s = static_cast<hipStream_t>(stream);
  st = ensure_scratch(m, rows);
  if (st != IKF_OK) return st;
  if (cfg_reads_frag_image((m->tile_cfg >= 0) ? m->tile_cfg : fused_pick_cfg(rows, m->dims.width, m->tune))) {
    st = build_frag_weights(m);
    if (st != IKF_OK) return st;
  }
  if (rows > m->chunk_rows) rows = m->chunk_rows;
  const SubnetWeights& w = m->subnets[0];
  const int variant = pick_variant(m, rows);
  const bool fused = fused_ok(m) && m->dims.n_hidden >= 3;
  hipEvent_t e0, e1;
  IKF_HIP(hipEventCreate(&e0));
  IKF_HIP(hipEventCreate(&e1));
  FusedGemmArgs g{};
  if (fused) {
    // the contraction that reads its A operand from HBM and reduces the last Linear in its epilogue (h -> partials)
    g.M = (int)rows; g.N = m->dims.width; g.K = m->dims.width; g.slope = m->dims.slope;
    g.tune = m->tune;
    g.A = m->hA; g.W = w.w_mid[m->dims.n_hidden - 2]; g.bias = w.b_mid[m->dims.n_hidden - 2];
    g.Wf = frag_image(m, 0, m->dims.n_hidden - 2);
    g.w_last = w.w_last; g.n_out = w.n_out; g.P_out = m->pbuf; g.p_slot_stride = m->chunk_rows * IKF_PSTRIDE;
  }
  auto launch = [&]() -> hipError_t {
    if (fused) return launch_flow_gemm(true, (m->tile_cfg >= 0) ? m->tile_cfg : fused_pick_cfg(rows, m->dims.width, m->tune), g, s);
    return launch_gemm_lrelu(variant, m->hA, w.w_mid[0], w.b_mid[0], m->hB, rows, m->dims.width, m->dims.width,
                             m->dims.slope, s);
  };
  IKF_HIP(launch());
  IKF_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) IKF_HIP(launch());
  IKF_HIP(hipEventRecord(e1, s));
  IKF_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  IKF_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_out = ms / iters;
  return IKF_OK;
}

extern "C" ikf_status ikf_profile_begin(ikf_model* m) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_profile_begin: null model");
  m->prof_on = true;
  m->prof_used = 0;
  return IKF_OK;
}

extern "C" ikf_status ikf_profile_end(ikf_model* m, int64_t* n_launches, double* total_ms, void* stream) {
  if (!m || !n_launches || !total_ms) return fail(IKF_ERR_NULL_POINTER, "ikf_profile_end: null argument");
  IKF_ON_DEVICE(m)
  hipStream_t s = static_cast<hipStream_t>(stream);
  m->prof_on = false;
  // calibrate what an (otherwise empty) event pair measures on this stream and take it off every bracketed launch
  const int ncal = 32;
  hipEvent_t cal[2 * ncal];
  for (int i = 0; i < 2 * ncal; ++i) IKF_HIP(hipEventCreate(&cal[i]));
  for (int i = 0; i < ncal; ++i) {
    IKF_HIP(hipEventRecord(cal[2 * i], s));
    IKF_HIP(hipEventRecord(cal[2 * i + 1], s));
  }
  IKF_HIP(hipStreamSynchronize(s));
  double empty = 0.0;
  for (int i = 0; i < ncal; ++i) {
    float ms = 0.f;
    IKF_HIP(hipEventElapsedTime(&ms, cal[2 * i], cal[2 * i + 1]));
    empty += ms;
  }
  empty /= ncal;
  for (int i = 0; i < 2 * ncal; ++i) (void)hipEventDestroy(cal[i]);
  double tot = 0.0;
  const size_t pairs = m->prof_used / 2;
  for (size_t i = 0; i < pairs; ++i) {
    float ms = 0.f;
    IKF_HIP(hipEventElapsedTime(&ms, m->prof_ev[2 * i], m->prof_ev[2 * i + 1]));
    tot += (ms > empty ? ms - empty : 0.0);
  }
  *n_launches = (int64_t)pairs;
  *total_ms = tot;
  m->prof_used = 0;
  m->last_event_overhead_ms = empty;
  return IKF_OK;
}
extern "C" double ikf_profile_event_overhead_ms(const ikf_model* m) { return m ? m->last_event_overhead_ms : 0.0; }

extern "C" ikf_status ikf_set_precision(ikf_model* m, int mode) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_precision: null model");
  if (mode != 0 && mode != 1) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_precision: mode must be 0 (f32 MFMA) or 1 (3x f16 split)");
  if (mode == 1 && (m->dims.n_hidden < 2 || m->dims.width % 128 != 0))
    return fail(IKF_ERR_BAD_SHAPE, "ikf_set_precision: the f16-split contraction needs a width that is a multiple of 128 and >= 2 hidden layers");
  m->precision = mode;
  if (mode == 1) {
    IKF_ON_DEVICE(m)
    return build_split_weights(m);
  }
  return IKF_OK;
}
extern "C" int ikf_get_precision(const ikf_model* m) { return m ? m->precision : -1; }
extern "C" ikf_status ikf_set_lm_precision(ikf_model* m, int mode) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_lm_precision: null model");
  if (mode != 0 && mode != 1) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_lm_precision: mode must be 0 (fp32, the reference's arithmetic) or 1 (fp64 inside the step)");
  m->lm_precision = mode;
  return IKF_OK;
}
extern "C" int ikf_get_lm_precision(const ikf_model* m) { return m ? m->lm_precision : -1; }

extern "C" ikf_status ikf_set_split_guard(ikf_model* m, int guard) {
  if (!m) return fail(IKF_ERR_NULL_POINTER, "ikf_set_split_guard: null model");
  if (guard != 0 && guard != 1) return fail(IKF_ERR_BAD_ARGUMENT, "ikf_set_split_guard: guard must be 0 or 1");
  m->split_guard = guard;
  return IKF_OK;
}
extern "C" int64_t ikf_split_fallback_count(const ikf_model* m) { return m ? (int64_t)m->split_fallbacks : 0; }
extern "C" int ikf_split_overflow_pending(ikf_model* m, void* stream) {
  if (!m) return 0;
  DeviceGuard dev_guard_(m->device);
  if (dev_guard_.err != hipSuccess) return 0;
  bool bad = false;
  const int saved = m->precision;
  if (m->split_arena) m->precision = 1;  // the flag is meaningful whenever the split images exist
  (void)split_overflowed(m, static_cast<hipStream_t>(stream), &bad);
  m->precision = saved;
  return bad ? 1 : 0;
}
