// Batched forward kinematics, pose error, Levenberg-Marquardt step and the exact-IK bookkeeping for gfx950.
//
// Replaces the jrl.Robot / jrl.math_utils calls on the exact-IK path (jrl is third-party, pinned 2ba7c39, not in tree):
//   robot.forward_kinematics                          call site ikflow/ikflow_solver.py:114
//   geodesic_distance_between_quaternions             call site ikflow/ikflow_solver.py:116
//   robot.inverse_kinematics_step_levenburg_marquardt call site ikflow/ikflow_solver.py:205,208
//   robot.clamp_to_joint_limits                       call site ikflow/ikflow_solver.py:101-102
// and the Python bookkeeping of _generate_exact_ik_solutions (ikflow_solver.py:211-233): validity mask, "highest
// valid repeat wins" selection, and the compaction of still-unsolved poses between retry rounds (:387-400).
//
// One thread per row; the whole chain, the 6 x ndof Jacobian and the ndof x ndof normal equations live in registers
// (every loop is unrolled on the compile-time NDOF so nothing is runtime-indexed).  These kernels are latency/VALU
// bound and tiny next to the flow (~3 kFLOP per row).  FK / pose error are fp32 like the reference; the LM step is
// evaluated in fp64 internally (J^T J + 1e-4 I has condition numbers up to ~1e5, where an fp32 solve - the
// reference's included - carries 1e-3 relative noise) and rounded to fp32 at the end.
#include "ikf_internal.h"   // (kin_math.h through it: the per-row arithmetic)

namespace ikf {

// ---------------------------------------------------------------------------------------------------------------
// public-API kernels
// ---------------------------------------------------------------------------------------------------------------
template <int NDOF>
__global__ __launch_bounds__(256) void k_fk(const Chain* __restrict__ ch, const float* __restrict__ q, long long n,
                                            float* __restrict__ poses) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float qv[NDOF], pose[7];
  load_q<NDOF>(q, row, qv);
  fk_pose_f32<NDOF>(ch, qv, pose);
#pragma unroll
  for (int i = 0; i < 7; ++i) poses[(size_t)row * 7 + i] = pose[i];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_pose_error(const Chain* __restrict__ ch, const float* __restrict__ q,
                                                    const float* __restrict__ tgt, long long n,
                                                    float* __restrict__ pos_err, float* __restrict__ rot_err) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float qv[NDOF], pe, re;
  load_q<NDOF>(q, row, qv);
  pose_error_f32<NDOF>(ch, qv, tgt + (size_t)row * 7, &pe, &re);
  pos_err[row] = pe;
  rot_err[row] = re;
}

template <int NDOF, typename T>
__global__ __launch_bounds__(256) void k_lm_step(const Chain* __restrict__ ch, const float* __restrict__ tgt,
                                                 const float* __restrict__ q, long long n, float* __restrict__ q_out) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float qv[NDOF];
  load_q<NDOF>(q, row, qv);
  lm_step_row<NDOF, T>(ch, tgt + (size_t)row * 7, qv);
#pragma unroll
  for (int j = 0; j < NDOF; ++j) q_out[(size_t)row * NDOF + j] = qv[j];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_jacobian(const Chain* __restrict__ ch, const float* __restrict__ q,
                                                  long long n, float* __restrict__ jac) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float qv[NDOF];
  load_q<NDOF>(q, row, qv);
  float R[9], p[3], axw[NDOF][3], orw[NDOF][3];
  fk_walk<float, NDOF, true>(ch, qv, R, p, axw, orw);
  float* Jo = jac + (size_t)row * 6 * NDOF;
#pragma unroll
  for (int j = 0; j < NDOF; ++j) {
    if (ch->joints[j].kind == 1) {
      const float rx = p[0] - orw[j][0], ry = p[1] - orw[j][1], rz = p[2] - orw[j][2];
      Jo[0 * NDOF + j] = axw[j][0]; Jo[1 * NDOF + j] = axw[j][1]; Jo[2 * NDOF + j] = axw[j][2];
      Jo[3 * NDOF + j] = axw[j][1] * rz - axw[j][2] * ry;
      Jo[4 * NDOF + j] = axw[j][2] * rx - axw[j][0] * rz;
      Jo[5 * NDOF + j] = axw[j][0] * ry - axw[j][1] * rx;
    } else {
      Jo[0 * NDOF + j] = Jo[1 * NDOF + j] = Jo[2 * NDOF + j] = 0.f;
      Jo[3 * NDOF + j] = axw[j][0]; Jo[4 * NDOF + j] = axw[j][1]; Jo[5 * NDOF + j] = axw[j][2];
    }
  }
}

__global__ __launch_bounds__(256) void k_clamp(const Chain* __restrict__ ch, int ndof, const float* __restrict__ q,
                                               long long total, float* __restrict__ q_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % ndof);
  q_out[i] = fminf(fmaxf(q[i], ch->lo[j]), ch->hi[j]);
}

__global__ __launch_bounds__(256) void k_limits_exceeded(const Chain* __restrict__ ch, int ndof,
                                                         const float* __restrict__ q, long long n,
                                                         uint8_t* __restrict__ out) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  bool ex = false;
  for (int j = 0; j < ndof; ++j) {
    const float v = q[(size_t)row * ndof + j];
    ex = ex || (v > ch->hi[j]) || (v < ch->lo[j]);  // strict, evaluation_utils.py:110-112
  }
  out[row] = ex ? 1 : 0;
}

// evaluation_utils.pose_errors (ikflow/evaluation_utils.py:37-51): L2 of the positions, geodesic of the quaternions.
// acos_eps < 0 selects the jrl default clamp (1e-7, as in k_pose_error).
__global__ __launch_bounds__(256) void k_pose_distance(const float* __restrict__ a, const float* __restrict__ b,
                                                       long long n, float acos_eps, float* __restrict__ pos_err,
                                                       float* __restrict__ rot_err) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const float* pa = a + (size_t)row * 7;
  const float* pb = b + (size_t)row * 7;
  const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
  pos_err[row] = sqrtf(dx * dx + dy * dy + dz * dz);
  if (acos_eps < 0.f) {
    rot_err[row] = geodesic_f32(pa + 3, pb + 3);
  } else {
    float dot = pa[3] * pb[3] + pa[4] * pb[4] + pa[5] * pb[5] + pa[6] * pb[6];
    dot = fminf(fmaxf(dot, -1.0f + acos_eps), 1.0f - acos_eps);
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    float m = fmodf(2.0f * acosf(dot) + PI_F, TWO_PI_F);
    if (m < 0.f) m += TWO_PI_F;
    rot_err[row] = fabsf(m - PI_F);
  }
}

// evaluation_utils.calculate_joint_limits_exceeded for an arbitrary limits table (evaluation_utils.py:100-112)
struct LimitsTable {
  float lo[IKF_MAX_LIMIT_COLS], hi[IKF_MAX_LIMIT_COLS];
};
__global__ __launch_bounds__(256) void k_limits_exceeded_table(LimitsTable t, int ncols, const float* __restrict__ q,
                                                               long long n, uint8_t* __restrict__ out) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  bool ex = false;
  for (int j = 0; j < ncols; ++j) {
    const float v = q[(size_t)row * ncols + j];
    ex = ex || (v > t.hi[j]) || (v < t.lo[j]);  // strict
  }
  out[row] = ex ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Capsule self-collision (the mechanism behind evaluation_utils.calculate_self_collisions, ikflow/evaluation_utils.py:
// 115-126, which the reference delegates to jrl/Klampt).  Capsules live in the frame that follows an actuated joint
// (frame 0 = base, frame j+1 = after joint j; fixed URDF offsets are folded on the host); a configuration collides when
// a listed capsule pair comes closer than the sum of its radii.  One thread per row; closest points of two segments
// after Ericson, "Real-Time Collision Detection", 5.1.9.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float segment_segment_dist(const float* p1, const float* q1, const float* p2, const float* q2) {
  const float d1[3] = {q1[0] - p1[0], q1[1] - p1[1], q1[2] - p1[2]};
  const float d2[3] = {q2[0] - p2[0], q2[1] - p2[1], q2[2] - p2[2]};
  const float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const float a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
  const float e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
  const float f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
  const float EPS = 1e-12f;
  float sN, tN;
  if (a <= EPS && e <= EPS) {
    sN = 0.f; tN = 0.f;
  } else if (a <= EPS) {
    sN = 0.f; tN = fminf(fmaxf(f / e, 0.f), 1.f);
  } else {
    const float c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
    if (e <= EPS) {
      tN = 0.f; sN = fminf(fmaxf(-c / a, 0.f), 1.f);
    } else {
      const float b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
      const float denom = a * e - b * b;
      sN = denom > EPS ? fminf(fmaxf((b * f - c * e) / denom, 0.f), 1.f) : 0.f;
      tN = (b * sN + f) / e;
      if (tN < 0.f) { tN = 0.f; sN = fminf(fmaxf(-c / a, 0.f), 1.f); }
      else if (tN > 1.f) { tN = 1.f; sN = fminf(fmaxf((b - c) / a, 0.f), 1.f); }
    }
  }
  const float dx = r[0] + d1[0] * sN - d2[0] * tN, dy = r[1] + d1[1] * sN - d2[1] * tN, dz = r[2] + d1[2] * sN - d2[2] * tN;
  return sqrtf(dx * dx + dy * dy + dz * dz);
}

template <int NDOF>
__global__ __launch_bounds__(64) void k_self_collision(const Chain* __restrict__ ch, const CollisionModel* __restrict__ cm,
                                                       const float* __restrict__ q, long long n,
                                                       float* __restrict__ min_dist, uint8_t* __restrict__ colliding) {
  __shared__ float W[64][IKF_MAX_CAPSULES * 6 + 1];  // world end points of this thread's capsules (+1: bank spread)
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float* w = W[threadIdx.x];
  float qv[NDOF];
  load_q<NDOF>(q, row, qv);
  float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, p[3] = {0.f, 0.f, 0.f};
  const int nc = cm->n_caps;
  for (int f = 0; f <= NDOF; ++f) {
    if (f > 0) {
      compose<float>(R, p, ch->joints[f - 1].pre);
      apply_joint<float>(R, p, ch->joints[f - 1].kind, ch->joints[f - 1].axis, qv[f - 1]);
    }
    for (int c = 0; c < nc; ++c) {
      if (cm->frame[c] != f) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float* pl = e == 0 ? cm->p0[c] : cm->p1[c];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[c * 6 + e * 3 + r] = R[3 * r + 0] * pl[0] + R[3 * r + 1] * pl[1] + R[3 * r + 2] * pl[2] + p[r];
      }
    }
  }
  float best = 3.0e38f;
  for (int k = 0; k < cm->n_pairs; ++k) {
    const int a = cm->pair_a[k], b = cm->pair_b[k];
    const float d = segment_segment_dist(w + a * 6, w + a * 6 + 3, w + b * 6, w + b * 6 + 3) - cm->radius[a] - cm->radius[b];
    best = fminf(best, d);
  }
  if (min_dist) min_dist[row] = best;
  if (colliding) colliding[row] = best < 0.f ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// exact-IK round kernels.  Row layout of q is the reference's tile-major one: row = r * n_active + j  <->  repeat r of
// active pose j (cond.repeat((R,1)), ikflow_solver.py:185).  Poses solved in an earlier iteration are masked instead
// of physically compacted (rows are independent, so the results are identical to the reference's q[mask] compaction).
// ---------------------------------------------------------------------------------------------------------------
// All LM iterations of a retry round in ONE launch (r03; one launch + one selection per iteration before: the exact path is a latency
// chain at small sizes and six launches of ~6 us were a third of a converged call's overhead).  The reference's loop
// (ikflow_solver.py:199-233) runs, per iteration: one LM step on every row of the still-unsolved poses, validity of every row, per pose the
// pick among its valid repeats, and drops the rows of solved poses.  Row-wise that is: a row keeps stepping until IT is valid (its pose is
// then solved in that iteration at the latest), and a pose is solved in the FIRST iteration in which any of its repeats is valid, by the
// highest such repeat.  So each row records the iteration at which it first became valid (1-based; 0 = never) and keeps that q; the
// selection takes the earliest iteration over a pose's repeats and the highest repeat among those.  A repeat stops stepping once a
// sibling of its pose has been valid at an iteration it has already completed (pose_first[j], an atomicMin over the repeats' first valid
// iterations, 0xffffffff = none yet): whatever it found later could not be selected - the reference drops such rows from the batch
// (q[mask] compaction, ikflow_solver.py:231-233).  A hint only: a late or missed update costs iterations, never a result.
template <int NDOF, typename T>
__global__ __launch_bounds__(256) void k_exact_lm_iters(const Chain* __restrict__ ch, const float* __restrict__ poses,
                                                        const int* __restrict__ pose_idx, int n_active, int repeat, int n_steps,
                                                        const float* q_in, float* q, uint8_t* __restrict__ row_valid_iter,
                                                        unsigned* pose_first, float pos_thr, float rot_thr) {  // q_in: the seeds (may be q itself)
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= (long long)n_active * repeat) return;
  const int j = (int)(row % n_active);
  const float* tgt = poses + (size_t)pose_idx[j] * 7;
  float qv[NDOF];
  load_q<NDOF>(q_in, row, qv);
  int first = 0;
  const bool siblings = repeat > 1 && pose_first != nullptr;
  for (int it = 0; it < n_steps; ++it) {
    if (siblings && it > 0 && __hip_atomic_load(pose_first + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)it) break;
    lm_step_row<NDOF, T>(ch, tgt, qv);
    float pe, re;
    pose_error_f32<NDOF>(ch, qv, tgt, &pe, &re);
    if (pe < pos_thr && re < rot_thr) {  // ikflow_solver.py:211
      first = it + 1;
      if (siblings) atomicMin(pose_first + j, (unsigned)first);
      break;
    }
  }
#pragma unroll
  for (int k = 0; k < NDOF; ++k) q[(size_t)row * NDOF + k] = qv[k];
  row_valid_iter[row] = (uint8_t)first;
}

// init (round 0, where every pose is active): a pose without a valid repeat gets its row of zeros and valid = 0 here (ikflow_solver.py:197),
// so the caller's output buffers need no memset before the call.
__global__ __launch_bounds__(256) void k_exact_select_first(int ndof, const int* __restrict__ pose_idx, int n_active,
                                                            int repeat, const float* __restrict__ q,
                                                            const uint8_t* __restrict__ row_valid_iter,
                                                            float* __restrict__ q_out, uint8_t* __restrict__ valid_out, int init) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_active) return;
  // ascending scan of valid_idxs with sol_idx = idx % n_invalid: the highest valid repeat of the iteration wins (ikflow_solver.py:217-222)
  int best_it = 256, best_r = -1;
  for (int r = repeat - 1; r >= 0; --r) {
    const int it = row_valid_iter[(long long)r * n_active + j];
    if (it != 0 && it < best_it) {
      best_it = it;
      best_r = r;
    }
  }
  const int dst = pose_idx[j];
  if (best_r < 0) {
    if (init) {
      for (int k = 0; k < ndof; ++k) q_out[(size_t)dst * ndof + k] = 0.f;
      valid_out[dst] = 0;
    }
    return;
  }
  const long long row = (long long)best_r * n_active + j;
  for (int k = 0; k < ndof; ++k) q_out[(size_t)dst * ndof + k] = q[(size_t)row * ndof + k];
  valid_out[dst] = 1;
}

// round 0's active list: every pose, in order
__global__ __launch_bounds__(256) void k_iota(int* __restrict__ idx, long long n, int* __restrict__ count_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (int)i;
  if (i == 0) *count_out = (int)n;
}

// ordered compaction of the indices with valid[i] == 0 (boolean-mask indexing, ikflow_solver.py:389).
// Up to kCompactSingle poses: one workgroup (one launch - the exact path is a latency chain at these sizes).
__global__ __launch_bounds__(1024) void k_compact_invalid(const uint8_t* __restrict__ valid, long long n,
                                                          int* __restrict__ idx_out, int* __restrict__ count_out) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const long long per = (n + 1023) / 1024;
  const long long b = (long long)t * per;
  long long e = b + per;
  if (e > n) e = n;
  int c = 0;
  for (long long i = b; i < e; ++i) c += valid[i] ? 0 : 1;
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int pos = part[t] - c;
  for (long long i = b; i < e; ++i)
    if (!valid[i]) idx_out[pos++] = (int)i;
  if (t == 1023) *count_out = part[1023];
}

// Larger batches (the 1M-pose sharded exact path): three short launches over 4096-pose blocks - per-block counts, an
// exclusive scan of the block counts by one workgroup, ordered per-block writes.  Thread t of a block owns 16 consecutive
// poses, so thread order = pose order.
constexpr int kCompactSingle = 32768;
constexpr int kCompactPerThread = 16;
constexpr int kCompactBlock = 256 * kCompactPerThread;

__device__ __forceinline__ int compact_thread_count(const uint8_t* __restrict__ valid, long long n, long long base) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < kCompactPerThread; ++i)
    if (base + i < n && !valid[base + i]) ++c;
  return c;
}

// exclusive prefix of `c` over the 256 threads of the block (scan[] = 256 ints of LDS); returns the block total via *total
__device__ __forceinline__ int block_exclusive_scan_256(int c, int* scan, int* total) {
  const int t = threadIdx.x;
  scan[t] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = (t >= off) ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  *total = scan[255];
  return scan[t] - c;
}

__global__ __launch_bounds__(256) void k_compact_count(const uint8_t* __restrict__ valid, long long n,
                                                       int* __restrict__ block_count) {
  __shared__ int scan[256];
  const long long base = (long long)blockIdx.x * kCompactBlock + (long long)threadIdx.x * kCompactPerThread;
  int total;
  (void)block_exclusive_scan_256(compact_thread_count(valid, n, base), scan, &total);
  if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_compact_offsets(const int* __restrict__ block_count, int nb,
                                                          int* __restrict__ block_off, int* __restrict__ count_out) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b = t * per;
  const int e = (b + per < nb) ? b + per : nb;
  int c = 0;
  for (int i = b; i < e; ++i) c += block_count[i];
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - c;
  for (int i = b; i < e; ++i) {
    block_off[i] = run;
    run += block_count[i];
  }
  if (t == 1023) *count_out = part[1023];
}

__global__ __launch_bounds__(256) void k_compact_write(const uint8_t* __restrict__ valid, long long n,
                                                       const int* __restrict__ block_off, int* __restrict__ idx_out) {
  __shared__ int scan[256];
  const long long base = (long long)blockIdx.x * kCompactBlock + (long long)threadIdx.x * kCompactPerThread;
  int total;
  int pos = block_off[blockIdx.x] + block_exclusive_scan_256(compact_thread_count(valid, n, base), scan, &total);
#pragma unroll
  for (int i = 0; i < kCompactPerThread; ++i)
    if (base + i < n && !valid[base + i]) idx_out[pos++] = (int)(base + i);
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

#define IKF_NDOF_DISPATCH(ndof, CALL) \
  switch (ndof) {                     \
    case 4: { constexpr int ND = 4; CALL; break; } \
    case 5: { constexpr int ND = 5; CALL; break; } \
    case 6: { constexpr int ND = 6; CALL; break; } \
    case 7: { constexpr int ND = 7; CALL; break; } \
    case 8: { constexpr int ND = 8; CALL; break; } \
    default: return hipErrorInvalidValue;          \
  }

hipError_t launch_fk(const Chain* ch, int ndof, const float* q, long long n, float* poses, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_fk<ND>), dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, q, n, poses));
  return hipGetLastError();
}
hipError_t launch_pose_error(const Chain* ch, int ndof, const float* q, const float* tgt, long long n, float* pe,
                             float* re, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_pose_error<ND>), dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, q, tgt,
                                             n, pe, re));
  return hipGetLastError();
}
hipError_t launch_lm_step(const Chain* ch, int ndof, const float* tgt, const float* q, long long n, float* q_out,
                          int lm_precision, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (lm_precision == 0) {
    IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_lm_step<ND, float>), dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, tgt, q, n, q_out));
  } else {
    IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_lm_step<ND, double>), dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, tgt, q, n, q_out));
  }
  return hipGetLastError();
}
hipError_t launch_jacobian(const Chain* ch, int ndof, const float* q, long long n, float* jac, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_jacobian<ND>), dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, q, n, jac));
  return hipGetLastError();
}
hipError_t launch_clamp(const Chain* ch, int ndof, const float* q, long long n, float* q_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const long long total = n * ndof;
  hipLaunchKernelGGL(k_clamp, dim3(blocks_for(total, 256)), dim3(256), 0, s, ch, ndof, q, total, q_out);
  return hipGetLastError();
}
hipError_t launch_limits_exceeded(const Chain* ch, int ndof, const float* q, long long n, uint8_t* out,
                                  hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_limits_exceeded, dim3(blocks_for(n, 256)), dim3(256), 0, s, ch, ndof, q, n, out);
  return hipGetLastError();
}
hipError_t launch_pose_distance(const float* a, const float* b, long long n, float acos_eps, float* pe, float* re,
                                hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pose_distance, dim3(blocks_for(n, 256)), dim3(256), 0, s, a, b, n, acos_eps, pe, re);
  return hipGetLastError();
}
hipError_t launch_limits_exceeded_table(const float* lo, const float* hi, int ncols, const float* q, long long n,
                                        uint8_t* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (ncols < 1 || ncols > IKF_MAX_LIMIT_COLS) return hipErrorInvalidValue;
  LimitsTable t{};
  for (int j = 0; j < ncols; ++j) { t.lo[j] = lo[j]; t.hi[j] = hi[j]; }
  hipLaunchKernelGGL(k_limits_exceeded_table, dim3(blocks_for(n, 256)), dim3(256), 0, s, t, ncols, q, n, out);
  return hipGetLastError();
}
hipError_t launch_self_collision(const Chain* ch, const CollisionModel* cm, int ndof, const float* q, long long n,
                                 float* min_dist, uint8_t* colliding, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_self_collision<ND>), dim3(blocks_for(n, 64)), dim3(64), 0, s, ch, cm, q, n,
                                             min_dist, colliding));
  return hipGetLastError();
}
hipError_t launch_exact_lm_iters(const Chain* ch, int ndof, const float* poses, const int* pose_idx, int n_active, int repeat,
                                 int n_steps, const float* q_in, float* q, uint8_t* row_valid_iter, unsigned* pose_first, float pos_thr,
                                 float rot_thr, int lm_precision, hipStream_t s) {
  const long long rows = (long long)n_active * repeat;
  if (rows <= 0) return hipSuccess;
  if (n_steps > 255) return hipErrorInvalidValue;  // (the first-valid iteration is recorded in a byte)
  if (repeat > 1 && pose_first != nullptr) {  // "no repeat of this pose valid yet"
    if (hipError_t e = hipMemsetAsync(pose_first, 0xff, sizeof(unsigned) * (size_t)n_active, s); e != hipSuccess) return e;
  }
  unsigned* const pf = repeat > 1 ? pose_first : nullptr;
  if (lm_precision == 0) {
    IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_exact_lm_iters<ND, float>), dim3(blocks_for(rows, 256)), dim3(256), 0, s, ch, poses, pose_idx,
                                               n_active, repeat, n_steps, q_in, q, row_valid_iter, pf, pos_thr, rot_thr));
  } else {
    IKF_NDOF_DISPATCH(ndof, hipLaunchKernelGGL((k_exact_lm_iters<ND, double>), dim3(blocks_for(rows, 256)), dim3(256), 0, s, ch, poses, pose_idx,
                                               n_active, repeat, n_steps, q_in, q, row_valid_iter, pf, pos_thr, rot_thr));
  }
  return hipGetLastError();
}
hipError_t launch_exact_select_first(int ndof, const int* pose_idx, int n_active, int repeat, const float* q,
                                     const uint8_t* row_valid_iter, float* q_out, uint8_t* valid_out, int init, hipStream_t s) {
  if (n_active <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_exact_select_first, dim3(blocks_for(n_active, 256)), dim3(256), 0, s, ndof, pose_idx, n_active,
                     repeat, q, row_valid_iter, q_out, valid_out, init);
  return hipGetLastError();
}
hipError_t launch_all_active(long long n, int* idx_out, int* count_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_iota, dim3(blocks_for(n, 256)), dim3(256), 0, s, idx_out, n, count_out);
  return hipGetLastError();
}
long long compact_blocks(long long n) { return (n + kCompactBlock - 1) / kCompactBlock; }
hipError_t launch_compact_invalid(const uint8_t* valid, long long n, int* idx_out, int* count_out, int* block_scratch,
                                  hipStream_t s) {
  if (n <= kCompactSingle || block_scratch == nullptr) {
    hipLaunchKernelGGL(k_compact_invalid, dim3(1), dim3(1024), 0, s, valid, n, idx_out, count_out);
    return hipGetLastError();
  }
  const int nb = (int)compact_blocks(n);
  int* counts = block_scratch;
  int* offs = block_scratch + nb;
  hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, s, valid, n, counts);
  hipLaunchKernelGGL(k_compact_offsets, dim3(1), dim3(1024), 0, s, counts, nb, offs, count_out);
  hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(256), 0, s, valid, n, offs, idx_out);
  return hipGetLastError();
}

}  // namespace ikf
