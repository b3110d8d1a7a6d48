/*
 * ikflow_amd.h - C-ABI of the MI355X (gfx950) IKFlow inference engine.
 *
 * This is the drop-in boundary for the hot path of jstmn/ikflow:
 *     IKFlowSolver.generate_ik_solutions()        ikflow/ikflow_solver.py:254-343
 *     IKFlowSolver.generate_exact_ik_solutions()  ikflow/ikflow_solver.py:345-411
 * The reference has no FFI of its own (pure Python on FrEIA / jrl / torch); the entry points below are what a
 * ctypes binding inside ikflow/ikflow_solver.py would call instead of `self.nn_model(latent, c=cond, rev=True)`
 * (:98), `robot.forward_kinematics` (:114), `geodesic_distance_between_quaternions` (:116),
 * `robot.inverse_kinematics_step_levenburg_marquardt` (:205,208) and `robot.clamp_to_joint_limits` (:102).
 * INTEGRATION.md shows that binding.
 *
 * Measurement / tuning entry points (kernel timing, form selection) are in ikflow_amd_debug.h - not part of this boundary.
 *
 * Conventions
 *   - plain C types only: no torch / HIP types in signatures; a stream is passed as `void*` (hipStream_t, may be NULL
 *     for the default stream).
 *   - pointers prefixed d_ are DEVICE pointers owned by the caller; h_ are HOST pointers.
 *   - all matrices are row-major fp32; poses are [x y z qw qx qy qz] (ikflow README.md:76).
 *   - every function returns an ikf_status; ikf_last_error() returns a thread-local message for the last failure.
 *   - a handle is NOT thread-safe; use one handle per device.  Work is enqueued on the given stream; the approx
 *     path never synchronises (exception: the opt-in f16x3 precision with its range guard on reads one 4-byte flag per
 *     call), the exact path synchronises once per retry round to read a 4-byte count.
 *   - every entry point runs on the handle's device and restores the caller's current device before it returns.
 *   - the handle's scratch is shared by all of its calls.  Calls may arrive on different streams: a call on another
 *     stream than the previous one first waits (hipStreamWaitEvent) for the previous call's work; calls are never
 *     concurrent on one handle.  Use one handle per stream for concurrency.
 *   - the library owns: the handle, the packed weights and its scratch.  ikf_reserve() pre-sizes scratch so that
 *     steady-state calls allocate nothing.
 */
#ifndef IKFLOW_AMD_H
#define IKFLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IKF_ABI_VERSION 3
#define IKF_MAX_DOF 8     /* actuated joints on the chain                      */
#define IKF_MAX_DIM 16    /* flow width D (dim_latent_space)                   */
#define IKF_MAX_ROUNDS 8  /* len(repeat_counts)                                */

typedef enum ikf_status {
  IKF_OK = 0,
  IKF_ERR_NULL_POINTER = 1,
  IKF_ERR_BAD_SHAPE = 2,        /* dimension outside what the kernels are built for          */
  IKF_ERR_NOT_LOADED = 3,       /* weights not loaded (== the reference's assert, :310-311)   */
  IKF_ERR_MISSING_TENSOR = 4,   /* a state_dict key is absent or has the wrong shape         */
  IKF_ERR_HIP = 5,              /* a HIP runtime call failed; see ikf_last_error()           */
  IKF_ERR_NO_DEVICE = 6,        /* no gfx950 device visible                                  */
  IKF_ERR_BAD_ARGUMENT = 7
} ikf_status;

/* One actuated joint of the serial chain, with the fixed transforms that precede it already folded in
 * (replaces the per-joint walk of jrl.Robot.forward_kinematics; call site ikflow_solver.py:114). */
typedef struct ikf_joint {
  int32_t kind;          /* 1 = revolute, 2 = prismatic                                   */
  float axis[3];         /* unit axis in the joint frame                                  */
  float pre[12];         /* 3x4 row-major fixed transform applied before the joint motion */
} ikf_joint;

/* Replaces IkflowModelParameters (ikflow/model.py:17-41) + the jrl.Robot description (ikflow_solver.py:33). */
typedef struct ikf_model_desc {
  int32_t abi_version;        /* IKF_ABI_VERSION                                                    */
  int32_t nb_nodes;           /* coupling blocks                     model.py:338                   */
  int32_t dim;                /* D = dim_latent_space                ikflow_solver.py:54            */
  int32_t dim_cond;           /* 8 (softflow) or 7                   ikflow_solver.py:51-53         */
  int32_t width;              /* coeff_fn_internal_size, 1..4096     model.py:297  (run at the next multiple
                                 of 256 with zero padding - exact)                                    */
  int32_t n_hidden;           /* coeff_fn_config (1..4)              model.py:59                    */
  float clamp;                /* rnvp_clamp                          model.py:347                   */
  float leaky_slope;          /* 0.01                                model.py:63                    */
  int32_t ndof;               /* robot.ndof                                                         */
  float joint_lo[IKF_MAX_DOF];
  float joint_hi[IKF_MAX_DOF];
  ikf_joint chain[IKF_MAX_DOF];
  float tool[12];             /* 3x4 fixed transform after the last joint (end-effector frame)      */
  int32_t sigmoid_on_output;  /* model.py:304-307 graph variant: module_list.0 = scaling node, .1 = flipped
                                 sigmoid (applied before the linear transform in the inverse pass); permutation /
                                 coupling modules shift to 2i+2 / 2i+3                               */
} ikf_model_desc;

/* One named tensor of the reference's state_dict (FrEIA GraphINN key names, ikflow_solver.py:413-429). */
typedef struct ikf_tensor {
  const char* name;     /* e.g. "module_list.2.subnet1.0.weight"                 */
  const void* h_data;   /* host pointer; fp32 (dtype 0) or int64 (dtype 1)       */
  int32_t dtype;        /* 0 = float32, 1 = int64                                */
  int32_t ndim;
  int64_t shape[4];
} ikf_tensor;

typedef struct ikf_model ikf_model; /* opaque */

/* -- lifetime ---------------------------------------------------------------------------------------------- */
ikf_status ikf_create(const ikf_model_desc* desc, int device, ikf_model** out);
void ikf_destroy(ikf_model* m);
const char* ikf_last_error(void);
int ikf_abi_version(void);

/* Replaces IKFlowSolver.load_state_dict (ikflow_solver.py:413-429): packs the tensors once into the kernels'
 * HBM layout.  Required keys: module_list.0.M_inv [D,D]; module_list.0.b [1,D] (optional = 0 for the plain graph,
 * required for the sigmoid_on_output graph whose scaling node has a non-zero offset); per block i:
 * module_list.{2i+1}.perm_inv [D] (int64), module_list.{2i+2}.subnet{1,2}.{0,2,..}.{weight,bias}. */
ikf_status ikf_load_weights(ikf_model* m, const ikf_tensor* tensors, int n_tensors);
int ikf_weights_loaded(const ikf_model* m);

/* Pre-size the flow scratch for batches of up to max_rows flow rows, and build every weight image a call of that size can reach on this
 * handle (the small-batch per-layer kernels' image included - the form a cluster-form handle falls back to while another process holds
 * CUs): after ikf_load_weights + ikf_reserve no ikf_generate_approx call of <= max_rows rows allocates or synchronises the device. */
ikf_status ikf_reserve(ikf_model* m, int64_t max_rows);
/* Pre-size the exact-IK state for max_poses target poses x max_repeat repeats, so that ikf_generate_exact allocates nothing.
 * Without it a call reserves its own worst case (n * max(repeat_counts) rows) up front while that is at most
 * ikf_set_exact_upfront_rows rows (default 32 Mi; 0 = never), else it grows per round from the measured survivor count. */
ikf_status ikf_reserve_exact(ikf_model* m, int64_t max_poses, int max_repeat);
ikf_status ikf_set_exact_upfront_rows(ikf_model* m, int64_t max_rows);

/* -- approximate IK: replaces IKFlowSolver._run_inference (ikflow_solver.py:85-110) ----------------------- */
/* d_poses: [n x 7], or a single pose [7] when pose_broadcast != 0 (the `y.expand((n,7))` form, :333-336).
 * d_latent: [n x D].  d_q_out: [n x ndof].  clamp_to_limits: robot.clamp_to_joint_limits (:101-102).
 * softflow_scale: the 8th conditional entry (always 0.0 at inference, :335-338). */
ikf_status ikf_generate_approx(ikf_model* m, const float* d_poses, int pose_broadcast, const float* d_latent,
                               int64_t n, int clamp_to_limits, float softflow_scale, float* d_q_out, void* stream);

/* -- kinematics: replace the jrl.Robot calls ---------------------------------------------------------------- */
/* robot.forward_kinematics (ikflow_solver.py:114): [n x ndof] -> [n x 7]. */
ikf_status ikf_forward_kinematics(ikf_model* m, const float* d_q, int64_t n, float* d_poses_out, void* stream);
/* IKFlowSolver._calculate_pose_error (ikflow_solver.py:112-117): L2 position error and quaternion geodesic. */
ikf_status ikf_pose_error(ikf_model* m, const float* d_q, const float* d_target_poses, int64_t n,
                          float* d_pos_err, float* d_rot_err, void* stream);
/* robot.inverse_kinematics_step_levenburg_marquardt(target_poses, q) with jrl defaults (lambda 1e-4, alpha 1,
 * clamped) (ikflow_solver.py:205,208). d_q_out may alias d_q. */
ikf_status ikf_lm_step(ikf_model* m, const float* d_target_poses, const float* d_q, int64_t n, float* d_q_out,
                       void* stream);
/* robot.jacobian: [n x ndof] -> [n x 6 x ndof], rows = angular(3), linear(3). */
ikf_status ikf_jacobian(ikf_model* m, const float* d_q, int64_t n, float* d_jac_out, void* stream);
/* robot.clamp_to_joint_limits (ikflow_solver.py:101-102). d_q_out may alias d_q. */
ikf_status ikf_clamp_to_joint_limits(ikf_model* m, const float* d_q, int64_t n, float* d_q_out, void* stream);
/* evaluation_utils.calculate_joint_limits_exceeded (evaluation_utils.py:100-112): strict inequalities. */
ikf_status ikf_joint_limits_exceeded(ikf_model* m, const float* d_q, int64_t n, uint8_t* d_exceeded_out,
                                     void* stream);

/* Capsule self-collision: the mechanism behind evaluation_utils.calculate_self_collisions (ikflow/evaluation_utils.py:115-126;
 * the reference delegates to jrl / Klampt geometry that is not in this repository - the caller supplies the capsules).  A
 * capsule is a segment p0-p1 with a radius in the frame that follows an actuated joint (0 = base, j + 1 = after joint j);
 * pairs = 2 * n_pairs capsule indices; a configuration collides when any listed pair is closer than r_a + r_b. */
typedef struct ikf_capsule {
  int32_t frame;
  float p0[3];
  float p1[3];
  float radius;
} ikf_capsule;
#define IKF_MAX_CAPSULES 24
ikf_status ikf_set_collision_model(ikf_model* m, const ikf_capsule* h_capsules, int n_capsules, const int32_t* h_pairs,
                                   int n_pairs);
/* [n x ndof] -> signed clearance of the closest listed pair (d_min_dist_out, nullable) and the collision flag
 * (d_colliding_out, nullable).  IKF_ERR_BAD_ARGUMENT when no collision model has been set. */
ikf_status ikf_self_collision(ikf_model* m, const float* d_q, int64_t n, float* d_min_dist_out, uint8_t* d_colliding_out,
                              void* stream);

/* Model-free helpers of the evaluation path, on the current device.  evaluation_utils.pose_errors (evaluation_utils.py:
 * 37-51): [n x 7] vs [n x 7] -> L2 position error and quaternion geodesic; acos_epsilon < 0 = the jrl default (1e-7). */
ikf_status ikf_pose_distance(const float* d_poses_a, const float* d_poses_b, int64_t n, float acos_epsilon,
                             float* d_pos_err, float* d_rot_err, void* stream);
/* evaluation_utils.calculate_joint_limits_exceeded for any limits table (evaluation_utils.py:100-112; the reference's test
 * uses 3 columns): d_q [n x n_cols] on the device, h_lower / h_upper [n_cols] on the host, n_cols <= 32; strict. */
ikf_status ikf_limits_exceeded(const float* d_q, int64_t n, int n_cols, const float* h_lower, const float* h_upper,
                               uint8_t* d_exceeded_out, void* stream);

/* -- exact IK: replaces generate_exact_ik_solutions + _generate_exact_ik_solutions (:119-247, :345-411) ----- */
/* Callback that supplies the latent for retry round `round`: must fill (or return a pointer to) a device buffer of
 * [rows x D] fp32 laid out tile-major exactly as the reference draws it (`draw_latent(..., (n_tiled, D))`, :187).
 * The Python shim draws it with torch.randn on the device - the same generator call the reference makes. */
typedef const float* (*ikf_latent_fn)(void* user, int round, int64_t rows, int dim);

/* d_target_poses [n x 7]; repeat_counts[n_rounds]; thresholds as in :349-350.
 * Outputs: d_q_out [n x ndof] (rows never solved are 0.0, :197), d_valid_out [n] (0/1); every entry is written on success (the
 * buffers need no clearing by the caller), their contents are unspecified when the call returns an error.
 * n_lm_steps: the reference's n_opt_steps_max = 3 (:364); 1 .. 255 (all iterations of a round run in one launch).
 * h_stats (nullable, 4*n_rounds int64): per round {poses entering, flow rows, LM row-iterations, poses solved}. */
ikf_status ikf_generate_exact(ikf_model* m, const float* d_target_poses, int64_t n, const int32_t* repeat_counts,
                              int n_rounds, int n_lm_steps, float pos_error_threshold, float rot_error_threshold,
                              ikf_latent_fn latent_fn, void* latent_user, float* d_q_out, uint8_t* d_valid_out,
                              int64_t* h_stats, void* stream);

/* The same retry schedule with the flow taken out (parity runs on identical seeds): round `round`'s seeds come from the
 * callback.  d_active_idx [n_active] (device, int32, ascending) lists the still-unsolved poses; the callback returns a DEVICE
 * pointer to [n_active * repeat x ndof] clamped seeds, tile-major (row = r * n_active + j <-> repeat r of pose
 * d_active_idx[j]; `conditional.repeat((R,1))`, :185), read in place by kernels enqueued on `stream`.  Exercises everything
 * behind `self._run_inference` (:188): LM iterations, validity, "highest valid repeat wins", slot order, compaction, retry
 * rounds (:199-233, :383-408).  Needs no weights. */
typedef const float* (*ikf_seed_fn)(void* user, int round, int64_t n_active, int repeat, const int32_t* d_active_idx,
                                    int ndof);
ikf_status ikf_generate_exact_seeded(ikf_model* m, const float* d_target_poses, int64_t n, const int32_t* repeat_counts,
                                     int n_rounds, int n_lm_steps, float pos_error_threshold, float rot_error_threshold,
                                     ikf_seed_fn seed_fn, void* seed_user, float* d_q_out, uint8_t* d_valid_out,
                                     int64_t* h_stats, void* stream);
/* One round = IKFlowSolver._generate_exact_ik_solutions (:119-247) given its flow output: d_seeds_q [n * repeat x ndof]
 * (tile-major, not modified) -> d_q_out [n x ndof], d_valid_out [n]. */
ikf_status ikf_refine_exact(ikf_model* m, const float* d_target_poses, int64_t n, int repeat, const float* d_seeds_q,
                            int n_lm_steps, float pos_error_threshold, float rot_error_threshold, float* d_q_out,
                            uint8_t* d_valid_out, void* stream);

/* -- arithmetic of the hidden Linear contractions (99 % of the FLOPs) ---------------------------------------- */
/*   0 = exact f32 on the f32 matrix instructions (default: the reference's precision);
 *   1 = opt-in: error-compensated f16 split (a = hi + lo/2048 for both operands, three v_mfma_f32_32x32x16_f16 products,
 *       fp32 accumulate): measured closer to fp64 than mode 0, ~2x the throughput.  f16 holds magnitudes up to 65504: every
 *       kernel that produces a split operand flags a hidden activation that is non-finite or beyond that range. */
ikf_status ikf_set_precision(ikf_model* m, int mode);
int ikf_get_precision(const ikf_model* m);
/* -- arithmetic of the Levenberg-Marquardt step (ikf_lm_step and every exact-IK entry point) ----------------- */
/*   1 = (default) chain walk, Jacobian, J^T J + 1e-4 I and a Cholesky solve in fp64 inside the kernel, q rounded to fp32;
 *   0 = the reference's own arithmetic: every quantity fp32 (ikflow/config.py:8 DEFAULT_TORCH_DTYPE; the step is jrl's
 *       inverse_kinematics_step_levenburg_marquardt on fp32 tensors, ikflow_solver.py:205,208) and an LU solve with partial
 *       pivoting - what torch.linalg.solve runs (LAPACK sgesv).  Agrees with the reference's CPU result to cond(J^T J + 1e-4 I) x
 *       2^-24 per step, the reference's own rounding noise. */
ikf_status ikf_set_lm_precision(ikf_model* m, int mode);
int ikf_get_lm_precision(const ikf_model* m);
/* Range guard of mode 1.  guard = 1 (default): every call ends by reading the flag (4-byte copy + stream synchronisation) and,
 * if set, runs again on the exact-f32 path (counted).  guard = 0: no synchronisation, no re-run; the flag accumulates on the
 * device until ikf_split_overflow_pending() reads and clears it (synchronises `stream`). */
ikf_status ikf_set_split_guard(ikf_model* m, int guard);
/* Number of calls re-run on the f32 path because the f16 range was exceeded (guard = 1). */
int64_t ikf_split_fallback_count(const ikf_model* m);
/* 1 if an activation left the f16 range since the last check (then cleared), 0 if not; synchronises `stream`. */
int ikf_split_overflow_pending(ikf_model* m, void* stream);
/* Cluster form (workgroups of one launch exchange activations): number of calls in which a wait ran out - a peer workgroup was not
 * resident, i.e. the device is shared or partitioned.  Such a call's rows were recomputed by the row-owner launch queued behind it (the
 * caller's results are valid), and the handle stopped using the form.  Meaningful once the stream of those calls has been synchronised. */
int64_t ikf_cluster_repairs(ikf_model* m);

#ifdef __cplusplus
}
#endif
#endif /* IKFLOW_AMD_H */
