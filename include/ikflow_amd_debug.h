/*
 * ikflow_amd_debug.h - measurement and tuning entry points of libikflow_amd.so.  NOT part of the drop-in boundary: a binding of the
 * reference needs include/ikflow_amd.h only.  bench.py, tools/ and the tests use these to time the dominant kernel
 * inside real calls, to name it for a rocprofv3 trace, and to force one of the forms the engine otherwise chooses by batch size (every
 * setting computes the same function).  The priced-and-rejected forms of rounds 2 - 3 (codes 106 / 108 / 121 / 163 / 164 / 171) exist
 * only in the probes library (lib/libikflow_amd_probes.so, built with -DIKF_PROBES); the shipped library answers IKF_ERR_BAD_ARGUMENT.
 */
#ifndef IKFLOW_AMD_DEBUG_H
#define IKFLOW_AMD_DEBUG_H

#include "ikflow_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 if this library was built with -DIKF_PROBES (lib/libikflow_amd_probes.so), 0 for the shipped library. */
int ikf_probes_build(void);
/* Time `iters` launches of the dominant kernel (the width x width fused Linear+LeakyReLU contraction) on M rows with
 * hipEvents on `stream`; returns average milliseconds per launch in *ms_out. */
ikf_status ikf_time_gemm(ikf_model* m, int64_t rows, int iters, float* ms_out, void* stream);
/* Per-launch timing of the dominant kernel inside real calls: between _begin and _end every hidden-Linear contraction
 * launched by ikf_generate_approx/_exact on `stream` is bracketed by a hipEvent pair; _end synchronises the stream and
 * returns the number of launches and the sum of their elapsed times (ms), each reduced by the elapsed time of an empty
 * event pair calibrated on the same stream. Adds two event records per launch - use it on extra steps, not inside a
 * throughput-timed region. */
ikf_status ikf_profile_begin(ikf_model* m);
ikf_status ikf_profile_end(ikf_model* m, int64_t* n_launches, double* total_ms, void* stream);
/* What an empty hipEvent pair measured on that stream in the last ikf_profile_end (already subtracted per launch). */
double ikf_profile_event_overhead_ms(const ikf_model* m);
const char* ikf_split_kernel_name(void);
/* Name of the dominant kernel as it appears in a rocprofv3 kernel trace. */
const char* ikf_dominant_kernel_name(void);
/* ... of the kernel that carries (most of) a batch of `rows` rows on this handle with its current settings: "k_flow_rowowner" for
 * batches that take the one-launch row-owner form, else the per-layer contraction of the selected precision. */
const char* ikf_dominant_kernel_for(const ikf_model* m, int64_t rows);
/* The chunks a call of `rows` rows is cut into on this handle with its current settings, e.g. "rowowner:4096 cluster16:200"
 * (forms: perlayer, rowowner, cluster<G>). */
ikf_status ikf_plan_describe(ikf_model* m, int64_t rows, char* buf, int buf_len);
/* 1 while cluster launches with 4 / 8 / 16 members keep a row tile's members on one XCD and hand over through its L2 (a placement census at
 * load agreed, and no launch has met a member elsewhere since); 0: hand-over through memory. */
int ikf_cluster_local(ikf_model* m);
/* Calls the cluster form still sits out on this handle: a wait inside one of its launches ran out (another process's kernel held CUs; the
 * rows were recomputed by the repair launch and ikf_cluster_repairs counted it), so the form pauses for 16 plans, twice as many after every
 * further give-up (at most 65536), and is tried again afterwards; 64 clean cluster plans in a row forget the history.  A plan = one cut of a
 * batch into chunks: ikf_generate_approx makes one per call, an exact-IK call one per retry round.  0: in use. */
int64_t ikf_cluster_backoff(ikf_model* m);
/* What first use costs: host wall time (ms) of the last ikf_load_weights on this handle - the packing, the upload and the device-side
 * images of the resident-row forms included - and of building the small-batch per-layer kernels' weight image, which only the first
 * <= 512-row chunk on that path (or ikf_reserve on a handle that can reach it) builds; 0 until then. */
double ikf_load_time_ms(const ikf_model* m);
double ikf_frag_image_time_ms(const ikf_model* m);
/* The same decision as pure host logic - no handle, no device: a chip of n_cu CUs, the released shape in f32, the row-owner launch and
 * the cluster form allowed (1) or not (0).  (CPU tests of the planner.) */
ikf_status ikf_plan_describe_for(int n_cu, int64_t rows, int rowowner_allowed, int cluster_allowed, char* buf, int buf_len);
/* Select the flow pipeline (a tuning / test switch; every setting computes the same function):
 *   -1 auto (3-kernel-per-subnet fused form when the shape allows), 100 the same explicitly, 101..108 the fused form with tile
 *   configuration 0..7 forced, 160 with the 16 x 32 small-batch tiles forced; 0..8 the unfused 4-kernel form with that tile variant;
 *   110 / 111 / 112  small-batch one-launch subnet head (entry kernel + first hidden contraction): off / automatic (default) / forced;
 *   120 / 121        next subnet's entry phase inside the preceding launch (row-tile arrival counter): off (default) / on;
 *   130 .. 134       write-through (sc1) activation stores: none / contractions / entry kernel / both / by batch size (default);
 *   150 / 151        batches of <= 128 rows on 16 x 32 tiles (v_mfma_f32_16x16x4_f32): off / on (default)
 *   152 / 153        the 16-row kernels request their whole operand stream up front: off / on (default)
 *   158 / 159        batches of <= 64 rows on 16 x 16 tiles: off / on (default); 161 forced
 *   162 / 163        129 .. 256 rows on 32 x 32 tiles built from 16x16x4 MFMAs: off (default) / on; 164 forced
 *   170 / 171        <= 128 rows: the whole subnet chain in one launch, hand-over between layers inside each XCD: off (default) / on
 *   185 / 186 / 187  cluster form for 1 .. 3327 rows (G = 32 / 16 / 8 / 4 / 2 workgroups per 16-row tile split the hidden columns and
 *                    exchange activations inside the launch): never / by the cost model (default) / whenever its grid fits; 188: tests -
 *                    the next cluster launch runs one workgroup short (exercises the repair launch)
 *   189 / 190        cluster form with 4 / 8 / 16 members: a row tile's members spread over the XCDs / all on one XCD, hand-over through that
 *                    XCD's L2 (default; every launch publishes its members' XCC_IDs and the handle falls back to 189 by itself on a mismatch);
 *                    191: tests - the next such launch's workgroup 0 publishes a wrong XCC_ID (exercises that fall-back)
 *   192 / 193        cluster form with 2 .. 16 members: hand-over by a drain + an epoch word per member / by payload whose every float carries the
 *                    subnet's parity in its last mantissa bit (default: no drain, no epoch words, no memset in front of the launch)
 *   180 / 181 / 182  row-owner form (ONE launch per call; a workgroup keeps 16 rows on chip through every subnet, weights streamed
 *                    past them; width 1024, coeff_fn_config 3): never / by the cost model (default: full rounds of CUs x 16 rows, and a
 *                    last partial round when nothing cheaper covers it) / always
 * Returns IKF_ERR_BAD_ARGUMENT if unknown. */
ikf_status ikf_set_gemm_variant(ikf_model* m, int variant);


#ifdef __cplusplus
}
#endif
#endif /* IKFLOW_AMD_DEBUG_H */
