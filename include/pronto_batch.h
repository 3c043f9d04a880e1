/*
 * pronto_batch.h -- C ABI of the batched RBIS/RBIM EKF hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of openhumanoids/pronto: the propagate-and-correct loop
 * (RBISUpdateInterface::updateFilter of the IMU / indexed / indexed+orientation / reset updates, driven
 * by MavStateEstimator::addUpdate), batched over B independent filters.  Each entry point names the
 * reference interface it replaces (paths relative to the reference tree).  The reference has no FFI
 * today -- its boundary is C++; INTEGRATION.md shows the binding a maintainer would add,
 * and pronto_amd/csrc/mav_state_est_batch.hpp keeps the reference's class names on top of this ABI.
 *
 * Conventions
 *  - All functions return PB_OK (0) or an error code; they never throw and never exit().
 *    pb_last_error() gives the text of the last failure on that context.
 *  - The caller owns every buffer it passes; the context owns the device state.
 *  - One context is driven from one host thread at a time (the reference is single-threaded,
 *    lcm_front_end.cpp:223-229); distinct contexts (one per GPU) may be driven concurrently.
 *  - Batched arrays are SoA with the FILTER INDEX FASTEST: element (c, b) of an array [C][B] is at c*B + b.
 *  - `mem` says where the caller's buffers live: PB_HOST (copied into an internal device staging area on the
 *    context's stream, PCIe inclusive) or PB_DEVICE (HBM-resident; the kernels read them in place).
 *    PB_HOST_BROADCAST: host memory with ONE value per row, the same message for every filter (see enum pb_mem).
 *  - fp64 throughout.  State layout is RBIS's (rbis.hpp:22-30): vec = [omega(0-2) v_body(3-5) chi(6-8)
 *    pos(9-11) accel(12-14) | gyro_bias(15-17) accel_bias(18-20)], quat = (w,x,y,z), n_states = 15 keeps
 *    the first 15 (bias states and their covariance pinned to 0).
 *  - The covariance is stored symmetric-packed on the device; the reference does not symmetrise P
 *    (rbis.cpp:118,226) so results agree to rounding, not bit for bit (tolerances in tests/).
 */
#ifndef PRONTO_BATCH_H
#define PRONTO_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pb_ctx pb_ctx;

enum pb_status {
  PB_OK = 0,
  PB_ERR_ARG = 1,         /* bad argument (NULL, size, index out of range, unsupported m) */
  PB_ERR_HIP = 2,         /* a HIP runtime call failed; see pb_last_error */
  PB_ERR_NO_DEVICE = 3,   /* no gfx950 device / HIP extension unusable: there is NO CPU fallback */
  PB_ERR_STATE = 4        /* call order (e.g. step before reset, unknown snapshot slot) */
};

enum pb_mem {
  PB_HOST = 0,   /* caller's buffers are host memory, [rows][B] */
  PB_DEVICE = 1, /* caller's buffers are device memory, [rows][B] */
  /* host memory holding ONE value per row ([rows] doubles): the same message for every filter of the batch, the batch
   * differing in parameters / initial state only (the reference's own batch use: state-estimator/python/param_sweep.py:39-52
   * replays one log per parameter set).  Nothing of batch size crosses PCIe: the step kernels (pb_predict,
   * pb_step_legodo, pb_step_legodo_correct) and the handlers' index lists of pb_update_indexed(_orient) with a diagonal R
   * take the values as kernel arguments (no device block, no input traffic); any other block is expanded on the device.
   * Accepted by pb_predict, pb_update_indexed(_orient), pb_step_legodo(_correct), pb_legodo_update and pb_compose_delta;
   * mask must be NULL. */
  PB_HOST_BROADCAST = 2
};

/* how the measurement covariance R is passed to pb_update_indexed* */
enum pb_rkind {
  PB_R_DIAG_BROADCAST = 0, /* host double[m], same for every filter (handlers' cov_* members)       */
  PB_R_DIAG = 1,           /* [m][B]      per-filter diagonal (legodo certain/uncertain switch)       */
  PB_R_FULL = 2            /* [m*m][B]    per-filter full symmetric R, column-major (indexed_measurement_t) */
};

/* sensor ids, RBISUpdateInterface::sensor_enum (rbis_update_interface.hpp:10-12) */
enum pb_sensor {
  PB_SENSOR_INS = 0, PB_SENSOR_GPS, PB_SENSOR_VICON, PB_SENSOR_LASER, PB_SENSOR_LASER_GPF, PB_SENSOR_SCAN_MATCHER,
  PB_SENSOR_OPTICAL_FLOW, PB_SENSOR_RESET, PB_SENSOR_INVALID, PB_SENSOR_RGBD, PB_SENSOR_FOVIS, PB_SENSOR_LEGODO,
  PB_SENSOR_POSE_MEAS, PB_SENSOR_ALTIMETER, PB_SENSOR_AIRSPEED, PB_SENSOR_SIDESLIP, PB_SENSOR_INIT_MESSAGE,
  PB_SENSOR_VIEWER, PB_SENSOR_YAWLOCK
};

/* ---- lifetime ------------------------------------------------------------------------------------- */

/* Replaces MavStateEstimator::MavStateEstimator (mav_state_est.cpp:12-22) for B filters.
 * n_states in {15, 21}; device = HIP device ordinal; n_snapshots = history slots for pb_snapshot (>= 0). */
int pb_create(pb_ctx **out, int n_states, int batch, int device, int n_snapshots);
int pb_destroy(pb_ctx *ctx);
const char *pb_last_error(const pb_ctx *ctx); /* ctx may be NULL: last error of a failed pb_create */

/* A new context launches on its own non-blocking stream.  pb_set_stream switches to an existing hipStream_t, taken
 * literally: NULL is the null stream (torch's default stream) -- use the stream that PRODUCES the device buffers you
 * pass, otherwise nothing orders their producer kernels before these launches.  pb_use_own_stream switches back. */
int pb_set_stream(pb_ctx *ctx, void *hip_stream);
int pb_use_own_stream(pb_ctx *ctx);
/* eigen_utils constants that are not in the reference tree (g_vec magnitude, chiToQuat tolerance). */
int pb_set_constants(pb_ctx *ctx, double g, double chi_tol);
int pb_sync(pb_ctx *ctx);
/* name (as rocprofv3 prints it, without the pb:: prefix) of the kernel pb_step_legodo / pb_run_legodo launch for
 * this context: "k_step_coop<15,true,H>" (15 states, two waves per tile, up to 393 216 filters), "k_step<15,true,H>"
 * (15 states, one lane per filter, beyond that), "k_step_quad<true,H>" (21 states, four waves per tile) or
 * "k_step_coop<21,true,H>" (21 states with PRONTO_BATCH_QUAD21=0); H = 0/1/2 is the cache policy of the state round trip
 * the library picked from the state size (default / sc1 stores / non-temporal) */
const char *pb_hot_kernel(const pb_ctx *ctx);
/* filters per block of pb_run_legodo's cache-blocked order (0: the whole batch per launch; see pb_run_legodo) */
int pb_run_block(const pb_ctx *ctx);
int pb_batch(const pb_ctx *ctx);
int pb_n_states(const pb_ctx *ctx);

/* device memory helpers for hosts without torch (hipMalloc / hipMemcpy on the context's device) */
int pb_malloc(pb_ctx *ctx, uint64_t bytes, void **dev_ptr);
int pb_free(pb_ctx *ctx, void *dev_ptr);
int pb_memcpy_h2d(pb_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
int pb_memcpy_d2h(pb_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
/* Page-locked host memory for PB_HOST inputs.  Any host memory works; from pinned buffers the staging copy is a DMA at
 * link rate, and either way it runs on a second stream into one of two staging buffers, so the copy of message k+1
 * overlaps the kernels of message k.  A PB_HOST call returns when its buffers have been copied (reusable at once). */
int pb_host_alloc(pb_ctx *ctx, uint64_t bytes, void **host_ptr);
int pb_host_free(pb_ctx *ctx, void *host_ptr);

/* ---- chunked uploads (bulk replays of recorded logs, segment_stream.hpp) ------------------------------------------------
 * The reference's LCMFrontEnd::run hands one message at a time to the handlers (lcm_front_end.cpp:223-229); a replay of N recorded
 * logs as one batch instead moves CHUNKS of pre-decoded messages ([slot][rows][B] blocks in page-locked memory) to HBM on the
 * context's copy stream while the kernels of the previous chunk run, and the handlers are then given PB_DEVICE messages.
 *   pb_fence_create / pb_fence_record   a marker on the main stream: "every kernel enqueued so far" (at most 16 per context)
 *   pb_fence_wait                       the host waits until the marked work has run
 *   pb_upload_async                     host (pb_host_alloc) -> device on the copy stream; it starts once fence `after_fence` has
 *                                       been reached (-1: at once) -- the fence recorded behind the last kernel that READ dev_dst
 *   pb_upload_join                      the main stream waits for every upload issued so far (call it before the first kernel
 *                                       that reads the uploaded block)
 *   pb_upload_sync                      the HOST waits for every upload issued so far (the page-locked source may be refilled) */
int pb_fence_create(pb_ctx *ctx, int *fence_out);
int pb_fence_record(pb_ctx *ctx, int fence);
int pb_fence_wait(pb_ctx *ctx, int fence);
int pb_upload_async(pb_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes, int after_fence);
int pb_upload_join(pb_ctx *ctx);
int pb_upload_sync(pb_ctx *ctx);

/* ---- update objects (rbis_update_interface.hpp) ----------------------------------------------------- */

/* RBISResetUpdate::updateFilter (rbis_update_interface.cpp:23-28): posterior <- (state, cov), loglik <- 0.
 * vec [n][B], quat [4][B], cov [n*n][B] (full, column-major; the lower triangle is taken) -- or, with
 * broadcast != 0, vec [n], quat [4], cov [n*n] applied to every filter (host memory only). */
int pb_reset(pb_ctx *ctx, const double *vec, const double *quat, const double *cov, int broadcast, int mem);

/* The posterior of an update computed OUTSIDE the library becomes the head: posterior_state / posterior_covariance /
 * loglikelihood as an RBISUpdateInterface subclass fills them (rbis_update_interface.hpp:14-35; the estimator then reads them as
 * the next prior, mav_state_est.cpp:55-62).  This is the slow path behind the shim's RBISHostUpdate -- user code written against
 * the reference's updateFilter(const RBIS&, const RBIM&, double) runs on the host between a pb_get_head and this call.  Layout
 * as pb_reset (not broadcast); loglik [B] or NULL (= 0).  Honours pb_set_output_slot like every update.  PB_HOST or PB_DEVICE. */
int pb_set_head(pb_ctx *ctx, const double *vec, const double *quat, const double *cov, const double *loglik, int mem);

/* RBISIMUProcessStep::updateFilter (rbis_update_interface.cpp:30-52 -> rbis.cpp:37-122).
 * imu_block [7][B] = gyro xyz, accelerometer xyz (body frame, as InsHandler hands them over,
 * sensor_handlers.cpp:226-251), dt.  q = {q_gyro, q_accel, q_gyro_bias, q_accel_bias} (host). */
int pb_predict(pb_ctx *ctx, const double *imu_block, const double q[4], int mem);

/* RBISIndexedMeasurement::updateFilter (rbis_update_interface.cpp:54-95 -> rbis.cpp:160-178,124-143,219-227).
 * m in 1..6; idx host int[m] (state indices < n_states, distinct); z [m][B]; R per r_kind;
 * mask [B] uint8 or NULL: 0 = "handler returned NULL for this filter" (lcm_front_end.hpp:156) -> no update. */
int pb_update_indexed(pb_ctx *ctx, int m, const int *idx, const double *z, const double *R, int r_kind,
                      const uint8_t *mask, int mem);

/* RBISIndexedPlusOrientationMeasurement::updateFilter (rbis_update_interface.cpp:97-107 -> rbis.cpp:189-217).
 * quat_meas [4][B] (w,x,y,z).  z entries at chi indices (6..8) are ignored, as in the reference. */
int pb_update_indexed_orient(pb_ctx *ctx, int m, const int *idx, const double *z, const double *R, int r_kind,
                             const double *quat_meas, const uint8_t *mask, int mem);

/* Fused hot step = RBISIMUProcessStep then RBISIndexedMeasurement(idx = velocityInds = {3,4,5}) as produced
 * by LegOdoCommon::createMeasurement in mode lin_rate (rbis_legodo_common.cpp:153-156): ONE launch and ONE
 * round trip of the state through HBM (BASELINE.json's "predict+update step").
 * lo_block [6][B] = z xyz, Rdiag xyz; mask as above. */
int pb_step_legodo(pb_ctx *ctx, const double *imu_block, const double *lo_block, const uint8_t *mask,
                   const double q[4], int mem);
/* The same step with the IMU block and the leg-odometry block (+ mask) living in DIFFERENT spaces, e.g. one robot's IMU
 * message for every filter (PB_HOST_BROADCAST) and a per-filter measurement that pb_legodo_update_after_predict left on
 * the device (PB_DEVICE).  At most one of the two may be PB_HOST. */
int pb_step_legodo_split(pb_ctx *ctx, const double *imu_block, int imu_mem, const double *lo_block, const uint8_t *mask,
                         int lo_mem, const double q[4]);

/* The same step with ONE MORE measurement behind it, still one launch and one round trip of the state: what the
 * reference does as a third updateFilter call when a correction message follows the IMU / leg-odometry pair.
 *   PB_CORR_POS_ORIENT  idx = {9,10,11,6,7,8}  FovisHandler position_orient (rbis_fovis_update.cpp:299-305)
 *   PB_CORR_POS_YAW     idx = {9,10,11,8}      ScanMatcherHandler position_yaw (sensor_handlers.cpp:709-722)
 * both RBISIndexedPlusOrientationMeasurement (rbis.cpp:189-217) with a diagonal R: z2 [m][B] (entries at chi indices are
 * ignored), R2 [m][B] (r_kind2 = PB_R_DIAG) or host double[m] (PB_R_DIAG_BROADCAST), quat_meas2 [4][B], mask2 [B] or NULL.
 * `mem` says where imu_block / lo_block / mask live, `mem2` where z2 / R2 / quat_meas2 / mask2 live (a FovisHandler
 * measurement is composed on the device, pb_compose_delta, while the IMU message comes from the host).
 * Results equal pb_step_legodo followed by pb_update_indexed_orient to rounding (tests/: <= 1e-12 relative). */
enum pb_corr { PB_CORR_POS_ORIENT = 0, PB_CORR_POS_YAW = 1 };
int pb_step_legodo_correct(pb_ctx *ctx, const double *imu_block, const double *lo_block, const uint8_t *mask,
                           const double q[4], int mem, int corr_kind, const double *z2, const double *R2, int r_kind2,
                           const double *quat_meas2, const uint8_t *mask2, int mem2);

/* n_steps consecutive fused steps from HBM-resident streams: imu_stream [n_steps][7][B],
 * lo_stream [n_steps][6][B], mask_stream [n_steps][B] or NULL (device pointers).  One launch per step (the
 * posterior is materialised in HBM after every message, as MavStateEstimator::addUpdate does,
 * mav_state_est.cpp:50-70).  If elapsed_ms != NULL the call brackets the launches with HIP events on the
 * context's stream, synchronises, and returns the device time.
 * Launch ORDER: for a state that fits the memory-side cache, step by step over the whole batch.  Beyond it (pb_create: more than
 * 256 MB of state and a batch of whole tiles; pb_run_block() > 0) the call runs filter range OUTER, time INNER: a block of
 * pb_run_block() filters takes all n_steps steps -- its posteriors stay cache-resident from step to step -- before the next block
 * starts.  The filters are independent, so the results are those of the step-by-step order: bit for bit when the same step kernel
 * runs (PRONTO_BATCH_COOP15 forced, or any batch up to 393 216 filters), to rounding where the blocks run the two-wave 15-state
 * kernel that their size wants and the whole batch would run the one-lane kernel.  Same accounting as ever: one launch per step
 * and block, every posterior loaded and stored once per step.  PRONTO_BATCH_BLOCKED=0 keeps the step-by-step order. */
int pb_run_legodo(pb_ctx *ctx, int n_steps, const double *imu_stream, const double *lo_stream,
                  const uint8_t *mask_stream, const double q[4], float *elapsed_ms);

/* Time-fused replay of the same streams: steps_per_launch consecutive steps per kernel launch with the state and
 * covariance resident in registers; the posterior is written to HBM once per launch instead of once per message.  Same
 * arithmetic as pb_run_legodo, NOT the plugin semantics (no per-message posterior): meant for parameter sweeps and
 * likelihood evaluation over log segments (state-estimator/python/param_sweep.py:39-52).  15 states: k_replay_coop (two
 * waves per tile), 21 states: k_replay_quad (four).  Accounting differs from the T = 1 path (104 + 2240/T bytes per
 * 15-state filter-step, 104 + 4112/T for 21 states): see DESIGN.md section 6. */
int pb_replay_legodo_fused(pb_ctx *ctx, int n_steps, int steps_per_launch, const double *imu_stream,
                           const double *lo_stream, const uint8_t *mask_stream, const double q[4], float *elapsed_ms);

/* The same replay as a forward pass that KEEPS EVERY POSTERIOR -- what the reference's history does by value
 * (mav_state_est.cpp:50-70) and what EKFSmoothBackwardsPass reads (:98-189): the posterior of step t is also written into
 * checkpoint slot first_slot + t (pb_history_reserve; first_slot + n_steps <= slots), while the state stays in registers.
 * Per step one slot is written and nothing is read back (the per-message path with pb_set_output_slot reads the previous
 * slot and writes the next one).  Slots are bit-identical to pb_replay_legodo_fused run ONE step per launch, and equal to
 * rounding (1e-12 relative in the tests) to what pb_set_output_slot + pb_step_legodo leave -- another kernel, other FMA
 * contractions: do not mix the two paths under a bit-exact replay checksum.  The head
 * afterwards is the context's own array (= the last slot's content).  Accounting: (S_x + S_P + 8) + 104 + 2 (S_x + S_P + 8) / T
 * bytes per filter-step -- its own, never the headline metric. */
int pb_replay_legodo_checkpointed(pb_ctx *ctx, int n_steps, int steps_per_launch, const double *imu_stream,
                                  const double *lo_stream, const uint8_t *mask_stream, const double q[4], int first_slot,
                                  float *elapsed_ms);

/* ---- history look-up used by FovisHandler (rbis_fovis_update.cpp:184-223) ------------------------------ */

/* remember the head posterior's (position, quat) in `slot` (the reference finds it again by
 * history.updateMap.lower_bound(prev_timestamp)) */
int pb_snapshot(pb_ctx *ctx, int slot);
/* the same from a saved posterior (pb_history_reserve checkpoint) instead of the head: what FovisHandler does when the
 * keyframe changes -- history.updateMap.lower_bound(prev_timestamp)->posterior_state (rbis_fovis_update.cpp:184-207) */
int pb_snapshot_from_slot(pb_ctx *ctx, int slot, int checkpoint_slot);
/* T1 = T0(slot) * (t, q): z_out [3][B] = T1.translation, quat_out [4][B] = T1.rotation (device buffers) */
int pb_compose_delta(pb_ctx *ctx, int slot, const double *t, const double *q, double *z_out, double *quat_out,
                     int mem);

/* ---- noise identification / parameter sweeps (state-estimator/src/noise_id/noise_id.cpp:9-65) ------------------ */

/* Per-filter process noise: q_block [4][B] (device memory) = q_gyro, q_accel, q_gyro_bias, q_accel_bias of every
 * filter, used by all following pb_predict / pb_step_legodo / pb_run_legodo calls instead of their scalar q; NULL
 * switches back.  This is what lets ONE batch carry the windows x candidate-noise grid of a noise-ID run
 * (sampleProcessForward, noise_id.cpp:9-42: by linearity rolled_cov - start_window_cov is the predict chain from P = 0). */
int pb_set_process_noise_block(pb_ctx *ctx, const double *q_block_dev);
/* Window likelihood pieces (noise_id.cpp:37-38,44-65): e = head (-) truth with chi = Log(truth.quat^-1 * quat); over
 * the m <= 9 active indices: out3 [3][B] = log det P_aa, e_a^T P_aa^-1 e_a, -0.5*(m log 2pi + logdet + maha).
 * truth_vec [n][B], truth_quat [4][B]; err_out [n][B] or NULL. */
int pb_window_nll(pb_ctx *ctx, int m, const int *idx, const double *truth_vec, const double *truth_quat, double *out3,
                  double *err_out, int mem);

/* ---- leg kinematic odometry: the producer of the leg-odometry increment LegOdoHandler turns into a measurement ----
 * Replaces, per filter, leg_estimate::updateOdometry (motion_estimate/src/leg_estimate/leg_estimate.cpp:395-556: forward
 * kinematics :430-447, 30 ms reset, primary-foot selection through FootContactAlt::DetectFootTransition or -- contact mode
 * "standing" -- FootContact::DetectFootTransition, pelvis pose slaved to the FILTER's own orientation, :219-297) and
 * foot_contact_classify::update (foot_contact_classify.cpp:57-125: status -1 / 0 / 1).  Inputs are either the two
 * body-to-foot transforms (pb_legodo_update*: feet [14][B] = body-to-left-foot (t3, q4 = w,x,y,z), body-to-right-foot
 * (t3, q4); forces [2][B] = left, right |vertical foot force|, doubles) or the joint state itself (pb_legodo_update_joints,
 * after pb_legodo_set_chain).  `mem` as everywhere (PB_HOST_BROADCAST: one robot's message for a whole parameter sweep).
 * Outputs are DEVICE arrays, any may be NULL: delta_out [7][B] (pelvis increment t3, q4),
 * status_out [B], and -- LegOdoCommon::createMeasurement in mode lin_rate (rbis_legodo_common.cpp:99-169) -- lo_block_out
 * [6][B] + mask_out [B] in exactly the form pb_step_legodo / pb_update_indexed take with PB_DEVICE.
 * The thresholds are rounded to float like the reference's (leg_estimate.cpp:103-104); delays < 2e9 us. */
int pb_legodo_init(pb_ctx *ctx, double schmitt_low_threshold, double schmitt_high_threshold, int64_t schmitt_low_delay_us,
                   int64_t schmitt_high_delay_us, int filter_contact_events);
int pb_legodo_update(pb_ctx *ctx, int64_t utime, const double *feet, const double *forces, int mem, int zero_delta,
                     double r_vxyz, double r_vxyz_uncertain, double *delta_out, double *status_out, double *lo_block_out,
                     uint8_t *mask_out);
/* The same odometry, slaved to the orientation the filter WILL have after pb_predict(imu_block) -- computed from the head
 * state on the device, covariance untouched -- so that the caller can then run that IMU step and the leg-odometry update this
 * call produced as ONE fused kernel (pb_step_legodo_split with lo_block_out / mask_out as PB_DEVICE blocks) instead of
 * pb_predict, pb_legodo_update, pb_update_indexed: two state round trips become one.  Results equal that three-call sequence
 * to rounding.  imu_block: [7][B] (PB_DEVICE / PB_HOST) or [7] (PB_HOST_BROADCAST); not both it and the foot blocks PB_HOST. */
int pb_legodo_update_after_predict(pb_ctx *ctx, const double *imu_block, int imu_mem, int64_t utime, const double *feet,
                                   const double *forces, int mem, int zero_delta, double r_vxyz, double r_vxyz_uncertain,
                                   double *delta_out, double *status_out, double *lo_block_out, uint8_t *mask_out);
/* Contact mode (leg_estimate.cpp:113-121): standing != 0 = state_estimator.legodo.init_contact_mode "standing", the
 * conservative FootContact classifier (foot_contact/FootContact.cpp:29-54; total_force and standing_schmitt_level are floats
 * there and here); use_controller_input != 0 lets the controller's contact counts overrule FootContactAlt's standing foot
 * (leg_estimate.cpp:365-387).  Call after pb_legodo_init, which puts EVERYTHING back to the constructor's defaults: FootContactAlt,
 * no controller input, controller contact counts -1, no world constraint, all per-robot state reset. */
int pb_legodo_set_contact_mode(pb_ctx *ctx, int standing, double total_force, double standing_schmitt_level,
                               int use_controller_input);
/* Independent log segments (one recorded robot per filter; the reference's se-batch-process.sh runs such logs one after the
 * other): every filter's message carries its OWN time stamp, and a filter whose segment has ended has none.  For the NEXT
 * odometry / pair call only: utimes [B] (int64, may be NULL = the call's scalar utime for every filter) is what
 * leg_estimate::updateOdometry takes as `utime` per filter (elapsed time of the increment, the 30 ms reset, the Schmitt-trigger
 * clocks, foot_contact_classify's black-out windows); valid [B] (uint8, may be NULL = all) marks the filters that HAVE a message --
 * the others keep their odometry state and get no measurement (mask 0).  mem: PB_HOST (copied before the call returns) or
 * PB_DEVICE (read IN PLACE by the consuming launch: the arrays stay the caller's until that launch has run).  The arrays are
 * taken by the next pb_legodo_update* / pb_step_legodo_joints / _feet call whatever its outcome -- a call that fails its argument
 * checks has still consumed them.  (The joint Kalman filter of
 * pb_joint_filter keeps one clock per context: with per-filter times use the low-pass filter or none.) */
int pb_legodo_set_message_times(pb_ctx *ctx, const int64_t *utimes, const uint8_t *valid, int mem);
/* Which of LegOdoCommon's measurements (state_estimator.legodo.mode, rbis_legodo_common.cpp:5-23,110-169) the odometry calls
 * pb_legodo_update / _after_predict / _joints write into lo_block_out / mask_out -- formed on the device, where the increment
 * is, so that no batch-sized array crosses PCIe to form it (the reference forms it on the host right behind updateOdometry):
 *   0  lin_rate (default)  lo_block_out [6][B]  = z (v), Rdiag;                  mask_out [B]     idx {3,4,5}
 *   1  lin_rot_rate        lo_block_out [12][B] = z (v, rpy rate) [6], Rdiag [6]; mask_out [B]     idx {3,4,5,0,1,2}
 *   2  pos_and_lin_rate    lo_block_out [12][B] = z (position, v) [6], Rdiag [6]; mask_out [2][B]  idx {9,10,11,3,4,5}
 *      mask_out[0][b] = status valid AND position valid: the six-row update; mask_out[1][b] = status valid AND position NOT valid:
 *      the reference's per-message fall-back to lin_rate (:118-122), a three-row update on rows 3..5 (z) and 9..11 (Rdiag) of
 *      the same block with idx {3,4,5}.  (The position is leg_estimate's world constraint, which this mode switches on.)
 * r_xyz, r_vang, r_vang_uncertain: state_estimator.legodo.r_xyz / r_vang / r_vang_uncertain (r_vxyz and r_vxyz_uncertain stay
 * arguments of the odometry calls).  The pair calls pb_step_legodo_joints / _feet form AND APPLY the same measurement (their
 * lo_block_out / mask_out have the shapes above).  pb_legodo_init puts the mode back to 0. */
int pb_legodo_set_measurement_mode(pb_ctx *ctx, int mode, double r_xyz, double r_vang, double r_vang_uncertain);
/* LegOdoHandler's "ignore the calculated velocity at launch" (state_estimator.legodo.zero_initial_velocity,
 * rbis_legodo_update.cpp:58,264-268), counted PER FILTER on the device: the counter is decremented by every message whose
 * status is valid for that filter (the reference returns NULL before the decrement otherwise, :243-255) and while it stays
 * above zero the increment (and the position) handed on is the identity.  Independent of the per-call zero_delta flag.
 * ticks <= 65535 (the counter is a 16-bit field of the per-robot flag word), else PB_ERR_ARG. */
int pb_legodo_set_zero_initial_velocity(pb_ctx *ctx, int ticks);
/* The last CONTROLLER_FOOT_CONTACT message (LegOdoHandler::controllerInputHandler, rbis_legodo_update.cpp:190-193):
 * n_contacts = {num_left_foot_contacts, num_right_foot_contacts} for every filter (PB_HOST_BROADCAST) or [2][B]
 * (PB_HOST / PB_DEVICE); kept until the next call; before the first call both are -1 (rbis_legodo_update.cpp:100-101). */
int pb_legodo_set_control_contacts(pb_ctx *ctx, const int32_t *n_contacts, int mem);
/* Forward kinematics table: what kdl_parser + KDL::TreeFkSolverPosFull_recursive make of the URDF for the two standing links
 * (leg_estimate.cpp:68-73,430-447; state_estimator.legodo.{left,right}_standing_link).  KDL and the URDF are not part of the
 * reference tree, so the caller supplies, for the joints from the root link down to each standing link (left chain first,
 * n_left + n_right entries): joint_type (0 fixed, 1 revolute / continuous, 2 prismatic), joint_row (row of the joint-position
 * block that holds the joint, i.e. its index in joint_state_t.joint_name), origin_xyz_rpy [.][6] (the URDF <origin>),
 * axis [.][3] (the URDF <axis>, normalised here like KDL::Joint does), adjustment_gain [.] or NULL
 * (state_estimator.legodo.adjustment_gain of that joint, EstimateTools::TorqueAdjustment: 0, inf or nan = none).
 * body_to_foot = prod_j [R(rpy_j), xyz_j] * Rot(axis_j, position_j).  At most 8 joints per leg. */
int pb_legodo_set_chain(pb_ctx *ctx, int n_left, int n_right, const int *joint_type, const int *joint_row,
                        const double *origin_xyz_rpy, const double *axis, const float *adjustment_gain);
/* leg_estimate::updateOdometry from a joint state: joint_position [n_rows][B] floats (bot_core::joint_state_t carries
 * floats), joint_effort [n_rows][B] or NULL (torque adjustment off, rbis_legodo_update.cpp:236-238), forces [2][B] floats
 * (FootSensing::force_z).  PB_HOST_BROADCAST: [n_rows] / [2] values of ONE robot.  imu_block != NULL: slaved to the
 * orientation after that IMU step, as pb_legodo_update_after_predict.  position_out [3][B] + position_status_out [B]
 * (DEVICE, may be NULL): the pelvis position leg_estimate derives from the foot it last put down
 * (getLegOdometryWorldConstraint, leg_estimate.cpp:299-318,461-492: world_to_body_constraint_ and its _init_ flag), which
 * LegOdoCommon's mode pos_and_lin_rate measures; it is tracked from the first call that passes position_out.
 * Everything else as pb_legodo_update. */
int pb_legodo_update_joints(pb_ctx *ctx, const double *imu_block, int imu_mem, int64_t utime, int n_rows,
                            const float *joint_position, const float *joint_effort, const float *forces, int mem, int zero_delta,
                            double r_vxyz, double r_vxyz_uncertain, double *delta_out, double *status_out, double *lo_block_out,
                            uint8_t *mask_out, double *position_out, uint8_t *position_status_out);
/* ONE call per IMU + joint-state (or foot-state) message pair -- RBISIMUProcessStep::updateFilter, leg_estimate::updateOdometry
 * slaved to the head AFTER that step, LegOdoCommon::createMeasurement (the mode of pb_legodo_set_measurement_mode; default
 * lin_rate) and RBISIndexedMeasurement::updateFilter (rbis_update_interface.cpp:30-95, rbis_legodo_update.cpp:206-280) -- and ONE
 * kernel and one round trip of the filter state: k_step_leg for 15 states up to 393 216 filters, k_step_quad_leg for 21; other
 * contexts run pb_legodo_update_joints(imu_block, ...) followed by pb_step_legodo_split (lin_rate) or pb_predict and
 * pb_update_indexed (the six-row modes) internally.  Results equal that sequence (tests: bit-identical statuses, posterior to
 * rounding).  The six-row modes are applied as their two 3-row blocks with ONE summed correction -- R is diagonal, so that IS the
 * six-row update (equal to rounding, not bit for bit: the 6 x 6 S is never factored); mode 2's per-filter fall-back to lin_rate
 * is the velocity block alone.  imu_block / imu_mem and q as pb_step_legodo; the leg inputs as pb_legodo_update_joints /
 * pb_legodo_update; not both groups PB_HOST.  lo_block_out + mask_out (DEVICE, or both NULL; [6][B] + [B], six-row modes [12][B] +
 * [B] / [2][B]): the measurement that was applied, for a caller that may have to re-apply this update later (history replay) --
 * the odometry itself must not run twice. */
int pb_step_legodo_joints(pb_ctx *ctx, const double *imu_block, int imu_mem, const double q[4], int64_t utime, int n_rows,
                          const float *joint_position, const float *joint_effort, const float *forces, int mem, double r_vxyz,
                          double r_vxyz_uncertain, double *lo_block_out, uint8_t *mask_out);
int pb_step_legodo_feet(pb_ctx *ctx, const double *imu_block, int imu_mem, const double q[4], int64_t utime, const double *feet,
                        const double *forces, int mem, double r_vxyz, double r_vxyz_uncertain, double *lo_block_out,
                        uint8_t *mask_out);
/* The joint-position filters in front of the kinematics -- step 0 of leg_estimate::updateOdometry (leg_estimate.cpp:411-428,
 * state_estimator.legodo.filter_joint_positions): mode 1 = LowPassFilter (estimate_tools filter_tools/Filter.cpp:4-65: 14-tap
 * FIR, the first sample fills the window), mode 2 = SimpleKalmanFilter (kalman_filter_tools/simple_kalman_filter.cpp:11-50;
 * the three noise values are its constructor arguments -- leg_estimate.cpp:56 passes (joint_process_noise,
 * joint_observation_noise) and leaves the third at 5e-4 -- kept as floats like its members).  One filter per robot and joint
 * ROW THE KINEMATIC CHAINS READ (pb_legodo_set_chain first; setting another chain invalidates the filters) with row < 28
 * (NUM_FILT_JOINTS, leg_estimate.hpp:59); pb_joint_filter_init starts every filter over.
 * pb_joint_filter: one joint_state_t message per robot, [n_rows][B] floats (PB_HOST / PB_DEVICE; joint_position_out is a
 * DEVICE array [n_rows][B] to be handed to pb_legodo_update_joints / pb_step_legodo_joints as PB_DEVICE) or ONE robot's
 * [n_rows] values (PB_HOST_BROADCAST; joint_position_out is a HOST array [n_rows] to be passed on as PB_HOST_BROADCAST).
 * joint_velocity: read by the Kalman filter's first sample only (may be NULL for the low-pass).  joint_effort != NULL: the
 * handler's torque adjustment (rbis_legodo_update.cpp:231-241, gains of pb_legodo_set_chain) is applied BEFORE the filters
 * as in the reference -- pass joint_effort = NULL to the kinematics afterwards.  Filtered positions are rounded to float
 * like the reference's std::vector<float> (leg_estimate.hpp:91-93); rows no filter owns are copied. */
int pb_joint_filter_init(pb_ctx *ctx, int mode, double process_noise_pos, double process_noise_vel, double observation_noise);
int pb_joint_filter(pb_ctx *ctx, int64_t utime, int n_rows, const float *joint_position, const float *joint_velocity,
                    const float *joint_effort, int mem, float *joint_position_out);
/* forward kinematics alone (diagnostics, tests): feet_out [14][B] DEVICE array in pb_legodo_update's layout */
int pb_legodo_fk(pb_ctx *ctx, int n_rows, const float *joint_position, const float *joint_effort, int mem, double *feet_out);
/* one filter's odometry state, for diagnostics and tests: odom_to_body (t3, q4); info = primary_foot (0 left, 1 right),
 * leg_odo_init, walking-phase mode (foot_contact_classify.hpp:34-44), transitions the classifier did not know */
int pb_legodo_get(pb_ctx *ctx, int filter, double odom_to_body[7], int64_t info[4]);

/* ---- IMU front end of the Atlas path (InsHandler::doFilter, sensor_handlers.cpp:29-42,154-162) ---------------- */

/* Three cascaded 2nd-order IIR notches (iir_notch.cpp:3-61) at notch_freq * 2^i, i = 0..2, per accelerometer axis and
 * per filter.  pb_imu_notch_init allocates/zeroes the per-filter filter state and fixes the coefficients
 * (state_estimator.ins.atlas_filter_freq, fs = 1000 in the reference). */
int pb_imu_notch_init(pb_ctx *ctx, double notch_freq, double fs);
/* Filters n_packets consecutive NEW packets (oldest first) accel_packets [n_packets][3][B]; accel_out [3][B] receives
 * the newest filtered sample (what processMessageAtlas feeds to the process step, sensor_handlers.cpp:191-196).
 * accel_out must be device memory when mem == PB_DEVICE, host memory when PB_HOST. */
int pb_imu_notch(pb_ctx *ctx, int n_packets, const double *accel_packets, double *accel_out, int mem);
/* The same with a PER-FILTER packet count -- independent log segments: every filter has its own KVH stream and its own
 * IMUStream de-duplication (imu_stream.cpp:62-98), so a batched message carries a different number of NEW packets per filter.
 * counts [B] (int32): filter b runs its first min(counts[b], max_packets) packets of accel_packets [max_packets][3][B] (oldest
 * first) through its cascade; a filter with no new packet keeps its filter state and its accel_out entries are not written
 * (PB_HOST: they come back as 0) -- the reference returns NULL for such a message (sensor_handlers.cpp:181-187). */
int pb_imu_notch_counts(pb_ctx *ctx, int max_packets, const int32_t *counts, const double *accel_packets, double *accel_out, int mem);

/* InsHandler's per-message arithmetic for B robots at once, on the device: one robot's IMU sample -> the [7][B] block that
 * pb_predict / pb_step_legodo* take (gyro xyz | accelerometer xyz | dt, body frame), written to imu_block_out (DEVICE memory).
 *   Microstrain (InsHandler::processMessage, sensor_handlers.cpp:96-131): gyro [3][B] and accel [3][B] rotated by rot_quat
 *     (bot_quat_rotate_to), dt = dt_default.  raw_dt = NULL, trans_vec = NULL, dt_from_utimes = 0.
 *   Atlas KVH (processMessageAtlas, :199-252): gyro holds delta_rotation and is divided by raw_dt [B] (:207-210); accel goes through
 *     the whole ins_to_body transform (rotation + trans_vec, bot_trans_apply_vec :227); dt_from_utimes = 1: dt = (utime - this
 *     filter's previous utime) * 1E-6, dt_default on its first message (:239-249).  utimes [B] = every robot's own message time,
 *     or NULL = `utime` for all.
 *   valid [B] (or NULL = all): 0 = this filter has NO message (its log segment has ended, or its KVH batch carried no new packet:
 *     the reference's handler returns NULL) -- its block entry is its own last sample with dt = 0; its previous-utime is not
 *     advanced.  valid_out (DEVICE, [B], or NULL) receives the mask: hand it to pb_set_imu_valid in front of the call that takes the
 *     step, which makes that step a no-op for these filters.
 * mem (of the input arrays): PB_HOST or PB_DEVICE.  pb_ins_body_reset forgets the last samples / previous utimes. */
int pb_ins_body_block(pb_ctx *ctx, const double *gyro, const double *accel, const double *raw_dt, const int64_t *utimes, int64_t utime,
                      const uint8_t *valid, const double rot_quat[4], const double trans_vec[3], double dt_default, int dt_from_utimes,
                      int mem, double *imu_block_out, uint8_t *valid_out);
int pb_ins_body_reset(pb_ctx *ctx);
/* Filters WITHOUT an IMU message in the next IMU step (independent log segments: the reference's handler returned NULL for them, so
 * its estimator did nothing at all).  valid_dev [B] (DEVICE; 0 = no message) applies to the NEXT call that takes an IMU step
 * (pb_predict, pb_step_legodo, _split, _correct, pb_step_legodo_joints / _feet) and is consumed by it whatever its outcome.  A
 * step with dt = 0 already leaves pose, velocity, biases and covariance alone whatever the sample; with the mask the library
 * replaces, right in front of the step kernel, the sample of such a filter by the one that reproduces its angular-velocity and
 * acceleration entries too (omega + gyro bias, a + accel bias of the prior), and sets its dt to 0: the step is a no-op for it --
 * exactly for 15 states, to the last bit of those sums for 21 -- and a measurement update in the same kernel sees the state the
 * reference's would.  The step kernels themselves know nothing of it.  The array must stay valid until the step has run. */
int pb_set_imu_valid(pb_ctx *ctx, const uint8_t *valid_dev);

/* ---- posterior checkpoints for roll-forward replay (mav_state_est.cpp:28-80, update_history.cpp) ------------ */

/* The reference keeps every update's posterior (RBIS + RBIM, 3.7 KB) in a multimap so that a delayed measurement can
 * be inserted at its timestamp and everything after it re-applied.  Here the posterior of the WHOLE batch after a
 * chosen update is saved device-to-device into one of n_slots checkpoint slots (each = the full state array:
 * 73 MB for 64k 15-state filters -- sized for 288 GB of HBM); the time-ordered bookkeeping and the replay loop live
 * on the host (MavStateEstimator in mav_state_est_batch.hpp).  pb_history_reserve may be called again to resize
 * (contents are lost). */
int pb_history_reserve(pb_ctx *ctx, int n_slots);
int pb_state_save(pb_ctx *ctx, int slot);      /* slot <- head posterior (state, cov, loglik) */
/* Checkpoint without a copy: the NEXT update (pb_predict / pb_update_indexed* / pb_step_legodo) reads the head and
 * writes its posterior straight into checkpoint `slot`, which then IS the head (pb_state_save(slot) becomes a no-op).
 * The step moves the same bytes either way, so a forward pass that keeps every posterior -- what the reference's
 * history does by value (mav_state_est.cpp:55-61) and what the smoother needs -- costs no more than one that keeps none.
 * A saved posterior is never modified afterwards: an update that finds the head in a slot and has no output slot of its
 * own writes back into the context's array; pb_state_restore and pb_reset always land there.  slot = -1 cancels. */
int pb_set_output_slot(pb_ctx *ctx, int slot);
/* the checkpoint slot the head currently lives in, or -1 (the context's own array) */
int pb_head_slot(const pb_ctx *ctx);
int pb_state_restore(pb_ctx *ctx, int slot);   /* head posterior <- slot */

/* ekfSmoothingStep (rbis.cpp:234-266), one backward step of MavStateEstimator::EKFSmoothBackwardsPass
 * (mav_state_est.cpp:98-189), on checkpoint slots:
 *   slot_next_pred = posterior of the INS update at k+1 (prediction), slot_next = smoothed (or, for the last step,
 *   filtered) posterior at k+1, slot_cur = filtered posterior at k; slot_out <- smoothed posterior at k.
 * slot_out may be slot_cur (in place) but not one of the k+1 slots.  dt as passed to EKFSmoothBackwardsPass.
 * P^-_{k+1} is factorised WITHOUT the reference's diagonal pivot search (it is SPD; results agree with Eigen's .ldlt() to rounding,
 * tests: <= 1e-9 against the oracle, and as close to exact arithmetic as the pivoted oracle up to a condition number of 1e10).
 * Default kernels: 15 states k_smooth_wide (rbis_smooth_wide.hpp: persistent workgroups, rows through LDS), 21 states k_smooth_lane
 * (rbis_smooth_lane.hpp).  Environment, read once per process: PRONTO_SMOOTH_KERNEL=lane selects k_smooth_lane for 15 states as well,
 * =reg the kernel of rounds 2-4 (16 / 32 lanes per filter); PRONTO_SMOOTH_PIVOT=1 that kernel with Eigen's pivoting. */
int pb_smooth_step(pb_ctx *ctx, int slot_next_pred, int slot_next, int slot_cur, int slot_out, double dt);

/* EKFSmoothBackwardsPass over a WHOLE log with bounded memory (mav_state_est.cpp:98-189, lcm_front_end.cpp:168-203: the reference's
 * "-S" smooths the entire log; it keeps three posteriors per step by value, which for a batch is 2 T slots of the whole state).
 * Checkpoint and recompute: the forward pass filters the n_steps steps of the streams (as pb_run_legodo takes them; per step the
 * process step, then the velocity update -- the per-message kernels of pb_predict / pb_update_indexed) from the current head and
 * keeps only the posterior in front of every `stride`-th step; the backward pass re-runs the log stretch by stretch, newest
 * first, into a window of 2 * stride slots and applies pb_smooth_step to every step but the newest.  It uses the checkpoint slots
 * [first_slot, first_slot + pb_smooth_log_slots(n_steps, stride)) = n_steps / stride + 2 * stride + 4 slots instead of 2 * n_steps
 * (stride near sqrt(n_steps / 2) is the minimum).  sink(user, step, slot) is called, newest step first, once the smoother step
 * of `step` has been ENQUEUED: `slot` holds the smoothed posterior of that step until the NEXT call of the sink returns; read it with
 * stream-ordered calls that leave the head alone (pb_get_slot, pb_snapshot_from_slot).  The smoothed posteriors are, bit for bit,
 * those of the all-checkpoints pass (pb_set_output_slot per update + pb_smooth_step).  Afterwards the head is the newest filtered
 * posterior.  dt as pb_smooth_step.  elapsed_ms != NULL: device time of the whole call (forward + backward). */
typedef void (*pb_smooth_sink)(void *user, int step, int slot);
int pb_smooth_log_slots(int n_steps, int stride);
int pb_smooth_log(pb_ctx *ctx, int n_steps, int stride, const double *imu_stream, const double *lo_stream, const uint8_t *mask_stream,
                  const double q[4], double dt, int first_slot, pb_smooth_sink sink, void *user, float *elapsed_ms);

/* ---- estimator queries (mav_state_est.hpp:20-22) -------------------------------------------------------- */

/* MavStateEstimator::getHeadState + getMeasurementsLogLikelihood for filters [first, first+count):
 * vec_out [n][count], quat_out [4][count], cov_out [n*n][count] full column-major (as rbis.cpp:300 Map<RBIM>),
 * ll_out [count]; any of them may be NULL. */
int pb_get_head(pb_ctx *ctx, int first, int count, double *vec_out, double *quat_out, double *cov_out,
                double *ll_out, int mem);
/* rbisCreateFilterStateMessageCPP (rbis.cpp:287-304) for one filter: quat[4], state[21], cov[441] (host). */
/* the same read of a posterior that lives in checkpoint slot `slot` (the head stays what it is) */
int pb_get_slot(pb_ctx *ctx, int slot, int first, int count, double *vec_out, double *quat_out, double *cov_out, double *loglik_out,
                int mem);
int pb_get_filter_state(pb_ctx *ctx, int filter, double quat[4], double state[21], double cov[441]);
/* per-shard run summary for the end-of-run all-reduce:
 * out[0] = sum loglik, out[1] = sum |vec| + |quat| (checksum), out[2] = max | |quat|^2 - 1 |,
 * out[3] = number of non-finite state entries. */
int pb_summary(pb_ctx *ctx, double out[4]);
/* How many filters a DEVICE-resident update mask [B] lets through (synchronises).  The reference's handlers return NULL when a
 * message yields no update (rbis_legodo_update.cpp:242-255, rbis_fovis_update.cpp:160-164), so such a message never enters the
 * history; a batched handler whose per-filter validity is decided on the device cannot know that when it returns.  The shim asks
 * lazily -- only where the difference is observable: FovisHandler's history.updateMap.lower_bound look-up
 * (rbis_fovis_update.cpp:184-207) skips updates that applied to no filter. */
int pb_mask_count(pb_ctx *ctx, const uint8_t *mask_dev, int *count_out);
/* Bit-level checksum of the whole device state (every component of every filter, padding lanes of the last tile included):
 * out[0] a weighted wrapping sum, out[1] a xor of rotated words; slot < 0 = the head, else a checkpoint slot.  Two runs of
 * the same context over the same inputs must give the same pair (tests: replay identity over thousands of launches);
 * one pass over the state, ~15 us at 64k filters. */
int pb_state_checksum(pb_ctx *ctx, int slot, uint64_t out[2]);
/* Measurement aid (not on the reference path): `reps` plain copies of the whole state array with the step kernels' exact
 * access pattern (one tile row = 64 lanes x 16 bytes per buffer_load / buffer_store_dwordx4), to calibrate rocprofv3's
 * FETCH_SIZE / WRITE_SIZE on a known byte count and to measure this box's achievable copy rate.  elapsed_ms = HIP-event
 * time of the reps.  pb_calib_copy_checksum: the same copies, then pb_state_checksum's pair OF THE COPY (must equal the
 * head's: the regression test of the 16-byte store path, tests/test_gpu_edge_cases.py). */
int pb_calib_copy(pb_ctx *ctx, int reps, float *elapsed_ms);
int pb_calib_copy_checksum(pb_ctx *ctx, int reps, uint64_t out[2]);
/* head utime bookkeeping (posterior_state.utime = update->utime, mav_state_est.cpp:60) */
int pb_set_utime(pb_ctx *ctx, int64_t utime);
int64_t pb_get_utime(const pb_ctx *ctx);

/* library identification: "pronto_batch <version> gfx950" */
const char *pb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PRONTO_BATCH_H */
