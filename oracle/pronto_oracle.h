/*
 * pronto_oracle.h -- CPU restatement of Pronto's RBIS/RBIM EKF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (pronto_amd/, include/) never links it.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
 *  and cannot be compiled here (Eigen3, eigen_utils, lcm, libbot2 absent), so this
 * restatement is pinned only by (i) analytic known-answer tests and (ii) an independent numpy
 * restatement (oracle/numpy_restatement.py) agreeing to <=1e-12.
 *
 * Dense, op-for-op with the reference (all paths relative to /root/reference):
 *   state-estimator/src/mav_state_est/rbis.cpp:12-227          filter maths
 *   state-estimator/src/mav_state_est/rbis_update_interface.cpp:23-107   update objects
 *   state-estimator/src/mav_state_est/sensor_handlers.cpp:612-724        scan-matcher mapping
 *   motion_estimate/src/mav_est_legodo/rbis_legodo_common.cpp:34-169     leg-odometry measurement
 *   motion_estimate/src/mav_est_fovis/rbis_fovis_update.cpp:199-305      VO pose composition
 *   pronto-utils/src/conversions/pronto_conversions_lcm.hpp:38-87        getDeltaAsVelocity
 *   pronto-utils/src/pronto_math/pronto_math.cpp:25-61                   euler<->quat
 *
 * Third-party semantics NOT IN THE REFERENCE TREE (un-vendored pod `eigen-utils`, no version pinned;
 * Eigen3 for Quaternion/AngleAxis/LDLT/determinant) are restated from their published algorithms and
 * are explicit, overridable constants here (po_set_constants):
 *   g_vec       = -g * z_hat, g = 9.80665   (eigen_utils `g_val`; SURVEY recalled 9.8 -- see DESIGN.md)
 *   chiToQuat() folds chi into quat when |chi| > 1e-6, else leaves chi in vec
 *   addState(d): vec += d.vec; chiToQuat(); quat = quat * d.quat
 *   RigidBodyState(vec) ctor: quat = I; chiToQuat()
 *   subtractQuats(q1,q2) = axis*wrap_pi(angle) of q2^-1 * q1
 */
#ifndef PRONTO_ORACLE_H
#define PRONTO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_N 21 /* RBIS::rbis_num_states, rbis.hpp:23 */

/* state vector layout (eigen_utils::RigidBodyState enum + rbis.hpp:22-24) */
enum {
  PO_ANGVEL = 0, PO_VEL = 3, PO_CHI = 6, PO_POS = 9, PO_ACC = 12, PO_BASIC = 15,
  PO_GYRO_BIAS = 15, PO_ACCEL_BIAS = 18
};

typedef struct {
  double vec[PO_N];
  double quat[4]; /* w, x, y, z */
  int64_t utime;
} po_rbis;

typedef struct {
  double m[PO_N * PO_N]; /* column-major, as Eigen's default and rbis.cpp:300 Map<RBIM> */
} po_rbim;

void po_set_constants(double g, double chi_tol);
void po_get_constants(double *g, double *chi_tol);

/* ---- eigen_utils / Eigen primitives ---- */
void po_rbis_zero(po_rbis *s);                         /* RBIS(): vec=0, quat=I, utime=0 */
void po_rbis_from_vec(po_rbis *s, const double *vec);  /* RBIS(vec): quat=I then chiToQuat */
void po_chi_to_quat(po_rbis *s);
void po_add_state(po_rbis *s, const po_rbis *d);
void po_subtract_quats(const double *q1, const double *q2, double *out3);
void po_quat_mul(const double *a, const double *b, double *out);
void po_quat_rotate(const double *q, const double *v, double *out);     /* q * v          */
void po_quat_inv_rotate(const double *q, const double *v, double *out); /* q.inverse() * v */
void po_quat_to_rot(const double *q, double *R /* 3x3 row-major */);

/* ---- rbis.cpp ---- */
void po_get_imu_linearization(const po_rbis *state, po_rbim *Ac);                 /* rbis.cpp:12-35  */
void po_ins_update_state(const double *gyro, const double *accel, double dt, po_rbis *state); /* :37-75 */
void po_ins_update_covariance(double q_gyro, double q_accel, double q_gyro_bias, double q_accel_bias,
                              const po_rbis *state, po_rbim *cov, double dt);    /* rbis.cpp:77-122 */
/* rbis.cpp:124-143; R is m x m col-major, C is m x 21 col-major, K is 21 x m col-major */
double po_matrix_measurement_k_dcov(int m, const double *R, const double *C, const po_rbim *cov,
                                    const double *z_resid, po_rbim *dcov, double *K);
double po_indexed_measurement(int m, const double *z, const double *R, const int *idx, const po_rbis *state,
                              const po_rbim *cov, po_rbis *dstate, po_rbim *dcov); /* rbis.cpp:160-178 */
double po_indexed_plus_orientation_measurement(int m, const double *z, const double *quat, const double *R,
                                               const int *idx, const po_rbis *state, const po_rbim *cov,
                                               po_rbis *dstate, po_rbim *dcov);    /* rbis.cpp:189-217 */
void po_apply_delta(const po_rbis *prior, const po_rbim *prior_cov, const po_rbis *dstate, const po_rbim *dcov,
                    po_rbis *post, po_rbim *post_cov);                             /* rbis.cpp:219-227 */
/* rbis.cpp:234-266 (RTS smoother step; SURVEY 8f rank 2: checks pb_smooth_step in tests/test_smoother.py, tests/cpp/test_smooth_pass.cpp) */
void po_ekf_smoothing_step(const po_rbis *next_state_pred, const po_rbim *next_cov_pred, const po_rbis *next_state,
                           const po_rbim *next_cov, double dt, po_rbis *cur_state, po_rbim *cur_cov);

/* ---- rbis_update_interface.cpp: updateFilter() of each update object ---- */
void po_imu_process_step(const double *gyro, const double *accel, double dt, double q_gyro, double q_accel,
                         double q_gyro_bias, double q_accel_bias, const po_rbis *prior, const po_rbim *prior_cov,
                         double prior_ll, po_rbis *post, po_rbim *post_cov, double *post_ll); /* :30-52 */
void po_indexed_update(int m, const int *idx, const double *z, const double *R, const po_rbis *prior,
                       const po_rbim *prior_cov, double prior_ll, po_rbis *post, po_rbim *post_cov,
                       double *post_ll);                                                     /* :54-95 */
void po_indexed_orient_update(int m, const int *idx, const double *z, const double *R, const double *quat,
                              const po_rbis *prior, const po_rbim *prior_cov, double prior_ll, po_rbis *post,
                              po_rbim *post_cov, double *post_ll);                           /* :97-107 */

/* ---- INS initialisation (InsHandler::processMessageInitCommon, sensor_handlers.cpp:283-364, without the GPS/magnetometer
 * yaw branch) ---- */
/* Eigen's Quaternion::setFromTwoVectors(a, b) [Eigen NOT IN TREE, restated from its documentation/algorithm]: the
 * smallest rotation q with q * a parallel to b: v0 = a/|a|, v1 = b/|b|, c = v0.v1, axis = v0 x v1, s = sqrt(2(1+c)),
 * q = (s/2, axis/s).  For c < -1 + 1e-12 Eigen takes the axis from an SVD; here any unit axis orthogonal to v0. */
void po_quat_from_two_vectors(const double *a, const double *b, double *q);
/* g_vec_sum = sum of (-accelerometer), gyro_sum = sum of gyro over `count` body-frame samples (:289-291).
 * quat_out = quat_in * setFromTwoVectors(g_vec_sum/count, -z) (:305-313); gyro_bias_est = gyro_sum/count, or 0 if any
 * component exceeds max_gyro_bias in magnitude (:303-311). */
void po_ins_init(const double *g_vec_sum, const double *gyro_sum, int count, double max_gyro_bias, const double *quat_in,
                 double *quat_out, double *gyro_bias_est);

/* the GPS / magnetometer yaw branch (:338-351): quat_out = setFromTwoVectors((mag_sum/count with z = 0), +y) * quat_in */
void po_ins_init_yaw(const double *mag_vec_sum, int count, const double *quat_in, double *quat_out);

/* ---- measurement formers (handlers' arithmetic) ---- */
void po_euler_to_quat(double roll, double pitch, double yaw, double *q);    /* pronto_math.cpp:25-50 */
void po_quat_to_euler(const double *q, double *rpy);                        /* pronto_math.cpp:53-61 */
/* pronto_conversions_lcm.hpp:38-87: delta (t[3], q[4]) over dt_us -> velocity transform */
void po_delta_as_velocity(const double *t, const double *q, int64_t dt_us, double *t_vel, double *q_vel);
/* rbis_legodo_common.cpp:110-169.  mode: 0 lin_rate, 1 lin_rot_rate, 2 pos_and_lin_rate.
 * r[5] = {r_xyz, r_vxyz, r_vang, r_vxyz_uncertain, r_vang_uncertain}.
 * returns m (3 or 6) and fills idx[m], z[m], Rdiag[m]. */
int po_legodo_create_measurement(int mode, const double *r, const double *pos_t, const double *delta_t,
                                 const double *delta_q, int64_t utime, int64_t prev_utime, int pos_status,
                                 float delta_status, int *idx, double *z, double *Rdiag);
/* rbis_fovis_update.cpp:199-223,299-305: T1 = T0(pos0,quat0) * (t,q); z = T1.translation, q_meas = T1.rot */
void po_fovis_compose(const double *pos0, const double *quat0, const double *t, const double *q, double *z3,
                      double *q_meas);

/* ---- leg kinematic odometry (oracle/leg_odometry.c): leg_estimate.cpp:172-297,395-556, FootContactAlt.cpp:5-100,
 * foot_contact_classify.cpp:5-125,146-318, SignalTap.cpp:48-134.  Forward kinematics is the caller's. ---- */
typedef struct { int status; long timer, previous_time; int first_call; } po_schmitt;
void po_schmitt_reset(po_schmitt *s);
void po_schmitt_update(po_schmitt *s, double lt, double ht, long low_delay, long high_delay, long present_time, double value);
typedef struct po_leg po_leg;
size_t po_leg_sizeof(void);
void po_leg_init(po_leg *s, double schmitt_low, double schmitt_high, long low_delay, long high_delay, int filter_contact_events);
/* one joint-state message: body-to-foot transforms (t[3], q[4] = w,x,y,z), foot forces, the filter's head orientation
 * (setPoseBody).  Returns leg_estimate::updateOdometry's status (-1 / 0 / 1) and the pelvis increment. */
float po_leg_update(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                    double lforce, double rforce, const double *world_to_body_quat, double *delta_t, double *delta_q,
                    long *prev_utime);
/* contact mode "standing" (FootContact) and the controller-contact override (leg_estimate.cpp:113-121,322-393) */
void po_leg_set_contact_mode(po_leg *s, int standing, double total_force, double standing_schmitt_level, int use_controller_input);
float po_leg_update_cc(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                       double lforce, double rforce, int n_control_contacts_left, int n_control_contacts_right,
                       const double *world_to_body_quat, double *delta_t, double *delta_q, long *prev_utime);
float po_leg_update_wc(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                       double lforce, double rforce, int n_control_contacts_left, int n_control_contacts_right,
                       const double *world_to_body_pos, const double *world_to_body_quat, double *delta_t, double *delta_q,
                       long *prev_utime, double *constraint_pos, int *constraint_ok);
/* forward kinematics of one chain as KDL computes it (leg_estimate.cpp:430-447), torque adjustment (torque_adjustment.cpp) */
void po_fk(int n, const int *type, const double *origin_xyz_rpy, const double *axis, const double *angle, double *t_out, double *q_out);
float po_torque_adjust(float position, float effort, float gain);
void po_leg_get(const po_leg *s, double *odom_to_body_t, double *odom_to_body_q, int *primary_foot, int *leg_odo_init, int *mode,
                int *unknown_transitions);

/* ---- joint-position filters in front of the kinematics (joint_filter.c): leg_estimate.cpp:43-61,411-428,
 * estimate_tools filter_tools/Filter.cpp:4-65, kalman_filter_tools/simple_kalman_filter.cpp:11-50 ---- */
#define PO_LP_TAPS 14
#define PO_NUM_FILT_JOINTS 28
typedef struct {
  double coeffs[PO_LP_TAPS], buf[PO_LP_TAPS];
  int begin, firstsample;
} po_lowpass;
typedef struct {
  double P[2][2], x_est[2], tlast;
  float R, process_noise_pos, process_noise_vel, observation_noise;
  int init;
} po_skf;
void po_lowpass_init(po_lowpass *f);
double po_lowpass_sample(po_lowpass *f, double sample);
void po_skf_init(po_skf *k, double process_noise_pos, double process_noise_vel, double observation_noise);
void po_skf_sample(po_skf *k, double t, double x, double x_dot, double *x_filtered, double *x_dot_filtered);
/* mode 1 lowpass (lp[min(n,28)]), 2 kalman (kf[min(n,28)]): one robot's joint vector, filtered in place */
void po_joint_filter(int mode, po_lowpass *lp, po_skf *kf, long utime, int n, float *joint_position, const float *joint_velocity);

/* ---- IMU front end: estimate_tools/src/estimate_tools/iir_notch.cpp:3-61 (2nd-order IIR notch) ---- */
typedef struct {
  double b[3], a[3]; /* num, den */
  double x[2], y[2]; /* carried inputs / outputs */
} po_notch;
void po_notch_init(po_notch *f, double notch_freq, double fs);   /* iir_notch.cpp:3-32 */
double po_notch_process(po_notch *f, double input);              /* iir_notch.cpp:34-61 */
/* InsHandler::doFilter (sensor_handlers.cpp:154-162): cascade of 3 notches (freq * 2^i) per accelerometer axis.
 * filt is po_notch[3 axes][3 stages]; acc is filtered in place. */
void po_notch_cascade_init(po_notch *filt9, double notch_freq, double fs);
void po_notch_cascade(po_notch *filt9, double *acc3);

/* ---- noise identification: state-estimator/src/noise_id/noise_id.cpp:9-65 ---- */
/* sampleProcessForward for ONE window of N steps (noise_id.cpp:19-40): truth[0..N] are the logged filter states
 * (truth[k].vec holds omega and accel, which drive the roll-forward, :26); start_cov = logged covariance at truth[0].
 * err_out = rolled (-) truth[N] with chi = Log(quat difference) (:37-38); cov_out = rolled_cov - start_window_cov (:40). */
void po_noise_id_window(int N, const po_rbis *truth, const po_rbim *start_cov, double dt, double q_gyro, double q_accel,
                        po_rbis *err_out, po_rbim *cov_out);
/* the pieces of loglike_normalized (eigen_utils [NOT IN TREE]) over the active indices (noise_id.cpp:52-58):
 * log det cov_active and e^T cov_active^-1 e.  Returns -0.5*(m log 2pi + logdet + maha), the usual convention. */
double po_loglike_pieces(int m, const int *idx, const po_rbis *err, const po_rbim *cov, double *logdet, double *maha);

/* ---- batch drivers (fixtures + CPU baseline).  All arrays SoA with the filter index fastest. ----
 * State SoA: vec[21][B], quat[4][B], cov[441][B] (col-major index c*21+r), ll[B].
 * IMU block per step: gyro[3][B], accel[3][B], dt[B]  (7*B doubles)
 * legodo block per step: z[3][B], Rdiag[3][B]          (6*B doubles), mask[B] (uint8, 0 = handler returned NULL)
 */
typedef struct {
  int B;
  double *vec;  /* [21][B] */
  double *quat; /* [4][B]  */
  double *cov;  /* [441][B] */
  double *ll;   /* [B] */
} po_batch;

void po_batch_predict(po_batch *s, const double *imu_block, const double *q4 /* qg,qa,qbg,qba */, int nthreads);
void po_batch_update_indexed(po_batch *s, int m, const int *idx, const double *z /* [m][B] */,
                             const double *Rdiag /* [m][B] */, const double *quat_meas /* [4][B] or NULL */,
                             const uint8_t *mask /* [B] or NULL */, int nthreads);
/* T steps of predict + legodo m=3 (idx 3,4,5); returns seconds spent in the step loop */
double po_batch_run_legodo(po_batch *s, int T, const double *imu_stream /* [T][7][B] */,
                           const double *lo_stream /* [T][6][B] */, const uint8_t *mask_stream /* [T][B] or NULL */,
                           const double *q4, int nthreads);
int po_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
