/* joint_filter.c -- CPU restatement (TEST INFRASTRUCTURE, parity unpinned like the rest of oracle/: the reference cannot be
 * built here and holds no fixtures for these classes) of the two joint-position filters leg_estimate::updateOdometry runs in
 * front of the kinematics (motion_estimate/src/leg_estimate/leg_estimate.cpp:43-61,411-428):
 *   estimate_tools/src/filter_tools/Filter.cpp:4-65                         LowPassFilter
 *   estimate_tools/src/kalman_filter_tools/simple_kalman_filter.cpp:11-50   SimpleKalmanFilter (hpp:27-47: which members are float)
 * Written the way the reference states them -- an object per joint with a circular sample buffer; 2 x 2 matrices and
 * generic matrix products in Eigen's coefficient order (a(i,0) b(0,j) + a(i,1) b(1,j)), every product with a 0 or 1 entry
 * of F, Hk and I carried out -- i.e. NOT with the hand-expanded expressions of pronto_amd/csrc/rbis_jointfilt.hpp it checks.
 * Build without contraction into fused multiply-adds (gcc on x86-64 without -mfma does not contract). */
#include <string.h>

#include "pronto_oracle.h"

/* ---- LowPassFilter ---- */
void po_lowpass_init(po_lowpass *f)
{
  static const double taps[PO_LP_TAPS] = { 0.005271208909706, 0.05204636786996, 0.05315761628452, 0.07562063364867,
                                           0.09406855250555,  0.108343855546,   0.1160610649931,  0.1160610649931,
                                           0.108343855546,    0.09406855250555, 0.07562063364867, 0.05315761628452,
                                           0.05204636786996,  0.005271208909706 };
  double sum = 0; /* "the above values dont sum to 1, re-normalize here" (Filter.cpp:26-36) */
  for (int i = 0; i < PO_LP_TAPS; i++) sum += taps[i];
  for (int i = 0; i < PO_LP_TAPS; i++) f->coeffs[i] = taps[i] / sum;
  f->firstsample = 1;
  f->begin = 0;
  for (int i = 0; i < PO_LP_TAPS; i++) f->buf[i] = 0; /* the buffer starts full of zeros (:39-44) */
}
/* boost::circular_buffer<double>::push_back on a full buffer: overwrite the oldest, which becomes the newest */
static void lp_push(po_lowpass *f, double v)
{
  f->buf[f->begin] = v;
  f->begin = (f->begin + 1) % PO_LP_TAPS;
}
double po_lowpass_sample(po_lowpass *f, double sample)
{
  if (f->firstsample) { /* :46-52 */
    f->firstsample = 0;
    for (int i = 0; i < PO_LP_TAPS; i++) lp_push(f, sample);
  }
  lp_push(f, sample);
  double accumulator = 0.;
  for (int i = 0; i < PO_LP_TAPS; i++) /* samples_buf.at(i): i-th oldest (:61-64) */
    accumulator += f->coeffs[PO_LP_TAPS - i - 1] * f->buf[(f->begin + i) % PO_LP_TAPS];
  return accumulator;
}

/* ---- SimpleKalmanFilter ---- */
static void mul22(const double a[2][2], const double b[2][2], double c[2][2])
{
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j];
}
void po_skf_init(po_skf *k, double process_noise_pos, double process_noise_vel, double observation_noise)
{
  memset(k, 0, sizeof *k);
  k->process_noise_pos = (float) process_noise_pos; /* float members (hpp:39-40) */
  k->process_noise_vel = (float) process_noise_vel;
  k->observation_noise = (float) observation_noise;
  k->R = k->observation_noise;
  k->P[0][0] = 1; k->P[1][1] = 1; /* Matrix2d::Identity() (:19) */
}
void po_skf_sample(po_skf *k, double t, double x, double x_dot, double *x_filtered, double *x_dot_filtered)
{
  if (!k->init) { /* :27-34 */
    k->init = 1;
    k->x_est[0] = x; k->x_est[1] = x_dot;
    *x_filtered = x; *x_dot_filtered = x_dot;
    k->tlast = t;
    return;
  }
  const double dt = t - k->tlast;
  const double F[2][2] = { { 1, dt }, { 0, 1 } }, Ft[2][2] = { { 1, 0 }, { dt, 1 } };
  const double Q[2][2] = { { k->process_noise_pos * dt, 0 }, { 0, k->process_noise_vel / dt } };
  const double Hk[2] = { 1, 0 };
  double jprior[2], FP[2][2], FPFt[2][2], Pprior[2][2];
  for (int i = 0; i < 2; i++) jprior[i] = F[i][0] * k->x_est[0] + F[i][1] * k->x_est[1];
  mul22(F, k->P, FP);
  mul22(FP, Ft, FPFt);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) Pprior[i][j] = FPFt[i][j] + Q[i][j];
  const float meas_resid = (float) (x - (Hk[0] * jprior[0] + Hk[1] * jprior[1]));
  double HtP[2]; /* Hk^T Pprior */
  for (int j = 0; j < 2; j++) HtP[j] = Hk[0] * Pprior[0][j] + Hk[1] * Pprior[1][j];
  const float S = (float) ((HtP[0] * Hk[0] + HtP[1] * Hk[1]) + k->R);
  double K[2]; /* ( P*Hk ) / S -- the posterior of the previous step, as the reference has it (:42) */
  for (int i = 0; i < 2; i++) K[i] = (k->P[i][0] * Hk[0] + k->P[i][1] * Hk[1]) / S;
  for (int i = 0; i < 2; i++) k->x_est[i] = jprior[i] + K[i] * meas_resid;
  double M[2][2], Pn[2][2]; /* ( I - K Hk^T ) Pprior */
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) M[i][j] = (i == j ? 1.0 : 0.0) - K[i] * Hk[j];
  mul22(M, Pprior, Pn);
  memcpy(k->P, Pn, sizeof Pn);
  *x_filtered = k->x_est[0];
  *x_dot_filtered = k->x_est[1];
  k->tlast = t;
}

/* leg_estimate.cpp:411-428 on one robot's joint vector (std::vector<float>): the first min(n, 28) joints, in place */
void po_joint_filter(int mode, po_lowpass *lp, po_skf *kf, long utime, int n, float *joint_position, const float *joint_velocity)
{
  const int nf = n < PO_NUM_FILT_JOINTS ? n : PO_NUM_FILT_JOINTS;
  if (mode == 1) {
    for (int i = 0; i < nf; i++) joint_position[i] = (float) po_lowpass_sample(&lp[i], joint_position[i]);
  } else if (mode == 2) {
    for (int i = 0; i < nf; i++) {
      double xf, xdf;
      po_skf_sample(&kf[i], ((double) utime * 1E-6), joint_position[i], joint_velocity[i], &xf, &xdf);
      joint_position[i] = (float) xf;
    }
  }
}
