/* leg_sweep.c -- the reference's leg-odometry handler path (rbis_legodo_update.cpp:206-280: torque adjustment, joint filters,
 * KDL forward kinematics, leg_estimate::updateOdometry, LegOdoCommon's lin_rate measurement, the indexed update) as a
 * parameter sweep over the IMU process noise, in plain C on the C ABI: ONE robot's IMU + joint-state log (PB_HOST_BROADCAST)
 * drives every filter of the batch; each filter has its own (q_gyro, q_accel); one call -- one kernel -- per message pair.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/leg_sweep.c -Lpronto_amd/lib -lpronto_batch -lm -o leg_sweep
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pronto_batch.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != PB_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pb_last_error(ctx));                \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

static uint64_t rng = 0x1234567887654321ULL;
static double urand(void)
{
  rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
  return ((rng >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }
static double ramp(double x) { return x < 0 ? 0 : (x > 0.05 ? 1.0 : x / 0.05); }

int main(void)
{
  enum { NG = 32, NA = 32, B = NG * NA, T = 1500, N = 15, NJ = 12 };
  pb_ctx *ctx = NULL;
  if (pb_create(&ctx, N, B, 0, 0) != PB_OK) {
    fprintf(stderr, "pb_create: %s\n", pb_last_error(NULL));
    return 2;
  }
  double x0[N] = { 0 }, q0[4] = { 1, 0, 0, 0 }, P0[N * N] = { 0 };
  x0[11] = 0.86; /* pelvis height */
  for (int i = 3; i < 12; i++) P0[i * N + i] = (i < 6) ? 0.0225 : (i < 9 ? 0.0027 : 0.25);
  CHECK(pb_reset(ctx, x0, q0, P0, 1, PB_HOST));

  /* leg_estimate's parameters (leg_estimate.cpp:93-121) and the two kinematic chains: hip yaw / roll / pitch, knee, ankle pitch /
   * roll per leg (test values; pb_legodo_set_chain takes what kdl_parser reads out of the URDF: <origin xyz rpy>, <axis>) */
  CHECK(pb_legodo_init(ctx, 475.0, 525.0, 7000, 7000, 1));
  int type[2 * 6], row[2 * 6];
  double org[2 * 6 * 6] = { 0 }, axis[2 * 6 * 3] = { 0 };
  float gain[2 * 6];
  const double xyz[6][3] = { { 0, 0.089, 0 }, { 0, 0, 0 }, { 0.05, 0.0225, -0.066 }, { -0.05, 0, -0.374 }, { 0, 0, -0.422 }, { 0, 0, 0 } };
  const int ax[6] = { 2, 0, 1, 1, 1, 0 };
  for (int side = 0; side < 2; side++)
    for (int j = 0; j < 6; j++) {
      const int k = 6 * side + j;
      type[k] = 1; /* revolute */
      row[k] = k;  /* row of the joint-position block */
      for (int i = 0; i < 3; i++) org[6 * k + i] = xyz[j][i];
      if (side) org[6 * k + 1] = -org[6 * k + 1];
      axis[3 * k + ax[j]] = 1.0;
      gain[k] = (j == 3) ? 10000.0f : 0.0f; /* torque adjustment on the knees (rbis_legodo_update.cpp:29-53) */
    }
  CHECK(pb_legodo_set_chain(ctx, 6, 6, type, row, org, axis, gain));
  CHECK(pb_joint_filter_init(ctx, 1 /* lowpass */, 0.01, 5e-4, 5e-4)); /* state_estimator.legodo.filter_joint_positions */

  /* the candidates: q_gyro x q_accel */
  double *qblk = malloc(sizeof(double) * 4 * B);
  for (int b = 0; b < B; b++) {
    const double qg = 0.05 * pow(1.2, b / NA) * M_PI / 180.0, qa = 0.02 * pow(1.2, b % NA);
    qblk[b] = qg * qg; qblk[B + b] = qa * qa; qblk[2 * B + b] = 0; qblk[3 * B + b] = 0;
  }
  void *d_q;
  CHECK(pb_malloc(ctx, sizeof(double) * 4 * B, &d_q));
  CHECK(pb_memcpy_h2d(ctx, d_q, qblk, sizeof(double) * 4 * B));
  CHECK(pb_set_process_noise_block(ctx, d_q));

  /* one robot walks: 500 Hz IMU + joint-state pairs */
  const double g = 9.80665, dt = 0.002, period = 1.1, swing = 0.25;
  const double q_unused[4] = { 0, 0, 0, 0 };
  for (int k = 0; k < T; k++) {
    const int64_t utime = 1000000 + (int64_t) (k + 1) * 2000;
    const double t = (k + 1) * dt;
    double ph = t / period;
    ph -= floor(ph);
    double wl = ramp(ph) * ramp(0.6 - ph), wr = ramp(ph - 0.5) * ramp(1.1 - ph) + (ph < 0.1 ? ramp(0.1 - ph) : 0.0);
    if (t < 0.4) wl = wr = 1.0;
    const double sw = sin(2 * M_PI * ph);
    float jp[NJ], jv[NJ] = { 0 }, je[NJ], jf[NJ], ff[2] = { (float) fabs(900 * wl + 5 * nrand()), (float) fabs(900 * wr + 5 * nrand()) };
    for (int side = 0; side < 2; side++) {
      const double sgn = side ? -1.0 : 1.0, lift = fmax(0.0, -sgn * sw);
      float *p = jp + 6 * side;
      p[0] = (float) (0.05 * sgn * sw);
      p[1] = (float) (0.03 * sgn + 0.02 * sw);
      p[2] = (float) (-0.35 - sgn * swing * sw - 0.2 * lift);
      p[3] = (float) (0.7 + 0.5 * lift);
      p[4] = (float) (-0.35 + sgn * swing * sw * 0.5 - 0.3 * lift);
      p[5] = (float) (-0.03 * sgn - 0.02 * sw);
    }
    for (int j = 0; j < NJ; j++) { jp[j] += (float) (0.002 * nrand()); je[j] = (float) (40 * nrand()); }
    const double imu[7] = { 0.01 * nrand(), 0.01 * nrand(), 0.01 * nrand(), 0.2 * nrand(), 0.2 * nrand(), g + 0.2 * nrand(), dt };
    /* torque adjustment + joint filters (one robot: on the host, output [NJ]), then IMU step + odometry + update: one kernel */
    CHECK(pb_joint_filter(ctx, utime, NJ, jp, jv, je, PB_HOST_BROADCAST, jf));
    CHECK(pb_step_legodo_joints(ctx, imu, PB_HOST_BROADCAST, q_unused, utime, NJ, jf, NULL, ff, PB_HOST_BROADCAST, 5.0, 10.0, NULL, NULL));
  }
  double sum[4], pose[7], *ll = malloc(sizeof(double) * B);
  int64_t info[4];
  CHECK(pb_summary(ctx, sum));
  CHECK(pb_get_head(ctx, 0, B, NULL, NULL, NULL, ll, PB_HOST));
  CHECK(pb_legodo_get(ctx, B - 1, pose, info));
  int best = 0;
  for (int b = 1; b < B; b++)
    if (ll[b] > ll[best]) best = b;
  printf("%d candidates x %d message pairs (%s): best log-likelihood %.1f at q_gyro index %d, q_accel index %d; non-finite entries %.0f; "
         "odometry of the last filter: pelvis at (%.3f, %.3f, %.3f), standing on the %s foot\n",
         B, T, pb_hot_kernel(ctx), ll[best], best / NA, best % NA, sum[3], pose[0], pose[1], pose[2], info[0] ? "right" : "left");
  /* the odometry has walked (several foot changes move the pelvis), every state is finite, and the likelihood separates the
   * candidates */
  double lo_ = ll[0], hi_ = ll[0];
  for (int b = 1; b < B; b++) { lo_ = fmin(lo_, ll[b]); hi_ = fmax(hi_, ll[b]); }
  const int ok = sum[3] == 0 && info[1] == 1 && isfinite(pose[0]) && fabs(pose[0]) + fabs(pose[1]) > 1e-3 && hi_ - lo_ > 1.0;
  printf(ok ? "PASS\n" : "FAIL\n");
  pb_free(ctx, d_q);
  pb_destroy(ctx);
  free(qblk); free(ll);
  return ok ? 0 : 1;
}
