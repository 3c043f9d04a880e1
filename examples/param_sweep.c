/* param_sweep.c -- the reference's parameter sweep (state-estimator/python/param_sweep.py:39-52: 8 000 serial
 * `mav-state-estimator` runs, one per override set) as ONE batched context, in plain C on the C ABI.
 *
 * Every filter of the batch replays the same synthetic log segment with its own leg-odometry noise r_vxyz (the
 * measurement covariance is a per-filter input) and its own IMU process noise (pb_set_process_noise_block); the
 * accumulated measurement log-likelihood per variation is what the reference writes with `-M` (lcm_front_end.cpp:133-142).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/param_sweep.c -Lpronto_amd/lib -lpronto_batch -lm -o param_sweep
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

static uint64_t rng = 88172645463325252ULL;
static double urand(void)
{
  rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
  return ((rng >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(void)
{
  enum { NR = 16, NQ = 8, B = NR * NQ, T = 400, N = 15 };
  pb_ctx *ctx = NULL;
  if (pb_create(&ctx, N, B, 0, 0) != PB_OK) {
    fprintf(stderr, "pb_create: %s\n", pb_last_error(NULL));
    return 2;
  }
  /* x0 / P0 broadcast to all variations */
  double x0[N] = { 0 }, q0[4] = { 1, 0, 0, 0 }, P0[N * N] = { 0 };
  for (int i = 3; i < 12; i++) P0[i * N + i] = (i < 6) ? 0.0225 : (i < 9 ? 0.0027 : 0.25);
  CHECK(pb_reset(ctx, x0, q0, P0, 1, PB_HOST));

  /* the log segment: a gentle yaw oscillation while standing; true leg-odometry noise 0.1 m/s */
  const double g = 9.80665, dt = 1e-3, true_r = 0.1;
  double *imu = malloc(sizeof(double) * (size_t) T * 7 * B), *lo = malloc(sizeof(double) * (size_t) T * 6 * B);
  double *qblk = malloc(sizeof(double) * 4 * B);
  for (int k = 0; k < T; k++) {
    double gyro[3] = { 0.01 * nrand(), 0.01 * nrand(), 0.3 * sin(0.01 * k) + 0.01 * nrand() };
    double acc[3] = { 0.1 * nrand(), 0.1 * nrand(), g + 0.1 * nrand() };
    double z[3] = { true_r * nrand(), true_r * nrand(), true_r * nrand() };
    for (int b = 0; b < B; b++) {
      const double r = 0.02 * pow(1.25, b % NR); /* candidate r_vxyz: 0.02 ... 0.57 */
      for (int i = 0; i < 3; i++) {
        imu[((size_t) k * 7 + i) * B + b] = gyro[i];
        imu[((size_t) k * 7 + 3 + i) * B + b] = acc[i];
        lo[((size_t) k * 6 + i) * B + b] = z[i];
        lo[((size_t) k * 6 + 3 + i) * B + b] = r * r;
      }
      imu[((size_t) k * 7 + 6) * B + b] = dt;
    }
  }
  for (int b = 0; b < B; b++) {
    const double qg = 0.5 * pow(1.3, b / NR) * M_PI / 180.0; /* candidate q_gyro (deg/s -> rad/s) */
    qblk[b] = qg * qg; qblk[B + b] = 0.01; qblk[2 * B + b] = 0; qblk[3 * B + b] = 0;
  }
  void *d_imu, *d_lo, *d_q;
  CHECK(pb_malloc(ctx, sizeof(double) * (size_t) T * 7 * B, &d_imu));
  CHECK(pb_malloc(ctx, sizeof(double) * (size_t) T * 6 * B, &d_lo));
  CHECK(pb_malloc(ctx, sizeof(double) * 4 * B, &d_q));
  CHECK(pb_memcpy_h2d(ctx, d_imu, imu, sizeof(double) * (size_t) T * 7 * B));
  CHECK(pb_memcpy_h2d(ctx, d_lo, lo, sizeof(double) * (size_t) T * 6 * B));
  CHECK(pb_memcpy_h2d(ctx, d_q, qblk, sizeof(double) * 4 * B));
  CHECK(pb_set_process_noise_block(ctx, d_q));

  const double q_unused[4] = { 0, 0, 0, 0 };
  float ms = 0;
  CHECK(pb_run_legodo(ctx, T, d_imu, d_lo, NULL, q_unused, &ms));
  double ll[B];
  CHECK(pb_get_head(ctx, 0, B, NULL, NULL, NULL, ll, PB_HOST));

  int best = 0;
  for (int b = 1; b < B; b++)
    if (ll[b] > ll[best]) best = b;
  const double best_r = 0.02 * pow(1.25, best % NR);
  printf("%d variations x %d steps in %.2f ms (%s); best log-likelihood %.1f at r_vxyz = %.3f (true %.3f), q_gyro index %d\n", B, T,
         ms, pb_hot_kernel(ctx), ll[best], best_r, true_r, best / NR);
  /* the likelihood must peak near the noise the data was generated with */
  const int ok = best_r > 0.06 && best_r < 0.17;
  printf(ok ? "PASS\n" : "FAIL\n");
  pb_free(ctx, d_imu); pb_free(ctx, d_lo); pb_free(ctx, d_q);
  pb_destroy(ctx);
  free(imu); free(lo); free(qblk);
  return ok ? 0 : 1;
}
