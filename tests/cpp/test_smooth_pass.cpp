// test_smooth_pass.cpp -- MavStateEstimator::EKFSmoothBackwardsPass (mav_state_est.cpp:98-189) through the shim: forward pass
// with a checkpoint on every update, backward pass on the device (pb_smooth_step), every smoothed step compared with the
// oracle's po_ekf_smoothing_step recursion over the oracle's own forward pass.
//   argv[1]: 15 | 21 states    argv[2]: state_estimator.history_checkpoint_every (default 1 = a posterior per update, like the reference).
//   With K > 1 only every K-th update keeps its posterior and the pass re-derives the others stretch by stretch (checkpoint and
//   recompute): a 60-step window in ~1/K of the slots, the same smoothed posteriors.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "../../oracle/pronto_oracle.h"
#include "../../pronto_amd/csrc/mav_state_est_batch.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x777ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = (argc > 1) ? atoi(argv[1]) : 15;
  const int every = (argc > 2) ? atoi(argv[2]) : 1;
  const int B = 20, T = every > 1 ? 60 : 16;
  const double dt = 0.001;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "100000000");
  // every update checkpointed: 2 T + 5 slots; sparse: one per `every` updates, the window of the longest run without one, two for
  // the smoothed posteriors, one for the head, a few spare
  param.set("state_estimator.history_slots", (double) (every > 1 ? (2 * T) / every + every + 8 : 2 * T + 5));
  param.set("state_estimator.history_checkpoint_every", (double) every);
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0);
  for (int b = 0; b < B; b++) {
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    const double sig[21] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0, .008, .008, .008, .1, .1, .1 };
    for (int i = 0; i < n; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
  }
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  const double q4[4] = { 7.6e-5, 0.01, n == 21 ? 3e-10 : 0, n == 21 ? 1e-8 : 0 }, r_lo[3] = { 0.01, 0.01, 0.01 };
  const int vel_idx[3] = { 3, 4, 5 };
  // oracle forward pass keeps pred / filtered per step
  std::vector<std::vector<po_rbis>> pred_x(T), filt_x(T);
  std::vector<std::vector<po_rbim>> pred_P(T), filt_P(T);
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    std::vector<double> imu(7 * B), lo(3 * B);
    for (int b = 0; b < B; b++) {
      for (int i = 0; i < 3; i++) {
        imu[i * B + b] = 0.3 * sin(0.1 * k + b + i);
        imu[(3 + i) * B + b] = 0.3 * nrand() + (i == 2 ? g : 0.0);
        lo[i * B + b] = 0.1 * nrand();
      }
      imu[6 * B + b] = dt;
    }
    est.addUpdate(new RBISIMUProcessStep(std::vector<double>(imu), q4[0], q4[1], q4[2], q4[3], utime), true);
    // every third step has no measurement: the step's filtered posterior is then the INS posterior itself
    const bool meas = (k % 3 != 2);
    if (meas)
      est.addUpdate(new RBISIndexedMeasurement(RBIS::velocityInds(), std::vector<double>(lo), std::vector<double>(r_lo, r_lo + 3),
                                               PB_R_DIAG_BROADCAST, std::vector<uint8_t>(), RBISUpdateInterface::legodo, utime), true);
    pred_x[k].resize(B); filt_x[k].resize(B); pred_P[k].resize(B); filt_P[k].resize(B);
    for (int b = 0; b < B; b++) {
      double gy[3] = { imu[b], imu[B + b], imu[2 * B + b] }, ac[3] = { imu[3 * B + b], imu[4 * B + b], imu[5 * B + b] };
      po_imu_process_step(gy, ac, dt, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      pred_x[k][b] = ox[b]; pred_P[k][b] = oP[b];
      if (meas) {
        double z[3] = { lo[b], lo[B + b], lo[2 * B + b] }, R[9] = { r_lo[0], 0, 0, 0, r_lo[1], 0, 0, 0, r_lo[2] };
        po_indexed_update(3, vel_idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
      filt_x[k][b] = ox[b]; filt_P[k][b] = oP[b];
    }
  }
  // oracle backward recursion
  std::vector<std::vector<po_rbis>> sm_x(T);
  std::vector<std::vector<po_rbim>> sm_P(T);
  sm_x[T - 1] = filt_x[T - 1]; sm_P[T - 1] = filt_P[T - 1];
  for (int k = T - 2; k >= 0; k--) {
    sm_x[k] = filt_x[k]; sm_P[k] = filt_P[k];
    for (int b = 0; b < B; b++)
      po_ekf_smoothing_step(&pred_x[k + 1][b], &pred_P[k + 1][b], &sm_x[k + 1][b], &sm_P[k + 1][b], dt, &sm_x[k][b], &sm_P[k][b]);
  }
  // device backward pass
  double worst = 0;
  int calls = 0;
  const int steps = est.EKFSmoothBackwardsPass(dt, [&](int64_t utime, int slot) {
    const int k = (int) (utime / 1000) - 1;
    pb_state_restore(est.ctx, slot);
    RBIS h; RBIM c;
    est.getHeadState(h, c);
    double ev = 0, sv = 0, eP = 0, sP = 0, eq = 0;
    for (int b = 0; b < B; b++) {
      for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(h(i, b) - sm_x[k][b].vec[i])); sv = fmax(sv, fabs(sm_x[k][b].vec[i])); }
      for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(h.q(i, b) - sm_x[k][b].quat[i]));
      for (int cc = 0; cc < n; cc++)
        for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(c(r, cc, b) - sm_P[k][b].m[cc * 21 + r])); sP = fmax(sP, fabs(sm_P[k][b].m[cc * 21 + r])); }
    }
    worst = fmax(worst, fmax(ev / sv, fmax(eq, eP / sP)));
    calls++;
  });
  printf("n=%d, a checkpoint every %d update(s), %d slots: %d smoothing steps (%d callbacks), %lld updates re-applied to re-derive posteriors, worst rel err vs oracle %.2e\n",
         n, every, est.history_slots, steps, calls, (long long) est.smoother_reapplied_updates, worst);
  const bool ok = steps == T - 1 && calls == T - 1 && worst < 1e-9 && est.last_status == PB_OK &&
                  (every > 1 ? est.smoother_reapplied_updates > T / 2 && est.history_slots < T : est.smoother_reapplied_updates == 0);
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
