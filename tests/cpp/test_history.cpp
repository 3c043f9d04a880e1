// test_history.cpp -- delayed / out-of-order measurements through the shim's MavStateEstimator (history window +
// roll-forward replay, mav_state_est.cpp:28-80 / update_history.cpp:16-55).  Position fixes arrive 7 steps late;
// the estimator must end exactly where an in-order pass over the same updates ends (here: the oracle, which
// processes them in timestamp order).  Also: sparse checkpoints give the same answer, too-old updates are discarded.
#include <cinttypes>
#include <cstdio>
#include <deque>
#include <string>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x1234567ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

struct Late {
  int64_t utime;
  int deliver_at_step;
  std::vector<double> z;  // [3][B]
};

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const int B = 96, T = 80; const int DELAY = (argc > 2) ? atoi(argv[2]) : 7;
  const int every = (argc > 1) ? atoi(argv[1]) : 1;
  // argv[3] = "fuse": state_estimator.fuse_ins_legodo with checkpoints -- pairs run as one kernel and are checkpointed behind
  // their second half; a late fix restores a checkpoint and the replay fuses again where the pairs are still adjacent
  const bool fuse = argc > 3 && std::string(argv[3]) == "fuse";
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "30000");  // 30 ms window
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  if (every > 0) {
    param.set("state_estimator.history_slots", every == 1 ? "40" : "12");
    param.set("state_estimator.history_checkpoint_every", (double) every);
  }  // every == 0: ONLY utime_history_span, like a reference .cfg -- the estimator derives its checkpoint pool from it

  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0);
  for (int b = 0; b < B; b++) {
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
  }
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);

  const double q4[4] = { 7.6e-5, 0.01, n == 21 ? 3e-10 : 0.0, n == 21 ? 1e-8 : 0.0 }, r_lo[3] = { 0.01, 0.01, 0.01 }, r_pos[3] = { 4e-4, 4e-4, 4e-4 };
  // the whole timeline is generated up front so the oracle can run it in timestamp order afterwards
  std::vector<std::vector<double>> imu(T), lo(T);
  std::vector<Late> fixes;
  for (int k = 0; k < T; k++) {
    imu[k].resize(7 * B);
    lo[k].resize(3 * B);
    for (int b = 0; b < B; b++) {
      for (int i = 0; i < 3; i++) {
        imu[k][i * B + b] = 0.2 * sin(0.05 * k + b + i) + 0.01 * nrand();
        imu[k][(3 + i) * B + b] = 0.3 * nrand() + (i == 2 ? g : 0.0);
        lo[k][i * B + b] = 0.1 * nrand();
      }
      imu[k][6 * B + b] = 0.001;
    }
    if (k % 10 == 4 && k + DELAY < T) {
      Late f{ (int64_t) (k + 1) * 1000, k + DELAY, std::vector<double>(3 * B) };
      for (auto &v : f.z) v = 0.02 * nrand();
      fixes.push_back(f);
    }
  }
  // ---- arrival order: IMU_k, legodo_k, then any fix whose delivery step is k ----
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    est.addUpdate(new RBISIMUProcessStep(std::vector<double>(imu[k]), q4[0], q4[1], q4[2], q4[3], utime), true);
    est.addUpdate(new RBISIndexedMeasurement(RBIS::velocityInds(), std::vector<double>(lo[k]), std::vector<double>(r_lo, r_lo + 3),
                                             PB_R_DIAG_BROADCAST, std::vector<uint8_t>(), RBISUpdateInterface::legodo, utime), true);
    for (auto &f : fixes)
      if (f.deliver_at_step == k)
        est.addUpdate(new RBISIndexedMeasurement(RBIS::positionInds(), std::vector<double>(f.z), std::vector<double>(r_pos, r_pos + 3),
                                                 PB_R_DIAG_BROADCAST, std::vector<uint8_t>(), RBISUpdateInterface::fovis, f.utime), true);
  }
  const int64_t replayed = est.replayed_updates;
  // a fix older than the window must be discarded
  {
    std::vector<double> z(3 * B, 1.0);
    est.addUpdate(new RBISIndexedMeasurement(RBIS::positionInds(), std::move(z), std::vector<double>(r_pos, r_pos + 3), PB_R_DIAG_BROADCAST,
                                             std::vector<uint8_t>(), RBISUpdateInterface::fovis, 2000), true);
  }
  // ---- oracle: timestamp order (fix after the legodo update of the same utime) ----
  const int vel_idx[3] = { 3, 4, 5 }, pos_idx[3] = { 9, 10, 11 };
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    for (int b = 0; b < B; b++) {
      double gy[3] = { imu[k][b], imu[k][B + b], imu[k][2 * B + b] }, ac[3] = { imu[k][3 * B + b], imu[k][4 * B + b], imu[k][5 * B + b] };
      po_imu_process_step(gy, ac, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      double z[3] = { lo[k][b], lo[k][B + b], lo[k][2 * B + b] }, R[9] = { r_lo[0], 0, 0, 0, r_lo[1], 0, 0, 0, r_lo[2] };
      po_indexed_update(3, vel_idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    }
    for (auto &f : fixes)
      if (f.utime == utime)
        for (int b = 0; b < B; b++) {
          double z[3] = { f.z[b], f.z[B + b], f.z[2 * B + b] }, R[9] = { r_pos[0], 0, 0, 0, r_pos[1], 0, 0, 0, r_pos[2] };
          po_indexed_update(3, pos_idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        }
  }
  RBIS head;
  RBIM cov;
  est.getHeadState(head, cov);
  std::vector<double> ll = est.getMeasurementsLogLikelihood();
  double ev = 0, eq = 0, eP = 0, el = 0, sv = 0, sP = 0, sl = 0;
  for (int b = 0; b < B; b++) {
    for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(head(i, b) - ox[b].vec[i])); sv = fmax(sv, fabs(ox[b].vec[i])); }
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(head.q(i, b) - ox[b].quat[i]));
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(cov(r, c, b) - oP[b].m[c * 21 + r])); sP = fmax(sP, fabs(oP[b].m[c * 21 + r])); }
    el = fmax(el, fabs(ll[b] - oll[b]));
    sl = fmax(sl, fabs(oll[b]));
  }
  printf("checkpoint_every=%d%s: %zu updates in window, %" PRId64 " replayed, %" PRId64 " fused pairs; rel err vec %.2e quat %.2e cov %.2e ll %.2e\n",
         every, fuse ? " fused" : "", est.history.updateMap.size(), replayed, (int64_t) est.fused_pairs, ev / sv, eq, eP / sP, el / sl);
  const bool ok = (fuse ? est.fused_pairs > T / 2 : est.fused_pairs == 0) && est.last_status == PB_OK && (replayed > 0 || DELAY == 0) && head.utime == (int64_t) T * 1000 && ev / sv < 1e-9 && eq < 1e-9 &&
                  eP / sP < 1e-9 && el / sl < 1e-9 && est.history.updateMap.size() < 80;
  (void) DELAY;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
