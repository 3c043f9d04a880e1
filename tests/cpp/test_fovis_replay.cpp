// test_fovis_replay.cpp -- FovisHandler updates OWN their measurement: two VO updates are in the history window when a
// measurement older than both arrives late; MavStateEstimator::addUpdate restores the checkpoint before it and re-applies
// everything after it (mav_state_est.cpp:28-80), including both VO updates, each with the z / quaternion it was built with
// (rbis_fovis_update.cpp:299-305 copies them into the update object).  A handler-wide device scratch would make the first
// VO update re-apply the second one's measurement.  The oracle replays the same way.  Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x5245504C4159ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const int B = 48, T = 40;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "60000");
  param.set("state_estimator.history_slots", "128");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.fovis.mode=position_orient|state_estimator.fovis.r_pxyz=0.02|state_estimator.fovis.r_chi=0.01");
  for (const char *s : { "ins", "fovis" }) {
    param.set(std::string("state_estimator.") + s + ".downsample_factor", "1");
    param.set(std::string("state_estimator.") + s + ".roll_forward_on_receive", "true");
    param.set(std::string("state_estimator.") + s + ".utime_offset", "0");
  }
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0);
  for (int b = 0; b < B; b++) {
    double q[4];
    po_euler_to_quat(0.1 * (urand() - 0.5), 0.1 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    for (int i = 0; i < 4; i++) { x0.q(i, b) = q[i]; ox[b].quat[i] = q[i]; }
    for (int i = 0; i < 3; i++) { x0(3 + i, b) = 0.3 * nrand(); ox[b].vec[3 + i] = x0(3 + i, b); }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  int rc = 1;
  {
    FovisHandler fovis_handler(&param, 0);  // destroyed BEFORE the estimator: its updates stay valid in the history
    auto on_fovis = front_end.addSensor("fovis", &FovisHandler::processMessage, &fovis_handler);
    const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
    struct Snap { std::vector<po_rbis> x; std::vector<po_rbim> P; std::vector<double> ll; };
    auto snap = [&]() { return Snap{ ox, oP, oll }; };
    std::vector<std::vector<double>> imu_log;   // v[6] per step
    std::vector<Snap> after_imu;                // oracle posterior right after IMU step k (message-time timeline)
    // VO measurements as built at message time: z [6], qm [4] per filter
    struct Vo { int64_t utime; std::vector<double> z, qm; };
    std::vector<Vo> vos;
    auto apply_vo = [&](const Vo &v) {
      for (int b = 0; b < B; b++) {
        double R[36] = { 0 };
        const int idx[6] = { 9, 10, 11, 6, 7, 8 };
        for (int i = 0; i < 6; i++) R[i * 6 + i] = (i < 3) ? 0.02 * 0.02 : 0.01 * 0.01;
        po_indexed_orient_update(6, idx, &v.z[6 * b], R, &v.qm[4 * b], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
    };
    auto apply_imu = [&](const std::vector<double> &v) {
      for (int b = 0; b < B; b++)
        po_imu_process_step(v.data(), v.data() + 3, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    };
    std::vector<double> vt(3 * B), vq(4 * B), late_z(3 * B), late_R(3 * B);
    for (int k = 0; k < T; k++) {
      const int64_t utime = (int64_t) (k + 1) * 1000;
      std::vector<double> v = { 0.3 * sin(0.05 * k), 0.1, -0.2 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
      msgs::ins_t im{ utime, BatchArray(v.data(), PB_HOST_BROADCAST), BatchArray(v.data() + 3, PB_HOST_BROADCAST) };
      on_ins(&im);
      apply_imu(v);
      imu_log.push_back(v);
      after_imu.push_back(snap());
      if (k == 11 || k == 23) {  // VO message: keyframe 4.7 ms back -> T0 = posterior of the IMU update 4 ms back
        const int64_t prev_ts = utime - 4700;
        const Snap &t0 = after_imu[k - 4];
        Vo vo{ utime, std::vector<double>(6 * B, 0.0), std::vector<double>(4 * B) };
        for (int b = 0; b < B; b++) {
          double dq[4];
          po_euler_to_quat(0.01 * nrand(), 0.01 * nrand(), 0.02 * nrand(), dq);
          for (int i = 0; i < 3; i++) vt[i * B + b] = 0.02 * nrand();
          for (int i = 0; i < 4; i++) vq[i * B + b] = dq[i];
          const double t3[3] = { vt[b], vt[B + b], vt[2 * B + b] };
          po_fovis_compose(t0.x[b].vec + 9, t0.x[b].quat, t3, dq, &vo.z[6 * b], &vo.qm[4 * b]);
        }
        msgs::update_t m{ utime, prev_ts, nullptr, BatchArray(vt.data(), PB_HOST), BatchArray(vq.data(), PB_HOST) };
        on_fovis(&m);
        apply_vo(vo);
        vos.push_back(vo);
      }
      if (k == 30) {  // a velocity measurement stamped 9.5 ms arrives now: older than both VO updates, inside the window
        for (int b = 0; b < B; b++)
          for (int i = 0; i < 3; i++) { late_z[i * B + b] = after_imu[8].x[b].vec[3 + i] + 0.05 * nrand(); late_R[i * B + b] = 0.01; }
        est.addUpdate(new RBISIndexedMeasurement(RBIS::velocityInds(), std::vector<double>(late_z), std::vector<double>(late_R), PB_R_DIAG,
                                                 std::vector<uint8_t>(), RBISUpdateInterface::legodo, 9500), true);
        // oracle: back to the posterior after the IMU update at 9000, insert, re-apply everything up to now in time order
        ox = after_imu[8].x; oP = after_imu[8].P; oll = after_imu[8].ll;
        for (int b = 0; b < B; b++) {
          const int idx[3] = { 3, 4, 5 };
          const double z[3] = { late_z[b], late_z[B + b], late_z[2 * B + b] };
          double R[9] = { 0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01 };
          po_indexed_update(3, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        }
        for (int j = 9; j <= k; j++) {
          apply_imu(imu_log[j]);
          for (auto &vo : vos)
            if (vo.utime == (int64_t) (j + 1) * 1000) apply_vo(vo);
        }
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
    printf("2 VO updates re-applied behind a late measurement (%lld updates replayed): rel err vec %.2e quat %.2e cov %.2e ll %.2e (status %d)\n",
           (long long) est.replayed_updates, ev / sv, eq, eP / sP, el / sl, est.last_status);
    const bool ok = est.last_status == PB_OK && est.replayed_updates >= 24 && ev / sv < 1e-9 && eq < 1e-9 && eP / sP < 1e-9 && el / sl < 1e-9;
    rc = ok ? 0 : 1;
  }
  printf(rc == 0 ? "PASS\n" : "FAIL\n");
  return rc;
}
