// test_legodo_modes.cpp -- LegOdoCommon::createMeasurement (rbis_legodo_common.cpp:110-169) through the shim for the modes
// the miniature se-fusion of test_shim.cpp does not use: pos_and_lin_rate with its per-message fall-back to lin_rate when
// the position is not valid (:118-122; per filter here: RBISEitherUpdate) and lin_rot_rate; delta status -1 / 0 / 1 per
// filter; with posterior checkpoints on (argv[2] = history slots) the two halves of the either-update must land in the
// same checkpoint slot.  Checked filter by filter against the oracle.  Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x4C45474F444FULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const std::string mode = argc > 1 ? argv[1] : "pos_and_lin_rate";
  const int slots = argc > 2 ? atoi(argv[2]) : 0;
  const bool fuse = argc > 3 && std::string(argv[3]) == "fuse";  // state_estimator.fuse_ins_legodo
  const int omode = mode == "lin_rate" ? 0 : (mode == "lin_rot_rate" ? 1 : 2);
  const int B = 150, T = 40;
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", std::to_string(slots));
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.legodo.mode=" + mode + "|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=0.1|"
                       "state_estimator.legodo.r_vang=0.3|state_estimator.legodo.r_vxyz_uncertain=0.5|"
                       "state_estimator.legodo.r_vang_uncertain=0.9");
  for (const char *s : { "ins", "legodo" }) {
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
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  LegOdoHandler legodo_handler(&param);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_legodo = front_end.addSensor("legodo", &LegOdoHandler::processMessageDelta, &legodo_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
  const double r5[5] = { 0.2, 0.1, 0.3, 0.5, 0.9 };
  double g;
  po_get_constants(&g, nullptr);
  std::vector<double> pos(3 * B), dtr(3 * B), dq(4 * B);
  std::vector<int> pstat(B);
  std::vector<float> status(B);
  int n_fallback = 0, n_full = 0, n_skip = 0;
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    const double v[6] = { 0.2 * sin(0.1 * k), 0.1, -0.15 * cos(0.07 * k), 0.2 * nrand(), 0.2 * nrand(), g + 0.2 * nrand() };
    std::vector<double> gyb(3 * B), acb(3 * B);
    for (int i = 0; i < 3; i++)
      for (int b = 0; b < B; b++) { gyb[i * B + b] = v[i]; acb[i * B + b] = v[3 + i]; }
    // fused pairs need the IMU and the leg odometry in the same memory space: per-filter host blocks when fusing
    msgs::ins_t im = fuse ? msgs::ins_t{ utime, BatchArray(gyb.data(), PB_HOST), BatchArray(acb.data(), PB_HOST) }
                          : msgs::ins_t{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
    on_ins(&im);
    for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    for (int b = 0; b < B; b++) {
      double qq[4];
      po_euler_to_quat(0.002 * nrand(), 0.002 * nrand(), 0.003 * nrand(), qq);
      for (int i = 0; i < 3; i++) {
        pos[i * B + b] = ox[b].vec[9 + i] + 0.2 * nrand();
        dtr[i * B + b] = (ox[b].vec[3 + i] + 0.1 * nrand()) * 0.001;
      }
      for (int i = 0; i < 4; i++) dq[i * B + b] = qq[i];
      const double u = urand();
      status[b] = u < 0.15 ? -1.f : (u < 0.45 ? 1.f : 0.f);
      // every 5th message: the position is bad for everybody, every 7th: good for everybody, else mixed
      pstat[b] = (k % 5 == 4) ? 0 : ((k % 7 == 6) ? 1 : (urand() < 0.6));
    }
    msgs::legodo_delta_t lo{ utime, utime - 1000, pos.data(), dtr.data(), dq.data(), pstat.data(), status.data() };
    on_legodo(&lo);
    for (int b = 0; b < B; b++) {
      if (status[b] < 0) { n_skip++; continue; }
      int idx[6];
      double z[6], Rd[6], R[36] = { 0 };
      const double p3[3] = { pos[b], pos[B + b], pos[2 * B + b] }, t3[3] = { dtr[b], dtr[B + b], dtr[2 * B + b] };
      const double qi[4] = { dq[b], dq[B + b], dq[2 * B + b], dq[3 * B + b] };
      const int m = po_legodo_create_measurement(omode, r5, p3, t3, qi, utime, utime - 1000, pstat[b], status[b], idx, z, Rd);
      (m == 3 && omode == 2 ? n_fallback : n_full)++;
      for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
      po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    }
  }
  // ---- one Vicon pose (position_orient, with a frame) on top: filter 3 reports a near-zero translation -> no update ----
  {
    param.applyOverrides("state_estimator.vicon.mode=position_orient|state_estimator.vicon.apply_frame=true|"
                         "state_estimator.vicon.r_xyz=0.01|state_estimator.vicon.r_chi=2.0|state_estimator.vicon.downsample_factor=1|"
                         "state_estimator.vicon.roll_forward_on_receive=true|state_estimator.vicon.utime_offset=0");
    BotTrans body_to_vicon;
    body_to_vicon.rot_quat[0] = cos(0.2); body_to_vicon.rot_quat[3] = sin(0.2);
    body_to_vicon.trans_vec[0] = 0.05; body_to_vicon.trans_vec[2] = -0.1;
    ViconHandler vicon_handler(&param, &body_to_vicon);
    auto on_vicon = front_end.addSensor("vicon", &ViconHandler::processMessage, &vicon_handler);
    std::vector<double> vt(3 * B), vq(4 * B);
    for (int b = 0; b < B; b++) {
      double qq[4];
      po_euler_to_quat(0.05 * nrand(), 0.05 * nrand(), 3.0 * (urand() - 0.5), qq);
      for (int i = 0; i < 3; i++) vt[i * B + b] = (b == 3) ? 1e-7 : ox[b].vec[9 + i] + 0.05 * nrand();
      for (int i = 0; i < 4; i++) vq[i * B + b] = qq[i];
    }
    msgs::rigid_transform_t vm{ (int64_t) (T + 1) * 1000, BatchArray(vt.data(), PB_HOST), BatchArray(vq.data(), PB_HOST) };
    on_vicon(&vm);
    for (int b = 0; b < B; b++) {
      if (b == 3) continue;
      const double tv[3] = { vt[b], vt[B + b], vt[2 * B + b] }, qv[4] = { vq[b], vq[B + b], vq[2 * B + b], vq[3 * B + b] };
      double tb[3], qb[4], z[6] = { 0 }, R[36] = { 0 };
      po_quat_rotate(qv, body_to_vicon.trans_vec, tb);
      for (int i = 0; i < 3; i++) z[i] = tb[i] + tv[i];
      po_quat_mul(qv, body_to_vicon.rot_quat, qb);
      const int idx[6] = { 9, 10, 11, 6, 7, 8 };
      for (int i = 0; i < 6; i++) R[i * 6 + i] = (i < 3) ? 1e-4 : bot_sq(bot_to_radians(2.0));
      po_indexed_orient_update(6, idx, z, R, qb, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    }
  }
  // ---- PoseMeasHandler (pose_meas.cpp:53-96): no_corrections = 3 lets two messages through (decrement, then compare), filter 5
  // reports a pose at the origin -> no update for it; the third and fourth message change nothing ----
  {
    param.applyOverrides("state_estimator.pose_meas.mode=position_orient|state_estimator.pose_meas.no_corrections=3|"
                         "state_estimator.pose_meas.r_xyz=0.02|state_estimator.pose_meas.r_chi=3.0|state_estimator.pose_meas.downsample_factor=1|"
                         "state_estimator.pose_meas.roll_forward_on_receive=true|state_estimator.pose_meas.utime_offset=0");
    PoseMeasHandler pose_handler(&param);
    auto on_pose = front_end.addSensor("pose_meas", &PoseMeasHandler::processMessage, &pose_handler);
    for (int k = 0; k < 4; k++) {
      std::vector<double> pp(3 * B), pv(3 * B, 0.0), pq(4 * B);
      for (int b = 0; b < B; b++) {
        double qq[4];
        po_euler_to_quat(0.05 * nrand(), 0.05 * nrand(), 3.0 * (urand() - 0.5), qq);
        for (int i = 0; i < 3; i++) pp[i * B + b] = (b == 5) ? 1e-7 : ox[b].vec[9 + i] + 0.05 * nrand();
        for (int i = 0; i < 4; i++) pq[i * B + b] = qq[i];
      }
      msgs::pose_t pm{ (int64_t) (T + 1) * 1000, BatchArray(pp.data(), PB_HOST), BatchArray(pv.data(), PB_HOST), BatchArray(pq.data(), PB_HOST) };
      on_pose(&pm);
      if (k >= 2) continue;  // silent
      for (int b = 0; b < B; b++) {
        if (b == 5) continue;
        double z[6] = { pp[b], pp[B + b], pp[2 * B + b], 0, 0, 0 }, R[36] = { 0 };
        const double qb[4] = { pq[b], pq[B + b], pq[2 * B + b], pq[3 * B + b] };
        const int idx[6] = { 9, 10, 11, 6, 7, 8 };
        for (int i = 0; i < 6; i++) R[i * 6 + i] = (i < 3) ? 4e-4 : bot_sq(bot_to_radians(3.0));
        po_indexed_orient_update(6, idx, z, R, qb, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
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
  printf("fused ins+legodo pairs: %lld\n", (long long) est.fused_pairs);
  printf("mode %s, %d history slots: %d full, %d fall-back, %d skipped filter-updates; rel err vec %.2e quat %.2e cov %.2e ll %.2e "
         "(status %d)\n", mode.c_str(), slots, n_full, n_fallback, n_skip, ev / sv, eq, eP / sP, el / sl, est.last_status);
  const bool ok = est.last_status == PB_OK && head.utime == (int64_t) (T + 1) * 1000 && ev / sv < 1e-9 && eq < 1e-9 && eP / sP < 1e-9 &&
                  el / sl < 1e-9 && n_skip > 0 && (omode != 2 || (n_fallback > 0 && n_full > 0)) &&
                  // fusible: every lin_rate message, and in pos_and_lin_rate the messages whose position is bad for
                  // everybody (k % 5 == 4: a pure lin_rate fall-back)
                  est.fused_pairs == (fuse ? (omode == 0 ? T : (omode == 2 ? T / 5 : 0)) : 0);
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
