// test_handler_modes.cpp -- every mode of the measurement handlers that tests/cpp/test_shim.cpp does not reach, one per run
// (argv[1]), through the handlers and MavStateEstimator::addUpdate on the GPU, against the oracle's restatement of the same
// handler arithmetic:
//   sm_position | sm_velocity | sm_yaw | sm_position_yaw | sm_velocity_yaw   ScanMatcherHandler (sensor_handlers.cpp:612-724)
//   fovis_velocity | fovis_position                                           FovisHandler (rbis_fovis_update.cpp:93-117,262-293)
//   legodo_zero3   state_estimator.legodo.zero_initial_velocity = 3: the first TWO ticks integrate a zero delta
//                  (decrement first, then compare: rbis_legodo_update.cpp:264-268)
//   legodo_ft      nothing is integrated before the first force/torque message (rbis_legodo_update.cpp:208-211)
// Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <string>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x4D4F444553ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const std::string mode = argc > 1 ? argv[1] : "sm_position";
  const int B = 90, T = 40;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", "0");  // in-order only (no posterior checkpoints)
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  const bool is_sm = mode.rfind("sm_", 0) == 0, is_fovis = mode.rfind("fovis_", 0) == 0, is_lo = mode.rfind("legodo_", 0) == 0;
  param.applyOverrides("state_estimator.scan_matcher.mode=" + (is_sm ? mode.substr(3) : std::string("position")) +
                       "|state_estimator.scan_matcher.r_pxy=0.05|state_estimator.scan_matcher.r_pz=0.07|"
                       "state_estimator.scan_matcher.r_vxy=0.11|state_estimator.scan_matcher.r_vz=0.13|state_estimator.scan_matcher.r_yaw=1.5");
  param.applyOverrides("state_estimator.fovis.mode=" + (is_fovis ? mode.substr(6) : std::string("velocity")) +
                       "|state_estimator.fovis.r_pxyz=0.03|state_estimator.fovis.r_vxyz=0.09|state_estimator.fovis.r_chi=0.01|state_estimator.fovis.r_vang=0.2");
  param.applyOverrides("state_estimator.legodo.mode=lin_rate|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=0.1|"
                       "state_estimator.legodo.r_vang=0.3|state_estimator.legodo.r_vxyz_uncertain=0.5|state_estimator.legodo.r_vang_uncertain=0.9");
  if (mode == "legodo_zero3") param.set("state_estimator.legodo.zero_initial_velocity", "3");
  for (const char *s : { "ins", "legodo", "fovis", "scan_matcher" }) {
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
  ScanMatcherHandler sm_handler(&param);
  FovisHandler fovis_handler(&param, 0);
  LegOdoHandler legodo_handler(&param);
  if (mode == "legodo_ft") legodo_handler.force_torque_init_ = false;
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_sm = front_end.addSensor("scan_matcher", &ScanMatcherHandler::processMessage, &sm_handler);
  auto on_fovis = front_end.addSensor("fovis", &FovisHandler::processMessage, &fovis_handler);
  auto on_legodo = front_end.addSensor("legodo", &LegOdoHandler::processMessageDelta, &legodo_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  fovis_handler.markKeyframe(&est);
  std::vector<po_rbis> key = ox;
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
  const double r5[5] = { 0.2, 0.1, 0.3, 0.5, 0.9 };
  std::vector<double> pos(3 * B), vel(3 * B), quat(4 * B), vt(3 * B), vq(4 * B), dtr(3 * B);
  std::vector<float> status(B, 0.f);
  int n_meas = 0, n_null = 0, lo_ticks = 0;
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    const double v[6] = { 0.3 * sin(0.05 * k), 0.1, -0.2 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
    msgs::ins_t im{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
    on_ins(&im);
    for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    if (is_lo) {  // leg odometry on every tick
      if (mode == "legodo_ft" && k == 10) legodo_handler.forceTorqueHandler();
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) dtr[i * B + b] = (ox[b].vec[3 + i] + 0.1 * nrand()) * 0.001;
      msgs::legodo_delta_t lo{ utime, utime - 1000, nullptr, dtr.data(), nullptr, nullptr, status.data() };
      on_legodo(&lo);
      if (mode == "legodo_ft" && k < 10) { n_null++; continue; }
      lo_ticks++;
      const bool zeroed = mode == "legodo_zero3" && lo_ticks < 3;
      for (int b = 0; b < B; b++) {
        int idx[6];
        double z[6], Rd[6], R[36] = { 0 }, t3[3] = { dtr[b], dtr[B + b], dtr[2 * B + b] }, qi[4] = { 1, 0, 0, 0 }, p3[3] = { 0, 0, 0 };
        if (zeroed) t3[0] = t3[1] = t3[2] = 0.0;
        const int m = po_legodo_create_measurement(0, r5, p3, t3, qi, utime, utime - 1000, 1, status[b], idx, z, Rd);
        for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
        po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
      n_meas++;
      continue;
    }
    if (k % 5 != 4) continue;
    n_meas++;
    if (is_sm) {
      for (int b = 0; b < B; b++) {
        double dq[4], qo[4];
        po_euler_to_quat(0, 0, 0.03 * nrand(), dq);
        po_quat_mul(dq, ox[b].quat, qo);
        for (int i = 0; i < 3; i++) { pos[i * B + b] = ox[b].vec[9 + i] + 0.05 * nrand(); vel[i * B + b] = ox[b].vec[3 + i] + 0.1 * nrand(); }
        for (int i = 0; i < 4; i++) quat[i * B + b] = qo[i];
      }
      msgs::pose_t pose{ utime, BatchArray(pos.data(), PB_HOST), BatchArray(vel.data(), PB_HOST), BatchArray(quat.data(), PB_HOST) };
      on_sm(&pose);
      const std::string sm = mode.substr(3);
      const bool use_pos = sm == "position" || sm == "position_yaw", yaw = sm == "yaw" || sm == "position_yaw" || sm == "velocity_yaw";
      for (int b = 0; b < B; b++) {
        int idx[4], m = 0;
        double z[4] = { 0, 0, 0, 0 }, R[16] = { 0 }, Rd[4], qm[4] = { quat[b], quat[B + b], quat[2 * B + b], quat[3 * B + b] };
        if (sm != "yaw") {
          for (int i = 0; i < 3; i++) { idx[m] = (use_pos ? 9 : 3) + i; z[m] = use_pos ? pos[i * B + b] : vel[i * B + b]; m++; }
          Rd[0] = Rd[1] = use_pos ? 0.05 * 0.05 : 0.11 * 0.11;
          Rd[2] = use_pos ? 0.07 * 0.07 : 0.13 * 0.13;
        }
        if (yaw) { idx[m] = 8; z[m] = 0; Rd[m] = bot_sq(bot_to_radians(1.5)); m++; }
        for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
        if (yaw) po_indexed_orient_update(m, idx, z, R, qm, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        else po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
    } else {  // fovis velocity / position
      for (int b = 0; b < B; b++) {
        double dq[4];
        po_euler_to_quat(0.01 * nrand(), 0.01 * nrand(), 0.02 * nrand(), dq);
        for (int i = 0; i < 3; i++) vt[i * B + b] = 0.004 * nrand() + ox[b].vec[3 + i] * 0.005;
        for (int i = 0; i < 4; i++) vq[i * B + b] = dq[i];
      }
      msgs::update_t vo{ utime, mode == "fovis_velocity" ? utime - 5000 : fovis_handler.prev_t0_body_utime_, nullptr,
                         BatchArray(vt.data(), PB_HOST), BatchArray(vq.data(), PB_HOST) };
      on_fovis(&vo);
      for (int b = 0; b < B; b++) {
        const double t3[3] = { vt[b], vt[B + b], vt[2 * B + b] }, q[4] = { vq[b], vq[B + b], vq[2 * B + b], vq[3 * B + b] };
        double z[6] = { 0 }, qm[4], R[9] = { 0 };
        if (mode == "fovis_velocity") {
          double tv[3], qv[4];
          po_delta_as_velocity(t3, q, 5000, tv, qv);  // getTransAsVelocityTrans: translation / elapsed
          const int idx[3] = { 3, 4, 5 };
          R[0] = R[4] = R[8] = 0.09 * 0.09;
          po_indexed_update(3, idx, tv, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        } else {
          po_fovis_compose(key[b].vec + 9, key[b].quat, t3, q, z, qm);
          const int idx[3] = { 9, 10, 11 };
          R[0] = R[4] = R[8] = 0.03 * 0.03;
          po_indexed_update(3, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        }
      }
      if (mode == "fovis_position") { fovis_handler.markKeyframe(&est); key = ox; }
    }
  }
  RBIS head;
  RBIM cov;
  est.getHeadState(head, cov);
  std::vector<double> ll = est.getMeasurementsLogLikelihood();
  double ev = 0, eq = 0, eP = 0, el = 0, sv = 0, sP = 0, sl = 1e-300;
  for (int b = 0; b < B; b++) {
    for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(head(i, b) - ox[b].vec[i])); sv = fmax(sv, fabs(ox[b].vec[i])); }
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(head.q(i, b) - ox[b].quat[i]));
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(cov(r, c, b) - oP[b].m[c * 21 + r])); sP = fmax(sP, fabs(oP[b].m[c * 21 + r])); }
    el = fmax(el, fabs(ll[b] - oll[b]));
    sl = fmax(sl, fabs(oll[b]));
  }
  printf("%s: %d measurements applied, %d refused: rel err vec %.2e quat %.2e cov %.2e ll %.2e (status %d)\n", mode.c_str(), n_meas,
         n_null, ev / sv, eq, eP / sP, el / sl, est.last_status);
  const bool counts = mode == "legodo_ft" ? (n_null == 10 && n_meas == T - 10) : (is_lo ? n_meas == T : n_meas == T / 5);
  const bool ok = counts && est.last_status == PB_OK && ev / sv < 1e-9 && eq < 1e-9 && eP / sP < 1e-9 && el / sl < 1e-9;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
