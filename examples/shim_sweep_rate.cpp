// shim_sweep_rate.cpp -- throughput of the DROP-IN path itself: the reference's handler API (BotParam keys -> InsHandler,
// LegOdoHandler -> FrontEnd::addSensor -> MavStateEstimator::addUpdate, pronto_amd/csrc/mav_state_est_batch.hpp) driving
// the batched filters, not the bare C ABI.  The workload is the reference's own batch use (state-estimator/python/
// param_sweep.py:39-52): ONE robot's recorded IMU + foot-state stream replayed into B filters that differ in their
// initial state, every message passed once (PB_HOST_BROADCAST), leg kinematic odometry and its contact classifier run
// per filter on the device (LegOdoHandler::processMessageFeet -> pb_legodo_update), the IMU step and the leg-odometry
// update fused into one launch (state_estimator.fuse_ins_legodo).
//
//   g++ -std=c++17 -O2 -Iinclude -Ipronto_amd/csrc examples/shim_sweep_rate.cpp -Lpronto_amd/lib -lpronto_batch
//       -Wl,-rpath,$PWD/pronto_amd/lib -o shim_sweep_rate
//   ./shim_sweep_rate [filters=65536] [messages=2000] [n_states=15] [history_slots=0|derived] [utime_history_span=1000000] [vo_every=0]
//                     [input=feet|joints] [pairs=one|two] [legodo mode=lin_rate|lin_rot_rate|pos_and_lin_rate]
// input = joints: the log is a bot_core::joint_state_t stream and LegOdoHandler::processMessage(joint_state_t) runs the
// forward kinematics per filter on the device too (the reference's own handler signature); pairs = two: the leg odometry as
// its own launch in front of the fused step (round 2's path) instead of inside the step kernel.
//
// Prints messages/s and filter-steps/s (one step = IMU predict + leg-odometry update of one filter).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mav_state_est_batch.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x5357454550ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return std::sqrt(-2 * std::log(urand())) * std::cos(2 * M_PI * urand()); }
static double ramp(double x) { return x < 0 ? 0 : (x > 0.05 ? 1.0 : x / 0.05); }
static void euler_to_quat(double r, double p, double y, double q[4])
{
  const double cr = std::cos(r / 2), sr = std::sin(r / 2), cp = std::cos(p / 2), sp = std::sin(p / 2), cy = std::cos(y / 2), sy = std::sin(y / 2);
  q[0] = cr * cp * cy + sr * sp * sy; q[1] = sr * cp * cy - cr * sp * sy; q[2] = cr * sp * cy + sr * cp * sy; q[3] = cr * cp * sy - sr * sp * cy;
}

int main(int argc, char **argv)
{
  const int B = argc > 1 ? std::atoi(argv[1]) : 65536;
  const int T = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int n = argc > 3 ? std::atoi(argv[3]) : 15;
  const std::string slots = argc > 4 ? argv[4] : "0";
  const std::string span = argc > 5 ? argv[5] : "1000000";  // us of update history kept (update_history.cpp:28-39)
  const int vo_every = argc > 6 ? std::atoi(argv[6]) : 0;     // a visual-odometry delta every N-th pair (0 = none): config 3
  const bool joints = argc > 7 && std::string(argv[7]) == "joints";
  const bool two_launches = argc > 8 && std::string(argv[8]) == "two";
  const std::string lomode = argc > 9 ? argv[9] : "lin_rate";   // state_estimator.legodo.mode: lin_rate | lin_rot_rate | pos_and_lin_rate
  BotParam param;
  if (const char *ov = getenv("SHIM_SWEEP_OVERRIDES")) param.applyOverrides(ov);   // "key=value|key=value": extra BotParam keys (experiments)
  param.set("state_estimator.utime_history_span", span);
  // "derived": only utime_history_span is set, like every reference .cfg -- the estimator then derives its checkpoint pool
  // (32 slots spaced over the span, mav_state_est_batch.hpp)
  if (slots != "derived") param.set("state_estimator.history_slots", slots);
  param.set("state_estimator.fuse_ins_legodo", "true");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.q_gyro_bias", n == 21 ? 0.001 : 0.0);
  param.set("state_estimator.ins.q_accel_bias", n == 21 ? 0.0001 : 0.0);
  param.set("state_estimator.ins.timestep_dt", 0.002);
  param.set("state_estimator.ins.atlas_filter", "false");
  param.set("state_estimator.ins.accel_bias_update_online", n == 21 ? "true" : "false");
  param.set("state_estimator.ins.gyro_bias_update_online", n == 21 ? "true" : "false");
  param.applyOverrides("state_estimator.legodo.mode=" + lomode + "|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=0.1|"
                       "state_estimator.legodo.r_vang=0.3|state_estimator.legodo.r_vxyz_uncertain=0.5|state_estimator.legodo.r_vang_uncertain=0.9|"
                       "state_estimator.legodo.schmitt_low_threshold=475|state_estimator.legodo.schmitt_high_threshold=525|"
                       "state_estimator.legodo.schmitt_low_delay=7000|state_estimator.legodo.schmitt_high_delay=7000|"
                       "state_estimator.legodo.filter_contact_events=true|state_estimator.legodo.zero_initial_velocity=3|"
                       "state_estimator.legodo.initialization_mode=zero|state_estimator.legodo.filter_joint_positions=none|"
                       "state_estimator.legodo.init_contact_mode=walking|state_estimator.legodo.total_force=1500|"
                       "state_estimator.legodo.standing_schmitt_level=0.65|state_estimator.legodo.use_controller_input=false|"
                       "state_estimator.legodo.torque_adjustment=true|state_estimator.legodo.adjustment_joints=l_leg_hpz,l_leg_kny,r_leg_hpz,r_leg_kny|"
                       "state_estimator.legodo.adjustment_gain=7000,10000,7000,10000");
  param.applyOverrides("state_estimator.fovis.mode=position_orient|state_estimator.fovis.r_pxyz=0.02|state_estimator.fovis.r_chi=0.01");
  for (const char *s : { "ins", "legodo", "fovis" }) {
    param.set(std::string("state_estimator.") + s + ".downsample_factor", "1");
    param.set(std::string("state_estimator.") + s + ".roll_forward_on_receive", "true");
    param.set(std::string("state_estimator.") + s + ".utime_offset", "0");
  }
  // the sweep: every filter its own initial attitude and velocity
  RBIS x0(n, B);
  RBIM P0(n, B);
  for (int b = 0; b < B; b++) {
    double q[4];
    euler_to_quat(0.05 * (urand() - 0.5), 0.05 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    for (int i = 0; i < 4; i++) x0.q(i, b) = q[i];
    for (int i = 0; i < 3; i++) x0(3 + i, b) = 0.1 * nrand();
    const double sig[21] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0, .008, .008, .008, .1, .1, .1 };
    for (int i = 0; i < n; i++) P0(i, i, b) = sig[i] * sig[i];
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  // the robot model: two 6-DoF legs (hip yaw / roll / pitch, knee, ankle pitch / roll), what ModelClient::fromURDFString
  // extracts from a URDF for the two standing links
  ModelClient model;
  const char *jn[6] = { "leg_hpz", "leg_hpx", "leg_hpy", "leg_kny", "leg_aky", "leg_akx" };
  const double jo[6][3] = { { 0, 0.089, 0 }, { 0, 0, 0 }, { 0.05, 0.0225, -0.066 }, { -0.05, 0, -0.374 }, { 0, 0, -0.422 }, { 0, 0, 0 } };
  const double ja[6][3] = { { 0, 0, 1 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 0 }, { 1, 0, 0 } };
  std::vector<std::string> joint_names;
  for (int side = 0; side < 2; side++)
    for (int j = 0; j < 6; j++) {
      ModelClient::Joint J;
      J.name = std::string(side ? "r_" : "l_") + jn[j];
      for (int i = 0; i < 3; i++) { J.xyz[i] = jo[j][i]; J.axis[i] = ja[j][i]; }
      if (side) J.xyz[1] = -J.xyz[1];
      (side ? model.right_chain : model.left_chain).push_back(J);
      joint_names.push_back(J.name);
    }
  LegOdoHandler legodo_handler(&param, &model);
  legodo_handler.one_kernel_pairs = !two_launches;
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_feet = front_end.addSensor("legodo", &LegOdoHandler::processMessageFeet, &legodo_handler);
  RBISUpdateInterface *(LegOdoHandler::*joint_fn)(const msgs::joint_state_t *, MavStateEstimator *) = &LegOdoHandler::processMessage;
  auto on_joints = front_end.addSensor("legodo", joint_fn, &legodo_handler);
  FovisHandler fovis_handler(&param, /*snapshot_slot=*/0);
  auto on_fovis = front_end.addSensor("fovis", &FovisHandler::processMessage, &fovis_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  if (est.last_status != PB_OK) { std::fprintf(stderr, "shim_sweep_rate: %s\n", pb_last_error(est.ctx)); return 1; }
  front_end.setStateEstimator(&est);
  if (vo_every > 0) fovis_handler.markKeyframe(&est);

  // one robot's log: a walking gait (left / right foot poses in the body frame, vertical foot forces) and its IMU
  std::vector<double> imu(6 * (size_t) T), feet(14 * (size_t) T), forces(2 * (size_t) T);
  std::vector<float> jpos(12 * (size_t) T), jeff(12 * (size_t) T);
  for (int k = 0; k < T; k++) {
    const double t = (k + 1) * 0.002;
    double ph = t / 1.1 + 0.3;
    ph -= std::floor(ph);
    double wl = ramp(ph) * ramp(0.6 - ph), wr = ramp(ph - 0.5) * ramp(1.1 - ph) + (ph < 0.1 ? ramp(0.1 - ph) : 0.0);
    if (t < 0.4) wl = wr = 1.0;
    const double sw = std::sin(2 * M_PI * ph), v[6] = { 0.2 * std::sin(0.05 * k), 0.05, -0.1 * std::cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), 9.80665 + 0.3 * nrand() };
    for (int i = 0; i < 6; i++) imu[6 * (size_t) k + i] = v[i];
    double ql[4], qr[4];
    euler_to_quat(0.02 * sw, 0.05 * sw, 0, ql);
    euler_to_quat(-0.02 * sw, -0.05 * sw, 0, qr);
    const double lt[3] = { 0.15 * sw, 0.11, -0.86 + 0.02 * std::fmax(0, -sw) }, rt[3] = { -0.15 * sw, -0.11, -0.86 + 0.02 * std::fmax(0, sw) };
    double *f = &feet[14 * (size_t) k];
    for (int i = 0; i < 3; i++) { f[i] = lt[i]; f[7 + i] = rt[i]; }
    for (int i = 0; i < 4; i++) { f[3 + i] = ql[i]; f[10 + i] = qr[i]; }
    forces[2 * (size_t) k] = 900 * wl + 5 * nrand();
    forces[2 * (size_t) k + 1] = 900 * wr + 5 * nrand();
    for (int side = 0; side < 2; side++) {
      const double sgn = side ? -1.0 : 1.0, lift = std::fmax(0.0, -sgn * sw);
      float *p = &jpos[12 * (size_t) k + 6 * side];
      p[0] = (float) (0.05 * sgn * sw); p[1] = (float) (0.03 * sgn + 0.02 * sw); p[2] = (float) (-0.35 - sgn * 0.25 * sw - 0.2 * lift);
      p[3] = (float) (0.7 + 0.5 * lift); p[4] = (float) (-0.35 + sgn * 0.125 * sw - 0.3 * lift); p[5] = (float) (-0.03 * sgn - 0.02 * sw);
      for (int j = 0; j < 6; j++) jeff[12 * (size_t) k + 6 * side + j] = (float) (40 * nrand());
    }
  }
  msgs::joint_state_t js;
  js.joint_name = joint_names;
  js.mem = PB_HOST_BROADCAST;
  auto feed = [&](int k) {
    const int64_t utime = 1000000 + (int64_t) (k + 1) * 2000;
    msgs::ins_t im{ utime, BatchArray(&imu[6 * (size_t) k], PB_HOST_BROADCAST), BatchArray(&imu[6 * (size_t) k + 3], PB_HOST_BROADCAST) };
    on_ins(&im);
    if (joints) {
      msgs::six_axis_force_torque_array_t ft{ utime, BatchArray(&forces[2 * (size_t) k], PB_HOST_BROADCAST) };
      legodo_handler.forceTorqueHandler(&ft, B);
      js.utime = utime;
      js.joint_position = &jpos[12 * (size_t) k];
      js.joint_effort = &jeff[12 * (size_t) k];
      on_joints(&js);
    } else {
      msgs::foot_state_t fs{ utime, BatchArray(&feet[14 * (size_t) k], PB_HOST_BROADCAST), BatchArray(&forces[2 * (size_t) k], PB_HOST_BROADCAST) };
      legodo_handler.forceTorqueHandler();
      on_feet(&fs);
    }
    if (vo_every > 0 && k % vo_every == vo_every - 1) {  // one camera's delta since the last keyframe, for every filter
      const double vt[3] = { 0.004 * std::sin(0.01 * k), 0.002, -0.001 * std::cos(0.02 * k) };
      double vq[4];
      euler_to_quat(0.002 * std::sin(0.03 * k), -0.001, 0.004 * std::cos(0.01 * k), vq);
      msgs::update_t vo{ utime, fovis_handler.prev_t0_body_utime_, nullptr, BatchArray(vt, PB_HOST_BROADCAST), BatchArray(vq, PB_HOST_BROADCAST) };
      on_fovis(&vo);
      fovis_handler.markKeyframe(&est);
    }
  };
  // The rate quoted is the steady state: with a history window the warm-up runs until the window has filled and slid once (messages are
  // 2 ms apart).  While it fills, every message takes new device blocks from the driver (hipMalloc: ~10 us mostly, sporadically 10-20 ms
  // once a few GB are held); SHIM_SWEEP_TRACE=1 prints the rate of every 100 messages and shows that transient.
  int warm = T / 10;
  if (slots != "0") warm = std::max(warm, (int) std::min<long long>(T / 2, atoll(span.c_str()) / 2000 + 100));
  for (int k = 0; k < warm; k++) feed(k);
  pb_sync(est.ctx);
  const auto t0 = std::chrono::steady_clock::now();
  if (getenv("SHIM_SWEEP_TRACE")) {
    auto tp = t0;
    for (int k = warm; k < T; k++) {
      feed(k);
      if ((k + 1) % 100 == 0) {
        pb_sync(est.ctx);
        const auto tn = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  messages ..%d: %.1f us each\n", k + 1, std::chrono::duration<double>(tn - tp).count() / 100 * 1e6);
        tp = tn;
      }
    }
  } else
    for (int k = warm; k < T; k++) feed(k);
  pb_sync(est.ctx);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (est.last_status != PB_OK) { std::fprintf(stderr, "shim_sweep_rate: %s\n", pb_last_error(est.ctx)); return 1; }
  RBIS head;
  RBIM cov;
  est.getHeadState(head, cov);
  double sum = 0;
  bool finite = true;
  for (int b = 0; b < B; b++)
    for (int i = 0; i < n; i++) { sum += std::fabs(head(i, b)); finite = finite && std::isfinite(head(i, b)); }
  std::printf("shim sweep: %d filters x %d message pairs (n=%d, history_slots=%s%s, %s, %s, legodo mode %s): %.1f us per IMU + %s pair, "
              "%.3e filter-steps/s, one-kernel pairs %lld of %lld fused, dropped %lld, checksum %.6g %s\n", B, T - warm, n, slots.c_str(),
              vo_every > 0 ? (", VO every " + std::to_string(vo_every)).c_str() : "", joints ? "joint-state log" : "foot-state log",
              two_launches ? "odometry as its own launch" : "odometry inside the step kernel", lomode.c_str(), dt / (T - warm) * 1e6, joints ? "joint-state" : "foot-state",
              (double) B * (T - warm) / dt, (long long) est.leg_kernel_pairs, (long long) est.fused_pairs, (long long) est.dropped_updates, sum,
              finite ? "finite" : "NON-FINITE");
  return finite ? 0 : 1;
}
