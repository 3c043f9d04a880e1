// test_leg_joints.cpp -- LegOdoHandler::processMessage(const joint_state_t*, MavStateEstimator*): the reference's leg-odometry
// handler (rbis_legodo_update.cpp:206-280) end to end from a joint-state log -- robot model from URDF text (ModelClient),
// force/torque and controller-contact messages, torque adjustment, forward kinematics, contact logic, pelvis integration and
// LegOdoCommon's measurement, on the GPU for every filter -- against the oracle chain po_torque_adjust -> po_fk ->
// po_leg_update_wc (given the oracle filter's own head pose) -> po_legodo_create_measurement -> po_indexed_update.
//   argv[1]: legodo mode (lin_rate | lin_rot_rate | pos_and_lin_rate)      argv[2]: contact mode (alt | standing | ctrl)
//   argv[3]: "fuse" = state_estimator.fuse_ins_legodo ("fuse3": + fuse_corrections)                     argv[4]: "bcast" = one robot's log for every filter,
//                                                                                   "device" = per-filter blocks already in HBM
//   argv[5]: state_estimator.legodo.filter_joint_positions (none | lowpass | kalman; leg_estimate.cpp:411-428), the oracle chain
//            then has po_joint_filter between the torque adjustment and the kinematics
//   "n21" anywhere: the 21-state filter (biases estimated online) instead of the 15-state one
// Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <string>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x4a4f494e54ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }
static double ramp(double x) { return x < 0 ? 0 : (x > 0.05 ? 1.0 : x / 0.05); }

// a biped with 6-DoF legs (test values; the reference's URDF is not in its tree), with the clutter a real URDF has around
// the elements the forward kinematics reads
static const char *URDF = R"(<?xml version="1.0"?>
<robot name="biped">
  <link name="pelvis"><inertial><mass value="17.8"/><origin xyz="0 0 0" rpy="0 0 0"/></inertial></link>
  <link name="l_uglut"/><link name="l_lglut"/><link name="l_uleg"/><link name="l_lleg"/><link name="l_talus"/><link name="l_foot"/>
  <link name="r_uglut"/><link name="r_lglut"/><link name="r_uleg"/><link name="r_lleg"/><link name="r_talus"/><link name="r_foot"/>
  <link name="utorso"/><link name="l_sole"/>
  <joint name="back_bkz" type="revolute"><origin xyz="-0.0125 0 0" rpy="0 0 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="utorso"/>
    <limit effort="124" lower="-0.6" upper="0.6" velocity="12"/></joint>
  <joint name="l_leg_hpz" type="revolute"><origin xyz="0 0.089 0" rpy="0 0 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="l_uglut"/>
    <dynamics damping="0.1" friction="0"/><limit effort="110" lower="-0.17" upper="1.1" velocity="12"/></joint>
  <joint name="l_leg_hpx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="l_uglut"/><child link="l_lglut"/></joint>
  <joint name="l_leg_hpy" type="revolute"><origin rpy="0 0 0" xyz="0.05 0.0225 -0.066"/><axis xyz="0 1 0"/><parent link="l_lglut"/><child link="l_uleg"/></joint>
  <joint name="l_leg_kny" type="revolute"><origin xyz="-0.05 0 -0.374" rpy="0 0.02 0"/><axis xyz="0 1 0"/><parent link="l_uleg"/><child link="l_lleg"/></joint>
  <joint name="l_leg_aky" type="continuous"><origin xyz="0 0 -0.422" rpy="0 0 0"/><axis xyz="0 1 0"/><parent link="l_lleg"/><child link="l_talus"/></joint>
  <joint name="l_leg_akx" type="revolute"><origin xyz="0 0 0" rpy="0 0 0"/><parent link="l_talus"/><child link="l_foot"/></joint>
  <joint name="l_sole_fixed" type="fixed"><origin xyz="0.05 0 -0.081" rpy="0 0 0"/><parent link="l_foot"/><child link="l_sole"/></joint>
  <joint name="r_leg_hpz" type="revolute"><origin xyz="0 -0.089 0" rpy="0 0 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="r_uglut"/></joint>
  <joint name="r_leg_hpx" type="revolute"><origin xyz="0 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/><parent link="r_uglut"/><child link="r_lglut"/></joint>
  <joint name="r_leg_hpy" type="revolute"><origin xyz="0.05 -0.0225 -0.066" rpy="0 0 0"/><axis xyz="0 1 0"/><parent link="r_lglut"/><child link="r_uleg"/></joint>
  <joint name="r_leg_kny" type="revolute"><origin xyz="-0.05 0 -0.374" rpy="0 0.02 0"/><axis xyz="0 1 0"/><parent link="r_uleg"/><child link="r_lleg"/></joint>
  <joint name="r_leg_aky" type="revolute"><origin xyz="0 0 -0.422" rpy="0 0 0"/><axis xyz="0 1 0"/><parent link="r_lleg"/><child link="r_talus"/></joint>
  <joint name="r_leg_akx" type="revolute"><origin xyz="0 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/><parent link="r_talus"/><child link="r_foot"/></joint>
  <transmission name="l_leg_kny_trans" type="pr2_mechanism_model/SimpleTransmission"><actuator name="l_leg_kny_motor"/><joint name="l_leg_kny"/>
    <mechanicalReduction>1</mechanicalReduction></transmission>
</robot>)";

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);
  // "late" anywhere: legodo.roll_forward_on_receive = false and posterior checkpoints on.  On ordinary ticks the joint state is
  // stamped 300 us behind its IMU sample and simply stays unapplied until the next IMU message rolls forward.  On every 10th tick
  // the messages arrive OUT OF ORDER: a scan-match pose stamped 200 us BEFORE the IMU sample is delivered and applied first, then
  // the IMU sample (held back by fuse_ins_legodo), then the joint state, stamped 300 us before the IMU sample -- older than the
  // held INS step AND older than an update that has been applied, without roll_forward, and (with device / broadcast joint blocks
  // under "fuse") with its odometry still deferred.  ADVICE r04: the estimator must neither apply it on top of the head nor apply
  // the pose twice; it restores the checkpoint in front of it, makes the odometry from THAT state and replays.  The oracle runs the
  // same messages in time order.
  bool late = false;
  {
    int w = 1;
    for (int i = 1; i < argc; i++) {
      if (std::string(argv[i]) == "late") late = true;
      else argv[w++] = argv[i];
    }
    argc = w;
  }
  const std::string lomode = argc > 1 ? argv[1] : "lin_rate";
  const std::string cmode = argc > 2 ? argv[2] : "alt";
  // "fuse3": fuse_ins_legodo + fuse_corrections -- the estimator holds the finished [INS, leg odometry] pair back until the next
  // message; with "device" blocks the test then OVERWRITES the blocks as soon as the handler has returned (they are the
  // caller's, valid until the next message): the odometry must have consumed them by then
  const bool fuse3 = argc > 3 && std::string(argv[3]) == "fuse3";
  const bool fuse = fuse3 || (argc > 3 && std::string(argv[3]) == "fuse");
  const bool bcast = argc > 4 && std::string(argv[4]) == "bcast";
  const bool device = argc > 4 && std::string(argv[4]) == "device";
  const std::string jfilt = argc > 5 ? argv[5] : "none";
  const int jmode = jfilt == "lowpass" ? 1 : (jfilt == "kalman" ? 2 : 0);
  const int B = 64, T = 900, NJ = 16, ZERO = 3;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", late ? "12" : "0");
  if (late) param.set("state_estimator.history_checkpoint_every", "1");
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  param.set("state_estimator.fuse_corrections", fuse3 ? "true" : "false");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.002);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  // r_vxyz = 5 / 10 m/s: the synthetic gait is not the motion the synthetic IMU measures; with the reference's 0.1 m/s the
  // orientation feedback through the 0.86 m lever arm (430 m/s per radian at 2 ms) makes the closed loop chaotic
  // (tests/test_leg_odometry.py R_VXYZ)
  param.applyOverrides("state_estimator.legodo.mode=" + lomode + "|state_estimator.legodo.r_xyz=2.0|state_estimator.legodo.r_vxyz=5|"
                       "state_estimator.legodo.r_vang=3|state_estimator.legodo.r_vxyz_uncertain=10|state_estimator.legodo.r_vang_uncertain=9|"
                       "state_estimator.legodo.schmitt_low_threshold=475|state_estimator.legodo.schmitt_high_threshold=525|"
                       "state_estimator.legodo.schmitt_low_delay=7000|state_estimator.legodo.schmitt_high_delay=7000|"
                       "state_estimator.legodo.filter_contact_events=true|state_estimator.legodo.zero_initial_velocity=3|"
                       "state_estimator.legodo.initialization_mode=zero|state_estimator.legodo.left_standing_link=l_foot|"
                       "state_estimator.legodo.right_standing_link=r_foot|state_estimator.legodo.filter_joint_positions=" + jfilt + "|"
                       "state_estimator.legodo.joint_process_noise=0.01|state_estimator.legodo.joint_observation_noise=0.0005|"
                       "state_estimator.legodo.total_force=900|state_estimator.legodo.standing_schmitt_level=0.65|"
                       "state_estimator.legodo.torque_adjustment=true|state_estimator.legodo.adjustment_joints=l_leg_hpz,l_leg_kny,r_leg_kny,r_leg_akx,back_bkz|"
                       "state_estimator.legodo.adjustment_gain=7000,10000,10000,0,5000");
  param.set("state_estimator.legodo.init_contact_mode", cmode == "standing" ? "standing" : "walking");
  param.set("state_estimator.legodo.use_controller_input", cmode == "ctrl" ? "true" : "false");
  for (const char *s : { "ins", "legodo", "scan_matcher" }) {
    param.set(std::string("state_estimator.") + s + ".downsample_factor", "1");
    param.set(std::string("state_estimator.") + s + ".roll_forward_on_receive", (late && std::string(s) == "legodo") ? "false" : "true");
    param.set(std::string("state_estimator.") + s + ".utime_offset", "0");
  }
  param.applyOverrides("state_estimator.scan_matcher.mode=position_yaw|state_estimator.scan_matcher.r_pxy=0.05|"
                       "state_estimator.scan_matcher.r_pz=0.05|state_estimator.scan_matcher.r_yaw=1.0");
  // the joint_state_t layout of the log: 12 leg joints among 4 others
  const std::vector<std::string> names = { "back_bkz", "l_leg_hpz", "l_leg_hpx", "l_leg_hpy", "neck_ay", "l_leg_kny", "l_leg_aky", "l_leg_akx",
                                           "l_arm_shz", "r_leg_hpz", "r_leg_hpx", "r_leg_hpy", "r_arm_shz", "r_leg_kny", "r_leg_aky", "r_leg_akx" };
  ModelClient model;
  if (!model.fromURDFString(URDF, bot_param_get_str_or_fail(&param, "state_estimator.legodo.left_standing_link"),
                            bot_param_get_str_or_fail(&param, "state_estimator.legodo.right_standing_link")) ||
      model.left_chain.size() != 6 || model.right_chain.size() != 6 || model.left_chain[3].name != "l_leg_kny" || model.left_chain[3].rpy[1] != 0.02 ||
      model.left_chain[5].axis[0] != 1.0 /* URDF default axis */ || model.left_chain[2].xyz[2] != -0.066) {
    printf("FAIL: URDF chains\n");
    return 1;
  }
  std::vector<ModelClient::Joint> sole;
  if (!ModelClient::chainTo(URDF, "l_sole", sole) || sole.size() != 7 || sole[6].type != 0) { printf("FAIL: fixed joint chain\n"); return 1; }
  // the oracle's view of the same chains (row, type, origin, axis, gain)
  const float gains[5] = { 7000.f, 10000.f, 10000.f, 0.f, 5000.f };
  const char *adj[5] = { "l_leg_hpz", "l_leg_kny", "r_leg_kny", "r_leg_akx", "back_bkz" };
  struct OChain { int n; int type[8], row[8]; double org[48], axis[24]; float gain[8]; } och[2];
  for (int side = 0; side < 2; side++) {
    const auto &ch = side ? model.right_chain : model.left_chain;
    och[side].n = (int) ch.size();
    for (int j = 0; j < och[side].n; j++) {
      och[side].type[j] = ch[j].type;
      och[side].row[j] = (int) (std::find(names.begin(), names.end(), ch[j].name) - names.begin());
      for (int i = 0; i < 3; i++) { och[side].org[6 * j + i] = ch[j].xyz[i]; och[side].org[6 * j + 3 + i] = ch[j].rpy[i]; och[side].axis[3 * j + i] = ch[j].axis[i]; }
      och[side].gain[j] = 0.f;
      for (int a = 0; a < 5; a++) if (ch[j].name == adj[a]) och[side].gain[j] = gains[a];
    }
  }
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0), period(B), phase(B), swing(B);
  std::vector<std::vector<char>> legs(B, std::vector<char>(po_leg_sizeof()));
  std::vector<int> zc(B, ZERO);
  // leg_estimate's lpfilter_ / joint_kf_ (one per joint, leg_estimate.cpp:45-58: SimpleKalmanFilter(process, observation) -- the
  // second value lands in process_noise_vel_, the observation noise keeps its default)
  std::vector<std::vector<po_lowpass>> olp(B, std::vector<po_lowpass>(PO_NUM_FILT_JOINTS));
  std::vector<std::vector<po_skf>> okf(B, std::vector<po_skf>(PO_NUM_FILT_JOINTS));
  for (int b = 0; b < B; b++)
    for (int i = 0; i < PO_NUM_FILT_JOINTS; i++) { po_lowpass_init(&olp[b][i]); po_skf_init(&okf[b][i], 0.01, 0.0005, 5E-4); }
  for (int b = 0; b < B; b++) {
    double q[4];
    po_euler_to_quat(0.05 * (urand() - 0.5), 0.05 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    for (int i = 0; i < 4; i++) { x0.q(i, b) = q[i]; ox[b].quat[i] = q[i]; }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
    const int src = bcast ? 0 : b;
    period[b] = bcast && b ? period[0] : 0.9 + 0.4 * urand();
    phase[b] = bcast && b ? phase[0] : urand();
    swing[b] = bcast && b ? swing[0] : 0.15 + 0.2 * urand();
    (void) src;
    po_leg_init((po_leg *) legs[b].data(), 475, 525, 7000, 7000, 1);
    if (cmode != "alt") po_leg_set_contact_mode((po_leg *) legs[b].data(), cmode == "standing", 900, 0.65, cmode == "ctrl");
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  ScanMatcherHandler sm_handler(&param);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_pose = front_end.addSensor("scan_matcher", &ScanMatcherHandler::processMessage, &sm_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  int n_status[3] = { 0, 0, 0 }, n_pos = 0, n_before_ft = 0;
  {
    LegOdoHandler legodo_handler(&param, &model);
    auto on_joints = front_end.addSensor("legodo", &LegOdoHandler::processMessage, &legodo_handler);
    const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
    const double r5[5] = { 2.0, 5.0, 3.0, 10.0, 9.0 };
    const int omode = lomode == "lin_rate" ? 0 : (lomode == "lin_rot_rate" ? 1 : 2);
    const int W = bcast ? 1 : B;
    std::vector<float> jp((size_t) NJ * W), je((size_t) NJ * W), jv((size_t) NJ * W);
    std::vector<double> fz(2 * (size_t) W);
    // "device": the log segment's blocks are uploaded by the caller and the handler is given device pointers (a device-resident replay)
    float *d_jp = nullptr, *d_je = nullptr, *d_jv = nullptr, *d_fz = nullptr;
    std::vector<float> fabs_fz(2 * (size_t) W);
    if (device) {
      void *p = nullptr;
      const size_t blk = sizeof(float) * (size_t) NJ * W;
      if (pb_malloc(est.ctx, 3 * blk + sizeof(float) * 2 * W, &p) != PB_OK) { printf("FAIL: pb_malloc\n"); return 1; }
      d_jp = (float *) p; d_je = d_jp + (size_t) NJ * W; d_jv = d_je + (size_t) NJ * W; d_fz = d_jv + (size_t) NJ * W;
    }
    int ncl = -1, ncr = -1;
    for (int k = 0; k < T; k++) {
      const int64_t utime = 1000000 + (int64_t) (k + 1) * 2000;
      const double t = (k + 1) * 0.002;
      const double v[6] = { 0.2 * sin(0.05 * k), 0.05, -0.1 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
      msgs::ins_t im{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
      const bool late_tick = late && k % 10 == 9;
      auto oracle_imu = [&]() {
        for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.002, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      };
      if (!late_tick) {
        on_ins(&im);
        oracle_imu();
      }
      for (int b = 0; b < W; b++) {
        double ph = t / period[b] + phase[b];
        ph -= floor(ph);
        double wl = ramp(ph) * ramp(0.6 - ph), wr = ramp(ph - 0.5) * ramp(1.1 - ph) + (ph < 0.1 ? ramp(0.1 - ph) : 0.0);
        if (t < 0.4) wl = wr = 1.0;
        fz[b] = -(900 * wl + 5 * nrand());      // the sensor's sign is not the handler's business: it takes fabs (:234-235)
        fz[W + b] = 900 * wr + 5 * nrand();
        const double sw = sin(2 * M_PI * ph);
        for (int j = 0; j < NJ; j++) {
          jp[(size_t) j * W + b] = (float) (0.3 * nrand());
          je[(size_t) j * W + b] = (float) (40 * nrand());
          jv[(size_t) j * W + b] = (float) (0.5 * nrand());
        }
        for (int side = 0; side < 2; side++) {
          const double sgn = side ? -1.0 : 1.0, lift = fmax(0.0, -sgn * sw);
          const int r0 = side ? 9 : 1, r1 = side ? 13 : 5;
          jp[(size_t) (r0 + 0) * W + b] = (float) (0.05 * sgn * sw);
          jp[(size_t) (r0 + 1) * W + b] = (float) (0.03 * sgn + 0.02 * sw);
          jp[(size_t) (r0 + 2) * W + b] = (float) (-0.35 - sgn * swing[b] * sw - 0.2 * lift);
          jp[(size_t) (r1 + 0) * W + b] = (float) (0.7 + 0.5 * lift);
          jp[(size_t) (r1 + 1) * W + b] = (float) (-0.35 + sgn * swing[b] * sw * 0.5 - 0.3 * lift);
          jp[(size_t) (r1 + 2) * W + b] = (float) (-0.03 * sgn - 0.02 * sw);
        }
      }
      const int64_t js_utime = late_tick ? utime - 300 : (late ? utime + 300 : utime);
      const bool pose_tick = late_tick;
      double ppos[3] = { 0, 0, 0 }, pquat[4] = { 1, 0, 0, 0 };
      if (pose_tick) {
        for (int i = 0; i < 3; i++) ppos[i] = 0.05 * nrand();
        po_euler_to_quat(0.0, 0.0, 0.3 * (urand() - 0.5), pquat);
      }
      msgs::joint_state_t js;
      js.utime = js_utime;
      js.joint_name = names;
      js.joint_position = jp.data();
      js.joint_effort = je.data();
      js.joint_velocity = jv.data();
      js.mem = bcast ? PB_HOST_BROADCAST : PB_HOST;
      if (device) {
        const size_t blk = sizeof(float) * (size_t) NJ * W;
        for (size_t i = 0; i < fabs_fz.size(); i++) fabs_fz[i] = (float) fabs(fz[i]);
        if (pb_memcpy_h2d(est.ctx, d_jp, jp.data(), blk) != PB_OK || pb_memcpy_h2d(est.ctx, d_je, je.data(), blk) != PB_OK ||
            pb_memcpy_h2d(est.ctx, d_jv, jv.data(), blk) != PB_OK || pb_memcpy_h2d(est.ctx, d_fz, fabs_fz.data(), sizeof(float) * 2 * W) != PB_OK) {
          printf("FAIL: upload\n");
          return 1;
        }
        js.joint_position = d_jp; js.joint_effort = d_je; js.joint_velocity = d_jv;
        js.mem = PB_DEVICE;
      }
      if (k == 0) {  // before the first force/torque message nothing is integrated (:208-211)
        const int before = (int) est.history.updateMap.size();
        on_joints(&js);
        n_before_ft = (int) est.history.updateMap.size() - before;
      }
      msgs::six_axis_force_torque_array_t ft{ utime, BatchArray(fz.data(), bcast ? PB_HOST_BROADCAST : PB_HOST) };
      if (device) legodo_handler.forceTorqueDevice(d_fz);
      else legodo_handler.forceTorqueHandler(&ft, B);
      if (cmode == "ctrl" && k >= 50) {
        ncl = 4; ncr = 4;
        if (k % 500 >= 300 && k % 500 < 360) ncl = 2;
        if (k % 500 >= 100 && k % 500 < 150) ncr = 1;
        msgs::controller_foot_contact_t cc{ utime, ncl, ncr };
        legodo_handler.controllerInputHandler(&cc);
      }
      if (late_tick) {   // the pose first (applied), then the IMU sample (held back), then the joint state that is older than both
        const double zero3[3] = { 0, 0, 0 };
        msgs::pose_t pm{ utime - 200, BatchArray(ppos, PB_HOST_BROADCAST), BatchArray(zero3, PB_HOST_BROADCAST), BatchArray(pquat, PB_HOST_BROADCAST) };
        on_pose(&pm);
        on_ins(&im);
      }
      on_joints(&js);
      if (device && fuse3) {  // the caller refills its blocks for the next message right away
        std::vector<float> junk((size_t) NJ * W, 1e9f);
        const size_t blk = sizeof(float) * (size_t) NJ * W;
        if (pb_memcpy_h2d(est.ctx, d_jp, junk.data(), blk) != PB_OK || pb_memcpy_h2d(est.ctx, d_je, junk.data(), blk) != PB_OK ||
            pb_memcpy_h2d(est.ctx, d_fz, junk.data(), sizeof(float) * 2 * W) != PB_OK) { printf("FAIL: upload\n"); return 1; }
      }
      for (int b = 0; b < B; b++) {
        const int s = bcast ? 0 : b;
        double ft_[2][3], fq_[2][4];
        // one robot's joint vector: torque adjustment (rbis_legodo_update.cpp:231-241), then the joint filters (leg_estimate.cpp:411-428)
        float pos[NJ], vel[NJ];
        for (int j = 0; j < NJ; j++) { pos[j] = jp[(size_t) j * W + s]; vel[j] = jv[(size_t) j * W + s]; }
        for (int side = 0; side < 2; side++)
          for (int j = 0; j < och[side].n; j++) {
            const size_t at = (size_t) och[side].row[j] * W + s;
            pos[och[side].row[j]] = po_torque_adjust(jp[at], je[at], och[side].gain[j]);
          }
        po_joint_filter(jmode, olp[b].data(), okf[b].data(), js_utime, NJ, pos, vel);
        for (int side = 0; side < 2; side++) {
          double ang[8];
          for (int j = 0; j < och[side].n; j++) ang[j] = (double) pos[och[side].row[j]];
          po_fk(och[side].n, och[side].type, och[side].org, och[side].axis, ang, ft_[side], fq_[side]);
        }
        double dt3[3], dq[4], cpos[3];
        long prev = 0;
        int cok = 0;
        float status = po_leg_update_wc((po_leg *) legs[b].data(), js_utime, ft_[0], fq_[0], ft_[1], fq_[1], fabs(fz[s]), fabs(fz[W + s]), ncl, ncr,
                                        &ox[b].vec[9], ox[b].quat, dt3, dq, &prev, cpos, &cok);
        n_status[status < 0 ? 0 : (status < 0.5 ? 1 : 2)]++;
        if (status >= 0) {                              // (else "return NULL", :243-255)
          zc[b]--;                                      // :264-268
          if (zc[b] > 0) { dt3[0] = dt3[1] = dt3[2] = 0; dq[0] = 1; dq[1] = dq[2] = dq[3] = 0; cpos[0] = cpos[1] = cpos[2] = 0; }
          int idx[6];
          double z[6], Rd[6], R[36] = { 0 };
          const int m = po_legodo_create_measurement(omode, r5, cpos, dt3, dq, js_utime, prev, cok, status, idx, z, Rd);
          if (m == 6 && omode == 2) n_pos++;
          for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
          po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        }
        if (pose_tick) {                                // in time order the pose FOLLOWS the leg odometry ...
          const int pidx[4] = { 9, 10, 11, 8 };
          double pz[4] = { ppos[0], ppos[1], ppos[2], 0.0 }, pR[16] = { 0 };
          pR[0] = pR[5] = pR[10] = 0.05 * 0.05;
          pR[15] = bot_sq(bot_to_radians(1.0));
          po_indexed_orient_update(4, pidx, pz, pR, pquat, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
        }
      }
      if (late_tick) oracle_imu();                      // ... and the IMU sample follows the pose
    }
    if (late) {   // one more IMU message rolls the last (unapplied) leg odometry forward
      const int64_t utime = 1000000 + (int64_t) (T + 1) * 2000;
      const double v[6] = { 0.01, 0.02, -0.01, 0.1, -0.1, g };
      msgs::ins_t im{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
      on_ins(&im);
      for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.002, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    }
    if (device) {
      est.flushPending();  // (a pending pair still reads the blocks)
      pb_sync(est.ctx);
      pb_free(est.ctx, d_jp);
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
  printf("n=%d mode %s / %s%s%s, joint filter %s: status skip/certain/uncertain %d/%d/%d, position updates %d: rel err vec %.2e quat %.2e cov %.2e ll %.2e (status %d, fused pairs %lld)\n",
         n, lomode.c_str(), cmode.c_str(), fuse ? " fused" : "", bcast ? " bcast" : (device ? " device blocks" : ""), jfilt.c_str(), n_status[0], n_status[1], n_status[2], n_pos, ev / sv, eq, eP / sP,
         el / sl, est.last_status, (long long) est.fused_pairs);
  // Every mode must have run its pairs as ONE kernel where the shim can: lin_rate always; the six-row modes whenever the joint block
  // outlives the handler call (broadcast, device blocks, or the joint filters' device output) -- inside the pair kernel
  // (leg_kernel_pairs).  The one exception: fuse_corrections holds the finished pair back, its measurement is made at once, and a
  // six-row measurement made ahead of time is applied by the stand-alone update.
  const bool six = lomode != "lin_rate";
  const bool expect_fused = fuse && (!six || (!fuse3 && !late && (bcast || device || jmode != 0)));
  const bool fused_ok = !expect_fused || (est.fused_pairs > T / 2 && (!six || est.leg_kernel_pairs > T / 2));
  printf("fused pairs %lld, of them inside the pair kernel %lld (expected to fuse: %s)\n", (long long) est.fused_pairs, (long long) est.leg_kernel_pairs, expect_fused ? "yes" : "no");
  const bool pos_ok = lomode != "pos_and_lin_rate" || n_pos > B * T / 20;
  if (late) printf("late mode: updates re-applied after late arrivals %lld, dropped %lld\n", (long long) est.replayed_updates, (long long) est.dropped_updates);
  if (late && (est.replayed_updates < T / 10 || est.dropped_updates != 0)) { printf("FAIL: the late leg odometry did not take the replay path\n"); return 1; }
  const bool ok = n_before_ft == 0 /* no leg odometry before the first F/T message */ && fused_ok && pos_ok && est.last_status == PB_OK && n_status[0] > 100 &&
                  n_status[1] > 50 && n_status[2] > 100 && ev / sv < 1e-8 && eq < 1e-8 && eP / sP < 1e-8 && el / sl < 1e-8;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
