// test_segments.cpp -- independent log segments as ONE batch (SegmentBatcher, pronto_amd/csrc/segment_batcher.hpp): the reference's
// many-runs workload (motion_estimate/scripts/se-batch-process.sh:17-26,58-74: one se-fusion run per recorded log, each opened
// with a start_timestamp, lcm_front_end.cpp:21-33) as B filters that share every launch.
//   * B DIFFERENT synthetic logs are written with LogWriter -- another robot gait, another IMU stream, another absolute time base,
//     message times that jitter by +-40 us, another length (ragged ends), some with events in front of their start_timestamp;
//     channels: IMU (bot_core::ins_t by run-time schema), FORCE_TORQUE, JOINT_STATE (bot_core types by schema), POSE (scan matcher),
//     a foreign channel;
//   * they are replayed as one batch through InsHandler::processMessage, LegOdoHandler::forceTorqueHandler / processMessage
//     (URDF model, torque adjustment, forward kinematics, contact logic, LegOdoCommon lin_rate; fuse_ins_legodo: one kernel per
//     IMU + joint-state pair) and ScanMatcherHandler::processMessage (position_yaw);
//   * every filter must equal the ORACLE run of ITS OWN log alone (po_imu_process_step, po_torque_adjust -> po_fk ->
//     po_leg_update_wc on the oracle filter's own pose with the log's own time stamps -> po_indexed_update), <= 1e-9.
// argv: "n21" = 21 states, "nofuse" = without fuse_ins_legodo, a directory for the logs;
//   "stream"  the same replay through SegmentStreamer (segment_stream.hpp): memory-mapped logs decoded ahead in chunks, one upload per
//             chunk, the handlers called with PB_DEVICE messages;  "chunk7": chunks of 7 batched messages (every chunk boundary case);
//   "kvh"     (with "stream") the IMU arrives as the reference's Atlas channel -- bot_core::kvh_raw_imu_batch_t on ATLAS_IMU_BATCH
//             (fusion.cpp:161-163): every message repeats packets of the one before, some carry NO new packet, the first ones are
//             shorter; InsHandler::processMessageAtlasSegments with atlas_filter = true (one IMUStream state per segment, the notch
//             cascade on the device with per-filter packet counts, raw_dt and message-time dt per filter); the oracle runs
//             imu_stream.cpp:62-98 + the notch cascade + sensor_handlers.cpp:199-252 per segment.
//   "kvhbatch" the same KVH logs through the per-message SegmentBatcher (page-locked host blocks, staged by the handler).
// Exit code 0 + "PASS".  Needs a GPU.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <string>
#include <vector>

#include "test_n.hpp"
#include "../../pronto_amd/csrc/segment_batcher.hpp"
#include "../../pronto_amd/csrc/segment_stream.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x5345474DULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }
static double ramp(double x) { return x < 0 ? 0 : (x > 0.05 ? 1.0 : x / 0.05); }

static const char *URDF = R"(<robot name="biped">
  <joint name="l_leg_hpz" type="revolute"><origin xyz="0 0.089 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="l_uglut"/></joint>
  <joint name="l_leg_hpx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="l_uglut"/><child link="l_lglut"/></joint>
  <joint name="l_leg_hpy" type="revolute"><origin xyz="0.05 0.0225 -0.066"/><axis xyz="0 1 0"/><parent link="l_lglut"/><child link="l_uleg"/></joint>
  <joint name="l_leg_kny" type="revolute"><origin xyz="-0.05 0 -0.374" rpy="0 0.02 0"/><axis xyz="0 1 0"/><parent link="l_uleg"/><child link="l_lleg"/></joint>
  <joint name="l_leg_aky" type="revolute"><origin xyz="0 0 -0.422"/><axis xyz="0 1 0"/><parent link="l_lleg"/><child link="l_talus"/></joint>
  <joint name="l_leg_akx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="l_talus"/><child link="l_foot"/></joint>
  <joint name="r_leg_hpz" type="revolute"><origin xyz="0 -0.089 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="r_uglut"/></joint>
  <joint name="r_leg_hpx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="r_uglut"/><child link="r_lglut"/></joint>
  <joint name="r_leg_hpy" type="revolute"><origin xyz="0.05 -0.0225 -0.066"/><axis xyz="0 1 0"/><parent link="r_lglut"/><child link="r_uleg"/></joint>
  <joint name="r_leg_kny" type="revolute"><origin xyz="-0.05 0 -0.374" rpy="0 0.02 0"/><axis xyz="0 1 0"/><parent link="r_uleg"/><child link="r_lleg"/></joint>
  <joint name="r_leg_aky" type="revolute"><origin xyz="0 0 -0.422"/><axis xyz="0 1 0"/><parent link="r_lleg"/><child link="r_talus"/></joint>
  <joint name="r_leg_akx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="r_talus"/><child link="r_foot"/></joint>
</robot>)";

// the bot_core definitions the logs use (libbot's; not in the reference tree -- written out here as the user of the library would)
static const char *BOT_CORE_LCM = R"(package bot_core;
struct ins_t { int64_t utime; int64_t device_time; double gyro[3]; double mag[3]; double accel[3]; double quat[4]; double pressure; double rel_alt; }
struct joint_state_t { int64_t utime; int16_t num_joints; string joint_name[num_joints]; float joint_position[num_joints];
  float joint_velocity[num_joints]; float joint_effort[num_joints]; }
struct six_axis_force_torque_t { int64_t utime; double force[3]; double moment[3]; }
struct six_axis_force_torque_array_t { int64_t utime; int32_t num_sensors; string names[num_sensors]; six_axis_force_torque_t sensors[num_sensors]; }
struct pose_t { int64_t utime; double pos[3]; double vel[3]; double orientation[4]; double rotation_rate[3]; double accel[3]; }
struct kvh_raw_imu_t { int64_t utime; int64_t packet_count; double delta_rotation[3]; double linear_acceleration[3]; }
struct kvh_raw_imu_batch_t { int64_t utime; int32_t num_packets; kvh_raw_imu_t raw_imu[num_packets]; }
)";

struct Packet { int64_t utime, count; double drot[3], lacc[3]; };
struct Tick {   // one tick of one segment, as the oracle replays it
  int n_new = 0, pk_hi = -1;   // "kvh": new packets in this tick's batch message, index of its newest packet
  int64_t imu_utime, js_utime;
  double gyro[3], accel[3], fz[2];
  float jp[16], je[16];
  bool pose;
  double pos[3], quat[4];
};

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);
  bool fuse = true, stream = false, kvh = false, chunk7 = false;
  std::string dir = "/tmp";
  for (int i = 1; i < argc; i++) {
    if (std::string(argv[i]) == "nofuse") fuse = false;
    else if (std::string(argv[i]) == "stream") stream = true;
    else if (std::string(argv[i]) == "kvh") kvh = stream = true;
    else if (std::string(argv[i]) == "kvhbatch") kvh = true;      // the KVH channel through the per-message SegmentBatcher
    else if (std::string(argv[i]) == "chunk7") chunk7 = true;
    else if (argv[i][0] == '/') dir = argv[i];
  }
  const int KVH_PACKETS = 5;
  // "rate <segments> <ticks>": no oracle, time the replay (segments x messages per second, PCIe and log decoding included)
  int rate_B = 0, rate_T = 0;
  for (int i = 1; i + 2 < argc; i++)
    if (std::string(argv[i]) == "rate") {
      rate_B = atoi(argv[i + 1]);
      rate_T = atoi(argv[i + 2]);
    }
  const bool rate = rate_B > 0;
  const int B = rate ? rate_B : 64, T = rate ? rate_T : 420, NJ = 16;
  double g;
  po_get_constants(&g, nullptr);
  pronto_wire::Schema schema;
  std::string err;
  if (!schema.parse(BOT_CORE_LCM, &err)) { printf("schema: %s\nFAIL\n", err.c_str()); return 1; }
  const std::vector<std::string> names = { "back_bkz", "l_leg_hpz", "l_leg_hpx", "l_leg_hpy", "neck_ay", "l_leg_kny", "l_leg_aky", "l_leg_akx",
                                           "l_arm_shz", "r_leg_hpz", "r_leg_hpx", "r_leg_hpy", "r_arm_shz", "r_leg_kny", "r_leg_aky", "r_leg_akx" };

  // ---- write B different logs ----
  // Rate runs of the streamer ("rate ... stream"): 16 long recordings and every segment its own WINDOW of T ticks of one of them
  // (a start_timestamp found by bisection and an end_timestamp) -- N runs over parts of a few long logs, the shape a batch of
  // se-fusion runs over recorded sessions has; 16 384 separate files of that length would not fit the box.
  const bool windows = rate && stream;
  const int n_logs = windows ? std::min(B, 16) : B;
  const int per_log = (B + n_logs - 1) / n_logs;
  std::vector<std::vector<Tick>> ticks((size_t) n_logs);
  std::vector<std::vector<Packet>> packets((size_t) n_logs);
  std::vector<int64_t> start_ts((size_t) B, 0), end_ts((size_t) B, 0), log_base((size_t) n_logs, 0);
  std::vector<std::string> paths((size_t) B), log_paths((size_t) n_logs);
  for (int s = 0; s < n_logs; s++) {
    const int Ts = windows ? T + 20 * per_log : (rate ? T : T - (s % 5) * 17);            // ragged ends
    const int64_t base = 1000000000LL * (s + 1) + 12345 * s;   // another absolute time base per recording
    log_base[(size_t) s] = base;
    const double period = 0.9 + 0.4 * urand(), phase = urand(), swing = 0.15 + 0.2 * urand();
    log_paths[(size_t) s] = dir + "/segment_" + std::to_string(s) + ".lcmlog";
    if (!windows) paths[(size_t) s] = log_paths[(size_t) s];
    pronto_wire::LogWriter log(log_paths[(size_t) s]);
    if (!log.good()) { printf("cannot write %s\nFAIL\n", log_paths[(size_t) s].c_str()); return 1; }
    if (!windows && s % 4 == 1) {   // events in front of the start_timestamp this segment is opened with: they must not be replayed
      for (int k = 0; k < 9; k++) {
        pronto_wire::Writer w;
        if (kvh) {
          w.u64(schema.fingerprint("bot_core.kvh_raw_imu_batch_t"));
          w.i64(base - 50000 + 2000 * k); w.i32(1);
          w.i64(base - 50000 + 2000 * k); w.i64(9000000 + k);   // (a packet count that would also freeze the de-duplication)
          for (int i = 0; i < 6; i++) w.f64(100.0);
        } else {
          w.u64(schema.fingerprint("bot_core.ins_t"));
          w.i64(base - 50000 + 2000 * k); w.i64(0);
          for (int i = 0; i < 16; i++) w.f64(100.0);   // nonsense that would wreck the filter
        }
        log.write(base - 50000 + 2000 * k, kvh ? "ATLAS_IMU_BATCH" : "IMU", w.buf);
      }
      start_ts[(size_t) s] = base - 1000;
    }
    ticks[(size_t) s].resize((size_t) Ts);
    for (int k = 0; k < Ts; k++) {
      Tick &tk = ticks[(size_t) s][(size_t) k];
      const int64_t nominal = base + (int64_t) (k + 1) * 2000;
      tk.imu_utime = nominal + (int64_t) (80 * (urand() - 0.5));        // +-40 us of time-stamp jitter, per segment
      tk.js_utime = nominal + 300 + (int64_t) (80 * (urand() - 0.5));
      const double t = (k + 1) * 0.002;
      for (int i = 0; i < 3; i++) tk.gyro[i] = 0.2 * sin(0.05 * k + s + i) + 0.01 * nrand();
      for (int i = 0; i < 3; i++) tk.accel[i] = 0.3 * nrand() + (i == 2 ? g : 0.0);
      double ph = t / period + phase;
      ph -= floor(ph);
      double wl = ramp(ph) * ramp(0.6 - ph), wr = ramp(ph - 0.5) * ramp(1.1 - ph) + (ph < 0.1 ? ramp(0.1 - ph) : 0.0);
      if (t < 0.4) wl = wr = 1.0;
      tk.fz[0] = -(900 * wl + 5 * nrand());
      tk.fz[1] = 900 * wr + 5 * nrand();
      const double sw = sin(2 * M_PI * ph);
      for (int j = 0; j < NJ; j++) { tk.jp[j] = (float) (0.3 * nrand()); tk.je[j] = (float) (40 * nrand()); }
      for (int side = 0; side < 2; side++) {
        const double sgn = side ? -1.0 : 1.0, lift = fmax(0.0, -sgn * sw);
        const int r0 = side ? 9 : 1, r1 = side ? 13 : 5;
        tk.jp[r0 + 0] = (float) (0.05 * sgn * sw);
        tk.jp[r0 + 1] = (float) (0.03 * sgn + 0.02 * sw);
        tk.jp[r0 + 2] = (float) (-0.35 - sgn * swing * sw - 0.2 * lift);
        tk.jp[r1 + 0] = (float) (0.7 + 0.5 * lift);
        tk.jp[r1 + 1] = (float) (-0.35 + sgn * swing * sw * 0.5 - 0.3 * lift);
        tk.jp[r1 + 2] = (float) (-0.03 * sgn - 0.02 * sw);
      }
      tk.pose = (k % 20 == 19);
      for (int i = 0; i < 3; i++) tk.pos[i] = 0.05 * nrand();
      po_euler_to_quat(0.0, 0.0, 0.3 * (urand() - 0.5), tk.quat);
      // the events of this tick, in the order every recording of this robot has them: IMU, force/torque, joint state, [pose]
      if (kvh) {
        // 1 kHz packets in 500 Hz batch messages: normally two new packets, sometimes one or three, one message in ten none at all
        // (imu_stream.cpp: "happens all the time"); each message repeats the packets in front, newest first
        const double u = urand();
        tk.n_new = k == 0 ? 2 : (u < 0.1 ? 0 : (u < 0.25 ? 1 : (u < 0.9 ? 2 : 3)));
        std::vector<Packet> &pk = packets[(size_t) s];
        for (int j = 0; j < tk.n_new; j++) {
          Packet p;
          p.count = 5000 + 3 * s + (int64_t) pk.size();
          p.utime = base + (int64_t) pk.size() * 1000 + (int64_t) (40 * (urand() - 0.5));
          for (int i = 0; i < 3; i++) { p.drot[i] = 0.001 * (0.2 * sin(0.025 * (double) pk.size() + s + i) + 0.01 * nrand()); p.lacc[i] = 0.3 * nrand() + (i == 2 ? g : 0.0); }
          pk.push_back(p);
        }
        tk.pk_hi = (int) pk.size() - 1;
        const int np = std::min(KVH_PACKETS, tk.pk_hi + 1);
        pronto_wire::Writer w;
        w.u64(schema.fingerprint("bot_core.kvh_raw_imu_batch_t"));
        w.i64(tk.imu_utime); w.i32(np);
        for (int j = 0; j < np; j++) {
          const Packet &p = pk[(size_t) (tk.pk_hi - j)];
          w.i64(p.utime); w.i64(p.count); w.f64s(p.drot, 3); w.f64s(p.lacc, 3);
        }
        log.write(tk.imu_utime, "ATLAS_IMU_BATCH", w.buf);
      } else {
        pronto_wire::Writer w;
        w.u64(schema.fingerprint("bot_core.ins_t"));
        w.i64(tk.imu_utime); w.i64(tk.imu_utime + 17);
        w.f64s(tk.gyro, 3);
        for (int i = 0; i < 3; i++) w.f64(0.1 * i);
        w.f64s(tk.accel, 3);
        for (int i = 0; i < 4; i++) w.f64(i == 0);
        w.f64(1013.0); w.f64(0.0);
        log.write(tk.imu_utime, "IMU", w.buf);
      }
      if (k % 11 == 3) log.write(tk.imu_utime + 50, "SOMETHING_ELSE", std::vector<uint8_t>(13, 0x5A));
      pronto_wire::Writer f;
      f.u64(schema.fingerprint("bot_core.six_axis_force_torque_array_t"));
      f.i64(tk.imu_utime + 100); f.i32(2); f.str("l_foot"); f.str("r_foot");
      for (int k2 = 0; k2 < 2; k2++) {
        f.i64(tk.imu_utime + 100);
        f.f64(1.0); f.f64(-2.0); f.f64(tk.fz[k2]);
        f.f64(0.1); f.f64(0.2); f.f64(0.3);
      }
      log.write(tk.imu_utime + 100, "FORCE_TORQUE", f.buf);
      pronto_wire::Writer j;
      j.u64(schema.fingerprint("bot_core.joint_state_t"));
      j.i64(tk.js_utime); j.i16((int16_t) NJ);
      for (int q = 0; q < NJ; q++) j.str(names[(size_t) q]);
      for (int q = 0; q < NJ; q++) j.f32(tk.jp[q]);
      for (int q = 0; q < NJ; q++) j.f32(0.0f);
      for (int q = 0; q < NJ; q++) j.f32(tk.je[q]);
      log.write(tk.js_utime, "JOINT_STATE", j.buf);
      if (tk.pose) {
        pronto_wire::Writer p;
        p.u64(schema.fingerprint("bot_core.pose_t"));
        p.i64(tk.js_utime + 200);
        p.f64s(tk.pos, 3);
        for (int i = 0; i < 3; i++) p.f64(0.0);
        p.f64s(tk.quat, 4);
        for (int i = 0; i < 6; i++) p.f64(0.0);
        log.write(tk.js_utime + 200, "POSE_SCAN", p.buf);
      }
    }
  }

  if (windows)
    for (int s = 0; s < B; s++) {
      const int l = s % n_logs, o = s / n_logs;
      paths[(size_t) s] = log_paths[(size_t) l];
      // between two ticks, a whole number of pose periods in: the window opens with an IMU message and every run sees its sparse
      // channel at the same place (messages are aligned by index per channel)
      start_ts[(size_t) s] = log_base[(size_t) l] + (int64_t) o * 20 * 2000 + 1000;
      end_ts[(size_t) s] = start_ts[(size_t) s] + (int64_t) T * 2000;
    }

  // ---- the estimator and the handlers, configured with the reference's keys ----
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", "0");
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.002);
  param.set("state_estimator.ins.atlas_filter", kvh ? "true" : "false");
  param.set("state_estimator.ins.atlas_filter_freq", 87.0);
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.legodo.mode=lin_rate|state_estimator.legodo.r_xyz=2.0|state_estimator.legodo.r_vxyz=5|"
                       "state_estimator.legodo.r_vang=3|state_estimator.legodo.r_vxyz_uncertain=10|state_estimator.legodo.r_vang_uncertain=9|"
                       "state_estimator.legodo.schmitt_low_threshold=475|state_estimator.legodo.schmitt_high_threshold=525|"
                       "state_estimator.legodo.schmitt_low_delay=7000|state_estimator.legodo.schmitt_high_delay=7000|"
                       "state_estimator.legodo.filter_contact_events=true|state_estimator.legodo.zero_initial_velocity=3|"
                       "state_estimator.legodo.initialization_mode=zero|state_estimator.legodo.left_standing_link=l_foot|"
                       "state_estimator.legodo.right_standing_link=r_foot|state_estimator.legodo.filter_joint_positions=none|"
                       "state_estimator.legodo.init_contact_mode=walking|state_estimator.legodo.use_controller_input=false|"
                       "state_estimator.legodo.total_force=900|state_estimator.legodo.standing_schmitt_level=0.65|"
                       "state_estimator.legodo.torque_adjustment=true|state_estimator.legodo.adjustment_joints=l_leg_hpz,l_leg_kny,r_leg_kny|"
                       "state_estimator.legodo.adjustment_gain=7000,10000,10000");
  param.applyOverrides("state_estimator.scan_matcher.mode=position_yaw|state_estimator.scan_matcher.r_pxy=0.05|"
                       "state_estimator.scan_matcher.r_pz=0.05|state_estimator.scan_matcher.r_yaw=1.0");
  for (const char *sn : { "ins", "legodo", "scan_matcher" }) {
    param.set(std::string("state_estimator.") + sn + ".downsample_factor", "1");
    param.set(std::string("state_estimator.") + sn + ".roll_forward_on_receive", "true");
    param.set(std::string("state_estimator.") + sn + ".utime_offset", "0");
  }
  ModelClient model;
  if (!model.fromURDFString(URDF, "l_foot", "r_foot") || model.left_chain.size() != 6) { printf("FAIL: URDF\n"); return 1; }
  const char *adj[3] = { "l_leg_hpz", "l_leg_kny", "r_leg_kny" };
  const float gains[3] = { 7000.f, 10000.f, 10000.f };
  struct OChain { int n; int type[8], row[8]; double org[48], axis[24]; float gain[8]; } och[2];
  for (int side = 0; side < 2; side++) {
    const auto &ch = side ? model.right_chain : model.left_chain;
    och[side].n = (int) ch.size();
    for (int j = 0; j < och[side].n; j++) {
      och[side].type[j] = ch[(size_t) j].type;
      och[side].row[j] = (int) (std::find(names.begin(), names.end(), ch[(size_t) j].name) - names.begin());
      for (int i = 0; i < 3; i++) { och[side].org[6 * j + i] = ch[(size_t) j].xyz[i]; och[side].org[6 * j + 3 + i] = ch[(size_t) j].rpy[i]; och[side].axis[3 * j + i] = ch[(size_t) j].axis[i]; }
      och[side].gain[j] = 0.f;
      for (int a = 0; a < 3; a++) if (ch[(size_t) j].name == adj[a]) och[side].gain[j] = gains[a];
    }
  }
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox((size_t) B);
  std::vector<po_rbim> oP((size_t) B);
  std::vector<double> oll((size_t) B, 0.0);
  for (int b = 0; b < B; b++) {
    double q[4];
    po_euler_to_quat(0.05 * (urand() - 0.5), 0.05 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    po_rbis_zero(&ox[(size_t) b]);
    memset(&oP[(size_t) b], 0, sizeof(po_rbim));
    for (int i = 0; i < 4; i++) { x0.q(i, b) = q[i]; ox[(size_t) b].quat[i] = q[i]; }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[(size_t) b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[(size_t) b], &oP[(size_t) b], urand);
  }
  BotTrans ins_to_body;   // a mounted IMU: 90 degrees about z
  ins_to_body.rot_quat[0] = sqrt(0.5); ins_to_body.rot_quat[3] = sqrt(0.5);
  if (kvh) { ins_to_body.trans_vec[0] = 0.01; ins_to_body.trans_vec[1] = -0.02; ins_to_body.trans_vec[2] = 0.03; }   // (the Atlas path applies the whole transform to the acceleration, :227)
  InsHandler ins_handler(&param, &ins_to_body);
  ScanMatcherHandler sm_handler(&param);
  FrontEnd front_end(&param);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  struct Totals {   // what both replayers count
    int64_t batches = 0, segment_messages = 0, ragged = 0, order_violations = 0, undecodable = 0, max_skew_us = 0;
    std::map<std::string, int64_t> per_channel;
  } st;
  RBIS head;
  RBIM cov;
  std::vector<double> ll;
  const char *imu_channel = kvh ? "ATLAS_IMU_BATCH" : "IMU";
  {
    LegOdoHandler legodo_handler(&param, &model);
    auto t0 = std::chrono::steady_clock::now();
    double sec = 0;
    int64_t nb = 0;
    if (stream) {
      SegmentStreamer batch(&est);
      if (chunk7) batch.max_slots = 7;
      if (getenv("SEGMENT_CHUNK_MB")) batch.chunk_budget_bytes = (uint64_t) atoi(getenv("SEGMENT_CHUNK_MB")) << 20;
      if (getenv("SEGMENT_MAX_SLOTS")) batch.max_slots = atoi(getenv("SEGMENT_MAX_SLOTS"));
      for (int s = 0; s < B; s++)
        if (!batch.addSegment(paths[(size_t) s], start_ts[(size_t) s], end_ts[(size_t) s])) { printf("FAIL: cannot open segment %d\n", s); return 1; }
      if (batch.addSegment(paths[0])) { printf("FAIL: a segment too many was accepted by a %d-filter batch\n", B); return 1; }
      if (kvh)
        batch.subscribeKvhBatch(imu_channel, &schema, "bot_core.kvh_raw_imu_batch_t", ins_handler.atlas_filter, KVH_PACKETS,
                                front_end.addSensor("ins", &InsHandler::processMessageAtlasSegments, &ins_handler));
      else
        batch.subscribeIns(imu_channel, &schema, "bot_core.ins_t", front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler));
      batch.subscribeForceTorque("FORCE_TORQUE", &schema, "bot_core.six_axis_force_torque_array_t", [&](const float *abs_fz) { legodo_handler.forceTorqueDevice(abs_fz); });
      batch.subscribeJointState("JOINT_STATE", &schema, "bot_core.joint_state_t", front_end.addSensor("legodo", &LegOdoHandler::processMessage, &legodo_handler));
      batch.subscribePose("POSE_SCAN", &schema, "bot_core.pose_t", front_end.addSensor("scan_matcher", &ScanMatcherHandler::processMessage, &sm_handler));
      t0 = std::chrono::steady_clock::now();
      nb = batch.run();
      est.flushPending();
      pb_sync(est.ctx);
      sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const SegmentStreamer::Stats &ss = batch.stats;
      st.batches = ss.batches; st.segment_messages = ss.segment_messages; st.ragged = ss.ragged; st.order_violations = ss.order_violations;
      st.undecodable = ss.undecodable; st.max_skew_us = ss.max_skew_us; st.per_channel = ss.per_channel;
      if (rate) {
        const double steps = (double) B * T;
        printf("segment stream rate: %d segments x %d ticks (%s + force/torque + joint state per tick, pose every 20th; %d logs, every segment its own "
               "window), n=%d, %s: %.2f s -> %.3g segment-messages/s, %.3g filter-steps/s (memory-mapped logs decoded through the run-time schema, "
               "page-locked chunks, PCIe and kernels all inside), %lld batched messages in %lld chunks of <= %.1f MB, fused pairs %lld, status %d\n",
               B, T, kvh ? "KVH batch IMU" : "IMU", n_logs, n, fuse ? "fused pairs" : "unfused", sec, ss.segment_messages / sec, steps / sec, (long long) ss.batches,
               (long long) ss.chunks, ss.chunk_bytes / 1048576.0, (long long) est.fused_pairs, est.last_status);
        printf("  decode-ahead thread [s]: lead pass %.3f | parallel decode + assembly %.3f | waiting for a free ring slot %.3f\n", ss.t_lead, ss.t_decode, ss.t_wait_free);
        printf("  dispatching thread [s]: waiting for a chunk %.3f | uploads (issue + host wait) %.3f | handlers (enqueues) %.3f | heads of finished runs %.3f\n",
               ss.t_wait_chunk, ss.t_upload, ss.t_handlers, ss.t_final);
        printf("  bytes: %.1f MB uploaded = %.1f B per segment-tick; ragged columns %lld, order violations %lld, undecodable %lld, read-ahead capped %lld, max skew %lld us\n",
               ss.uploaded_bytes / 1048576.0, ss.uploaded_bytes / steps, (long long) ss.ragged, (long long) ss.order_violations, (long long) ss.undecodable,
               (long long) ss.readahead_capped, (long long) ss.max_skew_us);
      }
      batch.finalState(head, cov);           // every run's result: its filter's head at the end of ITS log
      ll = batch.finalLogLikelihood();
    } else {
      SegmentBatcher batch(&est);
      for (int s = 0; s < B; s++)
        if (!batch.addSegment(paths[(size_t) s], start_ts[(size_t) s])) { printf("FAIL: cannot open segment %d\n", s); return 1; }
      if (batch.addSegment(paths[0])) { printf("FAIL: a 65th segment was accepted by a 64-filter batch\n"); return 1; }
      if (kvh)
        batch.subscribeKvhBatch(imu_channel, &schema, "bot_core.kvh_raw_imu_batch_t", ins_handler.atlas_filter, KVH_PACKETS,
                                front_end.addSensor("ins", &InsHandler::processMessageAtlasSegments, &ins_handler));
      else
        batch.subscribeIns("IMU", &schema, "bot_core.ins_t", front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler));
      batch.subscribeForceTorque("FORCE_TORQUE", &schema, "bot_core.six_axis_force_torque_array_t",
                                 [&](const msgs::six_axis_force_torque_array_t *m) { legodo_handler.forceTorqueHandler(m, B); });
      batch.subscribeJointState("JOINT_STATE", &schema, "bot_core.joint_state_t", front_end.addSensor("legodo", &LegOdoHandler::processMessage, &legodo_handler));
      batch.subscribePose("POSE_SCAN", &schema, "bot_core.pose_t", front_end.addSensor("scan_matcher", &ScanMatcherHandler::processMessage, &sm_handler));
      t0 = std::chrono::steady_clock::now();
      nb = batch.run();
      pb_sync(est.ctx);
      sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const SegmentBatcher::Stats &bs = batch.stats;
      st.batches = bs.batches; st.segment_messages = bs.segment_messages; st.ragged = bs.ragged; st.order_violations = bs.order_violations;
      st.undecodable = bs.undecodable; st.max_skew_us = bs.max_skew_us; st.per_channel = bs.per_channel;
      if (rate) {
        printf("segment batch rate: %d segments x %d ticks (IMU + force/torque + joint state per tick, pose every 20th), n=%d, %s: %.2f s -> "
               "%.3g segment-messages/s, %.3g filter-steps/s (log decoding through the run-time schema, page-locked assembly, PCIe and "
               "kernels all inside), %lld batched messages, fused pairs %lld, status %d\n",
               B, T, n, fuse ? "fused pairs" : "unfused", sec, bs.segment_messages / sec, (double) B * T / sec, (long long) bs.batches,
               (long long) est.fused_pairs, est.last_status);
        printf("  where the time went [s]: lead %.3f | read + decode %.3f | book-keeping %.3f | assembly + handlers %.3f | read-ahead %.3f | heads %.3f\n",
               bs.t_lead, bs.t_pull, bs.t_book, bs.t_dispatch, bs.t_fill, bs.t_final);
        printf("  of assembly + handlers: inside the handlers' callbacks %.3f, joint blocks to HBM %.3f\n", bs.t_handler, bs.t_upload);
      }
      batch.finalState(head, cov);
      ll = batch.finalLogLikelihood();
    }
    if (rate) {
      for (int s2 = 0; s2 < n_logs; s2++) remove(log_paths[(size_t) s2].c_str());
      return est.last_status == PB_OK ? 0 : 1;
    }
    if (nb != st.batches) { printf("FAIL: run() returned %lld\n", (long long) nb); return 1; }
  }

  // ---- the oracle: every segment ALONE, from its own log ----
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
  const double r5[5] = { 2.0, 5.0, 3.0, 10.0, 9.0 };
  int n_status[3] = { 0, 0, 0 }, n_kvh_new[4] = { 0, 0, 0, 0 };
  for (int b = 0; b < B; b++) {
    std::vector<char> leg(po_leg_sizeof());
    po_leg_init((po_leg *) leg.data(), 475, 525, 7000, 7000, 1);
    int zc = 3;
    // "kvh": this segment's IMUStream + notch cascade + processMessageAtlas state (imu_stream.hpp:10-37, sensor_handlers.hpp)
    std::vector<po_notch> notch(9);
    po_notch_cascade_init(notch.data(), 87.0, 1000);
    int64_t last_packet = -1, last_packet_utime = 0, prev_utime_atlas = 0;
    for (const Tick &tk : ticks[(size_t) b]) {
      double gb[3], ab[3];
      if (kvh) {
        // IMUStream::convertFromLCMBatch on the message as it was written (newest first, at most KVH_PACKETS packets): the new ones,
        // oldest first, through the cascade; the newest filtered one drives the step (sensor_handlers.cpp:173-197)
        const std::vector<Packet> &pk = packets[(size_t) b];
        const int np = std::min(KVH_PACKETS, tk.pk_hi + 1);
        int n_new = 0;
        double filt[3] = { 0, 0, 0 }, drot[3] = { 0, 0, 0 };
        int64_t utime_delta = 0;
        for (int j = np - 1; j >= 0; j--) {
          const Packet &p = pk[(size_t) (tk.pk_hi - j)];
          if (p.count <= last_packet) continue;
          utime_delta = p.utime - last_packet_utime;
          last_packet = p.count;
          last_packet_utime = p.utime;
          for (int i = 0; i < 3; i++) { filt[i] = p.lacc[i]; drot[i] = p.drot[i]; }
          po_notch_cascade(notch.data(), filt);
          n_new++;
        }
        if (n_new != tk.n_new) { printf("FAIL: the test's own packet bookkeeping (%d new, expected %d)\n", n_new, tk.n_new); return 1; }
        n_kvh_new[std::min(n_new, 3)]++;
        if (n_new > 0) {   // (else: "No new IMU packets detected ... Skipping iteration", return NULL)
          const double raw_dt = utime_delta * 1E-6;
          const double sg[3] = { drot[0] / raw_dt, drot[1] / raw_dt, drot[2] / raw_dt };
          bot_trans_apply_vec(&ins_to_body, filt, ab);            // :227
          bot_quat_rotate_to(ins_to_body.rot_quat, sg, gb);      // :235
          const double integration_dt = prev_utime_atlas == 0 ? 0.002 : (tk.imu_utime - prev_utime_atlas) * 1E-6;   // :239-249
          prev_utime_atlas = tk.imu_utime;
          po_imu_process_step(gb, ab, integration_dt, q4[0], q4[1], q4[2], q4[3], &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
        }
      } else {
        bot_quat_rotate_to(ins_to_body.rot_quat, tk.gyro, gb);
        bot_quat_rotate_to(ins_to_body.rot_quat, tk.accel, ab);
        po_imu_process_step(gb, ab, 0.002, q4[0], q4[1], q4[2], q4[3], &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
      }
      double ft_[2][3], fq_[2][4];
      for (int side = 0; side < 2; side++) {
        double ang[8];
        for (int j = 0; j < och[side].n; j++) {
          const int r = och[side].row[j];
          ang[j] = (double) po_torque_adjust(tk.jp[r], tk.je[r], och[side].gain[j]);
        }
        po_fk(och[side].n, och[side].type, och[side].org, och[side].axis, ang, ft_[side], fq_[side]);
      }
      double dt3[3], dq[4], cpos[3];
      long prev = 0;
      int cok = 0;
      const float status = po_leg_update_wc((po_leg *) leg.data(), tk.js_utime, ft_[0], fq_[0], ft_[1], fq_[1], fabs(tk.fz[0]), fabs(tk.fz[1]), -1, -1,
                                            &ox[(size_t) b].vec[9], ox[(size_t) b].quat, dt3, dq, &prev, cpos, &cok);
      n_status[status < 0 ? 0 : (status < 0.5 ? 1 : 2)]++;
      if (status >= 0) {
        zc--;
        if (zc > 0) { dt3[0] = dt3[1] = dt3[2] = 0; dq[0] = 1; dq[1] = dq[2] = dq[3] = 0; }
        int idx[6];
        double z[6], Rd[6], R[36] = { 0 };
        const int m = po_legodo_create_measurement(0, r5, cpos, dt3, dq, tk.js_utime, prev, cok, status, idx, z, Rd);
        for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
        po_indexed_update(m, idx, z, R, &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
      }
      if (tk.pose) {
        const int idx[4] = { 9, 10, 11, 8 };
        double z[4] = { tk.pos[0], tk.pos[1], tk.pos[2], 0.0 }, R[16] = { 0 };
        R[0] = R[5] = R[10] = 0.05 * 0.05;
        R[15] = bot_sq(bot_to_radians(1.0));
        po_indexed_orient_update(4, idx, z, R, tk.quat, &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
      }
    }
  }
  double ev = 0, eq = 0, eP = 0, el = 0, sv = 0, sP = 0, sl = 1e-300;
  for (int b = 0; b < B; b++) {
    for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(head(i, b) - ox[(size_t) b].vec[i])); sv = fmax(sv, fabs(ox[(size_t) b].vec[i])); }
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(head.q(i, b) - ox[(size_t) b].quat[i]));
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(cov(r, c, b) - oP[(size_t) b].m[c * 21 + r])); sP = fmax(sP, fabs(oP[(size_t) b].m[c * 21 + r])); }
    el = fmax(el, fabs(ll[(size_t) b] - oll[(size_t) b]));
    sl = fmax(sl, fabs(oll[(size_t) b]));
  }
  long long want_msgs = 0;
  for (int s = 0; s < B; s++)
    for (const Tick &tk : ticks[(size_t) s]) want_msgs += 3 + (tk.pose ? 1 : 0);
  if (kvh) printf("KVH batch messages with 0 / 1 / 2 / 3 new packets: %d / %d / %d / %d\n", n_kvh_new[0], n_kvh_new[1], n_kvh_new[2], n_kvh_new[3]);
  printf("n=%d %s%s: %d segments, %lld batched messages carrying %lld segment messages (IMU %lld, joint %lld, force/torque %lld, pose %lld), ragged columns %lld, "
         "order violations %lld, undecodable %lld, max skew %lld us, fused pairs %lld; status skip/certain/uncertain %d/%d/%d; "
         "rel err vs %d single-segment oracle runs: vec %.2e quat %.2e cov %.2e ll %.2e (status %d)\n",
         n, fuse ? "fused" : "unfused", stream ? (kvh ? ", streamed, KVH batch IMU" : ", streamed") : "", B, (long long) st.batches, (long long) st.segment_messages, (long long) st.per_channel[imu_channel],
         (long long) st.per_channel["JOINT_STATE"], (long long) st.per_channel["FORCE_TORQUE"], (long long) st.per_channel["POSE_SCAN"], (long long) st.ragged,
         (long long) st.order_violations, (long long) st.undecodable, (long long) st.max_skew_us, (long long) est.fused_pairs, n_status[0], n_status[1], n_status[2], B,
         ev / sv, eq, eP / sP, el / sl, est.last_status);
  const bool kvh_ok = !kvh || (n_kvh_new[0] > B * T / 40 && n_kvh_new[1] > B * T / 40 && n_kvh_new[3] > B * T / 40);
  const bool ok = kvh_ok && est.last_status == PB_OK && st.segment_messages == want_msgs && st.per_channel[imu_channel] == T && st.per_channel["JOINT_STATE"] == T &&
                  st.order_violations == 0 && st.undecodable == 0 && st.ragged > 0 && st.max_skew_us > 20 && st.max_skew_us < 200 &&
                  (!fuse || est.fused_pairs > T / 2) && n_status[0] > 100 && n_status[1] > 100 && n_status[2] > 100 && ev / sv < 1e-9 && eq < 1e-9 &&
                  eP / sP < 1e-9 && el / sl < 1e-9;
  printf(ok ? "PASS\n" : "FAIL\n");
  for (int s = 0; s < B; s++) remove(paths[(size_t) s].c_str());
  return ok ? 0 : 1;
}
