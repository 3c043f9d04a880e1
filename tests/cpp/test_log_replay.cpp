// test_log_replay.cpp -- a recorded segment (LCM event log) replayed into the batch through the C++ shim's LogPlayer,
// the way se-fusion replays a log for a parameter sweep (param_sweep.py:39-52): ONE robot's messages feed every filter,
// the filters differ in their initial state.  The log carries pronto::indexed_measurement_t (channel GPF_MEASUREMENT,
// m = 3 with a full R) and pronto::update_t (KINECT_REL_ODOMETRY, VO position_orient); the IMU travels on a raw test
// channel (7 big-endian doubles) because bot_core::ins_t's schema is not in the reference tree.  Every event is applied to
// the oracle filter by filter; afterwards the head is published as pronto::filter_state_t into a second log, read back
// and compared with getHeadState.  Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x4C434D4C4F47ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const std::string in_log = dir + "/segment.lcmlog", out_log = dir + "/published.lcmlog";
  const int B = 96, T = 100;
  double g;
  po_get_constants(&g, nullptr);

  // The IMU messages use a type defined by .lcm TEXT handed over at run time (lcm_schema.hpp) -- the situation of libbot's
  // bot_core types, whose definitions are not in the reference tree.  (This definition is the test's own.)
  pronto_wire::Schema ins_schema;
  {
    std::string err;
    const bool parsed = ins_schema.parse(
        "package test_core;\n"
        "struct ins_t {\n"
        "  int64_t utime;  int64_t device_time;\n"
        "  double gyro[3];  double mag[3];  double accel[3];  double quat[4];\n"
        "  double pressure;  double rel_alt;  // rad/s, gauss, m/s^2, -, mbar, m\n"
        "}\n", &err);
    if (!parsed || ins_schema.fingerprint("ins_t") == 0) { printf("schema: %s\nFAIL\n", err.c_str()); return 1; }
  }

  // ---- record the segment ----
  {
    pronto_wire::LogWriter log(in_log);
    if (!log.good()) { printf("cannot write %s\nFAIL\n", in_log.c_str()); return 1; }
    std::vector<uint8_t> buf;
    int64_t key_utime = 0;
    for (int k = 0; k < T; k++) {
      const int64_t utime = (int64_t) (k + 1) * 1000;
      // an INS message of a type the estimator only knows from its .lcm text (ins_schema above): fingerprint + fields
      pronto_wire::Writer w;
      w.u64(ins_schema.fingerprint("test_core.ins_t"));
      w.i64(utime);          // utime
      w.i64(utime + 17);     // device_time
      for (int i = 0; i < 3; i++) w.f64(0.3 * sin(0.02 * k + i) + 0.01 * nrand());      // gyro
      for (int i = 0; i < 3; i++) w.f64(0.1 * i);                                        // mag
      for (int i = 0; i < 3; i++) w.f64(0.3 * nrand() + (i == 2 ? g : 0.0));             // accel
      for (int i = 0; i < 4; i++) w.f64(i == 0);                                         // quat
      w.f64(1013.0);         // pressure
      w.f64(0.0);            // rel_alt
      log.write(utime, "IMU_TICK", w.buf);
      if (k % 7 == 3) log.write(utime, "SOMETHING_ELSE", std::vector<uint8_t>(11, 0x5A));
      if (k % 4 == 3) {
        pronto_wire::indexed_measurement_t im;
        im.utime = utime;
        im.state_utime = utime - 1000;
        im.measured_dim = 3;
        im.z_indices = { 3, 4, 5 };
        for (int i = 0; i < 3; i++) im.z_effective.push_back(0.2 * nrand());
        im.measured_cov_dim = 9;
        const double R[9] = { 0.012, 0.002, -0.001, 0.002, 0.011, 0.003, -0.001, 0.003, 0.014 };
        im.R_effective.assign(R, R + 9);
        im.encode(buf);
        log.write(utime, "GPF_MEASUREMENT", buf);
      }
      if (k % 25 == 24) {
        pronto_wire::update_t up;
        up.timestamp = utime;
        up.prev_timestamp = key_utime;
        double dq[4];
        po_euler_to_quat(0.01 * nrand(), 0.01 * nrand(), 0.02 * nrand(), dq);
        for (int i = 0; i < 3; i++) up.translation[i] = 0.02 * nrand();
        for (int i = 0; i < 4; i++) up.rotation[i] = dq[i];
        up.estimate_status = (k == 49) ? pronto_wire::update_t::ESTIMATE_DEGENERATE : pronto_wire::update_t::ESTIMATE_VALID;
        up.encode(buf);
        log.write(utime, "KINECT_REL_ODOMETRY", buf);
        key_utime = utime;
      }
    }
  }

  // ---- the estimator, configured with the reference's keys ----
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", "0");  // in-order only (no posterior checkpoints)
  param.set("state_estimator.ins.channel", "IMU_TICK");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.fovis.mode=position_orient|state_estimator.fovis.r_pxyz=0.02|state_estimator.fovis.r_chi=0.01");
  param.set("state_estimator.filter_state_channel", "STATE_ESTIMATOR_STATE");
  param.set("state_estimator.publish_filter_state", "true");
  for (const char *s : { "ins", "gpf", "fovis" }) {
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
    for (int i = 0; i < 3; i++) { x0(3 + i, b) = 0.2 * nrand(); ox[b].vec[3 + i] = x0(3 + i, b); }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i] * (1.0 + 0.01 * b); oP[b].m[i * 21 + i] = P0(i, i, b); }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  IndexedMeasurementHandler gpf_handler(RBISUpdateInterface::laser_gpf);
  FovisHandler fovis_handler(&param, 0);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_gpf = front_end.addSensor("gpf", &IndexedMeasurementHandler::processMessage, &gpf_handler);
  auto on_fovis = front_end.addSensor("fovis", &FovisHandler::processMessage, &fovis_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  fovis_handler.markKeyframe(&est);
  std::vector<po_rbis> key = ox;
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };

  // ---- replay: every subscription applies the event to the batch AND to the oracle ----
  LogPlayer player(B);
  int n_imu = 0, n_gpf = 0, n_vo = 0, n_vo_invalid = 0, n_other = 0;
  // decoded by field name through the run-time schema; handed over as PB_HOST_BROADCAST blocks (one IMU for every filter)
  player.subscribeIns("IMU_TICK", &ins_schema, "test_core.ins_t", [&](const msgs::ins_t *m) {
    if (m->gyro.mem != PB_HOST_BROADCAST) { printf("LogPlayer did not broadcast the INS message\n"); exit(1); }
    const double v[6] = { m->gyro.p[0], m->gyro.p[1], m->gyro.p[2], m->accel.p[0], m->accel.p[1], m->accel.p[2] };
    on_ins(m);
    for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    n_imu++;
  });
  player.subscribeRaw("SOMETHING_ELSE", [&](const pronto_wire::LogEvent &) { n_other++; });
  player.subscribeIndexedMeasurement("GPF_MEASUREMENT", [&](const msgs::indexed_measurement_t *m) {
    on_gpf(m);
    const int mm = (int) m->z_indices.size();
    if (m->z_effective.mem != PB_HOST_BROADCAST) { printf("LogPlayer did not broadcast\n"); exit(1); }
    for (int b = 0; b < B; b++)
      po_indexed_update(mm, m->z_indices.data(), m->z_effective.p, m->R_effective, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    n_gpf++;
  });
  player.subscribeUpdate("KINECT_REL_ODOMETRY", [&](const msgs::update_t *m) {
    on_fovis(m);
    n_vo++;
    if (m->estimate_valid && !m->estimate_valid[0]) {  // rbis_fovis_update.cpp:160-164: anything but ESTIMATE_VALID is dropped
      n_vo_invalid++;
    } else {
      for (int b = 0; b < B; b++) {
        const double *t3 = m->translation.p, *q = m->rotation.p;  // broadcast blocks: [3], [4]
        double z[6] = { 0 }, qm[4], R[36] = { 0 };
        po_fovis_compose(key[b].vec + 9, key[b].quat, t3, q, z, qm);
        const int idx[6] = { 9, 10, 11, 6, 7, 8 };
        for (int i = 0; i < 6; i++) R[i * 6 + i] = (i < 3) ? 0.02 * 0.02 : 0.01 * 0.01;
        po_indexed_orient_update(6, idx, z, R, qm, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
    }
    fovis_handler.markKeyframe(&est);
    key = ox;
  });
  const int64_t dispatched = player.run(in_log);

  // ---- compare the head with the oracle ----
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
  printf("replayed %" PRId64 " events (imu %d, gpf %d, vo %d of which %d not valid, undecodable %" PRId64 "): rel err vec %.2e "
         "quat %.2e cov %.2e ll %.2e (status %d)\n", dispatched, n_imu, n_gpf, n_vo, n_vo_invalid, player.undecodable(),
         ev / sv, eq, eP / sP, el / sl, est.last_status);
  bool ok = est.last_status == PB_OK && dispatched == n_imu + n_gpf + n_vo + n_other && n_other > 0 && n_imu == T && n_gpf == T / 4 && n_vo == T / 25 &&
            n_vo_invalid == 1 && player.undecodable() == 0 && head.utime == (int64_t) T * 1000 && ev / sv < 1e-9 && eq < 1e-9 &&
            eP / sP < 1e-9 && el / sl < 1e-9;

  // ---- publish the head of three filters, read the log back: bit-exact against getHeadState ----
  {
    pronto_wire::LogWriter out(out_log);
    FilterStatePublisher pub(&param, &out, { 0, 17, B - 1 });
    pub.publishHead(&est);
  }
  {
    pronto_wire::LogReader rd(out_log);
    pronto_wire::LogEvent e;
    const int which[3] = { 0, 17, B - 1 };
    const char *chan[3] = { "STATE_ESTIMATOR_STATE", "STATE_ESTIMATOR_STATE_1", "STATE_ESTIMATOR_STATE_2" };
    int seen = 0;
    while (rd.next(e)) {
      pronto_wire::filter_state_t fs;
      const int b = which[seen < 3 ? seen : 0];
      bool good = seen < 3 && e.channel == chan[seen] && e.timestamp == head.utime && e.data.size() == 3752 &&
                  fs.decode(e.data.data(), e.data.size()) == 3752 && fs.utime == head.utime && fs.num_states == 21 &&
                  fs.num_cov_elements == 441;
      for (int i = 0; good && i < 4; i++) good = fs.quat[i] == head.q(i, b);
      for (int i = 0; good && i < 21; i++) good = fs.state[(size_t) i] == (i < n ? head(i, b) : 0.0);
      for (int c = 0; good && c < 21; c++)
        for (int r = 0; good && r < 21; r++) good = fs.cov[(size_t) c * 21 + r] == ((r < n && c < n) ? cov(r, c, b) : 0.0);
      if (!good) { printf("published filter_state_t %d differs from the head\n", seen); ok = false; }
      seen++;
    }
    if (seen != 3) { printf("expected 3 published states, read %d\n", seen); ok = false; }
  }
  // ---- resume: a second, 3-filter estimator initialised from the published messages (InitMessageHandler), one more
  //      IMU + GPF step on both: the resumed filters must stay bit-identical to filters 0, 17, B-1 of the original ----
  {
    std::vector<pronto_wire::filter_state_t> msgs;
    pronto_wire::LogReader rd(out_log);
    pronto_wire::LogEvent e;
    while (rd.next(e)) {
      pronto_wire::filter_state_t fs;
      if (fs.decode(e.data.data(), e.data.size()) > 0) msgs.push_back(fs);
    }
    RBIS z0(n, 3);
    RBIM zP(n, 3);
    MavStateEstimator est2(new RBISResetUpdate(z0, zP, RBISUpdateInterface::reset, 0), &param, 0);
    InitMessageHandler init_handler;
    RBISUpdateInterface *reset = init_handler.processMessages(msgs, &est2);
    if (reset == nullptr) { printf("InitMessageHandler refused the published messages\n"); ok = false; }
    else {
      est2.addUpdate(reset, true);
      FrontEnd fe2(&param);
      fe2.setStateEstimator(&est2);
      InsHandler ins2(&param, &ins_to_body);
      IndexedMeasurementHandler gpf2(RBISUpdateInterface::laser_gpf);
      auto on_ins2 = fe2.addSensor("ins", &InsHandler::processMessage, &ins2);
      auto on_gpf2 = fe2.addSensor("gpf", &IndexedMeasurementHandler::processMessage, &gpf2);
      const double v[6] = { 0.11, -0.07, 0.23, 0.3, -0.2, g };
      const double zz[3] = { 0.05, -0.02, 0.01 }, RR[9] = { 0.012, 0.002, -0.001, 0.002, 0.011, 0.003, -0.001, 0.003, 0.014 };
      const int64_t t1 = head.utime + 1000;
      msgs::ins_t im{ t1, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
      msgs::indexed_measurement_t gm;
      gm.utime = t1;
      gm.z_indices = { 3, 4, 5 };
      gm.z_effective = BatchArray(zz, PB_HOST_BROADCAST);
      gm.R_effective = RR;
      on_ins(&im); on_gpf(&gm);
      on_ins2(&im); on_gpf2(&gm);
      RBIS h1, h2;
      RBIM c1, c2;
      est.getHeadState(h1, c1);
      est2.getHeadState(h2, c2);
      const int which[3] = { 0, 17, B - 1 };
      bool same = h1.utime == h2.utime && h2.utime == t1;
      for (int k = 0; same && k < 3; k++) {
        for (int i = 0; same && i < n; i++) same = h2(i, k) == h1(i, which[k]);
        for (int i = 0; same && i < 4; i++) same = h2.q(i, k) == h1.q(i, which[k]);
        for (int c = 0; same && c < n; c++)
          for (int r = 0; same && r < n; r++) same = c2(r, c, k) == c1(r, c, which[k]);
      }
      if (!same) { printf("resumed filters diverge from the originals\n"); ok = false; }
    }
  }
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
