// test_atlas_imu.cpp -- InsHandler::processMessageAtlas with atlas_filter = true (sensor_handlers.cpp:165-252): KVH batch
// messages that repeat packets are de-duplicated (imu_stream.cpp:62-98), every NEW packet goes through the 3-stage notch
// cascade on the device (pb_imu_notch), the newest filtered packet drives the process step.  Checked against the oracle's
// restatement of the same chain, filter by filter.
#include <cinttypes>
#include <cstdio>
#include <deque>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0xABCDEF12345ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const int B = 70, NMSG = 60;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", "0");  // in-order only (no posterior checkpoints)
  param.set("state_estimator.ins.channel", "ATLAS_IMU_BATCH");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.003);
  param.set("state_estimator.ins.atlas_filter", "true");
  param.set("state_estimator.ins.atlas_filter_freq", 87.0);
  set_ins_bias_keys(param, n);
  param.set("state_estimator.ins.downsample_factor", "1");
  param.set("state_estimator.ins.roll_forward_on_receive", "true");
  param.set("state_estimator.ins.utime_offset", "0");

  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0);
  std::vector<po_notch> onotch(9 * B);
  for (int b = 0; b < B; b++) {
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    for (int i = 3; i < 12; i++) { P0(i, i, b) = 0.01; oP[b].m[i * 21 + i] = 0.01; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
    po_notch_cascade_init(&onotch[9 * b], 87.0, 1000);
  }
  InsHandler ins_handler(&param);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessageAtlas, &ins_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);

  // 1 kHz packets; every message carries the 5 newest packets (so 2 of them repeat) and arrives every 3 packets,
  // except that now and then a message is a pure repeat (no new packet -> handler returns NULL).
  struct Pkt { int64_t utime, count; std::vector<double> drot, acc; };
  std::deque<Pkt> ring;
  int64_t count = 0, prev_upd_utime = 0, last_seen_utime = 0;
  int n_updates = 0;
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
  for (int m = 0; m < NMSG; m++) {
    const int fresh = (m % 7 == 6) ? 0 : 3;
    std::vector<const Pkt *> fresh_ptrs;
    for (int j = 0; j < fresh; j++) {
      Pkt p;
      p.count = ++count;
      p.utime = p.count * 1000;
      p.drot.resize(3 * B);
      p.acc.resize(3 * B);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) {
          p.drot[i * B + b] = (0.2 * sin(0.002 * p.count + b + i)) * 0.001;
          // gravity + an 87 Hz vibration the notch is there to remove + noise
          p.acc[i * B + b] = (i == 2 ? g : 0.0) + 1.5 * sin(2 * M_PI * 87.0 * p.count * 1e-3 + b) + 0.05 * nrand();
        }
      ring.push_back(std::move(p));
      if (ring.size() > 5) ring.pop_front();
    }
    msgs::kvh_raw_imu_batch_t msg;
    msg.utime = count * 1000 + 200;  // batch stamp a little after the newest packet
    for (int i = (int) ring.size() - 1; i >= 0; i--)
      msg.raw_imu.push_back({ ring[i].utime, ring[i].count, ring[i].drot.data(), ring[i].acc.data() });
    on_ins(&msg);
    // ---- oracle: same chain ----
    if (fresh == 0) continue;
    double accf[3] = { 0, 0, 0 };
    for (int b = 0; b < B; b++) {
      const Pkt *newest = nullptr;
      int64_t prev_packet_utime = last_seen_utime;
      for (size_t i = ring.size() - fresh; i < ring.size(); i++) {
        double a[3] = { ring[i].acc[b], ring[i].acc[B + b], ring[i].acc[2 * B + b] };
        po_notch_cascade(&onotch[9 * b], a);
        accf[0] = a[0]; accf[1] = a[1]; accf[2] = a[2];
        if (i + 1 < ring.size()) prev_packet_utime = ring[i].utime;
        newest = &ring[i];
      }
      const double raw_dt = (newest->utime - prev_packet_utime) * 1E-6;  // utime_delta of the newest packet
      double gyro[3] = { newest->drot[b] / raw_dt, newest->drot[B + b] / raw_dt, newest->drot[2 * B + b] / raw_dt };
      const double dt = (prev_upd_utime == 0) ? 0.003 : (msg.utime - prev_upd_utime) * 1E-6;
      po_imu_process_step(gyro, accf, dt, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    }
    last_seen_utime = ring.back().utime;
    prev_upd_utime = msg.utime;
    n_updates++;
  }
  RBIS head;
  RBIM cov;
  est.getHeadState(head, cov);
  double ev = 0, eq = 0, eP = 0, sv = 0, sP = 0, vib = 0;
  for (int b = 0; b < B; b++) {
    for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(head(i, b) - ox[b].vec[i])); sv = fmax(sv, fabs(ox[b].vec[i])); }
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(head.q(i, b) - ox[b].quat[i]));
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(cov(r, c, b) - oP[b].m[c * 21 + r])); sP = fmax(sP, fabs(oP[b].m[c * 21 + r])); }
    vib = fmax(vib, fabs(head(14, b) - g));  // filtered vertical acceleration: the 1.5 m/s^2 vibration must be gone
  }
  printf("atlas IMU front end: %d messages, %d process steps, head utime %" PRId64 "; rel err vec %.2e quat %.2e cov %.2e; residual "
         "vibration in a_z %.3f m/s^2\n", NMSG, n_updates, head.utime, ev / sv, eq, eP / sP, vib);
  const bool ok = est.last_status == PB_OK && ev / sv < 1e-9 && eq < 1e-9 && eP / sP < 1e-9 && vib < 0.4 && n_updates == NMSG - NMSG / 7;
  // ---- phase 2: ONE robot's KVH batches for every filter, from a recorded log whose message type is known only through
  //      its .lcm text: LogPlayer::subscribeKvhBatch -> PB_HOST_BROADCAST packets -> de-dup + notch on the host ->
  //      broadcast process step.  Every filter must equal the oracle's chain on that one stream. ----
  bool ok2 = false;
  {
    pronto_wire::Schema sc;
    std::string err;
    const bool parsed = sc.parse("package test_core;\n"
                                 "struct kvh_raw_imu_t { int64_t utime; int64_t packet_count; double delta_rotation[3]; double linear_acceleration[3]; }\n"
                                 "struct kvh_raw_imu_batch_t { int64_t utime; int32_t num_packets; kvh_raw_imu_t raw_imu[num_packets]; }\n", &err);
    const std::string path = std::string(argc > 1 ? argv[1] : "/tmp") + "/kvh.lcmlog";
    const int B2 = 70, NM2 = 60;
    struct P2 { int64_t utime, count; double drot[3], acc[3]; };
    std::deque<P2> ring2;
    std::vector<std::vector<P2>> fresh_of;  // the NEW packets of each message, for the oracle
    std::vector<int64_t> msg_utime;
    {
      pronto_wire::LogWriter log(path);
      int64_t cnt = 0;
      for (int m = 0; parsed && m < NM2; m++) {
        const int fresh = (m % 5 == 4) ? 0 : 3;
        std::vector<P2> fr;
        for (int j = 0; j < fresh; j++) {
          P2 p;
          p.count = ++cnt;
          p.utime = p.count * 1000;
          for (int i = 0; i < 3; i++) {
            p.drot[i] = 0.2 * sin(0.002 * p.count + i) * 0.001;
            p.acc[i] = (i == 2 ? g : 0.0) + 1.5 * sin(2 * M_PI * 87.0 * p.count * 1e-3) + 0.05 * nrand();
          }
          ring2.push_back(p);
          if (ring2.size() > 5) ring2.pop_front();
          fr.push_back(p);
        }
        pronto_wire::Writer w;
        w.u64(sc.fingerprint("test_core.kvh_raw_imu_batch_t"));
        w.i64(cnt * 1000 + 200);
        w.i32((int32_t) ring2.size());
        for (int i = (int) ring2.size() - 1; i >= 0; i--) {  // newest first
          w.i64(ring2[i].utime); w.i64(ring2[i].count);
          w.f64s(ring2[i].drot, 3); w.f64s(ring2[i].acc, 3);
        }
        log.write(cnt * 1000 + 200, "ATLAS_IMU_BATCH", w.buf);
        fresh_of.push_back(fr);
        msg_utime.push_back(cnt * 1000 + 200);
      }
    }
    RBIS y0(n, B2);
    RBIM Q0(n, B2);
    for (int b = 0; b < B2; b++)
      for (int i = 3; i < 12; i++) Q0(i, i, b) = 0.01;
    InsHandler h2(&param);
    FrontEnd fe2(&param);
    auto on_ins2 = fe2.addSensor("ins", &InsHandler::processMessageAtlas, &h2);
    MavStateEstimator est2(new RBISResetUpdate(y0, Q0, RBISUpdateInterface::reset, 0), &param, 0);
    fe2.setStateEstimator(&est2);
    LogPlayer player(B2);
    int seen = 0;
    player.subscribeKvhBatch("ATLAS_IMU_BATCH", &sc, "test_core.kvh_raw_imu_batch_t", [&](const msgs::kvh_raw_imu_batch_t *m) {
      if (m->mem != PB_HOST_BROADCAST) exit(1);
      on_ins2(m);
      seen++;
    });
    const int64_t dispatched = parsed ? player.run(path) : -1;
    // oracle: one filter on the one stream
    po_rbis o1;
    po_rbim oP1;
    double oll1 = 0.0;
    po_rbis_zero(&o1);
    memset(&oP1, 0, sizeof oP1);
    for (int i = 3; i < 12; i++) oP1.m[i * 21 + i] = 0.01;
    std::vector<po_notch> nt(9);
    po_notch_cascade_init(nt.data(), 87.0, 1000);
    int64_t last_pkt = 0, prev_upd = 0;
    int steps = 0;
    for (size_t m = 0; m < fresh_of.size(); m++) {
      if (fresh_of[m].empty()) continue;
      double af[3] = { 0, 0, 0 };
      int64_t prev_pkt = last_pkt;
      for (size_t i = 0; i < fresh_of[m].size(); i++) {
        double a3[3] = { fresh_of[m][i].acc[0], fresh_of[m][i].acc[1], fresh_of[m][i].acc[2] };
        po_notch_cascade(nt.data(), a3);
        af[0] = a3[0]; af[1] = a3[1]; af[2] = a3[2];
        if (i + 1 < fresh_of[m].size()) prev_pkt = fresh_of[m][i].utime;
      }
      const P2 &nw = fresh_of[m].back();
      const double raw_dt = (nw.utime - prev_pkt) * 1E-6;
      const double gy[3] = { nw.drot[0] / raw_dt, nw.drot[1] / raw_dt, nw.drot[2] / raw_dt };
      const double dt2 = (prev_upd == 0) ? 0.003 : (msg_utime[m] - prev_upd) * 1E-6;
      po_imu_process_step(gy, af, dt2, q4[0], q4[1], q4[2], q4[3], &o1, &oP1, oll1, &o1, &oP1, &oll1);
      last_pkt = nw.utime;
      prev_upd = msg_utime[m];
      steps++;
    }
    RBIS h;
    RBIM c;
    est2.getHeadState(h, c);
    double e2 = 0, s2 = 0, eq2 = 0, eP2 = 0, sP2 = 0;
    for (int b = 0; b < B2; b++) {
      for (int i = 0; i < n; i++) { e2 = fmax(e2, fabs(h(i, b) - o1.vec[i])); s2 = fmax(s2, fabs(o1.vec[i])); }
      for (int i = 0; i < 4; i++) eq2 = fmax(eq2, fabs(h.q(i, b) - o1.quat[i]));
      for (int cc = 0; cc < n; cc++)
        for (int r = 0; r < n; r++) { eP2 = fmax(eP2, fabs(c(r, cc, b) - oP1.m[cc * 21 + r])); sP2 = fmax(sP2, fabs(oP1.m[cc * 21 + r])); }
    }
    printf("broadcast KVH log: %lld events, %d messages, %d process steps; rel err vec %.2e quat %.2e cov %.2e (status %d, undecodable %lld)\n",
           (long long) dispatched, seen, steps, e2 / s2, eq2, eP2 / sP2, est2.last_status, (long long) player.undecodable());
    ok2 = parsed && dispatched == NM2 && seen == NM2 && steps == NM2 - NM2 / 5 && est2.last_status == PB_OK && player.undecodable() == 0 &&
          e2 / s2 < 1e-9 && eq2 < 1e-9 && eP2 / sP2 < 1e-9;
  }
  printf((ok && ok2) ? "PASS\n" : "FAIL\n");
  return (ok && ok2) ? 0 : 1;
}
