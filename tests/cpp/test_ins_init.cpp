// test_ins_init.cpp -- InsHandler::processMessageInit (gravity / gyro-bias initialisation, sensor_handlers.cpp:254-364) of
// the shim against the oracle's po_ins_init, filter by filter, plus the property the arithmetic exists for: the roll and
// pitch of a stationary IMU are recovered.  Host-only code path: runs WITHOUT a GPU.  Exit code 0 + "PASS".
#include <cstdio>
#include <map>
#include <vector>

#include "../../oracle/pronto_oracle.h"
#include "../../pronto_amd/csrc/mav_state_est_batch.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x494E4954ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main()
{
  const int n = 21, B = 40, N = 50;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.q_gyro_bias", 0.001);
  param.set("state_estimator.ins.q_accel_bias", 0.0001);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  param.set("state_estimator.ins.accel_bias_update_online", "true");
  param.set("state_estimator.ins.gyro_bias_update_online", "true");
  param.set("state_estimator.ins.num_to_init", std::to_string(N));
  param.set("state_estimator.ins.max_initial_gyro_bias", "0.02");
  param.set("state_estimator.ins.accel_bias_initial", "[0.01, -0.02, 0.03]");
  param.set("state_estimator.ins.gyro_bias_initial", "0, 0, 0");
  param.set("state_estimator.ins.accel_bias_recalc_at_start", "false");
  param.set("state_estimator.ins.gyro_bias_recalc_at_start", "true");
  BotTrans ins_to_body;  // 90 deg about z
  ins_to_body.rot_quat[0] = sqrt(0.5); ins_to_body.rot_quat[3] = sqrt(0.5);
  InsHandler ins(&param, &ins_to_body);

  RBIS def_state(n, B), init_state(n, B);
  RBIM def_cov(n, B), init_cov(n, B);
  std::vector<double> roll(B), pitch(B), bias(3 * B);
  for (int b = 0; b < B; b++) {
    roll[b] = 0.4 * (urand() - 0.5);
    pitch[b] = 0.4 * (urand() - 0.5);
    for (int i = 0; i < 3; i++) bias[i * B + b] = (b % 7 == 3 && i == 1) ? 0.05 : 0.004 * nrand();  // filter 3, 10, ..: too large
    for (int i = 0; i < n; i++) {
      def_cov(i, i, b) = 0.01 * (1 + i);
      init_cov(i, i, b) = -1.0;  // "not initialised yet" (rbis_initializer.cpp:124-128)
    }
  }
  std::map<std::string, bool> sensors_initialized = { { "ins", false }, { "vicon", false } };
  std::vector<double> gs(3 * B, 0.0), ws(3 * B, 0.0), gy(3 * B), ac(3 * B);
  bool done = false;
  int calls = 0;
  for (int k = 0; k < N + 3 && !done; k++) {
    if (k == 3) sensors_initialized["vicon"] = true;  // the INS waits for everybody else (:266-267)
    for (int b = 0; b < B; b++) {
      double q[4], up_w[3] = { 0, 0, g }, up_b[3], qc[4];
      po_euler_to_quat(roll[b], pitch[b], 0.0, q);
      qc[0] = q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = -q[3];
      po_quat_rotate(qc, up_w, up_b);                      // specific force of a stationary body, body frame
      double a_b[3], w_b[3], a_s[3], w_s[3];
      for (int i = 0; i < 3; i++) { a_b[i] = up_b[i] + 0.05 * nrand(); w_b[i] = bias[i * B + b] + 0.002 * nrand(); }
      const double sc[4] = { ins_to_body.rot_quat[0], -ins_to_body.rot_quat[1], -ins_to_body.rot_quat[2], -ins_to_body.rot_quat[3] };
      po_quat_rotate(sc, a_b, a_s);                        // what the sensor reports (sensor frame)
      po_quat_rotate(sc, w_b, w_s);
      for (int i = 0; i < 3; i++) { ac[i * B + b] = a_s[i]; gy[i * B + b] = w_s[i]; }
      if (k >= 3) {                                        // the handler only accumulates once the others are done
        double a_r[3], w_r[3];
        po_quat_rotate(ins_to_body.rot_quat, a_s, a_r);
        po_quat_rotate(ins_to_body.rot_quat, w_s, w_r);
        for (int i = 0; i < 3; i++) { gs[i * B + b] += -a_r[i]; ws[i * B + b] += w_r[i]; }
      }
    }
    msgs::ins_t m{ (int64_t) (k + 1) * 1000, BatchArray(gy.data(), PB_HOST), BatchArray(ac.data(), PB_HOST) };
    done = ins.processMessageInit(&m, sensors_initialized, def_state, def_cov, init_state, init_cov);
    calls++;
  }
  bool ok = done && calls == N + 3 && ins.init_counter == N;
  double eq = 0, eb = 0, erp = 0;
  int zeroed = 0;
  for (int b = 0; b < B; b++) {
    const double g3[3] = { gs[b], gs[B + b], gs[2 * B + b] }, w3[3] = { ws[b], ws[B + b], ws[2 * B + b] };
    const double qi[4] = { 1, 0, 0, 0 };
    double qo[4], gb[3], rpy[3];
    po_ins_init(g3, w3, N, 0.02, qi, qo, gb);
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(init_state.q(i, b) - qo[i]));
    for (int i = 0; i < 3; i++) eb = fmax(eb, fabs(init_state(RBIS::gyro_bias_ind + i, b) - gb[i]));  // recalc_at_start = true
    if (gb[0] == 0 && gb[1] == 0 && gb[2] == 0) zeroed++;
    po_quat_to_euler(qo, rpy);
    erp = fmax(erp, fmax(fabs(rpy[0] - roll[b]), fabs(rpy[1] - pitch[b])));
    // covariance blocks: chi roll/pitch and the gyro bias from the defaults, everything else untouched
    ok = ok && init_cov(6, 6, b) == def_cov(6, 6, b) && init_cov(7, 7, b) == def_cov(7, 7, b) && init_cov(8, 8, b) == -1.0 &&
         init_cov(15, 15, b) == def_cov(15, 15, b) && init_cov(17, 17, b) == def_cov(17, 17, b) && init_cov(3, 3, b) == -1.0;
    ok = ok && init_state(RBIS::accel_bias_ind, b) == 0.01 && init_state(RBIS::accel_bias_ind + 1, b) == -0.02 &&
         init_state(RBIS::accel_bias_ind + 2, b) == 0.03;
  }
  printf("init after %d messages (%d accumulated): |quat - oracle| %.2e, |gyro bias - oracle| %.2e, roll/pitch error %.2e rad, "
         "%d filters with an out-of-range bias estimate zeroed\n", calls, ins.init_counter, eq, eb, erp, zeroed);
  ok = ok && eq < 1e-14 && eb < 1e-15 && erp < 5e-3 && zeroed == (B + 3) / 7;
  // ---- IndexedMeasurementHandler::processMessageInit (sensor_handlers.cpp:584-610): position + yaw from one message ----
  {
    IndexedMeasurementHandler h(RBISUpdateInterface::laser_gpf);
    RBIS st(15, 3);
    RBIM cv(15, 3);
    const double z[4] = { 1.5, -2.0, 0.25, 0.3 }, R[16] = { 4, 0.1, 0, 0, 0.1, 5, 0, 0, 0, 0, 6, 0.2, 0, 0, 0.2, 7 };
    msgs::indexed_measurement_t m;
    m.utime = 5;
    m.z_indices = { 9, 10, 11, 8 };
    m.z_effective = BatchArray(z, PB_HOST_BROADCAST);
    m.R_effective = R;
    const bool r = h.processMessageInit(&m, sensors_initialized, st, cv, st, cv);
    bool good = r;
    for (int b = 0; b < 3; b++) {
      good = good && st(9, b) == 1.5 && st(10, b) == -2.0 && st(11, b) == 0.25 && st(8, b) == 0.0 &&
             fabs(st.q(0, b) - cos(0.15)) < 1e-16 && fabs(st.q(3, b) - sin(0.15)) < 1e-16 && st.q(1, b) == 0 && st.q(2, b) == 0 &&
             cv(9, 9, b) == 4 && cv(9, 10, b) == 0.1 && cv(10, 10, b) == 5 && cv(11, 8, b) == 0.2 && cv(8, 8, b) == 7 && cv(3, 3, b) == 0;
    }
    if (!good) printf("IndexedMeasurementHandler::processMessageInit: unexpected initial state\n");
    ok = ok && good;
  }
  // ---- the GPS / magnetometer yaw branch (sensor_handlers.cpp:338-351): a filter that also initialises with "gps" takes
  //      its yaw from the mean magnetometer vector (ENU: horizontal field along +y) ----
  {
    InsHandler ins2(&param, &ins_to_body);
    RBIS st(n, B), dst(n, B);
    RBIM cv(n, B), dcv(n, B);
    std::vector<double> yaw(B), mg(3 * B), ms_sum(3 * B, 0.0), gs2(3 * B, 0.0), ws2(3 * B, 0.0);
    for (int b = 0; b < B; b++) {
      yaw[b] = 2.5 * (urand() - 0.5);
      for (int i = 0; i < n; i++) { dcv(i, i, b) = 0.01 * (1 + i); cv(i, i, b) = -1.0; }
    }
    std::map<std::string, bool> si = { { "ins", false }, { "gps", true } };
    bool fin = false;
    for (int k = 0; k < N && !fin; k++) {
      for (int b = 0; b < B; b++) {
        double q[4], qc[4], up_w[3] = { 0, 0, g }, m_w[3] = { 0.0, 0.21, -0.43 }, up_b[3], m_b[3];
        po_euler_to_quat(0.1 * roll[b], 0.1 * pitch[b], yaw[b], q);
        qc[0] = q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = -q[3];
        po_quat_rotate(qc, up_w, up_b);
        po_quat_rotate(qc, m_w, m_b);
        const double sc[4] = { ins_to_body.rot_quat[0], -ins_to_body.rot_quat[1], -ins_to_body.rot_quat[2], -ins_to_body.rot_quat[3] };
        double a_b[3], w_b[3], mb[3], a_s[3], w_s[3], m_s[3], a_r[3], w_r[3], m_r[3];
        for (int i = 0; i < 3; i++) { a_b[i] = up_b[i] + 0.02 * nrand(); w_b[i] = 0.001 * nrand(); mb[i] = m_b[i] + 0.003 * nrand(); }
        po_quat_rotate(sc, a_b, a_s); po_quat_rotate(sc, w_b, w_s); po_quat_rotate(sc, mb, m_s);
        for (int i = 0; i < 3; i++) { ac[i * B + b] = a_s[i]; gy[i * B + b] = w_s[i]; mg[i * B + b] = m_s[i]; }
        po_quat_rotate(ins_to_body.rot_quat, a_s, a_r); po_quat_rotate(ins_to_body.rot_quat, w_s, w_r); po_quat_rotate(ins_to_body.rot_quat, m_s, m_r);
        for (int i = 0; i < 3; i++) { gs2[i * B + b] += -a_r[i]; ws2[i * B + b] += w_r[i]; ms_sum[i * B + b] += m_r[i]; }
      }
      msgs::ins_t m{ (int64_t) (k + 1) * 1000, BatchArray(gy.data(), PB_HOST), BatchArray(ac.data(), PB_HOST), BatchArray(mg.data(), PB_HOST) };
      fin = ins2.processMessageInit(&m, si, dst, dcv, st, cv);
    }
    double eq2 = 0, eyaw = 0;
    bool good = fin;
    for (int b = 0; b < B; b++) {
      const double g3[3] = { gs2[b], gs2[B + b], gs2[2 * B + b] }, w3[3] = { ws2[b], ws2[B + b], ws2[2 * B + b] };
      const double m3[3] = { ms_sum[b], ms_sum[B + b], ms_sum[2 * B + b] }, qi[4] = { 1, 0, 0, 0 };
      double q1[4], q2[4], gb[3], rpy[3];
      po_ins_init(g3, w3, N, 0.02, qi, q1, gb);
      po_ins_init_yaw(m3, N, q1, q2);
      for (int i = 0; i < 4; i++) eq2 = fmax(eq2, fabs(st.q(i, b) - q2[i]));
      po_quat_to_euler(q2, rpy);
      eyaw = fmax(eyaw, fabs(remainder(rpy[2] - yaw[b], 2 * M_PI)));
      good = good && cv(8, 8, b) == dcv(8, 8, b) && cv(6, 6, b) == dcv(6, 6, b) && cv(9, 9, b) == -1.0;
    }
    printf("magnetometer yaw initialisation: |quat - oracle| %.2e, yaw error %.2e rad\n", eq2, eyaw);
    good = good && eq2 < 1e-14 && eyaw < 0.1;  // (the property is approximate: tilt couples the field's dip into the horizontal part)
    if (!good) printf("InsHandler::processMessageInit (gps / magnetometer yaw branch): unexpected result\n");
    ok = ok && good;
  }
  // ---- ViconHandler (sensor_handlers.cpp:406-574): frame composition, the near-zero drop, modes, initialisation ----
  {
    BotParam vp;
    vp.applyOverrides("state_estimator.vicon.mode=position_orient|state_estimator.vicon.apply_frame=true|"
                      "state_estimator.vicon.r_xyz=0.01|state_estimator.vicon.r_chi=2.0");
    BotTrans body_to_vicon;
    body_to_vicon.rot_quat[0] = sqrt(0.5); body_to_vicon.rot_quat[3] = sqrt(0.5);
    body_to_vicon.trans_vec[0] = 0.1;
    ViconHandler vh(&vp, &body_to_vicon);
    const double t[3] = { 1.0, 2.0, 3.0 }, q[4] = { 1, 0, 0, 0 };
    msgs::rigid_transform_t m{ 7, BatchArray(t, PB_HOST_BROADCAST), BatchArray(q, PB_HOST_BROADCAST) };
    RBIS st(15, 4);
    RBIM cv(15, 4);
    bool good = vh.processMessageInit(&m, sensors_initialized, st, cv, st, cv);
    for (int b = 0; b < 4; b++)
      good = good && fabs(st(9, b) - 1.1) < 1e-15 && st(10, b) == 2.0 && st(11, b) == 3.0 && fabs(st.q(0, b) - sqrt(0.5)) < 1e-16 &&
             fabs(st.q(3, b) - sqrt(0.5)) < 1e-16 && cv(9, 9, b) == 1e-4 && fabs(cv(6, 6, b) - bot_sq(bot_to_radians(2.0))) < 1e-18;
    if (!good) printf("ViconHandler::processMessageInit: unexpected initial state\n");
    ok = ok && good;
    // GpsHandler::processMessageInit: filters without lock keep their state
    BotParam gp;
    gp.applyOverrides("state_estimator.gps.r_xy=2.0|state_estimator.gps.r_z=3.0");
    GpsHandler gh(&gp);
    std::vector<double> xyz = { 10, 11, 12, 13, 20, 21, 22, 23, 30, 31, 32, 33 };
    const uint8_t lock[4] = { 1, 0, 1, 1 };
    msgs::gps_data_t gm{ 11, lock, BatchArray(xyz.data(), PB_HOST) };
    RBIS gs2(15, 4);
    RBIM gc(15, 4);
    good = gh.processMessageInit(&gm, sensors_initialized, gs2, gc, gs2, gc) && gs2(9, 0) == 10 && gs2(10, 2) == 22 && gs2(11, 3) == 33 &&
           gs2(9, 1) == 0 && gc(9, 9, 0) == 4.0 && gc(11, 11, 0) == 9.0 && gc(9, 9, 1) == 0.0;
    if (!good) printf("GpsHandler::processMessageInit: unexpected initial state\n");
    ok = ok && good;
  }
  // ---- PoseMeasHandler (pose_meas.cpp:7-131): initialisation from a pose_t; the "N corrections, then silent" counter and the
  // origin drop need no device (they return before an update object is made) ----
  {
    BotParam pp;
    pp.applyOverrides("state_estimator.pose_meas.mode=nonsense|state_estimator.pose_meas.no_corrections=2|"
                      "state_estimator.pose_meas.r_xyz=0.03|state_estimator.pose_meas.r_chi=4.0");
    PoseMeasHandler ph(&pp);
    const double pos[3] = { 0.5, -1.5, 0.9 }, vel[3] = { 0, 0, 0 }, q[4] = { 0.8, 0, 0.6, 0 }, origin[3] = { 1e-6, -1e-6, 0 };
    msgs::pose_t m{ 9, BatchArray(pos, PB_HOST_BROADCAST), BatchArray(vel, PB_HOST_BROADCAST), BatchArray(q, PB_HOST_BROADCAST) };
    RBIS st(15, 3);
    RBIM cv(15, 3);
    bool good = ph.mode == PoseMeasHandler::MODE_POSITION /* unrecognised -> position (:19-22) */ && ph.z_indices == std::vector<int>({ 9, 10, 11 }) &&
                ph.processMessageInit(&m, sensors_initialized, st, cv, st, cv) && st.utime == 9;
    for (int b = 0; b < 3; b++)
      good = good && st(9, b) == 0.5 && st(10, b) == -1.5 && st(11, b) == 0.9 && st.q(0, b) == 0.8 && st.q(2, b) == 0.6 &&
             fabs(cv(9, 9, b) - 9e-4) < 1e-18 && fabs(cv(6, 6, b) - bot_sq(bot_to_radians(4.0))) < 1e-18 && cv(9, 10, b) == 0.0;
    // a pose at the origin is dropped (:74-75) but still counts as a message (:56-64: the counter is decremented first)
    msgs::pose_t m0{ 10, BatchArray(origin, PB_HOST_BROADCAST), BatchArray(vel, PB_HOST_BROADCAST), BatchArray(q, PB_HOST_BROADCAST) };
    ph.no_corrections = 4;
    good = good && ph.processMessage(&m0, nullptr) == nullptr && ph.no_corrections == 3;
    // no_corrections = 3 now: two more messages produce updates (3 -> 2, 2 -> 1), the next one (-> 0) and all later ones are silent
    for (int k = 0; k < 4; k++) {
      RBISUpdateInterface *u = ph.processMessage(&m, nullptr);
      good = good && ((k < 2) == (u != nullptr)) && (u == nullptr || (u->sensor_id == RBISUpdateInterface::pose_meas && u->utime == 9));
      delete u;
    }
    if (!good) printf("PoseMeasHandler: unexpected initial state\n");
    ok = ok && good;
  }
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
