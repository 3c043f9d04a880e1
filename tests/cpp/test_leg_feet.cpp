// test_leg_feet.cpp -- LegOdoHandler::processMessageFeet: the reference's whole leg-odometry handler
// (rbis_legodo_update.cpp:206-280 with leg_estimate::updateOdometry, leg_estimate.cpp:395-556) from body-to-foot transforms
// and foot forces, on the GPU for every filter, against the oracle's restatement (po_leg_update with the oracle filter's own
// head orientation, po_legodo_create_measurement, po_indexed_update).  argv[1]: legodo mode (lin_rate: the measurement never
// leaves the device; lin_rot_rate: formed on the host from the fetched increment).  Exit code 0 + "PASS".  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <string>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x46454554ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }
static double ramp(double x) { return x < 0 ? 0 : (x > 0.05 ? 1.0 : x / 0.05); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const std::string lomode = argc > 1 ? argv[1] : "lin_rate";
  // argv[2] = "fuse": state_estimator.fuse_ins_legodo -- the INS step is held back, the odometry is slaved to the orientation
  // AFTER it (pb_legodo_update_after_predict) and the pair runs as one fused kernel (lin_rate only; same oracle sequence)
  const bool fuse = argc > 2 && std::string(argv[2]) == "fuse";
  const int B = 64, T = 900;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", "0");
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.002);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.legodo.mode=" + lomode + "|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=5|"
                       "state_estimator.legodo.r_vang=3|state_estimator.legodo.r_vxyz_uncertain=10|state_estimator.legodo.r_vang_uncertain=9|"
                       "state_estimator.legodo.schmitt_low_threshold=475|state_estimator.legodo.schmitt_high_threshold=525|"
                       "state_estimator.legodo.schmitt_low_delay=7000|state_estimator.legodo.schmitt_high_delay=7000|"
                       "state_estimator.legodo.filter_contact_events=true|state_estimator.legodo.zero_initial_velocity=3");
  for (const char *s : { "ins", "legodo" }) {
    param.set(std::string("state_estimator.") + s + ".downsample_factor", "1");
    param.set(std::string("state_estimator.") + s + ".roll_forward_on_receive", "true");
    param.set(std::string("state_estimator.") + s + ".utime_offset", "0");
  }
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox(B);
  std::vector<po_rbim> oP(B);
  std::vector<double> oll(B, 0.0), period(B), phase(B), stride(B);
  std::vector<std::vector<char>> legs(B, std::vector<char>(po_leg_sizeof()));
  for (int b = 0; b < B; b++) {
    double q[4];
    po_euler_to_quat(0.05 * (urand() - 0.5), 0.05 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    po_rbis_zero(&ox[b]);
    memset(&oP[b], 0, sizeof(po_rbim));
    for (int i = 0; i < 4; i++) { x0.q(i, b) = q[i]; ox[b].quat[i] = q[i]; }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[b], &oP[b], urand);
    period[b] = 0.9 + 0.4 * urand(); phase[b] = urand(); stride[b] = 0.1 + 0.15 * urand();
    po_leg_init((po_leg *) legs[b].data(), 475, 525, 7000, 7000, 1);
  }
  BotTrans ins_to_body;
  InsHandler ins_handler(&param, &ins_to_body);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  int n_status[3] = { 0, 0, 0 }, lo_ticks = 0;
  std::vector<int> zero_count(B, 3);
  {
    LegOdoHandler legodo_handler(&param);
    auto on_feet = front_end.addSensor("legodo", &LegOdoHandler::processMessageFeet, &legodo_handler);
    const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };
    const double r5[5] = { 0.2, 5.0, 3.0, 10.0, 9.0 };  // (r_vxyz = 5 / 10 m/s: see tests/test_leg_odometry.py R_VXYZ)
    const int omode = lomode == "lin_rate" ? 0 : 1;
    std::vector<double> feet(14 * B), forces(2 * B);
    for (int k = 0; k < T; k++) {
      const int64_t utime = 1000000 + (int64_t) (k + 1) * 2000;
      const double t = (k + 1) * 0.002;
      const double v[6] = { 0.2 * sin(0.05 * k), 0.05, -0.1 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
      msgs::ins_t im{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
      on_ins(&im);
      for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.002, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      for (int b = 0; b < B; b++) {
        double ph = t / period[b] + phase[b];
        ph -= floor(ph);
        double wl = ramp(ph) * ramp(0.6 - ph), wr = ramp(ph - 0.5) * ramp(1.1 - ph) + (ph < 0.1 ? ramp(0.1 - ph) : 0.0);
        if (t < 0.4) wl = wr = 1.0;
        forces[b] = 900 * wl + 5 * nrand();
        forces[B + b] = 900 * wr + 5 * nrand();
        const double sw = sin(2 * M_PI * ph);
        double ql[4], qr[4];
        po_euler_to_quat(0.02 * sw, 0.05 * sw, 0, ql);
        po_euler_to_quat(-0.02 * sw, -0.05 * sw, 0, qr);
        const double lt[3] = { stride[b] * sw, 0.11, -0.86 + 0.02 * fmax(0, -sw) }, rt[3] = { -stride[b] * sw, -0.11, -0.86 + 0.02 * fmax(0, sw) };
        for (int i = 0; i < 3; i++) { feet[i * B + b] = lt[i]; feet[(7 + i) * B + b] = rt[i]; }
        for (int i = 0; i < 4; i++) { feet[(3 + i) * B + b] = ql[i]; feet[(10 + i) * B + b] = qr[i]; }
      }
      msgs::foot_state_t fs{ utime, BatchArray(feet.data(), PB_HOST), BatchArray(forces.data(), PB_HOST) };
      on_feet(&fs);
      lo_ticks++;
      for (int b = 0; b < B; b++) {
        const double lt[3] = { feet[b], feet[B + b], feet[2 * B + b] }, ql[4] = { feet[3 * B + b], feet[4 * B + b], feet[5 * B + b], feet[6 * B + b] };
        const double rt[3] = { feet[7 * B + b], feet[8 * B + b], feet[9 * B + b] }, qr[4] = { feet[10 * B + b], feet[11 * B + b], feet[12 * B + b], feet[13 * B + b] };
        double dt3[3], dq[4];
        long prev = 0;
        float status = po_leg_update((po_leg *) legs[b].data(), utime, lt, ql, rt, qr, forces[b], forces[B + b], ox[b].quat, dt3, dq, &prev);
        n_status[status < 0 ? 0 : (status < 0.5 ? 1 : 2)]++;
        if (status < 0) continue;   // "return NULL" (rbis_legodo_update.cpp:243-255), before the counter is touched
        zero_count[b]--;            // zero_initial_velocity = 3 (:264-268), per filter: validity is per filter
        if (zero_count[b] > 0) { dt3[0] = dt3[1] = dt3[2] = 0; dq[0] = 1; dq[1] = dq[2] = dq[3] = 0; }
        int idx[6];
        double z[6], Rd[6], R[36] = { 0 }, p3[3] = { 0, 0, 0 };
        const int m = po_legodo_create_measurement(omode, r5, p3, dt3, dq, utime, prev, 1, status, idx, z, Rd);
        for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
        po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
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
  printf("mode %s%s: status skip/certain/uncertain %d/%d/%d: rel err vec %.2e quat %.2e cov %.2e ll %.2e (status %d, fused pairs %lld)\n",
         lomode.c_str(), fuse ? " fused" : "", n_status[0], n_status[1], n_status[2], ev / sv, eq, eP / sP, el / sl, est.last_status,
         (long long) est.fused_pairs);
  // Tolerance 1e-6: the device integrates the odometry with quaternions, the oracle with rotation matrices like the reference;
  // the two pelvis poses differ at the 1e-12 level after hundreds of steps and the measurement is increment / 0.002 s (and,
  // in lin_rot_rate, the Euler angles of a 1e-4 rad rotation divided by the same 0.002 s).
  const bool fused_ok = !fuse || lomode != "lin_rate" || est.fused_pairs > T / 2;
  const bool ok = fused_ok && est.last_status == PB_OK && n_status[0] > 100 && n_status[1] > 50 && n_status[2] > 100 && ev / sv < 1e-6 && eq < 1e-6 &&
                  eP / sP < 1e-6 && el / sl < 1e-6;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
