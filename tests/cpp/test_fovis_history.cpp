// test_fovis_history.cpp -- FovisHandler's keyframe look-up through the estimator's history
// (rbis_fovis_update.cpp:177-213: T0 = posterior of history.updateMap.lower_bound(prev_timestamp), at most 25 ms later,
// cached while prev_timestamp does not change), with posterior checkpoints on and NO manual markKeyframe.  The oracle keeps
// the posterior after every update and takes T0 from the same place.  Exit code 0 + "PASS".  Needs a GPU.
//   argv[1] = "derived": only utime_history_span is set (every reference .cfg): 32 checkpoint slots with a derived
//             history_checkpoint_every > 1, so most look-ups land on an update WITHOUT a saved posterior, which the
//             estimator re-derives from the nearest earlier checkpoint (MavStateEstimator::snapshotPosteriorOf);
//   argv[1] = "fuse": fuse_ins_legodo with a leg-odometry update behind every INS step (same utime): the look-up lands on
//             the INS half of a fused pair, which never has a checkpoint of its own.
//   argv[1] = "empty": 500 us after every INS message comes a leg-odometry update whose device-resident mask lets NO filter through
//             -- the message for which the reference's handler returns NULL (rbis_legodo_update.cpp:242-255), so the reference's
//             history never holds it: the look-up must step over it (the oracle does not record it).
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x464F564953ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);  // "n21" anywhere on the command line: the 21-state filter
  const std::string variant = argc > 1 ? argv[1] : "";
  const bool derived = variant == "derived", fuse = variant == "fuse", empty = variant == "empty";
  const int B = 70, T = 60;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "40000");  // 40 ms window
  if (!derived) param.set("state_estimator.history_slots", "64");
  param.set("state_estimator.fuse_ins_legodo", fuse ? "true" : "false");
  param.applyOverrides("state_estimator.legodo.mode=lin_rate|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=0.3|"
                       "state_estimator.legodo.r_vang=0.3|state_estimator.legodo.r_vxyz_uncertain=0.5|state_estimator.legodo.r_vang_uncertain=0.9|"
                       "state_estimator.legodo.zero_initial_velocity=0");
  param.set("state_estimator.ins.channel", "IMU");
  param.set("state_estimator.ins.q_gyro", 0.5);
  param.set("state_estimator.ins.q_accel", 0.1);
  param.set("state_estimator.ins.timestep_dt", 0.001);
  param.set("state_estimator.ins.atlas_filter", "false");
  set_ins_bias_keys(param, n);
  param.applyOverrides("state_estimator.fovis.mode=position_orient|state_estimator.fovis.r_pxyz=0.02|state_estimator.fovis.r_chi=0.01");
  for (const char *s : { "ins", "fovis", "legodo" }) {
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
  FovisHandler fovis_handler(&param, 0);
  FrontEnd front_end(&param);
  auto on_ins = front_end.addSensor("ins", &InsHandler::processMessage, &ins_handler);
  auto on_fovis = front_end.addSensor("fovis", &FovisHandler::processMessage, &fovis_handler);
  LegOdoHandler legodo_handler(&param);
  auto on_legodo = front_end.addSensor("legodo", &LegOdoHandler::processMessageDelta, &legodo_handler);
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  front_end.setStateEstimator(&est);
  const double q4[4] = { ins_handler.cov_gyro, ins_handler.cov_accel, ins_handler.cov_gyro_bias, ins_handler.cov_accel_bias };

  // oracle-side history: posterior (Delta, quat) after every update, keyed by utime in arrival order
  std::vector<int64_t> h_utime;
  std::vector<std::vector<po_rbis>> h_state;
  auto record = [&](int64_t utime) { h_utime.push_back(utime); h_state.push_back(ox); };
  record(0);
  int64_t key_utime = -1;      // prev_timestamp of the VO messages (the keyframe)
  std::vector<po_rbis> key;    // oracle's cached T0
  int n_vo = 0, n_rejected = 0, n_key_changes = 0;
  std::vector<double> vt(3 * B), vq(4 * B);
  double *d_empty = nullptr;
  for (int k = 0; k < T; k++) {
    const int64_t utime = (int64_t) (k + 1) * 1000;
    const double v[6] = { 0.3 * sin(0.05 * k), 0.1, -0.2 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
    if (fuse) {  // per-filter host blocks on both sides: the pair the estimator fuses (pb_step_legodo with PB_HOST inputs)
      std::vector<double> gy(3 * B), ac(3 * B);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) { gy[i * B + b] = v[i]; ac[i * B + b] = v[3 + i]; }
      msgs::ins_t im{ utime, BatchArray(gy.data(), PB_HOST), BatchArray(ac.data(), PB_HOST) };
      on_ins(&im);
    } else {
      msgs::ins_t im{ utime, BatchArray(v, PB_HOST_BROADCAST), BatchArray(v + 3, PB_HOST_BROADCAST) };
      on_ins(&im);
    }
    for (int b = 0; b < B; b++) po_imu_process_step(v, v + 3, 0.001, q4[0], q4[1], q4[2], q4[3], &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
    record(utime);
    if (empty) {
      if (d_empty == nullptr) {
        std::vector<double> blk((size_t) 7 * B, 1.0);  // z [3][B] | R diagonal [3][B] | mask [B] bytes (zero) in the 7th row
        memset(&blk[(size_t) 6 * B], 0, sizeof(double) * B);
        void *p = nullptr;
        if (pb_malloc(est.ctx, sizeof(double) * 7 * B, &p) != PB_OK || pb_memcpy_h2d(est.ctx, p, blk.data(), sizeof(double) * 7 * B) != PB_OK) { printf("FAIL: pb_malloc\n"); return 1; }
        d_empty = (double *) p;
      }
      est.addUpdate(new RBISIndexedMeasurement(RBIS::velocityInds(), BatchArray(d_empty, PB_DEVICE), d_empty + (size_t) 3 * B, PB_R_DIAG,
                                               (const uint8_t *) (d_empty + (size_t) 6 * B), RBISUpdateInterface::legodo, utime + 500), true);
    }
    if (fuse) {  // a leg-odometry increment with the INS message's utime: the pair runs as one fused kernel
      std::vector<double> dtr(3 * B);
      std::vector<float> st(B, 0.f);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) dtr[i * B + b] = 0.001 * (ox[b].vec[3 + i] + 0.1 * nrand());
      msgs::legodo_delta_t lm{ utime, utime - 1000, nullptr, dtr.data(), nullptr, nullptr, st.data() };
      on_legodo(&lm);
      const double r5[5] = { 0.2, 0.3, 0.3, 0.5, 0.9 }, dq[4] = { 1, 0, 0, 0 }, p3[3] = { 0, 0, 0 };
      for (int b = 0; b < B; b++) {
        const double d3[3] = { dtr[b], dtr[B + b], dtr[2 * B + b] };
        int idx[6];
        double z[6], Rd[6], R[36] = { 0 };
        const int m = po_legodo_create_measurement(0, r5, p3, d3, dq, utime, utime - 1000, 1, 0.f, idx, z, Rd);
        for (int i = 0; i < m; i++) R[i * m + i] = Rd[i];
        po_indexed_update(m, idx, z, R, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
      record(utime);
    }
    if (k % 6 == 5) {
      // the keyframe changes every second VO message: it is an instant 300 us after an IMU message 4-5 ms ago, so the
      // look-up lands on the NEXT update (700 us later); at k == 41 it points 30 ms back -> outside the 40 ms window's
      // ... no: inside the window but the nearest later update is <= 25 ms away only if it exists: use a gap instead
      int64_t prev_ts = key_utime;
      if ((n_vo % 2) == 0) prev_ts = utime - 4700;
      if (k == 41) prev_ts = utime + 26000;  // beyond the newest update: "at the end" -> rejected
      for (int b = 0; b < B; b++) {
        double dq[4];
        po_euler_to_quat(0.01 * nrand(), 0.01 * nrand(), 0.02 * nrand(), dq);
        for (int i = 0; i < 3; i++) vt[i * B + b] = 0.02 * nrand();
        for (int i = 0; i < 4; i++) vq[i * B + b] = dq[i];
      }
      msgs::update_t vo{ utime, prev_ts, nullptr, BatchArray(vt.data(), PB_HOST), BatchArray(vq.data(), PB_HOST) };
      on_fovis(&vo);
      n_vo++;
      // oracle: the reference's look-up
      bool use = true;
      if (prev_ts != key_utime) {
        size_t j = 0;
        while (j < h_utime.size() && h_utime[j] < prev_ts) j++;
        if (j == h_utime.size() || (double) (h_utime[j] - prev_ts) * 1E-6 > 0.025) use = false;
        else { key = h_state[j]; key_utime = prev_ts; n_key_changes++; }
      }
      if (!use) { n_rejected++; continue; }
      for (int b = 0; b < B; b++) {
        const double t3[3] = { vt[b], vt[B + b], vt[2 * B + b] }, q[4] = { vq[b], vq[B + b], vq[2 * B + b], vq[3 * B + b] };
        double z[6] = { 0 }, qm[4], R[36] = { 0 };
        po_fovis_compose(key[b].vec + 9, key[b].quat, t3, q, z, qm);
        const int idx[6] = { 9, 10, 11, 6, 7, 8 };
        for (int i = 0; i < 6; i++) R[i * 6 + i] = (i < 3) ? 0.02 * 0.02 : 0.01 * 0.01;
        po_indexed_orient_update(6, idx, z, R, qm, &ox[b], &oP[b], oll[b], &ox[b], &oP[b], &oll[b]);
      }
      record(utime);
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
  printf("%s: %d VO messages, %d keyframe changes looked up in the history (%lld posteriors re-derived, checkpoint every %d, %lld fused pairs), "
         "%d rejected: rel err vec %.2e quat %.2e cov %.2e ll %.2e (status %d)\n", variant.c_str(), n_vo, n_key_changes,
         (long long) est.rederived_posteriors, est.checkpoint_every, (long long) est.fused_pairs, n_rejected, ev / sv, eq, eP / sP, el / sl,
         est.last_status);
  const bool variant_ok = (!derived || (est.checkpoint_every > 1 && est.rederived_posteriors >= 2)) &&
                          (!fuse || (est.fused_pairs > T / 2 && est.rederived_posteriors >= 2));
  const bool ok = variant_ok && est.last_status == PB_OK && n_vo == T / 6 && n_key_changes >= 4 && n_rejected == 1 && ev / sv < 1e-9 && eq < 1e-9 &&
                  eP / sP < 1e-9 && el / sl < 1e-9;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
