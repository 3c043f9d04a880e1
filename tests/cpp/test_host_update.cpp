// test_host_update.cpp -- a USER-DEFINED update written against the reference's contract
//     virtual void updateFilter(const RBIS & prior_state, const RBIM & prior_cov, double prior_loglikelihood) = 0;
// (rbis_update_interface.hpp:14-35: "fill posterior_state, posterior_covariance, loglikelihood") -- the pattern of a third-party
// RBISUpdateInterface subclass such as RBISOpticalFlowMeasurement (rbis_update_interface.hpp:128-154) -- handed to
// MavStateEstimator::addUpdate next to the built-in updates.  RBISHostUpdate runs it on the slow path (head to the host, user code
// per filter, posterior back; pb_get_head / pb_set_head).  The user's update here is a scalar altimeter on position z written
// from the textbook equations (gain, Joseph-free downdate, delta folded into the quaternion) WITHOUT calling the library or the
// oracle; the expected result is the oracle's po_indexed_update in the same place of the same message sequence.
//   argv: "n21" = 21 states, "slots" = with posterior checkpoints (the host update then writes into its checkpoint slot) and one
//   altimeter message that arrives LATE (replayed from the checkpoint), "mask" = the update applies to every other filter only.
// Exit code 0 + "PASS".  Needs a GPU.
#include <cstdio>
#include <string>
#include <vector>

#include "test_n.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x484F5354ULL;
static double urand()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((rng_state >> 11) + 0.5) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand())) * cos(2 * M_PI * urand()); }

// what a user of the reference writes: one filter, the reference's signature
class UserAltimeter : public RBISHostUpdate {
public:
  double z, R;
  UserAltimeter(double z_, double R_, int64_t utime) : RBISHostUpdate(RBISUpdateInterface::altimeter, utime), z(z_), R(R_) {}
  void updateFilter(const RBIS &prior_state, const RBIM &prior_cov, double prior_loglikelihood) override
  {
    const int n = prior_state.n, i = RBIS::position_ind + 2;
    const double S = prior_cov(i, i, 0) + R, r = z - prior_state(i, 0);
    std::vector<double> K((size_t) n);
    for (int a = 0; a < n; a++) K[(size_t) a] = prior_cov(a, i, 0) / S;
    posterior_state = prior_state;
    posterior_covariance = prior_cov;
    for (int a = 0; a < n; a++)
      for (int b = 0; b < n; b++) posterior_covariance(a, b, 0) = prior_cov(a, b, 0) - K[(size_t) a] * prior_cov(i, b, 0);
    // dstate = RBIS(K * residual): a state whose attitude part is folded into its OWN quaternion when it exceeds the tolerance
    // (RigidBodyState(vec) -> chiToQuat); posterior = prior.addState(dstate): vec += dstate.vec, chiToQuat() on what is in chi then
    // (normally nothing), quat *= dstate.quat  (rbis.cpp:219-227, rbis.hpp's constructors)
    auto expq = [](const double c[3], double d[4]) {
      const double ang = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), sn = sin(0.5 * ang) / ang;
      d[0] = cos(0.5 * ang); d[1] = sn * c[0]; d[2] = sn * c[1]; d[3] = sn * c[2];
    };
    auto mulq = [](const double q[4], const double d[4], double o[4]) {
      o[0] = q[0] * d[0] - q[1] * d[1] - q[2] * d[2] - q[3] * d[3];
      o[1] = q[0] * d[1] + q[1] * d[0] + q[2] * d[3] - q[3] * d[2];
      o[2] = q[0] * d[2] - q[1] * d[3] + q[2] * d[0] + q[3] * d[1];
      o[3] = q[0] * d[3] + q[1] * d[2] - q[2] * d[1] + q[3] * d[0];
    };
    std::vector<double> d((size_t) n);
    for (int a = 0; a < n; a++) d[(size_t) a] = K[(size_t) a] * r;
    double dq[4] = { 1, 0, 0, 0 }, dchi[3] = { d[6], d[7], d[8] };
    if (sqrt(dchi[0] * dchi[0] + dchi[1] * dchi[1] + dchi[2] * dchi[2]) > 1e-6) {
      expq(dchi, dq);
      d[6] = d[7] = d[8] = 0.0;
    }
    for (int a = 0; a < n; a++) posterior_state(a, 0) += d[(size_t) a];
    double q[4] = { prior_state.q(0, 0), prior_state.q(1, 0), prior_state.q(2, 0), prior_state.q(3, 0) }, o[4];
    double chi[3] = { posterior_state(6, 0), posterior_state(7, 0), posterior_state(8, 0) };
    if (sqrt(chi[0] * chi[0] + chi[1] * chi[1] + chi[2] * chi[2]) > 1e-6) {
      double e[4];
      expq(chi, e);
      mulq(q, e, o);
      memcpy(q, o, sizeof q);
      for (int k = 0; k < 3; k++) posterior_state(RBIS::chi_ind + k, 0) = 0.0;
    }
    mulq(q, dq, o);
    for (int k = 0; k < 4; k++) posterior_state.q(k, 0) = o[k];
    loglikelihood = prior_loglikelihood - log(S) - r * r / S;
  }
};

int main(int argc, char **argv)
{
  const int n = take_n_states(argc, argv);
  bool slots = false, mask = false;
  for (int i = 1; i < argc; i++) {
    if (std::string(argv[i]) == "slots") slots = true;
    if (std::string(argv[i]) == "mask") mask = true;
  }
  const int B = 64, T = 60;
  double g;
  po_get_constants(&g, nullptr);
  BotParam param;
  param.set("state_estimator.utime_history_span", "1000000");
  param.set("state_estimator.history_slots", slots ? "16" : "0");
  if (slots) param.set("state_estimator.history_checkpoint_every", "1");
  RBIS x0(n, B);
  RBIM P0(n, B);
  std::vector<po_rbis> ox((size_t) B);
  std::vector<po_rbim> oP((size_t) B);
  std::vector<double> oll((size_t) B, 0.0);
  for (int b = 0; b < B; b++) {
    double q[4];
    po_euler_to_quat(0.1 * (urand() - 0.5), 0.1 * (urand() - 0.5), 6.0 * (urand() - 0.5), q);
    po_rbis_zero(&ox[(size_t) b]);
    memset(&oP[(size_t) b], 0, sizeof(po_rbim));
    for (int i = 0; i < 4; i++) { x0.q(i, b) = q[i]; ox[(size_t) b].quat[i] = q[i]; }
    const double sig[15] = { 0, 0, 0, .15, .15, .15, .05, .05, .05, .5, .5, .5, 0, 0, 0 };
    for (int i = 0; i < 15; i++) { P0(i, i, b) = sig[i] * sig[i]; oP[(size_t) b].m[i * 21 + i] = sig[i] * sig[i]; }
    init_bias_states(n, b, x0, P0, &ox[(size_t) b], &oP[(size_t) b], urand);
  }
  MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
  const double qg = bot_sq(bot_to_radians(0.5)), qa = 0.01, qbg = n == 21 ? bot_sq(bot_to_radians(0.001)) : 0.0, qba = n == 21 ? 1e-8 : 0.0;
  std::vector<uint8_t> apply;
  if (mask)
    for (int b = 0; b < B; b++) apply.push_back((uint8_t) (b % 2 == 0));
  int n_user = 0;
  struct Late { int64_t utime; double z; };
  std::vector<Late> held;   // ("slots": an altimeter message that is delivered three ticks late)
  auto oracle_alt = [&](double z) {
    for (int b = 0; b < B; b++) {
      if (mask && b % 2) continue;
      const int idx[1] = { 11 };
      const double R[1] = { 0.04 };
      po_indexed_update(1, idx, &z, R, &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
    }
  };
  for (int k = 0; k < T; k++) {
    const int64_t utime = 1000 + (int64_t) (k + 1) * 2000;
    std::vector<double> blk((size_t) 7 * B);
    for (int b = 0; b < B; b++) {
      const double v[6] = { 0.2 * sin(0.05 * k + b), 0.05, -0.1 * cos(0.03 * k), 0.3 * nrand(), 0.3 * nrand(), g + 0.3 * nrand() };
      for (int i = 0; i < 6; i++) blk[(size_t) i * B + b] = v[i];
      blk[(size_t) 6 * B + b] = 0.002;
      po_imu_process_step(v, v + 3, 0.002, qg, qa, qbg, qba, &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
    }
    // (the oracle applies everything in TIME order: a held-back altimeter message belongs right here, behind its IMU sample)
    const bool alt_tick = k % 7 == 3;
    const double z_alt = 0.3 * nrand();
    if (alt_tick) oracle_alt(z_alt);
    est.addUpdate(new RBISIMUProcessStep(std::move(blk), qg, qa, qbg, qba, utime), true);
    if (alt_tick) {
      if (slots && n_user == 2) held.push_back(Late{ utime + 100, z_alt });   // the third one arrives late
      else {
        auto *u = new UserAltimeter(z_alt, 0.04, utime + 100);
        u->apply = apply;
        est.addUpdate(u, true);
      }
      n_user++;
    }
    // a built-in velocity measurement on every third tick, so that the covariance has cross terms the user update must respect
    if (k % 3 == 1) {
      std::vector<double> z((size_t) 3 * B), R((size_t) 3 * B, 0.25);
      for (int b = 0; b < B; b++) {
        double zb[3];
        for (int i = 0; i < 3; i++) zb[i] = z[(size_t) i * B + b] = 0.2 * nrand();
        const int idx[3] = { 3, 4, 5 };
        const double Rf[9] = { 0.25, 0, 0, 0, 0.25, 0, 0, 0, 0.25 };
        po_indexed_update(3, idx, zb, Rf, &ox[(size_t) b], &oP[(size_t) b], oll[(size_t) b], &ox[(size_t) b], &oP[(size_t) b], &oll[(size_t) b]);
      }
      est.addUpdate(new RBISIndexedMeasurement(RBIS::velocityInds(), std::move(z), std::move(R), PB_R_DIAG, std::vector<uint8_t>(), RBISUpdateInterface::legodo, utime + 300), true);
    }
    if (!held.empty() && utime > held[0].utime + 5000) {
      auto *u = new UserAltimeter(held[0].z, 0.04, held[0].utime);
      u->apply = apply;
      est.addUpdate(u, true);   // inserted at its time stamp, everything behind it re-applied (mav_state_est.cpp:28-80)
      held.clear();
    }
  }
  RBIS head;
  RBIM cov;
  est.getHeadState(head, cov);
  std::vector<double> ll = est.getMeasurementsLogLikelihood();
  double ev = 0, eq = 0, eP = 0, el = 0, sv = 0, sP = 0, sl = 1e-300;
  for (int b = 0; b < B; b++) {
    for (int i = 0; i < n; i++) { ev = fmax(ev, fabs(head(i, b) - ox[(size_t) b].vec[i])); sv = fmax(sv, fabs(ox[(size_t) b].vec[i])); }
    for (int i = 0; i < 4; i++) eq = fmax(eq, fabs(head.q(i, b) - ox[(size_t) b].quat[i]));
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) { eP = fmax(eP, fabs(cov(r, c, b) - oP[(size_t) b].m[c * 21 + r])); sP = fmax(sP, fabs(oP[(size_t) b].m[c * 21 + r])); }
    el = fmax(el, fabs(ll[(size_t) b] - oll[(size_t) b]));
    sl = fmax(sl, fabs(oll[(size_t) b]));
  }
  printf("n=%d%s%s: %d user-defined updates (reference signature) among %d IMU steps, replayed after a late arrival %lld: rel err vs oracle vec %.2e quat %.2e cov %.2e ll %.2e (status %d)\n",
         n, slots ? " checkpoints" : "", mask ? " masked" : "", n_user, T, (long long) est.replayed_updates, ev / sv, eq, eP / sP, el / sl, est.last_status);
  const bool ok = est.last_status == PB_OK && n_user >= 8 && (!slots || est.replayed_updates > 0) && ev / sv < 1e-9 && eq < 1e-9 && eP / sP < 1e-9 && el / sl < 1e-9;
  printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
