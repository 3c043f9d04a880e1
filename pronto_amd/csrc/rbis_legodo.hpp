// rbis_legodo.hpp -- leg kinematic odometry for one robot per lane: which foot is the fixed one, the pelvis pose that
// follows from it, its increment since the previous message and how far that increment can be trusted.
//
// Restatement of (paths relative to the reference tree)
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:172-297   initializePose / prepInitialization /
//                                                               leg_odometry_gravity_slaved_always
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:395-556   updateOdometry (30 ms reset :402-408, delta :485, status :545-551)
//   motion_estimate/src/foot_contact_alt/FootContactAlt.cpp:35-100   DetectFootTransition (primary-foot selection)
//   motion_estimate/src/leg_estimate/foot_contact_classify.cpp:57-125,146-318   update / updateWalkingPhase (status -1 / 0 / 1)
//   estimate_tools/src/filter_tools/SignalTap.cpp:83-130        SchmittTrigger::UpdateState
// Forward kinematics stays with the caller (KDL + URDF in the reference, leg_estimate.cpp:430-444): the inputs are the two
// body-to-foot transforms it produces.  Poses are (translation, unit quaternion) pairs here, where the reference holds
// Eigen::Isometry3d and converts rotation matrices to quaternions and back at every step (:231-240); the two agree to
// rounding (a quaternion's overall sign never matters for a pose).  The joint filters (:411-428), the controller-contact
// override (use_controller_input, :365-387) and the "standing" control mode's classifier (:453-454) are not built.
//
// Per-robot state: NLD doubles + NLI 64-bit integers (flags packed into one of them), struct-of-arrays, robot index fastest.
#pragma once

#include <stdint.h>

#include "rbis_device.hpp"

namespace pb {

struct Pose {
  double t[3], q[4];
};
PB_HD void pose_identity(Pose &p)
{
  p.t[0] = p.t[1] = p.t[2] = 0.0;
  p.q[0] = 1.0; p.q[1] = p.q[2] = p.q[3] = 0.0;
}
PB_HD void quat_rot(const double (&q)[4], const double (&v)[3], double (&o)[3])
{
  double R[9];
  quat_to_rot(q, R);
#pragma unroll
  for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
// a * b
PB_HD void pose_mul(const Pose &a, const Pose &b, Pose &o)
{
  double rt[3], q[4];
  quat_rot(a.q, b.t, rt);
  quat_mul(a.q, b.q, q);
#pragma unroll
  for (int i = 0; i < 3; i++) o.t[i] = a.t[i] + rt[i];
#pragma unroll
  for (int i = 0; i < 4; i++) o.q[i] = q[i];
}
PB_HD void pose_inv(const Pose &a, Pose &o)
{
  const double n2 = a.q[0] * a.q[0] + a.q[1] * a.q[1] + a.q[2] * a.q[2] + a.q[3] * a.q[3];
  const double qi[4] = { a.q[0] / n2, -a.q[1] / n2, -a.q[2] / n2, -a.q[3] / n2 };
  double rt[3];
  quat_rot(qi, a.t, rt);
#pragma unroll
  for (int i = 0; i < 3; i++) o.t[i] = -rt[i];
#pragma unroll
  for (int i = 0; i < 4; i++) o.q[i] = qi[i];
}

// SchmittTrigger (SignalTap.cpp:48-134)
struct Schmitt {
  int64_t status, timer, previous_time, first_call;
};
struct SchmittPar {
  double low, high;
  int64_t low_delay, high_delay;
};
PB_HD void schmitt_reset(Schmitt &s) { s.status = 0; s.timer = 0; s.previous_time = 0; s.first_call = 1; }
PB_HD void schmitt_update(Schmitt &s, const SchmittPar &p, int64_t now, double value)
{
  if (s.first_call) {
    s.first_call = 0;
    s.previous_time = now;
  }
  if (s.status) {
    if (value <= p.low) {
      if (s.timer > p.low_delay) s.status = 0;
      else s.timer += now - s.previous_time;
    } else {
      s.timer = 0;
    }
  } else {
    if (value >= p.high) {
      if (s.timer > p.high_delay) s.status = 1;
      else s.timer += now - s.previous_time;
    } else {
      s.timer = 0;
    }
  }
  s.previous_time = now;
}

enum { LF_UNKNOWN = -1, LF_LEFT = 0, LF_RIGHT = 1 };                                         // footid_alt
enum { LC_UNKNOWN = -1, LC_LEFT_NEW = 0, LC_RIGHT_NEW = 1, LC_LEFT_FIXED = 2, LC_RIGHT_FIXED = 3 };  // contact_status_id

struct LegPar {
  SchmittPar alt;                    // state_estimator.legodo.schmitt_{low,high}_threshold / _delay (leg_estimate.cpp:103-108)
  int filter_contact_events;         // state_estimator.legodo.filter_contact_events (:63)
};

struct LegState {
  Pose odom_to_body, odom_to_primary, odom_to_secondary;
  int64_t utime, leg_odo_init, primary_foot;
  // FootContactAlt
  Schmitt alt_l, alt_r;
  int64_t standing_foot;
  // foot_contact_classify
  Schmitt weak_l, weak_r, strong_l, strong_r;
  int64_t mode, initialized, last_strike, last_break, unknown_transitions;
};
static constexpr int NLD = 21;  // three poses
// Stored integers: utime, 6 x (timer, previous_time), last_strike, last_break (true 64-bit times) + ONE word that packs every
// flag, enum and the transition counter (leg_pack_flags): 16 words instead of the 33 fields of LegState -- the kernel is
// bound by the bytes of this state (21 doubles + these, read and written per robot and message: 432 -> 296 bytes).
static constexpr int NLI = 1 + 12 + 2 + 1;

PB_HD void leg_reset(LegState &s)
{
  pose_identity(s.odom_to_body);
  pose_identity(s.odom_to_primary);
  pose_identity(s.odom_to_secondary);
  s.utime = 0;               // current_utime_ = 0 (leg_estimate.cpp: the first message always triggers the 30 ms reset, harmlessly)
  s.leg_odo_init = 0;        // :128
  s.primary_foot = LF_LEFT;  // :126 primary_foot_ = F_LEFT
  schmitt_reset(s.alt_l); schmitt_reset(s.alt_r);
  s.alt_l.status = 1; s.alt_l.timer = 0;  // forceHigh (FootContactAlt.cpp:28-29)
  s.alt_r.status = 1; s.alt_r.timer = 0;
  s.standing_foot = LF_UNKNOWN;
  schmitt_reset(s.weak_l); schmitt_reset(s.weak_r); schmitt_reset(s.strong_l); schmitt_reset(s.strong_r);
  s.mode = -1;  // UNKNOWN
  s.initialized = 0;
  s.last_strike = 0; s.last_break = 0;
  s.unknown_transitions = 0;
}

// FootContactAlt::DetectFootTransition (FootContactAlt.cpp:35-100).  The reference exit(-1)s when neither foot has ever been
// the standing one and nothing changes; here that returns LC_UNKNOWN (no odometry this tick).
PB_HD int alt_detect(LegState &s, const LegPar &p, int64_t utime, double lz, double rz)
{
  const bool l_last = s.alt_l.status != 0, r_last = s.alt_r.status != 0;
  schmitt_update(s.alt_l, p.alt, utime, lz);
  schmitt_update(s.alt_r, p.alt, utime, rz);
  const bool l = s.alt_l.status != 0, r = s.alt_r.status != 0;
  if (!l_last && l) { s.standing_foot = LF_LEFT; return LC_LEFT_NEW; }
  if (!r_last && r) { s.standing_foot = LF_RIGHT; return LC_RIGHT_NEW; }
  if (l_last && !l) {
    if (s.standing_foot == LF_LEFT) { s.standing_foot = LF_RIGHT; return LC_RIGHT_NEW; }
    return LC_RIGHT_FIXED;
  }
  if (r_last && !r) {
    if (s.standing_foot == LF_RIGHT) { s.standing_foot = LF_LEFT; return LC_LEFT_NEW; }
    return LC_LEFT_FIXED;
  }
  if (s.standing_foot == LF_LEFT) return LC_LEFT_FIXED;
  if (s.standing_foot == LF_RIGHT) return LC_RIGHT_FIXED;
  return LC_UNKNOWN;
}

// foot_contact_classify::updateWalkingPhase (foot_contact_classify.cpp:146-318).  Where the reference blocks on
// `cin >> blah` for a transition it does not know, the mode is kept and the event counted.
PB_HD void walking_phase(LegState &s, int64_t utime, bool lc, bool rc, bool ls, bool rs)
{
  enum { L_PRIME_R_STAND = 0, L_PRIME_R_BREAK, L_PRIME_R_SWING, L_PRIME_R_STRIKE, L_STAND_R_PRIME, L_BREAK_R_PRIME, L_SWING_R_PRIME,
         L_STRIKE_R_PRIME };
  if (!s.initialized) {
    if (lc && rc) { s.mode = L_PRIME_R_STAND; s.initialized = 1; }
    return;
  }
  switch (s.mode) {
  case L_PRIME_R_STAND:
    if (lc && !rs) { s.mode = L_PRIME_R_BREAK; s.last_break = utime; }
    else if (!ls && rc) { s.mode = L_BREAK_R_PRIME; s.last_break = utime; }
    else if (lc && rc) {}
    else s.unknown_transitions++;
    return;
  case L_PRIME_R_BREAK:
    if (lc && !rc) s.mode = L_PRIME_R_SWING;
    else if (lc && rs) s.mode = L_PRIME_R_STAND;
    else if (lc && !rs) {}
    else s.unknown_transitions++;
    return;
  case L_PRIME_R_SWING:
    if (lc && !rc) {}
    else if (lc && rc) { s.mode = L_PRIME_R_STRIKE; s.last_strike = utime; }
    else if (!lc && !rc) {}
    else s.unknown_transitions++;
    return;
  case L_PRIME_R_STRIKE:
    if (lc && rs) s.mode = L_PRIME_R_STAND;
    else if (lc && !rs) {}
    else s.unknown_transitions++;
    return;
  case L_STAND_R_PRIME:
    if (!ls && rc) { s.mode = L_BREAK_R_PRIME; s.last_break = utime; }
    else if (lc && !rs) { s.mode = L_PRIME_R_BREAK; s.last_break = utime; }
    else if (lc && rc) {}
    else s.unknown_transitions++;
    return;
  case L_BREAK_R_PRIME:
    if (!lc && rc) s.mode = L_SWING_R_PRIME;
    else if (ls && rc) s.mode = L_STAND_R_PRIME;
    else if (!ls && rc) {}
    else s.unknown_transitions++;
    return;
  case L_SWING_R_PRIME:
    if (!lc && rc) {}
    else if (lc && rc) { s.mode = L_STRIKE_R_PRIME; s.last_strike = utime; }
    else if (!lc && !rc) {}
    else s.unknown_transitions++;
    return;
  case L_STRIKE_R_PRIME:
    if (ls && rc) s.mode = L_STAND_R_PRIME;
    else if (!ls && rc) {}
    else s.unknown_transitions++;
    return;
  default:
    s.unknown_transitions++;
  }
}

// foot_contact_classify::update (foot_contact_classify.cpp:57-125): 0 accurate, -1 unusable (95 ms after a foot strike),
// 1 very inaccurate (800 ms after a foot break)
PB_HD double classify_update(LegState &s, int64_t utime, double lz, double rz)
{
  const SchmittPar weak = { 20.0, 30.0, 5000, 5000 }, strong = { 275.0, 375.0, 7000, 7000 };  // :34-37
  schmitt_update(s.weak_l, weak, utime, lz);
  schmitt_update(s.weak_r, weak, utime, rz);
  schmitt_update(s.strong_l, strong, utime, lz);
  schmitt_update(s.strong_r, strong, utime, rz);
  walking_phase(s, utime, s.weak_l.status != 0, s.weak_r.status != 0, s.strong_l.status != 0, s.strong_r.status != 0);
  const bool recent_strike = utime - s.last_strike < 95000;   // strike_blackout_duration_ (:41)
  const bool recent_break = utime - s.last_break < 800000;    // break_blackout_duration_ (:42)
  if (recent_strike) return -1.0;
  if (recent_break) return 1.0;
  return 0.0;
}

// odom_to_foot: translation kept, orientation = world_to_body's rotation * body_to_foot's (leg_estimate.cpp:230-240)
PB_HD void slave_foot_orientation(Pose &foot, const double (&wq)[4], const Pose &body_to_foot)
{
  double q[4];
  quat_mul(wq, body_to_foot.q, q);
#pragma unroll
  for (int i = 0; i < 4; i++) foot.q[i] = q[i];
}

// leg_estimate::updateOdometry (leg_estimate.cpp:395-556) without the joint filters and forward kinematics.
//   wq            rotation of world_to_body_ (setPoseBody: the filter's own head orientation, rbis_legodo_update.cpp:214-229)
//   returns the status (-1 no usable increment, 0 accurate, 1 inaccurate); delta = previous_odom_to_body^-1 * odom_to_body
PB_HD double leg_update(LegState &s, const LegPar &p, int64_t utime, const Pose &bl, const Pose &br, double lz, double rz,
                        const double (&wq)[4], Pose &delta, int64_t &prev_utime)
{
  prev_utime = s.utime;
  const Pose previous = s.odom_to_body;
  s.utime = utime;
  if ((double) (s.utime - prev_utime) * 1E-6 > 30E-3) s.leg_odo_init = 0;  // :402-408
  const double classification = classify_update(s, utime, lz, rz);        // :449-450
  const int cs = alt_detect(s, p, utime, lz, rz);                          // footTransitionAlt (:456)
  bool init_this_iteration = false;
  Pose inv;
  if (!s.leg_odo_init) {  // prepInitialization + initializePose, initialization_mode "zero" (:172-216)
    if (cs == LC_LEFT_FIXED || cs == LC_RIGHT_FIXED) {
      const Pose &prim = (cs == LC_LEFT_FIXED) ? bl : br, &sec = (cs == LC_LEFT_FIXED) ? br : bl;
      pose_identity(s.odom_to_primary);
      slave_foot_orientation(s.odom_to_primary, wq, prim);
      pose_inv(prim, inv);
      pose_mul(s.odom_to_primary, inv, s.odom_to_body);
      pose_mul(s.odom_to_body, sec, s.odom_to_secondary);
      s.primary_foot = (cs == LC_LEFT_FIXED) ? LF_LEFT : LF_RIGHT;
      s.leg_odo_init = 1;
      init_this_iteration = true;
    }
  } else if ((cs == LC_LEFT_FIXED && s.primary_foot == LF_LEFT) || (cs == LC_RIGHT_FIXED && s.primary_foot == LF_RIGHT)) {
    // the fixed foot keeps its position and is turned to agree with the pelvis orientation (:227-243, :259-275)
    const Pose &prim = (s.primary_foot == LF_LEFT) ? bl : br, &sec = (s.primary_foot == LF_LEFT) ? br : bl;
    slave_foot_orientation(s.odom_to_primary, wq, prim);
    pose_inv(prim, inv);
    pose_mul(s.odom_to_primary, inv, s.odom_to_body);
    pose_mul(s.odom_to_body, sec, s.odom_to_secondary);
  } else if ((cs == LC_RIGHT_NEW && s.primary_foot == LF_LEFT) || (cs == LC_LEFT_NEW && s.primary_foot == LF_RIGHT)) {
    // transition: the pelvis keeps its position, takes the filter's orientation, and the other foot becomes fixed where
    // forward kinematics puts it (:244-258, :276-291)
    const Pose &prim = (cs == LC_RIGHT_NEW) ? br : bl, &sec = (cs == LC_RIGHT_NEW) ? bl : br;
    Pose sw;
#pragma unroll
    for (int i = 0; i < 3; i++) sw.t[i] = s.odom_to_body.t[i];
#pragma unroll
    for (int i = 0; i < 4; i++) sw.q[i] = wq[i];
    pose_mul(sw, prim, s.odom_to_primary);
    pose_inv(prim, inv);
    pose_mul(s.odom_to_primary, inv, s.odom_to_body);
    pose_mul(s.odom_to_body, sec, s.odom_to_secondary);
    s.primary_foot = (cs == LC_RIGHT_NEW) ? LF_RIGHT : LF_LEFT;
  }  // else: "initialized but unknown update" (:292-294): nothing moves
  double status = -1.0;
  pose_identity(delta);
  if (s.leg_odo_init && !init_this_iteration) {
    pose_inv(previous, inv);
    pose_mul(inv, s.odom_to_body, delta);  // :485
    status = 0.0;
  }
  if (p.filter_contact_events && status > -1.0) status = classification;  // :545-551
  return status;
}

// every flag, enum and counter of LegState in one word: bit 0 leg_odo_init, 1-2 primary_foot + 1, 3-8 the six triggers'
// status, 9-14 their first_call, 15-16 standing_foot + 1, 17-24 mode + 1, 25 initialized, 32-63 unknown_transitions
PB_HD int64_t leg_pack_flags(const LegState &s)
{
  const Schmitt *ss[6] = { &s.alt_l, &s.alt_r, &s.weak_l, &s.weak_r, &s.strong_l, &s.strong_r };
  uint64_t w = (uint64_t) (s.leg_odo_init != 0) | ((uint64_t) (s.primary_foot + 1) & 3u) << 1;
  for (int k = 0; k < 6; k++) w |= (uint64_t) (ss[k]->status != 0) << (3 + k) | (uint64_t) (ss[k]->first_call != 0) << (9 + k);
  w |= ((uint64_t) (s.standing_foot + 1) & 3u) << 15 | ((uint64_t) (s.mode + 1) & 255u) << 17 | (uint64_t) (s.initialized != 0) << 25;
  w |= ((uint64_t) s.unknown_transitions & 0xFFFFFFFFu) << 32;
  return (int64_t) w;
}
PB_HD void leg_unpack_flags(LegState &s, int64_t word)
{
  const uint64_t w = (uint64_t) word;
  Schmitt *ss[6] = { &s.alt_l, &s.alt_r, &s.weak_l, &s.weak_r, &s.strong_l, &s.strong_r };
  s.leg_odo_init = (int64_t) (w & 1u);
  s.primary_foot = (int64_t) ((w >> 1) & 3u) - 1;
  for (int k = 0; k < 6; k++) { ss[k]->status = (int64_t) ((w >> (3 + k)) & 1u); ss[k]->first_call = (int64_t) ((w >> (9 + k)) & 1u); }
  s.standing_foot = (int64_t) ((w >> 15) & 3u) - 1;
  s.mode = (int64_t) ((w >> 17) & 255u) - 1;
  s.initialized = (int64_t) ((w >> 25) & 1u);
  s.unknown_transitions = (int64_t) (w >> 32);
}

// SoA <-> struct (robot index fastest; d: [NLD][stride] doubles, iw: [NLI][stride] 64-bit integers)
PB_HD void leg_load(LegState &s, const double *d, const int64_t *iw, long stride, long b)
{
  Pose *ps[3] = { &s.odom_to_body, &s.odom_to_primary, &s.odom_to_secondary };
  for (int k = 0; k < 3; k++) {
    for (int i = 0; i < 3; i++) ps[k]->t[i] = d[(long) (7 * k + i) * stride + b];
    for (int i = 0; i < 4; i++) ps[k]->q[i] = d[(long) (7 * k + 3 + i) * stride + b];
  }
  int c = 0;
  auto rd = [&]() { return iw[(long) (c++) * stride + b]; };
  s.utime = rd();
  Schmitt *ss[6] = { &s.alt_l, &s.alt_r, &s.weak_l, &s.weak_r, &s.strong_l, &s.strong_r };
  for (int k = 0; k < 6; k++) { ss[k]->timer = rd(); ss[k]->previous_time = rd(); }
  s.last_strike = rd(); s.last_break = rd();
  leg_unpack_flags(s, rd());
}
PB_HD void leg_store(const LegState &s, double *d, int64_t *iw, long stride, long b)
{
  const Pose *ps[3] = { &s.odom_to_body, &s.odom_to_primary, &s.odom_to_secondary };
  for (int k = 0; k < 3; k++) {
    for (int i = 0; i < 3; i++) d[(long) (7 * k + i) * stride + b] = ps[k]->t[i];
    for (int i = 0; i < 4; i++) d[(long) (7 * k + 3 + i) * stride + b] = ps[k]->q[i];
  }
  int c = 0;
  auto wr = [&](int64_t v) { iw[(long) (c++) * stride + b] = v; };
  wr(s.utime);
  const Schmitt *ss[6] = { &s.alt_l, &s.alt_r, &s.weak_l, &s.weak_r, &s.strong_l, &s.strong_r };
  for (int k = 0; k < 6; k++) { wr(ss[k]->timer); wr(ss[k]->previous_time); }
  wr(s.last_strike); wr(s.last_break);
  wr(leg_pack_flags(s));
}

#if defined(__HIPCC__)
// One robot per lane: leg_update on its state, with the filter's own head orientation as world_to_body_ (setPoseBody,
// rbis_legodo_update.cpp:214-229), then -- optionally -- LegOdoCommon::createMeasurement in mode lin_rate on the result
// (rbis_legodo_common.cpp:99-107,124-129,153-156): z = delta translation / elapsed time, R = r_vxyz^2 or r_vxyz_uncertain^2
// when status >= 0.5, no update (mask 0) when status < 0.  feet [14][B] = left (t3, q4), right (t3, q4); forces [2][B].
// ONE robot's foot poses and forces for every filter of the batch (PB_HOST_BROADCAST: a parameter sweep over one log): the 16
// values are kernel arguments.
struct LegBcast {
  double feet[14] = { 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0 }, forces[2] = { 0, 0 };
  double imu[7] = { 0, 0, 0, 0, 0, 0, 0 };
  int on = 0;  // bit 0: feet / forces are broadcast; bit 1: AHEAD (see k_legodo); bit 2: the IMU block of AHEAD is broadcast
};
// (launch bounds: without them the compiler budgets for 1024-thread blocks, 128 registers, and spilled 668 bytes per lane)
// AHEAD (bc.on & 2): the odometry is slaved to the orientation the filter WILL have after the IMU step in `imu` / bc.imu
// (rbis_update_interface.cpp:30-52 applied to the head), computed here from the head state with the step kernels' own
// ins_update_state -- the covariance is not touched.  That lets the estimator run the IMU step and the leg-odometry update
// it produces as ONE fused kernel afterwards instead of predict, odometry, update.
template <int NS>
static __global__ __launch_bounds__(64, 2) void k_legodo(const double *__restrict__ st, double *__restrict__ legd,
                                                         int64_t *__restrict__ legi, long stride, int B, int64_t utime, LegPar par,
                                                         const double *__restrict__ feet, const double *__restrict__ forces,
                                                         int zero_delta, double r2, double r2_uncertain,
                                                         double *__restrict__ delta_out, double *__restrict__ status_out,
                                                         double *__restrict__ lo_out, uint8_t *__restrict__ mask_out, LegBcast bc,
                                                         const double *__restrict__ imu, Consts k)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  LegState s;
  leg_load(s, legd, legi, stride, b);
  Pose bl, br, delta;
  double fl, fr;
  if (bc.on & 1) {  // wave-uniform
    for (int i = 0; i < 3; i++) { bl.t[i] = bc.feet[i]; br.t[i] = bc.feet[7 + i]; }
    for (int i = 0; i < 4; i++) { bl.q[i] = bc.feet[3 + i]; br.q[i] = bc.feet[10 + i]; }
    fl = bc.forces[0]; fr = bc.forces[1];
  } else {
    for (int i = 0; i < 3; i++) { bl.t[i] = feet[(long) i * B + b]; br.t[i] = feet[(long) (7 + i) * B + b]; }
    for (int i = 0; i < 4; i++) { bl.q[i] = feet[(long) (3 + i) * B + b]; br.q[i] = feet[(long) (10 + i) * B + b]; }
    fl = forces[b]; fr = forces[(long) B + b];
  }
  double wq[4];
  for (int i = 0; i < 4; i++) wq[i] = st[S::eidx(L::OFF_QUAT + i, b)];
  if (bc.on & 2) {
    double x[NS], gyro[3], accel[3], dt;
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = st[S::eidx(L::OFF_VEC + i, b)];
    if (bc.on & 4) {
      for (int i = 0; i < 3; i++) { gyro[i] = bc.imu[i]; accel[i] = bc.imu[3 + i]; }
      dt = bc.imu[6];
    } else {
      for (int i = 0; i < 3; i++) { gyro[i] = imu[(long) i * B + b]; accel[i] = imu[(long) (3 + i) * B + b]; }
      dt = imu[(long) 6 * B + b];
    }
    ins_update_state<NS>(x, wq, gyro, accel, dt, k);
  }
  int64_t prev = 0;
  const double status = leg_update(s, par, utime, bl, br, fl, fr, wq, delta, prev);
  leg_store(s, legd, legi, stride, b);
  if (zero_delta) pose_identity(delta);  // "Ignore the calculated velocity at launch" (rbis_legodo_update.cpp:264-268)
  if (delta_out != nullptr) {
    for (int i = 0; i < 3; i++) delta_out[(long) i * B + b] = delta.t[i];
    for (int i = 0; i < 4; i++) delta_out[(long) (3 + i) * B + b] = delta.q[i];
  }
  if (status_out != nullptr) status_out[b] = status;
  if (lo_out != nullptr) {
    const double elapsed = (double) (utime - prev) * 1E-6;  // pronto_conversions_lcm.hpp:38-87
    for (int i = 0; i < 3; i++) {
      lo_out[(long) i * B + b] = delta.t[i] / elapsed;
      lo_out[(long) (3 + i) * B + b] = (status >= 0.5) ? r2_uncertain : r2;
    }
    if (mask_out != nullptr) mask_out[b] = status < 0 ? 0 : 1;
  }
}
static __global__ void k_legodo_reset(double *legd, int64_t *legi, long stride, int B)
{
  const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  LegState s;
  leg_reset(s);
  leg_store(s, legd, legi, stride, b);
}
static __global__ void k_legodo_get(const double *legd, const int64_t *legi, long stride, long b, double *pose7, int64_t *info)
{
  LegState s;
  leg_load(s, legd, legi, stride, b);
  for (int i = 0; i < 3; i++) pose7[i] = s.odom_to_body.t[i];
  for (int i = 0; i < 4; i++) pose7[3 + i] = s.odom_to_body.q[i];
  info[0] = s.primary_foot; info[1] = s.leg_odo_init; info[2] = s.mode; info[3] = s.unknown_transitions;
}
#endif

}  // namespace pb
