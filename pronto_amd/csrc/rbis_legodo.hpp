// rbis_legodo.hpp -- leg kinematic odometry for one robot per lane: forward kinematics of the two legs from the joint
// angles, which foot is the fixed one, the pelvis pose that follows from it, its increment since the previous message and
// how far that increment can be trusted.
//
// Restatement of (paths relative to the reference tree)
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:430-447   forward kinematics (KDL TreeFkSolverPosFull_recursive over the
//                                                               URDF tree; KDL / kdl_parser / urdfdom are NOT in the tree: the
//                                                               published algorithm is restated, see leg_fk)
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:172-297   initializePose / prepInitialization /
//                                                               leg_odometry_gravity_slaved_always
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:322-393   footTransition ("standing" mode), footTransitionAlt incl. the
//                                                               controller-contact override (use_controller_input)
//   motion_estimate/src/leg_estimate/leg_estimate.cpp:395-556   updateOdometry (30 ms reset :402-408, delta :485, status :545-551)
//   motion_estimate/src/foot_contact/FootContact.cpp:29-83      DetectFootTransition of the "standing" mode (float arithmetic)
//   motion_estimate/src/foot_contact_alt/FootContactAlt.cpp:35-130   DetectFootTransition, forceLeft/RightStandingFoot
//   motion_estimate/src/leg_estimate/foot_contact_classify.cpp:57-125,146-318   update / updateWalkingPhase (status -1 / 0 / 1)
//   estimate_tools/src/filter_tools/SignalTap.cpp:64-130        SchmittTrigger
//   estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp:27-62   TorqueAdjustment::processSample (float arithmetic)
// Poses are (translation, unit quaternion) pairs here, where the reference holds Eigen::Isometry3d and converts rotation
// matrices to quaternions and back at every step (:231-240); the two agree to rounding (a quaternion's overall sign never
// matters for a pose; Isometry3d::inverse() transposes, which is the quaternion CONJUGATE, no division).  The joint
// low-pass / Kalman filters in front of the kinematics (:411-428) are rbis_jointfilt.hpp.
//
// Per-robot state (round 3: 136 bytes, was 296): NLD doubles + NLI 64-bit words, struct-of-arrays, robot index fastest.
//   * odom_to_secondary_foot_ is not kept: the reference only draws it (pc_vis_, determineContactPoints);
//   * odom_to_primary_foot_fixed_'s rotation is not kept: every branch that reads the pose overwrites it first (:233-240);
//   * the six triggers' previous_time is the previous message's utime for all of them (they are all updated with the same
//     clock in every call), so it is not stored six times; their timers are only ever compared with a delay and are kept as
//     SATURATING 32-bit values (identical decisions for |utime steps| < 2^31 us = 35 minutes and delays < 2^31 us).
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
// toRotationMatrix() * v without forming the matrix: v + 2 w (u x v) + 2 u x (u x v) -- the same polynomial in q as Eigen's
PB_HD void quat_rot(const double (&q)[4], const double (&v)[3], double (&o)[3])
{
  const double tx = 2.0 * (q[2] * v[2] - q[3] * v[1]), ty = 2.0 * (q[3] * v[0] - q[1] * v[2]), tz = 2.0 * (q[1] * v[1] - q[2] * v[0]);
  o[0] = v[0] + q[0] * tx + (q[2] * tz - q[3] * ty);
  o[1] = v[1] + q[0] * ty + (q[3] * tx - q[1] * tz);
  o[2] = v[2] + q[0] * tz + (q[1] * ty - q[2] * tx);
}
PB_HD void quat_conj(const double (&q)[4], double (&o)[4]) { o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3]; }
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
// Isometry3d::inverse(): (R^T, -R^T t)
PB_HD void pose_inv(const Pose &a, Pose &o)
{
  double qi[4], rt[3];
  quat_conj(a.q, qi);
  quat_rot(qi, a.t, rt);
#pragma unroll
  for (int i = 0; i < 3; i++) o.t[i] = -rt[i];
#pragma unroll
  for (int i = 0; i < 4; i++) o.q[i] = qi[i];
}

// (sincos_joint: rbis_device.hpp -- the filter's own exponential maps use it too)

// ---- forward kinematics of the two legs -------------------------------------------------------------------------------
// What KDL computes for leg_estimate.cpp:430-444 (JntToCart with flatten_tree = true, then the frames of the two standing
// links): kdl_parser turns every URDF joint into a KDL segment whose pose is
//     pose(theta) = (Rot(M a, theta) M, p)      M, p = the joint's <origin rpy xyz>, a = its <axis>
// and the solver multiplies the segment poses from the root link down, so body_to_foot = prod_j [M_j, p_j] * Rot(a_j, theta_j)
// (Rot(M a, theta) M = M Rot(a, theta)).  The chain table is supplied by the caller (the URDF is not in the tree):
// per leg the joints from the root link to the standing link, each with its origin, axis, type and the row of the
// joint-position block that holds its angle.  Here the product is formed with quaternions and half angles; the oracle
// forms KDL's 3x3 matrices with Rot2's Rodrigues formula (oracle/leg_odometry.c).
static constexpr int LEG_MAXJ = 8;  // joints per leg chain (Atlas: 6)
enum { LJ_FIXED = 0, LJ_REVOLUTE = 1, LJ_PRISMATIC = 2 };
// per-joint code word: bits 0-1 type, bit 2 the origin has a rotation (rpy != 0), bit 3 the origin has a translation,
// bits 4-5 revolute axis kind (0 general, 1 / 2 / 3 = x / y / z: half the multiplications), bit 6 that axis is negative
enum { LC_TYPE = 3, LC_ORG_ROT = 4, LC_ORG_T = 8, LC_AXIS_SHIFT = 4, LC_AXIS_NEG = 64 };
static constexpr int LEG_REC = 10;  // doubles per joint record: <origin xyz> (3), <origin rpy> as urdfdom's quaternion (4), <axis> normalised (3)
struct LegChain {
  int n[2];                        // joints of the left / right chain
  int code[2][LEG_MAXJ];           // LC_*
  int row[2][LEG_MAXJ];            // row of the joint-position block ([rows][B]); unused for LJ_FIXED
  float gain[2][LEG_MAXJ];         // TorqueAdjustment spring constant of the joint, 0 = none (torque_adjustment.cpp:52)
  double rec[2][LEG_MAXJ][LEG_REC];
};

// TorqueAdjustment::processSample for one joint, in float like the reference: position -= clamp(effort / gain, +-0.1)
// (branch-free: a select on gain == 0, which the host stores for a gain that is not std::isnormal, torque_adjustment.cpp:52)
PB_HD float torque_adjust(float position, float effort, float gain)
{
  float a = effort / gain;
  a = a > 0.1f ? 0.1f : (a < -0.1f ? -0.1f : a);
  return gain == 0.0f ? position : position - a;
}

// body_to_foot of leg `side`; ang[j] = joint j's (torque-adjusted) position; rec(j, f) = field f of joint j's record (the
// kernels read a copy of the table their wave has put into LDS: as scalar loads from memory the 200 table reads, each
// waited for on its own, cost more than the arithmetic -- measured 6.5 us of a 17 us kernel).  The loop is unrolled over
// the table's capacity with uniform guards, so the angles are a register array.
template <class REC>
PB_HD void leg_fk(const LegChain &ch, int side, const double (&ang)[LEG_MAXJ], REC &&rec, Pose &T)
{
  pose_identity(T);
  const int n = ch.n[side];
#pragma unroll
  for (int j = 0; j < LEG_MAXJ; j++) {
    if (j < n) {  // uniform over the batch, like every branch below
      const int code = ch.code[side][j];
      double rt[3];
      if (code & LC_ORG_T) {
        const double ot[3] = { rec(j, 0), rec(j, 1), rec(j, 2) };
        quat_rot(T.q, ot, rt);
#pragma unroll
        for (int i = 0; i < 3; i++) T.t[i] += rt[i];
      }
      if (code & LC_ORG_ROT) {
        const double oq[4] = { rec(j, 3), rec(j, 4), rec(j, 5), rec(j, 6) };
        double q[4];
        quat_mul(T.q, oq, q);
#pragma unroll
        for (int i = 0; i < 4; i++) T.q[i] = q[i];
      }
      const int ty = code & LC_TYPE;
      if (ty == LJ_REVOLUTE) {
        double s, c;
        sincos_joint(0.5 * ang[j], s, c);
        if (code & LC_AXIS_NEG) s = -s;
        const int kx = (code >> LC_AXIS_SHIFT) & 3;
        const double w = T.q[0], x = T.q[1], y = T.q[2], z = T.q[3];
        // q * (c, s e_k) = c q + s (q * e_k): the product with a unit axis only permutes q's components
        if (kx == 1) { T.q[0] = fma(-s, x, c * w); T.q[1] = fma(s, w, c * x); T.q[2] = fma(s, z, c * y); T.q[3] = fma(-s, y, c * z); }
        else if (kx == 2) { T.q[0] = fma(-s, y, c * w); T.q[1] = fma(-s, z, c * x); T.q[2] = fma(s, w, c * y); T.q[3] = fma(s, x, c * z); }
        else if (kx == 3) { T.q[0] = fma(-s, z, c * w); T.q[1] = fma(s, y, c * x); T.q[2] = fma(-s, x, c * y); T.q[3] = fma(s, w, c * z); }
        else {
          const double jq[4] = { c, s * rec(j, 7), s * rec(j, 8), s * rec(j, 9) };
          double q[4];
          quat_mul(T.q, jq, q);
#pragma unroll
          for (int i = 0; i < 4; i++) T.q[i] = q[i];
        }
      } else if (ty == LJ_PRISMATIC) {
        const double d = ang[j];
        const double av[3] = { d * rec(j, 7), d * rec(j, 8), d * rec(j, 9) };
        quat_rot(T.q, av, rt);
#pragma unroll
        for (int i = 0; i < 3; i++) T.t[i] += rt[i];
      }
    }
  }
}
// The angles of one chain through angle(j) (a load, a torque adjustment ...) for ALL table slots, without a branch: the
// rows of unused and fixed slots are 0 (a valid row), so every load can be requested before the first is used -- behind
// a uniform guard per joint each load was waited for on its own, twelve memory latencies in a row.
template <class ANGLE>
PB_HD void leg_angles(const LegChain &, int, ANGLE &&angle, double (&ang)[LEG_MAXJ])
{
#pragma unroll
  for (int j = 0; j < LEG_MAXJ; j++) ang[j] = angle(j);
}
// one joint's table entry from what pb_legodo_set_chain is given (shared with the test harness): the quaternion
// urdf::Rotation::setFromRPY makes of <origin rpy> (urdfdom_headers pose.h; NOT in the tree), the axis normalised as
// KDL::Joint does; returns false for a moving joint with a zero axis
inline bool leg_chain_entry(LegChain &ch, int side, int j, int type, int row, const double *o /* xyz rpy */, const double *a, float gain)
{
  double *r = ch.rec[side][j];
  for (int i = 0; i < 3; i++) r[i] = o[i];
  const double phi = o[3] / 2.0, the = o[4] / 2.0, psi = o[5] / 2.0;
  const double q[4] = { cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi), sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi),
                        cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi), cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi) };
  const double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) r[3 + i] = q[i] / qn;
  const double an = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (type != LJ_FIXED && !(an > 0.0)) return false;
  for (int i = 0; i < 3; i++) r[7 + i] = (type == LJ_FIXED) ? 0.0 : a[i] / an;
  int code = type | ((o[3] != 0.0 || o[4] != 0.0 || o[5] != 0.0) ? LC_ORG_ROT : 0) | ((o[0] != 0.0 || o[1] != 0.0 || o[2] != 0.0) ? LC_ORG_T : 0);
  if (type == LJ_REVOLUTE)
    for (int i = 0; i < 3; i++)
      if (a[(i + 1) % 3] == 0.0 && a[(i + 2) % 3] == 0.0) code |= ((i + 1) << LC_AXIS_SHIFT) | (a[i] < 0 ? LC_AXIS_NEG : 0);
  ch.code[side][j] = code;
  ch.row[side][j] = (type == LJ_FIXED) ? 0 : row;
  ch.gain[side][j] = (type != LJ_FIXED && std::isnormal(gain)) ? gain : 0.0f;  // torque_adjustment.cpp:52
  return true;
}

// ---- contact logic ------------------------------------------------------------------------------------------------------
struct SchmittPar {
  double low, high;
  int64_t low_delay, high_delay;
};
// SchmittTrigger::UpdateState (SignalTap.cpp:83-130); dt = present_time - previous_time (0 in the first call, :84-87)
PB_HD int32_t sat_add(int32_t timer, int64_t dt)
{
  const int64_t v = (int64_t) timer + dt;
  return (int32_t) (v > 2147483647LL ? 2147483647LL : (v < -2147483647LL ? -2147483647LL : v));
}
PB_HD void schmitt_update(bool &status, int32_t &timer, const SchmittPar &p, int64_t dt, double value)
{
  if (status) {
    if (value <= p.low) {
      if (timer > p.low_delay) status = false;
      else timer = sat_add(timer, dt);
    } else {
      timer = 0;
    }
  } else {
    if (value >= p.high) {
      if (timer > p.high_delay) status = true;
      else timer = sat_add(timer, dt);
    } else {
      timer = 0;
    }
  }
}

enum { LF_UNKNOWN = -1, LF_LEFT = 0, LF_RIGHT = 1 };                                         // footid / footid_alt
enum { LC_UNKNOWN = -1, LC_LEFT_NEW = 0, LC_RIGHT_NEW = 1, LC_LEFT_FIXED = 2, LC_RIGHT_FIXED = 3 };  // contact_status_id
enum { T_ALT_L = 0, T_ALT_R, T_WEAK_L, T_WEAK_R, T_STRONG_L, T_STRONG_R };                  // the six triggers

struct LegPar {
  SchmittPar alt;                    // state_estimator.legodo.schmitt_{low,high}_threshold / _delay (leg_estimate.cpp:103-108)
  int filter_contact_events;         // state_estimator.legodo.filter_contact_events (:63)
  int standing = 0;                  // init_contact_mode == "standing": FootContact instead of FootContactAlt (:113-118,453-457)
  float total_force = 0.f, standing_schmitt_level = 0.f;  // state_estimator.legodo.{total_force, standing_schmitt_level} (:93-97)
  int use_controller_input = 0;      // state_estimator.legodo.use_controller_input (:121)
  int world_constraint = 0;          // keep world_to_primary_foot_transition_ and form world_to_body_constraint_ (:299-318, 461-492):
                                     // the pelvis position LegOdoCommon's mode pos_and_lin_rate measures
};

struct LegState {
  double body_t[3], body_q[4];       // odom_to_body_
  double prim_t[3];                  // translation of odom_to_primary_foot_fixed_
  double trans_t[3];                 // translation of world_to_primary_foot_transition_ (only with LegPar::world_constraint)
  int64_t utime, last_strike, last_break;
  int32_t timer[6];                  // T_*; in "standing" mode timer[0] is FootContact's transition_timespan
  bool status[6];
  bool started;                      // the triggers have seen a first call
  bool leg_odo_init, initialized, fc_flag;   // fc_flag: FootContact's foottransitionintermediateflag
  bool trans_init;                   // world_to_primary_foot_transition_init_
  int primary_foot, standing_foot, mode;
  int zero_ticks;                    // LegOdoHandler::zero_initial_velocity, counted per robot (rbis_legodo_update.cpp:264-268)
  int unknown_transitions;
};
static constexpr int NLD = 10, NLD_WC = 3;  // + trans_t rows, read and written only with LegPar::world_constraint
static constexpr int NLI = 3 + 3 + 1;  // utime, last_strike, last_break | three timer pairs | the flag word

PB_HD void leg_reset(LegState &s)
{
#pragma unroll
  for (int i = 0; i < 3; i++) s.body_t[i] = s.prim_t[i] = s.trans_t[i] = 0.0;
  s.body_q[0] = 1.0; s.body_q[1] = s.body_q[2] = s.body_q[3] = 0.0;
  s.utime = 0;               // current_utime_ = 0 (the first message always triggers the 30 ms reset, harmlessly)
  s.last_strike = 0; s.last_break = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) { s.timer[k] = 0; s.status[k] = false; }
  s.status[T_ALT_L] = s.status[T_ALT_R] = true;  // forceHigh (FootContactAlt.cpp:28-29)
  s.started = false;
  s.leg_odo_init = false;    // leg_estimate.cpp:128
  s.initialized = false;
  s.fc_flag = true;          // FootContact.cpp:21
  s.trans_init = false;      // leg_estimate.cpp:140
  s.zero_ticks = 0;
  s.primary_foot = LF_LEFT;  // :126 primary_foot_ = F_LEFT
  s.standing_foot = LF_LEFT; // setStandingFoot(FOOT_LEFT) / setStandingFoot(F_LEFT) (:98, :110)
  s.mode = -1;               // UNKNOWN
  s.unknown_transitions = 0;
}

// FootContactAlt::DetectFootTransition (FootContactAlt.cpp:35-100).  The reference exit(-1)s when neither foot is the standing
// one and nothing changes; here that returns LC_UNKNOWN (no odometry this tick).
PB_HD int alt_detect(LegState &s, const LegPar &p, int64_t dt, double lz, double rz)
{
  const bool l_last = s.status[T_ALT_L], r_last = s.status[T_ALT_R];
  schmitt_update(s.status[T_ALT_L], s.timer[T_ALT_L], p.alt, dt, lz);
  schmitt_update(s.status[T_ALT_R], s.timer[T_ALT_R], p.alt, dt, rz);
  const bool l = s.status[T_ALT_L], r = s.status[T_ALT_R];
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

// leg_estimate::footTransitionAlt (leg_estimate.cpp:359-393): the detector above, overruled by the controller's contact
// counts (CONTROLLER_FOOT_CONTACT, rbis_legodo_update.cpp:190-193) when use_controller_input is set.  (standing_foot_ is a
// footid_alt compared with contact_status_id constants there: F_LEFT_NEW = 0 = F_LEFT and F_RIGHT_NEW = 1 = F_RIGHT are
// the only values that can match.)
PB_HD int foot_transition_alt(LegState &s, const LegPar &p, int64_t dt, double lz, double rz, int ncl, int ncr)
{
  int cs = alt_detect(s, p, dt, lz, rz);
  if (p.use_controller_input) {
    if (s.standing_foot == LF_LEFT) {
      if (ncl > -1 && ncl < 3 && ncr >= 3) {
        cs = LC_RIGHT_NEW;
        s.status[T_ALT_L] = false; s.timer[T_ALT_L] = 0;  // forceRightStandingFoot (FootContactAlt.cpp:125-129)
        s.status[T_ALT_R] = true; s.timer[T_ALT_R] = 0;
        s.standing_foot = LF_RIGHT;
      }
    } else if (s.standing_foot == LF_RIGHT) {
      if (ncr > -1 && ncr < 3 && ncl >= 3) {
        cs = LC_LEFT_NEW;
        s.status[T_ALT_L] = true; s.timer[T_ALT_L] = 0;   // forceLeftStandingFoot (:119-123)
        s.status[T_ALT_R] = false; s.timer[T_ALT_R] = 0;
        s.standing_foot = LF_LEFT;
      }
    }
  }
  return cs;
}

// secondary - level * total_force > primary, evaluated in float with a rounding after the product like the reference's
// compiled code (FootContact.cpp:38; no fused multiply-add on its x86-64 baseline target)
PB_HD bool fc_exceeds(float secondary, float level, float total, float primary)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __fsub_rn(secondary, __fmul_rn(level, total)) > primary;
#else
  volatile float prod = level * total;
  volatile float diff = secondary - prod;
  return diff > primary;
#endif
}
// leg_estimate::footTransition (leg_estimate.cpp:322-356) around FootContact::DetectFootTransition (FootContact.cpp:29-54):
// the "standing" contact mode.  dt = utime - lcmutime, and lcmutime is the previous message's utime (0 before the first).
PB_HD int foot_transition_standing(LegState &s, const LegPar &p, int64_t dt_since_prev, float lz, float rz)
{
  const float prim = (s.standing_foot == LF_LEFT) ? lz : rz, sec = (s.standing_foot == LF_LEFT) ? rz : lz;  // :72-82
  if (fc_exceeds(sec, p.standing_schmitt_level, p.total_force, prim)) {
    s.timer[0] = sat_add(s.timer[0], dt_since_prev);
  } else {
    s.timer[0] = 0;
    s.fc_flag = true;
  }
  if (s.timer[0] > 4000 && s.fc_flag) {  // transition_timeout_ (:13)
    s.fc_flag = false;
    s.standing_foot = (s.standing_foot == LF_LEFT) ? LF_RIGHT : LF_LEFT;  // getSecondaryFoot + setStandingFoot (:326-329)
    return s.standing_foot == LF_LEFT ? LC_LEFT_NEW : LC_RIGHT_NEW;
  }
  return s.standing_foot == LF_LEFT ? LC_LEFT_FIXED : LC_RIGHT_FIXED;
}

// foot_contact_classify::updateWalkingPhase (foot_contact_classify.cpp:146-318).  Where the reference blocks on
// `cin >> blah` for a transition it does not know, the mode is kept and the event counted.
PB_HD void walking_phase(LegState &s, int64_t utime, bool lc, bool rc, bool ls, bool rs)
{
  enum { L_PRIME_R_STAND = 0, L_PRIME_R_BREAK, L_PRIME_R_SWING, L_PRIME_R_STRIKE, L_STAND_R_PRIME, L_BREAK_R_PRIME, L_SWING_R_PRIME,
         L_STRIKE_R_PRIME };
  if (!s.initialized) {
    if (lc && rc) { s.mode = L_PRIME_R_STAND; s.initialized = true; }
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
PB_HD double classify_update(LegState &s, int64_t utime, int64_t dt, double lz, double rz)
{
  const SchmittPar weak = { 20.0, 30.0, 5000, 5000 }, strong = { 275.0, 375.0, 7000, 7000 };  // :34-37
  schmitt_update(s.status[T_WEAK_L], s.timer[T_WEAK_L], weak, dt, lz);
  schmitt_update(s.status[T_WEAK_R], s.timer[T_WEAK_R], weak, dt, rz);
  schmitt_update(s.status[T_STRONG_L], s.timer[T_STRONG_L], strong, dt, lz);
  schmitt_update(s.status[T_STRONG_R], s.timer[T_STRONG_R], strong, dt, rz);
  walking_phase(s, utime, s.status[T_WEAK_L], s.status[T_WEAK_R], s.status[T_STRONG_L], s.status[T_STRONG_R]);
  const bool recent_strike = utime - s.last_strike < 95000;   // strike_blackout_duration_ (:41)
  const bool recent_break = utime - s.last_break < 800000;    // break_blackout_duration_ (:42)
  if (recent_strike) return -1.0;
  if (recent_break) return 1.0;
  return 0.0;
}

// The part of leg_estimate::updateOdometry that does not need the pelvis orientation: clock, 30 ms reset, contact
// classification, contact status (leg_estimate.cpp:398-408,447-457).  lz / rz are FootSensing::force_z, i.e. floats
// (foot_contact_classify.hpp:24-31).  Returns the contact status; classification = foot_contact_classify's verdict.
PB_HD int leg_contacts(LegState &s, const LegPar &p, int64_t utime, float lz, float rz, int ncl, int ncr, double &classification,
                       int64_t &prev_utime)
{
  prev_utime = s.utime;
  s.utime = utime;
  if ((double) (s.utime - prev_utime) * 1E-6 > 30E-3) s.leg_odo_init = false;  // :402-408
  const int64_t dt = s.started ? utime - prev_utime : 0;                       // SignalTap.cpp:84-87
  s.started = true;
  classification = classify_update(s, utime, dt, (double) lz, (double) rz);    // :449-450
  return p.standing ? foot_transition_standing(s, p, utime - prev_utime, lz, rz)  // :453-454
                    : foot_transition_alt(s, p, dt, (double) lz, (double) rz, ncl, ncr);  // :456
}

// leg_odometry_gravity_slaved_always + the increment (leg_estimate.cpp:219-297,480-551).
//   wq   rotation of world_to_body_ (setPoseBody: the filter's own head orientation, rbis_legodo_update.cpp:214-229)
//   returns the status (-1 no usable increment, 0 accurate, 1 inaccurate); delta = previous_odom_to_body^-1 * odom_to_body
// All three moving branches of the reference (initialise :172-216, fixed foot :227-243/:259-275, transition :244-258/:276-291)
// end in  odom_to_body = odom_to_primary * body_to_primary^-1  with  rotation(odom_to_primary) = wq * rotation(body_to_primary);
// they differ in the primary foot and in where odom_to_primary's translation comes from, so the pose arithmetic runs once.
//   wpos / position / position_ok: with LegPar::world_constraint, world_to_body_'s translation (the head position) in,
//   world_to_body_constraint_'s translation and world_to_body_constraint_init_ out (determine_position_constraint_slaved_always)
PB_HD double leg_integrate(LegState &s, const LegPar &p, int cs, double classification, const Pose &bl, const Pose &br,
                           const double (&wq)[4], Pose &delta, const double (&wpos)[3], double (&position)[3], bool &position_ok)
{
  const double prev_t[3] = { s.body_t[0], s.body_t[1], s.body_t[2] };
  const double prev_q[4] = { s.body_q[0], s.body_q[1], s.body_q[2], s.body_q[3] };
  enum { NONE, INIT, FIXED, SWITCH };
  int act = NONE, pf = s.primary_foot;
  if (!s.leg_odo_init) {
    if (cs == LC_LEFT_FIXED || cs == LC_RIGHT_FIXED) { act = INIT; pf = (cs == LC_LEFT_FIXED) ? LF_LEFT : LF_RIGHT; }
  } else if ((cs == LC_LEFT_FIXED && pf == LF_LEFT) || (cs == LC_RIGHT_FIXED && pf == LF_RIGHT)) {
    act = FIXED;
  } else if ((cs == LC_RIGHT_NEW && pf == LF_LEFT) || (cs == LC_LEFT_NEW && pf == LF_RIGHT)) {
    act = SWITCH;
    pf = (cs == LC_RIGHT_NEW) ? LF_RIGHT : LF_LEFT;
  }  // else: "initialized but unknown update" (:292-294): nothing moves
  if (act != NONE) {
    Pose prim;
#pragma unroll
    for (int i = 0; i < 3; i++) prim.t[i] = (pf == LF_LEFT) ? bl.t[i] : br.t[i];
#pragma unroll
    for (int i = 0; i < 4; i++) prim.q[i] = (pf == LF_LEFT) ? bl.q[i] : br.q[i];
    double rt[3], fq[4], cq[4];
    quat_rot(wq, prim.t, rt);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      // INIT: the foot at the origin (:184); FIXED: where it was; SWITCH: where forward kinematics puts it from the pelvis'
      // position and the filter's orientation (:250-253)
      s.prim_t[i] = (act == INIT) ? 0.0 : ((act == SWITCH) ? s.body_t[i] + rt[i] : s.prim_t[i]);
    }
    quat_mul(wq, prim.q, fq);          // the foot turned to agree with the pelvis orientation (:230-240)
    quat_conj(prim.q, cq);
    quat_mul(fq, cq, s.body_q);        // odom_to_body = odom_to_primary * body_to_primary^-1 (:242)
    quat_rot(s.body_q, prim.t, rt);
#pragma unroll
    for (int i = 0; i < 3; i++) s.body_t[i] = s.prim_t[i] - rt[i];
    s.primary_foot = pf;
  }
  position[0] = position[1] = position[2] = 0.0;
  position_ok = false;
  Pose pfk;  // getPrimaryFootFK(primary_foot_, ...) with the primary foot as it is AFTER the update above
  if (p.world_constraint) {
#pragma unroll
    for (int i = 0; i < 3; i++) pfk.t[i] = (s.primary_foot == LF_LEFT) ? bl.t[i] : br.t[i];
#pragma unroll
    for (int i = 0; i < 4; i++) pfk.q[i] = (s.primary_foot == LF_LEFT) ? bl.q[i] : br.q[i];
    if (cs == LC_LEFT_NEW || cs == LC_RIGHT_NEW) {  // :461-466 world_to_primary_foot_transition_ = world_to_body_ * primary FK
      double rt[3];
      quat_rot(wq, pfk.t, rt);
#pragma unroll
      for (int i = 0; i < 3; i++) s.trans_t[i] = wpos[i] + rt[i];
      s.trans_init = true;
    }
  }
  double status = -1.0;
  pose_identity(delta);
  if (act == INIT) {
    s.leg_odo_init = true;
  } else if (s.leg_odo_init) {  // :482-486
    if (p.world_constraint && s.trans_init) {  // :488-492 -> :299-318: the foot where it was at the transition, turned to
      double fq[4], cq[4], bq[4], rt[3];       // agree with the pelvis orientation; the pelvis follows from it
      quat_mul(wq, pfk.q, fq);
      quat_conj(pfk.q, cq);
      quat_mul(fq, cq, bq);
      quat_rot(bq, pfk.t, rt);
#pragma unroll
      for (int i = 0; i < 3; i++) position[i] = s.trans_t[i] - rt[i];
      position_ok = true;
    }
    double cq[4];
    quat_conj(prev_q, cq);
    const double d[3] = { s.body_t[0] - prev_t[0], s.body_t[1] - prev_t[1], s.body_t[2] - prev_t[2] };
    quat_rot(cq, d, delta.t);
    quat_mul(cq, s.body_q, delta.q);
    status = 0.0;
  }
  if (p.filter_contact_events && status > -1.0) status = classification;  // :545-551
  return status;
}

// leg_estimate::updateOdometry (leg_estimate.cpp:395-556) from the two body-to-foot transforms
PB_HD double leg_update(LegState &s, const LegPar &p, int64_t utime, const Pose &bl, const Pose &br, float lz, float rz, int ncl, int ncr,
                        const double (&wq)[4], Pose &delta, int64_t &prev_utime, const double (&wpos)[3], double (&position)[3],
                        bool &position_ok)
{
  double classification;
  const int cs = leg_contacts(s, p, utime, lz, rz, ncl, ncr, classification, prev_utime);
  return leg_integrate(s, p, cs, classification, bl, br, wq, delta, wpos, position, position_ok);
}
PB_HD double leg_update(LegState &s, const LegPar &p, int64_t utime, const Pose &bl, const Pose &br, float lz, float rz, int ncl, int ncr,
                        const double (&wq)[4], Pose &delta, int64_t &prev_utime)
{
  const double wpos[3] = { 0.0, 0.0, 0.0 };
  double position[3];
  bool ok;
  return leg_update(s, p, utime, bl, br, lz, rz, ncl, ncr, wq, delta, prev_utime, wpos, position, ok);
}
// "Ignore the calculated velocity at launch" (rbis_legodo_update.cpp:264-268: `zero_initial_velocity--; if (... > 0)`), which
// the reference only reaches for a valid status (:243-255 return NULL first); counted per robot because validity is per robot
PB_HD bool leg_zero_velocity(LegState &s, double status)
{
  if (status >= 0.0 && s.zero_ticks > 0) {
    s.zero_ticks--;
    return s.zero_ticks > 0;
  }
  return false;
}

// every flag, enum and counter of LegState in one word: bit 0 leg_odo_init, 1-2 primary_foot + 1, 3-8 the six triggers'
// status, 9 started, 10 fc_flag, 11 trans_init, 15-16 standing_foot + 1, 17-24 mode + 1, 25 initialized, 32-47 unknown_transitions
// (saturating), 48-63 zero_ticks
PB_HD int64_t leg_pack_flags(const LegState &s)
{
  uint64_t w = (uint64_t) s.leg_odo_init | ((uint64_t) (s.primary_foot + 1) & 3u) << 1;
#pragma unroll
  for (int k = 0; k < 6; k++) w |= (uint64_t) s.status[k] << (3 + k);
  w |= (uint64_t) s.started << 9 | (uint64_t) s.fc_flag << 10 | (uint64_t) s.trans_init << 11;
  w |= ((uint64_t) (s.standing_foot + 1) & 3u) << 15 | ((uint64_t) (s.mode + 1) & 255u) << 17 | (uint64_t) s.initialized << 25;
  w |= (uint64_t) (s.unknown_transitions > 65535 ? 65535 : s.unknown_transitions) << 32 | ((uint64_t) s.zero_ticks & 0xFFFFu) << 48;
  return (int64_t) w;
}
PB_HD void leg_unpack_flags(LegState &s, int64_t word)
{
  const uint64_t w = (uint64_t) word;
  s.leg_odo_init = (w & 1u) != 0;
  s.primary_foot = (int) ((w >> 1) & 3u) - 1;
#pragma unroll
  for (int k = 0; k < 6; k++) s.status[k] = ((w >> (3 + k)) & 1u) != 0;
  s.started = ((w >> 9) & 1u) != 0;
  s.fc_flag = ((w >> 10) & 1u) != 0;
  s.trans_init = ((w >> 11) & 1u) != 0;
  s.standing_foot = (int) ((w >> 15) & 3u) - 1;
  s.mode = (int) ((w >> 17) & 255u) - 1;
  s.initialized = ((w >> 25) & 1u) != 0;
  s.unknown_transitions = (int) ((w >> 32) & 0xFFFFu);
  s.zero_ticks = (int) (w >> 48);
}

// SoA <-> struct (robot index fastest; d: [NLD][stride] doubles, iw: [NLI][stride] 64-bit words)
PB_HD void leg_load(LegState &s, const double *d, const int64_t *iw, long stride, long b, bool world_constraint = false)
{
  if (world_constraint) {
#pragma unroll
    for (int i = 0; i < 3; i++) s.trans_t[i] = d[(long) (NLD + i) * stride + b];
  } else {
    s.trans_t[0] = s.trans_t[1] = s.trans_t[2] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) s.body_t[i] = d[(long) i * stride + b];
#pragma unroll
  for (int i = 0; i < 4; i++) s.body_q[i] = d[(long) (3 + i) * stride + b];
#pragma unroll
  for (int i = 0; i < 3; i++) s.prim_t[i] = d[(long) (7 + i) * stride + b];
  s.utime = iw[b];
  s.last_strike = iw[stride + b];
  s.last_break = iw[2 * stride + b];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint64_t w = (uint64_t) iw[(long) (3 + k) * stride + b];
    s.timer[2 * k] = (int32_t) (uint32_t) (w & 0xFFFFFFFFu);
    s.timer[2 * k + 1] = (int32_t) (uint32_t) (w >> 32);
  }
  leg_unpack_flags(s, iw[6 * stride + b]);
}
PB_HD void leg_store(const LegState &s, double *d, int64_t *iw, long stride, long b, bool world_constraint = false)
{
  if (world_constraint) {
#pragma unroll
    for (int i = 0; i < 3; i++) d[(long) (NLD + i) * stride + b] = s.trans_t[i];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) d[(long) i * stride + b] = s.body_t[i];
#pragma unroll
  for (int i = 0; i < 4; i++) d[(long) (3 + i) * stride + b] = s.body_q[i];
#pragma unroll
  for (int i = 0; i < 3; i++) d[(long) (7 + i) * stride + b] = s.prim_t[i];
  iw[b] = s.utime;
  iw[stride + b] = s.last_strike;
  iw[2 * stride + b] = s.last_break;
#pragma unroll
  for (int k = 0; k < 3; k++)
    iw[(long) (3 + k) * stride + b] = (int64_t) ((uint64_t) (uint32_t) s.timer[2 * k] | (uint64_t) (uint32_t) s.timer[2 * k + 1] << 32);
  iw[6 * stride + b] = leg_pack_flags(s);
}

// ---- one message's inputs ----------------------------------------------------------------------------------------------
// Either the two body-to-foot transforms (kind 0: feet [14][B] = left (t3, q4), right (t3, q4), forces [2][B], doubles) or the
// joint state itself (kind 1: joint positions [rows][B], optionally efforts [rows][B] for the torque adjustment, forces
// [2][B], floats like bot_core::joint_state_t / six_axis_force_torque_t carry them), per filter in device memory -- or ONE
// robot's message for every filter (bcast: a parameter sweep over one log): v = feet[14] + forces[2] as kernel arguments (a
// broadcast JOINT state is reduced to the two foot transforms by the host side of the ABI: they are the same for every
// filter).
struct LegIn {
  int kind = 0, bcast = 0;
  const double *feet = nullptr, *forces = nullptr;
  const float *jpos = nullptr, *jeff = nullptr, *jforces = nullptr;
  // independent log segments: every filter has its OWN message, with its own time stamp -- utimes [B] (device) replaces the
  // call's scalar utime per filter -- and some filters may have none (their segment has ended): valid [B] (device), 0 = leave
  // this robot's state alone and mask its measurement.  NULL = one time for all / all valid.
  const int64_t *utimes = nullptr;
  const uint8_t *valid = nullptr;
  const int32_t *ncontacts = nullptr;  // controller contact counts [2][B] (device) or NULL: nc[] for every filter
  int nc[2] = { -1, -1 };              // (-1: no CONTROLLER_FOOT_CONTACT message yet, rbis_legodo_update.cpp:100-101)
  double v[16] = { 0 };
};

// chain: the table of pb_legodo_set_chain.  The kernels take it as a `const LegChain *__restrict__` KERNEL PARAMETER of its own
// and read it with compile-time offsets: a read-only, no-alias, wave-uniform address, i.e. scalar loads whose results are SGPR
// operands.  (Through a pointer inside a struct the same reads were per-lane vector loads; staged into LDS, flat loads with a
// wait each; passed BY VALUE inside LegIn -- 1.5 KB of kernel arguments -- every launch of every kernel that takes a LegIn got
// 6 us slower, measured on one box against the same code with the pointer: 24.4 -> 30.6 us for the 15-state foot-pose pair.)
PB_HD void leg_inputs(const LegIn &in, const LegChain *chain, long b, long B, Pose &bl, Pose &br, float &fl, float &fr, int &ncl, int &ncr)
{
  if (in.kind == 0) {
    if (in.bcast) {
#pragma unroll
      for (int i = 0; i < 3; i++) { bl.t[i] = in.v[i]; br.t[i] = in.v[7 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) { bl.q[i] = in.v[3 + i]; br.q[i] = in.v[10 + i]; }
      fl = (float) in.v[14]; fr = (float) in.v[15];
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) { bl.t[i] = in.feet[(long) i * B + b]; br.t[i] = in.feet[(long) (7 + i) * B + b]; }
#pragma unroll
      for (int i = 0; i < 4; i++) { bl.q[i] = in.feet[(long) (3 + i) * B + b]; br.q[i] = in.feet[(long) (10 + i) * B + b]; }
      fl = (float) in.forces[b]; fr = (float) in.forces[B + b];
    }
  } else {
    const LegChain &ch = *chain;
    double al[LEG_MAXJ], ar[LEG_MAXJ];
    if (in.jeff != nullptr) {  // (uniform) with the torque adjustment: positions and efforts of both chains requested together
      float pl[LEG_MAXJ], pr[LEG_MAXJ], el[LEG_MAXJ], er[LEG_MAXJ];
#pragma unroll
      for (int j = 0; j < LEG_MAXJ; j++) {
        const long atl = (long) ch.row[0][j] * B + b, atr = (long) ch.row[1][j] * B + b;
        pl[j] = in.jpos[atl]; pr[j] = in.jpos[atr]; el[j] = in.jeff[atl]; er[j] = in.jeff[atr];
      }
      leg_angles(ch, 0, [&](int j) { return (double) torque_adjust(pl[j], el[j], ch.gain[0][j]); }, al);
      leg_angles(ch, 1, [&](int j) { return (double) torque_adjust(pr[j], er[j], ch.gain[1][j]); }, ar);
    } else {
      leg_angles(ch, 0, [&](int j) { return (double) in.jpos[(long) ch.row[0][j] * B + b]; }, al);
      leg_angles(ch, 1, [&](int j) { return (double) in.jpos[(long) ch.row[1][j] * B + b]; }, ar);
    }
    fl = in.jforces[b]; fr = in.jforces[B + b];
    leg_fk(ch, 0, al, [&ch](int j, int f) { return ch.rec[0][j][f]; }, bl);
    leg_fk(ch, 1, ar, [&ch](int j, int f) { return ch.rec[1][j][f]; }, br);
  }
  if (in.ncontacts != nullptr) { ncl = in.ncontacts[b]; ncr = in.ncontacts[B + b]; }
  else { ncl = in.nc[0]; ncr = in.nc[1]; }
}

// One leg's forward kinematics from per-filter joint blocks (kind 1), and the rest of the message (forces, controller
// contacts) -- the pieces of leg_inputs for the kernels that give the two legs to two different waves of a tile
PB_HD void leg_fk_side(const LegIn &in, const LegChain *chain, int side, long b, long B, Pose &T)
{
  const LegChain &ch = *chain;
  double ang[LEG_MAXJ];
  if (in.jeff != nullptr) {
    float p[LEG_MAXJ], e[LEG_MAXJ];
#pragma unroll
    for (int j = 0; j < LEG_MAXJ; j++) {
      const long at = (long) ch.row[side][j] * B + b;
      p[j] = in.jpos[at]; e[j] = in.jeff[at];
    }
    leg_angles(ch, side, [&](int j) { return (double) torque_adjust(p[j], e[j], ch.gain[side][j]); }, ang);
  } else {
    leg_angles(ch, side, [&](int j) { return (double) in.jpos[(long) ch.row[side][j] * B + b]; }, ang);
  }
  leg_fk(ch, side, ang, [&ch, side](int j, int f) { return ch.rec[side][j][f]; }, T);
}
PB_HD void leg_inputs_rest(const LegIn &in, long b, long B, float &fl, float &fr, int &ncl, int &ncr)
{
  fl = in.jforces[b]; fr = in.jforces[B + b];
  if (in.ncontacts != nullptr) { ncl = in.ncontacts[b]; ncr = in.ncontacts[B + b]; }
  else { ncl = in.nc[0]; ncr = in.nc[1]; }
}

// LegOdoCommon::createMeasurement in mode lin_rate on the odometry's result (rbis_legodo_common.cpp:99-107,124-129,153-156,
// pronto_conversions_lcm.hpp:38-87): z = delta translation / elapsed time, R = r_vxyz^2 or r_vxyz_uncertain^2 when
// status >= 0.5, no update (mask 0) when status < 0
struct LegMeas {
  double z[3], r;
  bool valid;
};
PB_HD void leg_measurement(const Pose &delta, double status, int64_t utime, int64_t prev_utime, double r2, double r2_uncertain, LegMeas &m)
{
  const double elapsed = (double) (utime - prev_utime) * 1E-6;
#pragma unroll
  for (int i = 0; i < 3; i++) m.z[i] = delta.t[i] / elapsed;
  m.r = (status >= 0.5) ? r2_uncertain : r2;
  m.valid = !(status < 0);
}

// The same for LegOdoCommon's two six-row modes (rbis_legodo_common.cpp:131-165), formed where the increment is: in the
// reference createMeasurement runs on the host right behind updateOdometry; a batch would have to copy the increment, the status
// and the position of every filter over PCIe to do that.
//   mode 1  lin_rot_rate      idx {3,4,5,0,1,2}    z = (v, rpy(delta rotation) / elapsed)   R = (r_vxyz^2 x3, r_vang^2 x3), both
//                                                  "uncertain" values when status >= 0.5
//   mode 2  pos_and_lin_rate  idx {9,10,11,3,4,5}  z = (pelvis position, v)                 R = (r_xyz^2 x3, r_vxyz^2 x3)
//           per filter: position valid -> this update (valid6); else the reference falls back to lin_rate for this message
//           (:118-122) -> the three-row update on rows 3..5 of the same block (valid3).  Exactly one of the two per valid filter.
struct LegMeasPar {
  int mode = 0;
  double r_v2 = 0, r_v2_uncertain = 0, r_xyz2 = 0, r_a2 = 0, r_a2_uncertain = 0;
};
struct LegMeas6 {
  double z[6], r[6];
  bool valid6, valid3;
};
// bot_quat_to_roll_pitch_yaw (libbot, NOT IN TREE; the same formula as pronto_math.cpp:53-61 quat_to_euler)
PB_HD void quat_to_rpy(const double (&q)[4], double (&rpy)[3])
{
  rpy[0] = atan2(2.0 * (q[0] * q[1] + q[2] * q[3]), 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]));
  rpy[1] = asin(2.0 * (q[0] * q[2] - q[3] * q[1]));
  rpy[2] = atan2(2.0 * (q[0] * q[3] + q[1] * q[2]), 1.0 - 2.0 * (q[2] * q[2] + q[3] * q[3]));
}
PB_HD void leg_measurement6(const Pose &delta, double status, const double (&position)[3], bool position_ok, int64_t utime, int64_t prev_utime,
                            const LegMeasPar &mp, LegMeas6 &m)
{
  const double elapsed = (double) (utime - prev_utime) * 1E-6;   // getDeltaAsVelocity (pronto_conversions_lcm.hpp:45)
  const bool valid = !(status < 0), uncertain = status >= 0.5;
  const double rv = uncertain ? mp.r_v2_uncertain : mp.r_v2;
  double vel[3];
#pragma unroll
  for (int i = 0; i < 3; i++) vel[i] = delta.t[i] / elapsed;
  if (mp.mode == 1) {  // (wave-uniform)
    double rpy[3];
    quat_to_rpy(delta.q, rpy);
    const double el2 = ((double) utime - (double) prev_utime) / 1000000;   // (:143: its own elapsed_time)
    const double ra = uncertain ? mp.r_a2_uncertain : mp.r_a2;
#pragma unroll
    for (int i = 0; i < 3; i++) { m.z[i] = vel[i]; m.z[3 + i] = rpy[i] / el2; m.r[i] = rv; m.r[3 + i] = ra; }
    m.valid6 = valid;
    m.valid3 = false;
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) { m.z[i] = position[i]; m.z[3 + i] = vel[i]; m.r[i] = mp.r_xyz2; m.r[3 + i] = rv; }
    m.valid6 = valid && position_ok;
    m.valid3 = valid && !position_ok;
  }
}

#if defined(__HIPCC__)
// One robot per lane: the odometry on its state, with the filter's own head orientation as world_to_body_ (setPoseBody,
// rbis_legodo_update.cpp:214-229), then -- optionally -- the lin_rate measurement of it.
// AHEAD: the odometry is slaved to the orientation the filter WILL have after the IMU step in `imu` / imu_bc
// (rbis_update_interface.cpp:30-52 applied to the head), computed here from the head state with the step kernels' own
// ins_update_state -- the covariance is not touched.  That lets the estimator run the IMU step and the leg-odometry update
// it produces as ONE fused kernel afterwards instead of predict, odometry, update.
// (launch bounds: without them the compiler budgets for 1024-thread blocks, 128 registers, and spills)
struct LegAhead {
  int on = 0, bcast = 0;
  const double *imu = nullptr;  // [7][B]
  double v[7] = { 0, 0, 0, 0, 0, 0, 0 };
};
// SPLIT (per-filter joint blocks, 128-thread workgroups): a second wave runs the RIGHT leg's forward kinematics for the same 64
// robots and hands the foot pose over through LDS -- half of the kinematics leaves the one wave's dependent chain.
template <int NS, bool SPLIT = false>
static __global__ __launch_bounds__(SPLIT ? 128 : 64, 2) void k_legodo(const double *__restrict__ st, double *__restrict__ legd,
                                                         int64_t *__restrict__ legi, long stride, int B, int64_t utime, LegPar par,
                                                         LegIn in, const LegChain *__restrict__ chain, LegAhead ah, int zero_delta, LegMeasPar mp,
                                                         double *__restrict__ delta_out, double *__restrict__ status_out,
                                                         double *__restrict__ lo_out, uint8_t *__restrict__ mask_out,
                                                         double *__restrict__ pos_out, uint8_t *__restrict__ pos_ok_out, Consts k)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  __shared__ double foot_r[SPLIT ? 7 : 1][64];
  const long b_raw = (long) blockIdx.x * 64 + (threadIdx.x & 63u);
  const bool live = b_raw < B;
  if (!SPLIT && !live) return;
  const long b = live ? b_raw : (long) B - 1;   // (SPLIT: no lane returns before the barrier; lanes past the batch read the last robot)
  if constexpr (SPLIT) {
    if (threadIdx.x >= 64) {   // the helper wave
      Pose T;
      leg_fk_side(in, chain, 1, b, (long) B, T);
#pragma unroll
      for (int i = 0; i < 3; i++) foot_r[i][threadIdx.x & 63u] = T.t[i];
#pragma unroll
      for (int i = 0; i < 4; i++) foot_r[3 + i][threadIdx.x & 63u] = T.q[i];
      __syncthreads();
      return;
    }
  }
  LegState s;
  leg_load(s, legd, legi, stride, b, par.world_constraint != 0);
  Pose bl, br, delta;
  float fl, fr;
  int ncl, ncr;
  double wq[4], wpos[3] = { 0.0, 0.0, 0.0 };
  for (int i = 0; i < 4; i++) wq[i] = st[S::eidx(L::OFF_QUAT + i, b)];
  if (par.world_constraint && !ah.on)
    for (int i = 0; i < 3; i++) wpos[i] = st[S::eidx(L::OFF_VEC + 9 + i, b)];
  if (ah.on) {
    double gyro[3], accel[3], dt;
    if (ah.bcast) {
      for (int i = 0; i < 3; i++) { gyro[i] = ah.v[i]; accel[i] = ah.v[3 + i]; }
      dt = ah.v[6];
    } else {
      for (int i = 0; i < 3; i++) { gyro[i] = ah.imu[(long) i * B + b]; accel[i] = ah.imu[(long) (3 + i) * B + b]; }
      dt = ah.imu[(long) 6 * B + b];
    }
    if (par.world_constraint) {  // the pose after the IMU step: the whole state propagate
      double x[NS];
#pragma unroll
      for (int i = 0; i < NS; i++) x[i] = st[S::eidx(L::OFF_VEC + i, b)];
      ins_update_state<NS>(x, wq, gyro, accel, dt, k);
      for (int i = 0; i < 3; i++) wpos[i] = x[9 + i];
    } else {                     // only the orientation after it (bit-identical to the above)
      double chi[3], bg[3] = { 0.0, 0.0, 0.0 };
      for (int i = 0; i < 3; i++) chi[i] = st[S::eidx(L::OFF_VEC + 6 + i, b)];
      if (NS == 21)
        for (int i = 0; i < 3; i++) bg[i] = st[S::eidx(L::OFF_VEC + 15 + i, b)];
      ins_update_quat<NS>(chi, bg, wq, gyro, dt, k);
    }
  }
  if constexpr (SPLIT) {
    leg_fk_side(in, chain, 0, b, (long) B, bl);
    leg_inputs_rest(in, b, (long) B, fl, fr, ncl, ncr);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; i++) br.t[i] = foot_r[i][threadIdx.x & 63u];
#pragma unroll
    for (int i = 0; i < 4; i++) br.q[i] = foot_r[3 + i][threadIdx.x & 63u];
    if (!live) return;
  } else {
    leg_inputs(in, chain, b, B, bl, br, fl, fr, ncl, ncr);
  }
  if (in.utimes != nullptr) utime = in.utimes[b];                 // this filter's own message time (independent segments)
  const bool msg_ok = in.valid == nullptr || in.valid[b] != 0;    // ... or no message at all for it
  int64_t prev = 0;
  double position[3];
  bool position_ok;
  double status = leg_update(s, par, utime, bl, br, fl, fr, ncl, ncr, wq, delta, prev, wpos, position, position_ok);
  const bool zero = leg_zero_velocity(s, status) || zero_delta != 0;
  if (msg_ok) leg_store(s, legd, legi, stride, b, par.world_constraint != 0);
  else status = -1.0;
  if (zero) {  // odo_delta.setIdentity(); odo_position.setIdentity() (rbis_legodo_update.cpp:266-267)
    pose_identity(delta);
    position[0] = position[1] = position[2] = 0.0;
  }
  if (pos_out != nullptr) {
    for (int i = 0; i < 3; i++) pos_out[(long) i * B + b] = position[i];
    if (pos_ok_out != nullptr) pos_ok_out[b] = position_ok ? 1 : 0;
  }
  if (delta_out != nullptr) {
    for (int i = 0; i < 3; i++) delta_out[(long) i * B + b] = delta.t[i];
    for (int i = 0; i < 4; i++) delta_out[(long) (3 + i) * B + b] = delta.q[i];
  }
  if (status_out != nullptr) status_out[b] = status;
  if (lo_out != nullptr && mp.mode == 0) {
    LegMeas m;
    leg_measurement(delta, status, utime, prev, mp.r_v2, mp.r_v2_uncertain, m);
    for (int i = 0; i < 3; i++) {
      lo_out[(long) i * B + b] = m.z[i];
      lo_out[(long) (3 + i) * B + b] = m.r;
    }
    if (mask_out != nullptr) mask_out[b] = m.valid ? 1 : 0;
  } else if (lo_out != nullptr) {  // the six-row modes: z [6][B] | R diagonal [6][B]; masks [B] (six rows) | [B] (mode 2's lin_rate fall-back)
    LegMeas6 m;
    leg_measurement6(delta, status, position, position_ok, utime, prev, mp, m);
    for (int i = 0; i < 6; i++) {
      lo_out[(long) i * B + b] = m.z[i];
      lo_out[(long) (6 + i) * B + b] = m.r[i];
    }
    if (mask_out != nullptr) {
      mask_out[b] = m.valid6 ? 1 : 0;
      if (mp.mode == 2) mask_out[(long) B + b] = m.valid3 ? 1 : 0;
    }
  }
}
// forward kinematics alone: feet_out [14][B] (diagnostics, tests)
static __global__ __launch_bounds__(64) void k_leg_fk(LegIn in, const LegChain *__restrict__ chain, int B, double *__restrict__ feet_out)
{
  const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Pose bl, br;
  float fl, fr;
  int ncl, ncr;
  leg_inputs(in, chain, b, B, bl, br, fl, fr, ncl, ncr);
  for (int i = 0; i < 3; i++) { feet_out[(long) i * B + b] = bl.t[i]; feet_out[(long) (7 + i) * B + b] = br.t[i]; }
  for (int i = 0; i < 4; i++) { feet_out[(long) (3 + i) * B + b] = bl.q[i]; feet_out[(long) (10 + i) * B + b] = br.q[i]; }
}
// zero_ticks < 0: reset everything; otherwise only set the per-robot zero_initial_velocity counter
static __global__ void k_legodo_reset(double *legd, int64_t *legi, long stride, int B, int zero_ticks)
{
  const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  LegState s;
  if (zero_ticks < 0) leg_reset(s);
  else {
    leg_load(s, legd, legi, stride, b, true);
    s.zero_ticks = zero_ticks > 65535 ? 65535 : zero_ticks;
  }
  leg_store(s, legd, legi, stride, b, true);
}
static __global__ void k_legodo_get(const double *legd, const int64_t *legi, long stride, long b, double *pose7, int64_t *info)
{
  LegState s;
  leg_load(s, legd, legi, stride, b);
  for (int i = 0; i < 3; i++) pose7[i] = s.body_t[i];
  for (int i = 0; i < 4; i++) pose7[3 + i] = s.body_q[i];
  info[0] = s.primary_foot; info[1] = s.leg_odo_init; info[2] = s.mode; info[3] = s.unknown_transitions;
}
#endif

}  // namespace pb
