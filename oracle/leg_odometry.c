/* leg_odometry.c -- CPU restatement (TEST INFRASTRUCTURE, parity unpinned like the rest of oracle/) of the reference's leg
 * kinematic odometry, written with Eigen::Isometry3d-style 3x3 rotation matrices like the reference, i.e. NOT with the
 * quaternion arithmetic of pronto_amd/csrc/rbis_legodo.hpp it checks:
 *   motion_estimate/src/leg_estimate/leg_estimate.cpp:172-297,395-556
 *   motion_estimate/src/foot_contact_alt/FootContactAlt.cpp:5-100
 *   motion_estimate/src/leg_estimate/foot_contact_classify.cpp:5-125,146-318
 *   estimate_tools/src/filter_tools/SignalTap.cpp:48-134
 *   motion_estimate/src/leg_estimate/leg_estimate.cpp:322-393 (footTransition / footTransitionAlt incl. controller input)
 *   motion_estimate/src/foot_contact/FootContact.cpp:5-83 (the "standing" contact mode, float arithmetic)
 *   estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp:27-62
 * Inputs are the body-to-foot transforms, or the joint angles: po_fk restates the forward kinematics of
 * leg_estimate.cpp:430-447.  That code calls THIRD-PARTY libraries that are not in /root/reference and not in this image:
 * orocos KDL (TreeFkSolverPosFull_recursive, Segment::pose, Joint::pose, Rotation::Rot2 / GetQuaternion / Quaternion),
 * kdl_parser (treeFromString) and urdfdom (Rotation::setFromRPY), the versions ROS Indigo pinned for the 2014-15 Atlas
 * stack (orocos_kdl 1.3, urdfdom_headers 0.3).  Their published algorithms are restated from memory with 3x3 matrices --
 * unpinned like everything else in oracle/. */
#include <math.h>
#include <string.h>

#include "pronto_oracle.h"

/* ---- SignalTap.cpp:64-130 ---- */
void po_schmitt_reset(po_schmitt *s) { s->status = 0; s->previous_time = 0; s->timer = 0; s->first_call = 1; }
void po_schmitt_update(po_schmitt *s, double lt, double ht, long low_delay, long high_delay, long present_time, double value)
{
  if (s->first_call) { s->first_call = 0; s->previous_time = present_time; }
  if (s->status) {
    if (value <= lt) {
      if (s->timer > low_delay) s->status = 0;
      else s->timer += (present_time - s->previous_time);
    } else s->timer = 0;
  } else {
    if (value >= ht) {
      if (s->timer > high_delay) s->status = 1;
      else s->timer += (present_time - s->previous_time);
    } else s->timer = 0;
  }
  s->previous_time = present_time;
}

/* ---- Isometry3d ---- */
typedef struct { double R[9], t[3]; } iso;
static void iso_identity(iso *a) { memset(a, 0, sizeof *a); a->R[0] = a->R[4] = a->R[8] = 1.0; }
static void iso_mul(const iso *a, const iso *b, iso *o)
{
  iso r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.R[3 * i + j] = a->R[3 * i] * b->R[j] + a->R[3 * i + 1] * b->R[3 + j] + a->R[3 * i + 2] * b->R[6 + j];
    r.t[i] = a->R[3 * i] * b->t[0] + a->R[3 * i + 1] * b->t[1] + a->R[3 * i + 2] * b->t[2] + a->t[i];
  }
  *o = r;
}
static void iso_inv(const iso *a, iso *o)
{
  iso r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.R[3 * i + j] = a->R[3 * j + i];
  for (int i = 0; i < 3; i++) r.t[i] = -(r.R[3 * i] * a->t[0] + r.R[3 * i + 1] * a->t[1] + r.R[3 * i + 2] * a->t[2]);
  *o = r;
}
static void iso_from_tq(const double *t, const double *q, iso *o)
{
  po_quat_to_rot(q, o->R);
  memcpy(o->t, t, sizeof(double) * 3);
}
/* Eigen::Quaterniond(const Matrix3d&) [Eigen NOT IN TREE; restated from its documented algorithm: trace branch, else the
 * largest diagonal element] */
static void quat_from_rot(const double *R, double *q)
{
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}
/* X.setIdentity(); X.translation() = trans; X.rotate(Quaterniond(Rw * Rfoot))  (leg_estimate.cpp:230-240) */
static void slaved_foot(const double *trans, const double *Rw, const iso *body_to_foot, iso *o)
{
  iso w, wf;
  iso_identity(&w);
  memcpy(w.R, Rw, sizeof w.R);
  iso_mul(&w, body_to_foot, &wf);
  double q[4];
  quat_from_rot(wf.R, q);
  po_quat_to_rot(q, o->R);
  memcpy(o->t, trans, sizeof(double) * 3);
}

struct po_leg {
  iso odom_to_body, odom_to_primary, odom_to_secondary;
  long current_utime, previous_utime;
  int leg_odo_init, primary_foot;
  /* FootContactAlt */
  po_schmitt alt_l, alt_r;
  double alt_lt, alt_ht;
  long alt_ld, alt_hd;
  int standing_foot;
  /* foot_contact_classify */
  po_schmitt weak_l, weak_r, strong_l, strong_r;
  int mode, initialized;
  long last_strike, last_break;
  int filter_contact_events, unknown_transitions;
  /* leg_estimate.cpp:113-121: contact mode and controller input */
  int standing_mode, use_controller_input;
  /* FootContact (foot_contact/FootContact.h:22-36) */
  float fc_schmitt_level, fc_total_force;
  long fc_transition_timespan, fc_lcmutime;
  int fc_standing_foot, fc_flag;
  /* leg_estimate.hpp:193-215: the world-frame constraint */
  iso world_to_primary_foot_transition, world_to_body_constraint;
  int world_to_primary_foot_transition_init, world_to_body_constraint_init;
};
size_t po_leg_sizeof(void) { return sizeof(struct po_leg); }

void po_leg_init(po_leg *s, double schmitt_low, double schmitt_high, long low_delay, long high_delay, int filter_contact_events)
{
  memset(s, 0, sizeof *s);
  iso_identity(&s->odom_to_body);
  iso_identity(&s->odom_to_primary);
  iso_identity(&s->odom_to_secondary);
  s->primary_foot = 0;  /* F_LEFT (leg_estimate.cpp:126) */
  /* `float schmitt_low_threshold = bot_param_get_double_or_fail(...)` (leg_estimate.cpp:103-104) */
  s->alt_lt = (float) schmitt_low; s->alt_ht = (float) schmitt_high; s->alt_ld = low_delay; s->alt_hd = high_delay;
  po_schmitt_reset(&s->alt_l); po_schmitt_reset(&s->alt_r);
  s->alt_l.status = 1; s->alt_r.status = 1;  /* forceHigh (FootContactAlt.cpp:28-29) */
  s->standing_foot = 0;  /* foot_contact_logic_alt_->setStandingFoot(F_LEFT) (leg_estimate.cpp:110) */
  s->fc_standing_foot = 0;  /* foot_contact_logic_->setStandingFoot(FOOT_LEFT) (:98) */
  s->fc_flag = 1;           /* foottransitionintermediateflag (FootContact.cpp:21) */
  po_schmitt_reset(&s->weak_l); po_schmitt_reset(&s->weak_r); po_schmitt_reset(&s->strong_l); po_schmitt_reset(&s->strong_r);
  s->mode = -1;
  s->filter_contact_events = filter_contact_events;
}

void po_leg_set_contact_mode(po_leg *s, int standing, double total_force, double standing_schmitt_level, int use_controller_input)
{
  s->standing_mode = standing;
  s->fc_total_force = (float) total_force;            /* `float total_force = bot_param_get_double_or_fail` (leg_estimate.cpp:93-95) */
  s->fc_schmitt_level = (float) standing_schmitt_level;
  s->use_controller_input = use_controller_input;
}

/* FootContact::DetectFootTransition (FootContact.cpp:29-54): returns the new standing foot or -1 */
static int fc_detect_foot_transition(po_leg *s, long utime, float leftz, float rightz)
{
  const long deltautime = utime - s->fc_lcmutime;
  s->fc_lcmutime = utime;
  const float prim = s->fc_standing_foot == 0 ? leftz : rightz, sec = s->fc_standing_foot == 0 ? rightz : leftz;  /* :72-82 */
  /* `getSecondaryFootZforce() - schmitt_level_*total_force_ > getPrimaryFootZforce()` in float, no contraction */
  volatile float prod = s->fc_schmitt_level * s->fc_total_force;
  volatile float diff = sec - prod;
  if (diff > prim) s->fc_transition_timespan += deltautime;
  else { s->fc_transition_timespan = 0; s->fc_flag = 1; }
  if (s->fc_transition_timespan > 4000 && s->fc_flag) {
    s->fc_flag = 0;
    return s->fc_standing_foot == 0 ? 1 : (s->fc_standing_foot == 1 ? 0 : -1);  /* getSecondaryFoot */
  }
  return -1;
}
/* leg_estimate::footTransition (leg_estimate.cpp:322-356) */
static int foot_transition(po_leg *s, long utime, float leftz, float rightz)
{
  const int newstep = fc_detect_foot_transition(s, utime, leftz, rightz);
  if (newstep == 0 || newstep == 1) s->fc_standing_foot = newstep;
  int contact_status;
  if (newstep != -1) contact_status = s->fc_standing_foot == 0 ? 0 : 1;      /* F_LEFT_NEW / F_RIGHT_NEW */
  else contact_status = s->fc_standing_foot == 0 ? 2 : 3;                    /* F_LEFT_FIXED / F_RIGHT_FIXED */
  return contact_status;
}

/* FootContactAlt.cpp:35-100; -1 where the reference exits ("Situation unknown") */
static int detect_foot_transition(po_leg *s, long utime, double leftz, double rightz)
{
  const int lf_last = s->alt_l.status, rf_last = s->alt_r.status;
  po_schmitt_update(&s->alt_l, s->alt_lt, s->alt_ht, s->alt_ld, s->alt_hd, utime, leftz);
  po_schmitt_update(&s->alt_r, s->alt_lt, s->alt_ht, s->alt_ld, s->alt_hd, utime, rightz);
  const int lf = s->alt_l.status, rf = s->alt_r.status;
  if (!lf_last && lf) { s->standing_foot = 0; return 0; }
  else if (!rf_last && rf) { s->standing_foot = 1; return 1; }
  else if (lf_last && !lf) {
    if (s->standing_foot == 0) { s->standing_foot = 1; return 1; }
    return 3;
  } else if (rf_last && !rf) {
    if (s->standing_foot == 1) { s->standing_foot = 0; return 0; }
    return 2;
  } else {
    if (s->standing_foot == 0) return 2;
    if (s->standing_foot == 1) return 3;
  }
  return -1;
}

/* foot_contact_classify.cpp:146-318 (unknown transitions: counted, mode kept -- the reference waits on stdin there) */
static void update_walking_phase(po_leg *s, long utime, int left_contact, int right_contact, int left_strong, int right_strong)
{
  if (!s->initialized) {
    if (left_contact && right_contact) { s->mode = 0; s->initialized = 1; }
    return;
  }
  if (s->mode == 0) {
    if (left_contact && !right_strong) { s->mode = 1; s->last_break = utime; return; }
    else if (!left_strong && right_contact) { s->mode = 5; s->last_break = utime; return; }
    else if (left_contact && right_contact) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 1) {
    if (left_contact && !right_contact) { s->mode = 2; return; }
    else if (left_contact && right_strong) { s->mode = 0; return; }
    else if (left_contact && !right_strong) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 2) {
    if (left_contact && !right_contact) return;
    else if (left_contact && right_contact) { s->mode = 3; s->last_strike = utime; return; }
    else if (!left_contact && !right_contact) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 3) {
    if (left_contact && right_strong) { s->mode = 0; return; }
    else if (left_contact && !right_strong) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 4) {
    if (!left_strong && right_contact) { s->mode = 5; s->last_break = utime; return; }
    else if (left_contact && !right_strong) { s->mode = 1; s->last_break = utime; return; }
    else if (left_contact && right_contact) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 5) {
    if (!left_contact && right_contact) { s->mode = 6; return; }
    else if (left_strong && right_contact) { s->mode = 4; return; }
    else if (!left_strong && right_contact) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 6) {
    if (!left_contact && right_contact) return;
    else if (left_contact && right_contact) { s->mode = 7; s->last_strike = utime; return; }
    else if (!left_contact && !right_contact) return;
    s->unknown_transitions++; return;
  }
  if (s->mode == 7) {
    if (left_strong && right_contact) { s->mode = 4; return; }
    else if (!left_strong && right_contact) return;
    s->unknown_transitions++; return;
  }
  s->unknown_transitions++;
}

/* foot_contact_classify.cpp:57-125 */
static float classify_update(po_leg *s, long utime, double lforce, double rforce)
{
  po_schmitt_update(&s->weak_l, 20.0, 30.0, 5000, 5000, utime, lforce);
  po_schmitt_update(&s->weak_r, 20.0, 30.0, 5000, 5000, utime, rforce);
  po_schmitt_update(&s->strong_l, 275.0, 375.0, 7000, 7000, utime, lforce);
  po_schmitt_update(&s->strong_r, 275.0, 375.0, 7000, 7000, utime, rforce);
  update_walking_phase(s, utime, s->weak_l.status, s->weak_r.status, s->strong_l.status, s->strong_r.status);
  int recent_strike = 0, recent_break = 0;
  if (utime - s->last_strike < 95000) recent_strike = 1;
  if (utime - s->last_break < 800000) recent_break = 1;
  float odometry_status = 0.0f;
  if (recent_strike) odometry_status = -1.0f;
  else if (recent_break) odometry_status = 1.0f;
  return odometry_status;
}

/* leg_estimate::footTransitionAlt (leg_estimate.cpp:359-393) */
static int foot_transition_alt(po_leg *s, long utime, float leftz, float rightz, int n_control_contacts_left, int n_control_contacts_right)
{
  int contact_status = detect_foot_transition(s, utime, leftz, rightz);
  const int standing_foot = s->standing_foot;
  if (s->use_controller_input) {
    if (standing_foot == 0 /* F_LEFT_NEW */ || standing_foot == 2 /* F_LEFT_FIXED */) {
      if (n_control_contacts_left > -1 && n_control_contacts_left < 3 && n_control_contacts_right >= 3) {
        contact_status = 1;
        s->alt_l.status = 0; s->alt_l.timer = 0;  /* forceRightStandingFoot (FootContactAlt.cpp:125-129) */
        s->alt_r.status = 1; s->alt_r.timer = 0;
        s->standing_foot = 1;
      }
    } else if (standing_foot == 1 /* F_RIGHT_NEW */ || standing_foot == 3 /* F_RIGHT_FIXED */) {
      if (n_control_contacts_right > -1 && n_control_contacts_right < 3 && n_control_contacts_left >= 3) {
        contact_status = 0;
        s->alt_l.status = 1; s->alt_l.timer = 0;  /* forceLeftStandingFoot (:119-123) */
        s->alt_r.status = 0; s->alt_r.timer = 0;
        s->standing_foot = 0;
      }
    }
  }
  return contact_status;
}

float po_leg_update(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                    double lforce_in, double rforce_in, const double *world_to_body_quat, double *delta_t, double *delta_q,
                    long *prev_utime)
{
  return po_leg_update_cc(s, utime, l_t, l_q, r_t, r_q, lforce_in, rforce_in, -1, -1, world_to_body_quat, delta_t, delta_q, prev_utime);
}

float po_leg_update_cc(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                       double lforce_in, double rforce_in, int n_control_contacts_left, int n_control_contacts_right,
                       const double *world_to_body_quat, double *delta_t, double *delta_q, long *prev_utime)
{
  const double zero[3] = { 0, 0, 0 };
  double pos[3];
  int ok;
  return po_leg_update_wc(s, utime, l_t, l_q, r_t, r_q, lforce_in, rforce_in, n_control_contacts_left, n_control_contacts_right, zero,
                          world_to_body_quat, delta_t, delta_q, prev_utime, pos, &ok);
}

/* the same with world_to_body_'s translation (setPoseBody gets the whole head pose, rbis_legodo_update.cpp:218-229); also
 * returns getLegOdometryWorldConstraint's pose translation and flag (leg_estimate.hpp:104-108) */
float po_leg_update_wc(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                       double lforce_in, double rforce_in, int n_control_contacts_left, int n_control_contacts_right,
                       const double *world_to_body_pos, const double *world_to_body_quat, double *delta_t, double *delta_q,
                       long *prev_utime, double *constraint_pos, int *constraint_ok)
{
  /* FootSensing::force_z is a float (foot_contact_classify.hpp:24-31) */
  const float lforce = (float) lforce_in, rforce = (float) rforce_in;
  iso body_to_l, body_to_r, inv;
  iso_from_tq(l_t, l_q, &body_to_l);
  iso_from_tq(r_t, r_q, &body_to_r);
  double Rw[9];
  po_quat_to_rot(world_to_body_quat, Rw);
  /* :398-408 */
  s->previous_utime = s->current_utime;
  const iso previous_odom_to_body = s->odom_to_body;
  s->current_utime = utime;
  if ((s->current_utime - s->previous_utime) * 1E-6 > 30E-3) s->leg_odo_init = 0;
  /* :447-457 */
  const float contact_classification = classify_update(s, utime, lforce, rforce);
  const int contact_status = s->standing_mode ? foot_transition(s, utime, lforce, rforce)
                                              : foot_transition_alt(s, utime, lforce, rforce, n_control_contacts_left, n_control_contacts_right);
  /* leg_odometry_gravity_slaved_always (:219-297) */
  int init_this_iteration = 0;
  if (!s->leg_odo_init) {
    if (contact_status == 2 || contact_status == 3) {  /* prepInitialization + initializePose "zero" (:172-216) */
      const iso *foot = contact_status == 2 ? &body_to_l : &body_to_r, *other = contact_status == 2 ? &body_to_r : &body_to_l;
      const double zero[3] = { 0, 0, 0 };
      slaved_foot(zero, Rw, foot, &s->odom_to_primary);
      iso_inv(foot, &inv);
      iso_mul(&s->odom_to_primary, &inv, &s->odom_to_body);
      s->primary_foot = contact_status == 2 ? 0 : 1;
      iso_mul(&s->odom_to_body, other, &s->odom_to_secondary);
      s->leg_odo_init = 1;
      init_this_iteration = 1;
    }
  } else if (contact_status == 2 && s->primary_foot == 0) {
    double tr[3] = { s->odom_to_primary.t[0], s->odom_to_primary.t[1], s->odom_to_primary.t[2] };
    slaved_foot(tr, Rw, &body_to_l, &s->odom_to_primary);
    iso_inv(&body_to_l, &inv);
    iso_mul(&s->odom_to_primary, &inv, &s->odom_to_body);
    iso_mul(&s->odom_to_body, &body_to_r, &s->odom_to_secondary);
  } else if (contact_status == 1 && s->primary_foot == 0) {
    iso sw;
    memcpy(sw.R, Rw, sizeof sw.R);
    memcpy(sw.t, s->odom_to_body.t, sizeof sw.t);
    iso_mul(&sw, &body_to_r, &s->odom_to_primary);
    iso_inv(&body_to_r, &inv);
    iso_mul(&s->odom_to_primary, &inv, &s->odom_to_body);
    iso_mul(&s->odom_to_body, &body_to_l, &s->odom_to_secondary);
    s->primary_foot = 1;
  } else if (contact_status == 3 && s->primary_foot == 1) {
    double tr[3] = { s->odom_to_primary.t[0], s->odom_to_primary.t[1], s->odom_to_primary.t[2] };
    slaved_foot(tr, Rw, &body_to_r, &s->odom_to_primary);
    iso_inv(&body_to_r, &inv);
    iso_mul(&s->odom_to_primary, &inv, &s->odom_to_body);
    iso_mul(&s->odom_to_body, &body_to_l, &s->odom_to_secondary);
  } else if (contact_status == 0 && s->primary_foot == 1) {
    iso sw;
    memcpy(sw.R, Rw, sizeof sw.R);
    memcpy(sw.t, s->odom_to_body.t, sizeof sw.t);
    iso_mul(&sw, &body_to_l, &s->odom_to_primary);
    iso_inv(&body_to_l, &inv);
    iso_mul(&s->odom_to_primary, &inv, &s->odom_to_body);
    iso_mul(&s->odom_to_body, &body_to_r, &s->odom_to_secondary);
    s->primary_foot = 0;
  }
  /* :459-472 (world_to_body_init_ is true: setPoseBody was called) */
  iso world_to_body;
  memcpy(world_to_body.R, Rw, sizeof Rw);
  memcpy(world_to_body.t, world_to_body_pos, sizeof world_to_body.t);
  {
    const iso *prim = s->primary_foot == 0 ? &body_to_l : &body_to_r;   /* getPrimaryFootFK(primary_foot_, ...) */
    iso slide;
    iso_mul(&world_to_body, prim, &slide);
    if (contact_status == 0 || contact_status == 1) {
      s->world_to_primary_foot_transition = slide;
      s->world_to_primary_foot_transition_init = 1;
    }
  }
  /* :480-551 */
  float estimate_status = -1.0f;
  delta_t[0] = delta_t[1] = delta_t[2] = 0.0;
  delta_q[0] = 1.0; delta_q[1] = delta_q[2] = delta_q[3] = 0.0;
  if (s->leg_odo_init && !init_this_iteration) {
    iso d;
    iso_inv(&previous_odom_to_body, &inv);
    iso_mul(&inv, &s->odom_to_body, &d);
    memcpy(delta_t, d.t, sizeof d.t);
    quat_from_rot(d.R, delta_q);
    estimate_status = 0.0f;
    if (s->world_to_primary_foot_transition_init) {  /* determine_position_constraint_slaved_always (:299-318) */
      const iso *prim = s->primary_foot == 0 ? &body_to_l : &body_to_r;
      iso foot_constraint, pinv;
      slaved_foot(s->world_to_primary_foot_transition.t, Rw, prim, &foot_constraint);
      iso_inv(prim, &pinv);
      iso_mul(&foot_constraint, &pinv, &s->world_to_body_constraint);
      s->world_to_body_constraint_init = 1;
    } else {
      s->world_to_body_constraint_init = 0;
    }
  }
  if (s->filter_contact_events && estimate_status > -1) estimate_status = contact_classification;
  *prev_utime = s->previous_utime;
  memcpy(constraint_pos, s->world_to_body_constraint.t, sizeof(double) * 3);
  *constraint_ok = s->world_to_body_constraint_init;
  return estimate_status;
}

void po_leg_get(const po_leg *s, double *odom_to_body_t, double *odom_to_body_q, int *primary_foot, int *leg_odo_init, int *mode,
                int *unknown_transitions)
{
  memcpy(odom_to_body_t, s->odom_to_body.t, sizeof(double) * 3);
  quat_from_rot(s->odom_to_body.R, odom_to_body_q);
  *primary_foot = s->primary_foot;
  *leg_odo_init = s->leg_odo_init;
  *mode = s->mode;
  *unknown_transitions = s->unknown_transitions;
}

/* ---- forward kinematics (leg_estimate.cpp:430-447) ---- */
/* KDL::Rotation::Rot2(axis, angle): Rodrigues, axis assumed normalised */
static void kdl_rot2(const double *v, double angle, double *R)
{
  const double ct = cos(angle), st = sin(angle), vt = 1 - ct;
  const double m_vt_0 = vt * v[0], m_vt_1 = vt * v[1], m_vt_2 = vt * v[2];
  const double m_st_0 = v[0] * st, m_st_1 = v[1] * st, m_st_2 = v[2] * st;
  const double m_vt_0_1 = m_vt_0 * v[1], m_vt_0_2 = m_vt_0 * v[2], m_vt_1_2 = m_vt_1 * v[2];
  R[0] = ct + m_vt_0 * v[0]; R[1] = -m_st_2 + m_vt_0_1; R[2] = m_st_1 + m_vt_0_2;
  R[3] = m_st_2 + m_vt_0_1;  R[4] = ct + m_vt_1 * v[1]; R[5] = -m_st_0 + m_vt_1_2;
  R[6] = -m_st_1 + m_vt_0_2; R[7] = m_st_0 + m_vt_1_2;  R[8] = ct + m_vt_2 * v[2];
}
/* urdf::Rotation::setFromRPY followed by KDL::Rotation::Quaternion(x, y, z, w) (kdl_parser toKdl(urdf::Pose)) */
static void urdf_rpy_to_kdl(const double *rpy, double *R)
{
  const double phi = rpy[0] / 2.0, the = rpy[1] / 2.0, psi = rpy[2] / 2.0;
  double x = sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi);
  double y = cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi);
  double z = cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi);
  double w = cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi);
  const double s = sqrt(x * x + y * y + z * z + w * w);
  x /= s; y /= s; z /= s; w /= s;
  const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
  R[0] = w2 + x2 - y2 - z2;     R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z; R[4] = w2 - x2 + y2 - z2;     R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = w2 - x2 - y2 + z2;
}
/* KDL::Rotation::GetQuaternion */
static void kdl_get_quaternion(const double *R, double *x, double *y, double *z, double *w)
{
  const double trace = R[0] + R[4] + R[8];
  const double epsilon = 1E-12;
  if (trace > epsilon) {
    const double s = 0.5 / sqrt(trace + 1.0);
    *w = 0.25 / s;
    *x = (R[7] - R[5]) * s;
    *y = (R[2] - R[6]) * s;
    *z = (R[3] - R[1]) * s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
    *w = (R[7] - R[5]) / s;
    *x = 0.25 * s;
    *y = (R[1] + R[3]) / s;
    *z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
    *w = (R[2] - R[6]) / s;
    *x = (R[1] + R[3]) / s;
    *y = 0.25 * s;
    *z = (R[5] + R[7]) / s;
  } else {
    const double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
    *w = (R[3] - R[1]) / s;
    *x = (R[2] + R[6]) / s;
    *y = (R[5] + R[7]) / s;
    *z = 0.25 * s;
  }
}
/* TorqueAdjustment::processSample for one joint (torque_adjustment.cpp:27-62), float arithmetic */
float po_torque_adjust(float position, float effort, float gain)
{
  const float max_adjustment = 0.1f;
  if (isnormal(gain)) {
    volatile float val = effort / gain;
    float lim = val;
    if (val > max_adjustment) lim = max_adjustment;
    else if (val < -max_adjustment) lim = -max_adjustment;
    volatile float out = position - lim;
    return out;
  }
  return position;
}
/* One chain from the root link to a standing link.  Per joint what kdl_parser builds (addChildrenToTree / toKdl(JointPtr)):
 *   F_parent_jnt = (M, p) from <origin>;  joint = Joint(p, M * axis, RotAxis | TransAxis | None);  segment f_tip =
 *   joint.pose(0).Inverse() * F_parent_jnt;  Segment::pose(q) = joint.pose(q) * f_tip;
 * and what TreeFkSolverPosFull_recursive does with it: frame = parent_frame * segment.pose(q).  Then KDLToEigen
 * (common_conversions.hpp:5-13): translation copied, rotation through GetQuaternion and Eigen's rotate(q).
 * type: 0 fixed, 1 revolute, 2 prismatic; origin_xyz_rpy [n][6]; axis [n][3]; angle [n] (doubles holding the message's floats) */
void po_fk(int n, const int *type, const double *origin_xyz_rpy, const double *axis, const double *angle, double *t_out, double *q_out)
{
  iso T;
  iso_identity(&T);
  for (int j = 0; j < n; j++) {
    iso F, jp, jp0, jp0i, ftip, seg;
    urdf_rpy_to_kdl(origin_xyz_rpy + 6 * j + 3, F.R);
    memcpy(F.t, origin_xyz_rpy + 6 * j, sizeof F.t);
    double ax[3] = { 0, 0, 0 };
    if (type[j] != 0) {
      for (int i = 0; i < 3; i++) ax[i] = F.R[3 * i] * axis[3 * j] + F.R[3 * i + 1] * axis[3 * j + 1] + F.R[3 * i + 2] * axis[3 * j + 2];
      const double nrm = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  /* Joint::Joint: axis / axis.Norm() */
      for (int i = 0; i < 3; i++) ax[i] /= nrm;
    }
    /* Joint::pose(q) and Joint::pose(0) */
    iso_identity(&jp); iso_identity(&jp0);
    memcpy(jp.t, F.t, sizeof F.t); memcpy(jp0.t, F.t, sizeof F.t);
    if (type[j] == 1) { kdl_rot2(ax, angle[j], jp.R); kdl_rot2(ax, 0.0, jp0.R); }
    else if (type[j] == 2) for (int i = 0; i < 3; i++) jp.t[i] = F.t[i] + angle[j] * ax[i];
    iso_inv(&jp0, &jp0i);
    iso_mul(&jp0i, &F, &ftip);
    iso_mul(&jp, &ftip, &seg);
    iso_mul(&T, &seg, &T);
  }
  memcpy(t_out, T.t, sizeof T.t);
  double x, y, z, w;
  kdl_get_quaternion(T.R, &x, &y, &z, &w);
  q_out[0] = w; q_out[1] = x; q_out[2] = y; q_out[3] = z;
}
