/* leg_odometry.c -- CPU restatement (TEST INFRASTRUCTURE, parity unpinned like the rest of oracle/) of the reference's leg
 * kinematic odometry, written with Eigen::Isometry3d-style 3x3 rotation matrices like the reference, i.e. NOT with the
 * quaternion arithmetic of pronto_amd/csrc/rbis_legodo.hpp it checks:
 *   motion_estimate/src/leg_estimate/leg_estimate.cpp:172-297,395-556
 *   motion_estimate/src/foot_contact_alt/FootContactAlt.cpp:5-100
 *   motion_estimate/src/leg_estimate/foot_contact_classify.cpp:5-125,146-318
 *   estimate_tools/src/filter_tools/SignalTap.cpp:48-134
 * Inputs are the body-to-foot transforms forward kinematics produces (KDL + URDF in the reference: not in tree). */
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
};
size_t po_leg_sizeof(void) { return sizeof(struct po_leg); }

void po_leg_init(po_leg *s, double schmitt_low, double schmitt_high, long low_delay, long high_delay, int filter_contact_events)
{
  memset(s, 0, sizeof *s);
  iso_identity(&s->odom_to_body);
  iso_identity(&s->odom_to_primary);
  iso_identity(&s->odom_to_secondary);
  s->primary_foot = 0;  /* F_LEFT (leg_estimate.cpp:126) */
  s->alt_lt = schmitt_low; s->alt_ht = schmitt_high; s->alt_ld = low_delay; s->alt_hd = high_delay;
  po_schmitt_reset(&s->alt_l); po_schmitt_reset(&s->alt_r);
  s->alt_l.status = 1; s->alt_r.status = 1;  /* forceHigh (FootContactAlt.cpp:28-29) */
  s->standing_foot = -1;
  po_schmitt_reset(&s->weak_l); po_schmitt_reset(&s->weak_r); po_schmitt_reset(&s->strong_l); po_schmitt_reset(&s->strong_r);
  s->mode = -1;
  s->filter_contact_events = filter_contact_events;
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

float po_leg_update(po_leg *s, long utime, const double *l_t, const double *l_q, const double *r_t, const double *r_q,
                    double lforce, double rforce, const double *world_to_body_quat, double *delta_t, double *delta_q,
                    long *prev_utime)
{
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
  const int contact_status = detect_foot_transition(s, utime, lforce, rforce);
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
  }
  if (s->filter_contact_events && estimate_status > -1) estimate_status = contact_classification;
  *prev_utime = s->previous_utime;
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
