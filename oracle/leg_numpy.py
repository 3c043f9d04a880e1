"""A SECOND, independent statement of the leg-odometry path in numpy / scipy -- the witness that holds oracle/leg_odometry.c.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference holds no fixtures for this path and cannot be built here; KDL,
kdl_parser and urdfdom are not in its tree).  Never imported from pronto_amd/.

Written from the reference's sources with a different representation than both other statements: poses are 4 x 4 homogeneous
matrices (Eigen::Isometry3d) and every quaternion <-> matrix conversion goes through scipy.spatial.transform.Rotation, where
oracle/leg_odometry.c carries 3 x 3 matrices + translations by hand and the kernels carry (translation, quaternion) pairs.
    forward kinematics     leg_estimate.cpp:430-447 -> KDL TreeFkSolverPosFull_recursive / kdl_parser / urdfdom (published
                           algorithms: segment = [R(rpy), xyz] * Rot(axis, theta), fixed-axis roll-pitch-yaw, Rodrigues)
    SchmittTrigger         estimate_tools/src/filter_tools/SignalTap.cpp:55-134
    FootContactAlt         motion_estimate/src/foot_contact_alt/FootContactAlt.cpp:5-130
    FootContact            motion_estimate/src/foot_contact/FootContact.cpp:5-83 (float members)
    foot_contact_classify  motion_estimate/src/leg_estimate/foot_contact_classify.cpp:25-125,146-318
    leg_estimate           motion_estimate/src/leg_estimate/leg_estimate.cpp:147-556
    TorqueAdjustment       estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp:27-62
"""
import numpy as np
from scipy.spatial.transform import Rotation

F_UNKNOWN, F_LEFT, F_RIGHT = -1, 0, 1                                     # footid / footid_alt
F_STATUS_UNKNOWN, F_LEFT_NEW, F_RIGHT_NEW, F_LEFT_FIXED, F_RIGHT_FIXED = -1, 0, 1, 2, 3   # contact_status_id
(LEFT_PRIME_RIGHT_STAND, LEFT_PRIME_RIGHT_BREAK, LEFT_PRIME_RIGHT_SWING, LEFT_PRIME_RIGHT_STRIKE, LEFT_STAND_RIGHT_PRIME,
 LEFT_BREAK_RIGHT_PRIME, LEFT_SWING_RIGHT_PRIME, LEFT_STRIKE_RIGHT_PRIME) = range(8)          # walkmode
UNKNOWN_MODE = -1
f32 = np.float32


# ---- poses -----------------------------------------------------------------------------------------------------------
def iso(R=None, t=None):
    T = np.eye(4)
    if R is not None:
        T[:3, :3] = R
    if t is not None:
        T[:3, 3] = t
    return T


def iso_inv(T):
    """Isometry3d::inverse()"""
    R = T[:3, :3]
    return iso(R.T, -R.T @ T[:3, 3])


def quat_wxyz(T):
    """Eigen::Quaterniond(T.rotation()) as (w, x, y, z)"""
    x, y, z, w = Rotation.from_matrix(T[:3, :3]).as_quat()
    return np.array([w, x, y, z])


def from_tq(t, q_wxyz):
    w, x, y, z = q_wxyz
    return iso(Rotation.from_quat([x, y, z, w]).as_matrix(), t)


def rotation_only(T):
    """Isometry3d::Identity().rotate(Quaterniond(T.rotation()))"""
    return iso(Rotation.from_matrix(T[:3, :3]).as_matrix())


# ---- forward kinematics ----------------------------------------------------------------------------------------------
def fk(joint_type, origin_xyz_rpy, axis, angle):
    """body_to_link of one chain: prod_j [R(rpy_j), xyz_j] * joint_j(angle_j).  type 0 fixed, 1 revolute, 2 prismatic.
    urdfdom's setFromRPY is the fixed-axis (extrinsic x, y, z) convention; KDL::Joint normalises its axis."""
    T = np.eye(4)
    for ty, o, a, th in zip(joint_type, origin_xyz_rpy, axis, angle):
        T = T @ iso(Rotation.from_euler("xyz", o[3:6]).as_matrix(), o[0:3])
        if ty == 1:
            u = np.asarray(a, dtype=np.float64) / np.linalg.norm(a)
            T = T @ iso(Rotation.from_rotvec(u * th).as_matrix())
        elif ty == 2:
            u = np.asarray(a, dtype=np.float64) / np.linalg.norm(a)
            T = T @ iso(None, u * th)
    return T


def torque_adjust(position, effort, gain):
    """TorqueAdjustment::processSample for one joint (float arithmetic; a gain that is not std::isnormal is skipped)."""
    position, effort, gain = f32(position), f32(effort), f32(gain)
    if not (np.isfinite(gain) and abs(gain) >= np.finfo(np.float32).tiny):
        return position
    a = f32(effort / gain)
    a = f32(0.1) if a > f32(0.1) else (f32(-0.1) if a < f32(-0.1) else a)
    return f32(position - a)


# ---- contact logic ---------------------------------------------------------------------------------------------------
class SchmittTrigger:
    def __init__(self, lt, ht, low_delay, high_delay):
        self.lt, self.ht, self.low_delay, self.high_delay = float(lt), float(ht), int(low_delay), int(high_delay)
        self.status, self.previous_time, self.timer, self.first_call = False, 0, 0, True

    def force(self, high):
        self.status, self.timer = bool(high), 0

    def update(self, present_time, value):
        if self.first_call:
            self.first_call = False
            self.previous_time = present_time
        if self.status:
            if value <= self.lt:
                if self.timer > self.low_delay:
                    self.status = False
                else:
                    self.timer += present_time - self.previous_time
            else:
                self.timer = 0
        else:
            if value >= self.ht:
                if self.timer > self.high_delay:
                    self.status = True
                else:
                    self.timer += present_time - self.previous_time
            else:
                self.timer = 0
        self.previous_time = present_time


class FootContactAlt:
    def __init__(self, lt, ht, low_delay, high_delay):
        lt, ht = float(f32(lt)), float(f32(ht))     # `const float schmitt_low_threshold` (FootContactAlt.cpp:5)
        self.standing_foot = F_UNKNOWN
        self.left, self.right = SchmittTrigger(lt, ht, low_delay, high_delay), SchmittTrigger(lt, ht, low_delay, high_delay)
        self.left.force(True)
        self.right.force(True)

    def detect(self, utime, leftz, rightz):
        l_last, r_last = self.left.status, self.right.status
        self.left.update(utime, float(leftz))
        self.right.update(utime, float(rightz))
        l, r = self.left.status, self.right.status
        if not l_last and l:
            self.standing_foot = F_LEFT
            return F_LEFT_NEW
        if not r_last and r:
            self.standing_foot = F_RIGHT
            return F_RIGHT_NEW
        if l_last and not l:
            if self.standing_foot == F_LEFT:
                self.standing_foot = F_RIGHT
                return F_RIGHT_NEW
            return F_RIGHT_FIXED
        if r_last and not r:
            if self.standing_foot == F_RIGHT:
                self.standing_foot = F_LEFT
                return F_LEFT_NEW
            return F_LEFT_FIXED
        if self.standing_foot == F_LEFT:
            return F_LEFT_FIXED
        if self.standing_foot == F_RIGHT:
            return F_RIGHT_FIXED
        return F_STATUS_UNKNOWN      # (the reference exits here)

    def force_left(self):
        self.standing_foot = F_LEFT
        self.left.force(True)
        self.right.force(False)

    def force_right(self):
        self.left.force(False)
        self.right.force(True)
        self.standing_foot = F_RIGHT


class FootContact:
    """the "standing" contact mode; returns the new standing foot or F_UNKNOWN"""

    def __init__(self, total_force, schmitt_level):
        self.total_force, self.schmitt_level = f32(total_force), f32(schmitt_level)
        self.transition_timeout = 4000
        self.standing_foot = F_UNKNOWN
        self.lcmutime, self.transition_timespan, self.flag = 0, 0, True

    def detect(self, utime, leftz, rightz):
        delta = utime - self.lcmutime
        self.lcmutime = utime
        l, r = f32(leftz), f32(rightz)
        prim, sec = (l, r) if self.standing_foot == F_LEFT else (r, l)
        prod = f32(self.schmitt_level * self.total_force)        # float product, rounded, then the float difference
        if f32(sec - prod) > prim:
            self.transition_timespan += delta
        else:
            self.transition_timespan = 0
            self.flag = True
        if self.transition_timespan > self.transition_timeout and self.flag:
            self.flag = False
            return F_RIGHT if self.standing_foot == F_LEFT else (F_LEFT if self.standing_foot == F_RIGHT else F_UNKNOWN)
        return F_UNKNOWN


class FootContactClassify:
    def __init__(self):
        self.lw, self.rw = SchmittTrigger(20.0, 30.0, 5000, 5000), SchmittTrigger(20.0, 30.0, 5000, 5000)
        self.ls, self.rs = SchmittTrigger(275.0, 375.0, 7000, 7000), SchmittTrigger(275.0, 375.0, 7000, 7000)
        self.mode, self.initialized = UNKNOWN_MODE, False
        self.last_strike, self.last_break = 0, 0
        self.unknown = 0

    def walking_phase(self, utime, lc, rc, lcs, rcs):
        if not self.initialized:
            if lc and rc:
                self.mode, self.initialized = LEFT_PRIME_RIGHT_STAND, True
            return
        m = self.mode
        if m == LEFT_PRIME_RIGHT_STAND:
            if lc and not rcs:
                self.mode, self.last_break = LEFT_PRIME_RIGHT_BREAK, utime
            elif not lcs and rc:
                self.mode, self.last_break = LEFT_BREAK_RIGHT_PRIME, utime
            elif lc and rc:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_PRIME_RIGHT_BREAK:
            if lc and not rc:
                self.mode = LEFT_PRIME_RIGHT_SWING
            elif lc and rcs:
                self.mode = LEFT_PRIME_RIGHT_STAND
            elif lc and not rcs:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_PRIME_RIGHT_SWING:
            if lc and not rc:
                pass
            elif lc and rc:
                self.mode, self.last_strike = LEFT_PRIME_RIGHT_STRIKE, utime
            elif not lc and not rc:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_PRIME_RIGHT_STRIKE:
            if lc and rcs:
                self.mode = LEFT_PRIME_RIGHT_STAND
            elif lc and not rcs:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_STAND_RIGHT_PRIME:
            if not lcs and rc:
                self.mode, self.last_break = LEFT_BREAK_RIGHT_PRIME, utime
            elif lc and not rcs:
                self.mode, self.last_break = LEFT_PRIME_RIGHT_BREAK, utime
            elif lc and rc:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_BREAK_RIGHT_PRIME:
            if not lc and rc:
                self.mode = LEFT_SWING_RIGHT_PRIME
            elif lcs and rc:
                self.mode = LEFT_STAND_RIGHT_PRIME
            elif not lcs and rc:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_SWING_RIGHT_PRIME:
            if not lc and rc:
                pass
            elif lc and rc:
                self.mode, self.last_strike = LEFT_STRIKE_RIGHT_PRIME, utime
            elif not lc and not rc:
                pass
            else:
                self.unknown += 1
        elif m == LEFT_STRIKE_RIGHT_PRIME:
            if lcs and rc:
                self.mode = LEFT_STAND_RIGHT_PRIME
            elif not lcs and rc:
                pass
            else:
                self.unknown += 1
        else:
            self.unknown += 1

    def update(self, utime, lforce, rforce):
        lf, rf = float(f32(lforce)), float(f32(rforce))     # FootSensing::force_z is a float
        self.lw.update(utime, lf)
        self.rw.update(utime, rf)
        self.ls.update(utime, lf)
        self.rs.update(utime, rf)
        self.walking_phase(utime, self.lw.status, self.rw.status, self.ls.status, self.rs.status)
        if utime - self.last_strike < 95000:
            return -1.0
        if utime - self.last_break < 800000:
            return 1.0
        return 0.0


# ---- leg_estimate ----------------------------------------------------------------------------------------------------
class LegEstimate:
    def __init__(self, schmitt_low, schmitt_high, low_delay, high_delay, filter_contact_events, standing=None,
                 use_controller_input=False):
        """standing = (total_force, standing_schmitt_level) selects init_contact_mode "standing" (leg_estimate.cpp:113-118)"""
        self.filter_contact_events = bool(filter_contact_events)
        self.use_controller_input = bool(use_controller_input)
        self.control_standing = standing is not None
        self.alt = FootContactAlt(schmitt_low, schmitt_high, low_delay, high_delay)
        self.alt.standing_foot = F_LEFT                       # setStandingFoot(F_LEFT) (:110)
        self.logic = FootContact(*(standing or (0.0, 0.0)))
        self.logic.standing_foot = F_LEFT                     # setStandingFoot(FOOT_LEFT) (:98)
        self.classify = FootContactClassify()
        self.primary_foot, self.standing_foot = F_LEFT, F_LEFT
        self.leg_odo_init = False
        self.odom_to_body = np.eye(4)
        self.odom_to_primary_foot_fixed = np.eye(4)
        self.previous_utime = self.current_utime = 0
        self.world_to_body, self.world_to_body_init = np.eye(4), False
        self.world_to_primary_foot_transition, self.world_to_primary_foot_transition_init = np.eye(4), False
        self.world_to_body_constraint, self.world_to_body_constraint_init = np.eye(4), False
        self.lforce = self.rforce = f32(0)
        self.ncl = self.ncr = -1

    def set_pose_body(self, world_to_body):
        self.world_to_body, self.world_to_body_init = np.array(world_to_body, dtype=np.float64), True

    def set_foot_sensing(self, lforce, rforce):
        self.lforce, self.rforce = f32(lforce), f32(rforce)

    def set_control_contacts(self, ncl, ncr):
        self.ncl, self.ncr = int(ncl), int(ncr)

    def _foot_transition(self):
        newstep = self.logic.detect(self.current_utime, self.lforce, self.rforce)
        if newstep in (F_LEFT, F_RIGHT):
            self.logic.standing_foot = newstep
        sf = self.logic.standing_foot
        if newstep != F_UNKNOWN:
            cs = F_LEFT_NEW if sf == F_LEFT else F_RIGHT_NEW
        else:
            cs = F_LEFT_FIXED if sf == F_LEFT else F_RIGHT_FIXED
        self.standing_foot = sf
        return cs

    def _foot_transition_alt(self):
        cs = self.alt.detect(self.current_utime, self.lforce, self.rforce)
        self.standing_foot = self.alt.standing_foot
        if self.use_controller_input:
            # (standing_foot_ is compared with F_LEFT_NEW = 0 / F_LEFT_FIXED = 2 there: only 0 = F_LEFT can match, and 1 = F_RIGHT)
            if self.standing_foot == F_LEFT:
                if -1 < self.ncl < 3 and self.ncr >= 3:
                    cs = F_RIGHT_NEW
                    self.alt.force_right()
                    self.standing_foot = F_RIGHT
            elif self.standing_foot == F_RIGHT:
                if -1 < self.ncr < 3 and self.ncl >= 3:
                    cs = F_LEFT_NEW
                    self.alt.force_left()
                    self.standing_foot = F_LEFT
        return cs

    def _slave(self, body_to_foot, translation):
        """the foot turned to agree with the pelvis orientation, put at `translation`"""
        at_zero = rotation_only(self.world_to_body) @ body_to_foot
        return iso(Rotation.from_matrix(at_zero[:3, :3]).as_matrix(), translation)

    def _gravity_slaved_always(self, bl, br, cs):
        if not self.leg_odo_init:
            if cs in (F_LEFT_FIXED, F_RIGHT_FIXED):          # prepInitialization / initializePose ("zero")
                foot = bl if cs == F_LEFT_FIXED else br
                self.odom_to_primary_foot_fixed = self._slave(foot, np.zeros(3))
                self.odom_to_body = self.odom_to_primary_foot_fixed @ iso_inv(foot)
                self.primary_foot = F_LEFT if cs == F_LEFT_FIXED else F_RIGHT
                self.leg_odo_init = True
                return True
            return False
        pf = self.primary_foot
        if (cs == F_LEFT_FIXED and pf == F_LEFT) or (cs == F_RIGHT_FIXED and pf == F_RIGHT):
            foot = bl if pf == F_LEFT else br
            self.odom_to_primary_foot_fixed = self._slave(foot, self.odom_to_primary_foot_fixed[:3, 3].copy())
            self.odom_to_body = self.odom_to_primary_foot_fixed @ iso_inv(foot)
        elif (cs == F_RIGHT_NEW and pf == F_LEFT) or (cs == F_LEFT_NEW and pf == F_RIGHT):
            foot = br if cs == F_RIGHT_NEW else bl
            switch = iso(Rotation.from_matrix(self.world_to_body[:3, :3]).as_matrix(), self.odom_to_body[:3, 3].copy())
            self.odom_to_primary_foot_fixed = switch @ foot
            self.odom_to_body = self.odom_to_primary_foot_fixed @ iso_inv(foot)
            self.primary_foot = F_RIGHT if cs == F_RIGHT_NEW else F_LEFT
        return False

    def update_odometry(self, utime, body_to_l_foot, body_to_r_foot):
        """leg_estimate::updateOdometry behind the forward kinematics -> (status, odom_to_body_delta)"""
        self.previous_utime = self.current_utime
        previous_odom_to_body = self.odom_to_body.copy()
        self.current_utime = int(utime)
        if (self.current_utime - self.previous_utime) * 1E-6 > 30E-3:
            self.leg_odo_init = False
        classification = self.classify.update(self.current_utime, self.lforce, self.rforce)
        cs = self._foot_transition() if self.control_standing else self._foot_transition_alt()
        init_this_iteration = self._gravity_slaved_always(body_to_l_foot, body_to_r_foot, cs)
        primary_fk = body_to_l_foot if self.primary_foot == F_LEFT else body_to_r_foot
        if self.world_to_body_init and cs in (F_LEFT_NEW, F_RIGHT_NEW):
            self.world_to_primary_foot_transition = self.world_to_body @ primary_fk
            self.world_to_primary_foot_transition_init = True
        status = -1.0
        delta = np.eye(4)
        if self.leg_odo_init and not init_this_iteration:
            delta = iso_inv(previous_odom_to_body) @ self.odom_to_body
            status = 0.0
            if self.world_to_body_init and self.world_to_primary_foot_transition_init:
                at_zero = rotation_only(self.world_to_body) @ primary_fk
                constraint = iso(Rotation.from_matrix(at_zero[:3, :3]).as_matrix(), self.world_to_primary_foot_transition[:3, 3].copy())
                self.world_to_body_constraint = constraint @ iso_inv(primary_fk)
                self.world_to_body_constraint_init = True
            else:
                self.world_to_body_constraint_init = False
        if self.filter_contact_events and status > -1:
            status = classification
        self.delta = delta
        return float(f32(status)), delta
