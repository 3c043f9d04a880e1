"""Deterministic synthetic IMU / leg-odometry / VO / scan-match streams.

Counter-based: every sample is a pure function of (seed, filter b, step k, channel), so any slice of
the workload can be regenerated anywhere (host tests, the GPU box, a single filter for the CPU
baseline) without sequential state.  The truth trajectory is closed-form (sinusoidal roll/pitch/yaw and
world position), so gyro/accel/velocity are analytic derivatives -- no integration drift in the inputs.

Values marked (*) are this build's choices because the reference tree ships no .cfg
(SURVEY.md 8d); sensor sigmas follow state-estimator/src/noise_id/roll_forward.cpp:21-23.

Layout of every returned block is SoA with the filter index fastest, matching include/pronto_batch.h:
  IMU block    [7, B]  = gyro xyz, accel xyz, dt
  legodo block [6, B]  = z (body velocity) xyz, Rdiag xyz     + mask [B] uint8
  VO block     z [3,B], quat [4,B]; scan-match block z [3,B], quat [4,B]
"""
import numpy as np

SEED = 0x50524F4E544F  # "PRONTO" (*)
G_VAL = 9.80665

U64 = np.uint64
_M1 = U64(0xBF58476D1CE4E5B9)
_M2 = U64(0x94D049BB133111EB)
_GOLD = U64(0x9E3779B97F4A7C15)
_CH = U64(0xD1B54A32D192ED03)


def _mix(x):
    x = x ^ (x >> U64(30))
    x = x * _M1
    x = x ^ (x >> U64(27))
    x = x * _M2
    return x ^ (x >> U64(31))


def _uniform(b, k, ch, seed=SEED):
    """U(0,1) from (filter, step, channel); b, k broadcastable integer arrays."""
    with np.errstate(over="ignore"):
        key = _mix(U64(seed) ^ np.asarray(b, dtype=U64))
        u = _mix(key + np.asarray(k, dtype=U64) * _GOLD + U64(ch) * _CH)
    return ((u >> U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _normal(b, k, ch, seed=SEED):
    u1 = _uniform(b, k, 2 * ch, seed)
    u2 = _uniform(b, k, 2 * ch + 1, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


_PARAM_K = (1 << 40)  # "step" index reserved for per-filter constants


class Workload:
    """Synthetic workload for filters [b0, b0+B)."""

    # sensor model (*) -- SURVEY.md 8d
    SIG_GYRO = np.deg2rad(0.5)
    SIG_ACCEL = 0.1
    SIG_LEGODO = 0.1
    R_VXYZ = 0.1
    R_VXYZ_UNCERTAIN = 0.5
    SIG_VO_POS, SIG_VO_ROT = 0.02, 0.01
    SIG_SM_POS, SIG_SM_YAW = 0.05, np.deg2rad(1.0)
    STRIKE_BLACKOUT = 0.095   # foot_contact_classify.cpp:34
    BREAK_UNCERTAIN = 0.250   # (*) shortened from 0.8 s so all three branches are exercised

    def __init__(self, B, b0=0, dt_us=1000, n_states=15, seed=SEED):
        self.B, self.b0, self.dt_us, self.n, self.seed = B, b0, dt_us, n_states, seed
        b = np.arange(b0, b0 + B, dtype=np.int64)
        self.b = b
        P = lambda ch: _uniform(b, _PARAM_K, ch, seed)
        # attitude: rpy_i(t) = c_i + A_i sin(2 pi f_i t + phi_i), |omega| <~ 1 rad/s
        self.rpy_c = np.stack([(P(0) - 0.5) * 0.2, (P(1) - 0.5) * 0.2, (P(2) - 0.5) * 2 * np.pi])
        self.rpy_f = np.stack([0.2 + 1.8 * P(3 + i) for i in range(3)])
        amp_rate = np.stack([0.1 + 0.4 * P(6 + i) for i in range(3)])        # peak rate rad/s per axis
        self.rpy_A = amp_rate / (2 * np.pi * self.rpy_f)
        self.rpy_phi = np.stack([2 * np.pi * P(9 + i) for i in range(3)])
        # world position p_i(t) = Ap_i sin(2 pi fp_i t + php_i), peak accel <= 2 m/s^2
        self.pos_f = np.stack([0.2 + 0.8 * P(12 + i) for i in range(3)])
        acc_pk = np.stack([0.3 + 1.7 * P(15 + i) for i in range(3)]) * np.array([1.0, 1.0, 0.3])[:, None]
        self.pos_A = acc_pk / (2 * np.pi * self.pos_f) ** 2
        self.pos_phi = np.stack([2 * np.pi * P(18 + i) for i in range(3)])
        # biases (only excite them for the 21-state filter)
        if n_states == 21:
            self.bg = np.stack([np.deg2rad(0.1) * _normal(b, _PARAM_K, 30 + i, seed) for i in range(3)])
            self.ba = np.stack([0.02 * _normal(b, _PARAM_K, 33 + i, seed) for i in range(3)])
        else:
            self.bg = np.zeros((3, B))
            self.ba = np.zeros((3, B))
        # gait
        self.gait_T = 0.8 + 0.4 * P(40)
        self.gait_off = P(41) * self.gait_T

    # ---------------- truth ----------------
    def truth(self, t):
        """t scalar seconds -> dict of truth quantities, each [3 or 4, B]."""
        w2 = 2 * np.pi * self.rpy_f
        ang = w2 * t + self.rpy_phi
        rpy = self.rpy_c + self.rpy_A * np.sin(ang)
        rpy_d = self.rpy_A * w2 * np.cos(ang)
        wp = 2 * np.pi * self.pos_f
        angp = wp * t + self.pos_phi
        pos = self.pos_A * np.sin(angp)
        vel_w = self.pos_A * wp * np.cos(angp)
        acc_w = -self.pos_A * wp * wp * np.sin(angp)
        r, p, y = rpy
        cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        # R = Rz(y) Ry(p) Rx(r), body -> world
        R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                      [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                      [-sp, cp * sr, cp * cr]])
        rd, pd, yd = rpy_d
        omega = np.stack([rd - yd * sp, pd * cr + yd * sr * cp, -pd * sr + yd * cr * cp])
        # quaternion of the same ZYX rotation (pronto_math.cpp:39-49 convention)
        ch = lambda a: np.cos(0.5 * a)
        sh = lambda a: np.sin(0.5 * a)
        quat = np.stack([ch(r) * ch(p) * ch(y) + sh(r) * sh(p) * sh(y),
                         sh(r) * ch(p) * ch(y) - ch(r) * sh(p) * sh(y),
                         ch(r) * sh(p) * ch(y) + sh(r) * ch(p) * sh(y),
                         ch(r) * ch(p) * sh(y) - sh(r) * sh(p) * ch(y)])
        vel_b = np.einsum('jib,jb->ib', R, vel_w)
        f_b = np.einsum('jib,jb->ib', R, acc_w + np.array([0.0, 0.0, G_VAL])[:, None])
        return dict(rpy=rpy, R=R, quat=quat, omega=omega, pos=pos, vel_w=vel_w, vel_b=vel_b, f_b=f_b)

    def time_s(self, k):
        return (k * self.dt_us) * 1e-6

    # ---------------- sensor blocks ----------------
    def imu_block(self, k):
        """[7,B]: gyro, accel, dt for the predict from t_k to t_{k+1}.
        dt is (utime - prev_utime)*1E-6 in double, as sensor_handlers.cpp:243."""
        tr = self.truth(self.time_s(k))
        out = np.empty((7, self.B))
        for i in range(3):
            out[i] = tr["omega"][i] + self.bg[i] + self.SIG_GYRO * _normal(self.b, k, i, self.seed)
            out[3 + i] = tr["f_b"][i] + self.ba[i] + self.SIG_ACCEL * _normal(self.b, k, 3 + i, self.seed)
        utime, prev = (k + 1) * self.dt_us, k * self.dt_us
        out[6] = (utime - prev) * 1E-6
        return out

    def legodo_status(self, k):
        """leg_estimate status per filter at t_{k+1}: -1 skip, 0 certain, 1 uncertain
        (leg_estimate.hpp:84-93, foot_contact_classify.cpp:86-105)."""
        t = self.time_s(k + 1)
        ph = np.mod(t + self.gait_off, self.gait_T)
        st = np.zeros(self.B, dtype=np.int32)
        brk = 0.5 * self.gait_T
        st[(ph >= brk) & (ph < brk + self.BREAK_UNCERTAIN)] = 1
        st[ph < self.STRIKE_BLACKOUT] = -1
        return st

    def legodo_velocity(self, k):
        """[3,B] noisy body-frame velocity at t_{k+1} (what LegOdoCommon derives from the pelvis delta)."""
        tr = self.truth(self.time_s(k + 1))
        return np.stack([tr["vel_b"][i] + self.SIG_LEGODO * _normal(self.b, k, 10 + i, self.seed) for i in range(3)])

    def legodo_block(self, k):
        """([6,B] z|Rdiag, mask[B]) after LegOdoCommon::createMeasurement in mode lin_rate
        (rbis_legodo_common.cpp:124-129,153-156; rbis_legodo_update.cpp:242-255 for the skip)."""
        st = self.legodo_status(k)
        out = np.empty((6, self.B))
        out[0:3] = self.legodo_velocity(k)
        r = np.where(st >= 0.5, self.R_VXYZ_UNCERTAIN ** 2, self.R_VXYZ ** 2)
        out[3:6] = r
        return out, (st >= 0).astype(np.uint8)

    def vo_block(self, k):
        """VO pose measurement at t_{k+1} for FovisHandler mode position_orient (rbis_fovis_update.cpp:299-305):
        z [3,B] world position, quat [4,B], Rdiag [6,B] = (r_pxyz^2 x3, r_chi^2 x3)."""
        tr = self.truth(self.time_s(k + 1))
        z = np.stack([tr["pos"][i] + self.SIG_VO_POS * _normal(self.b, k, 20 + i, self.seed) for i in range(3)])
        rv = np.stack([self.SIG_VO_ROT * _normal(self.b, k, 23 + i, self.seed) for i in range(3)])
        q = _quat_mul(tr["quat"], _quat_exp(rv))
        Rd = np.empty((6, self.B))
        Rd[0:3] = self.SIG_VO_POS ** 2
        Rd[3:6] = self.SIG_VO_ROT ** 2
        return z, q, Rd

    def scanmatch_block(self, k):
        """ScanMatcherHandler mode position_yaw (sensor_handlers.cpp:709-722): z [3,B], quat [4,B], Rdiag [4,B]."""
        tr = self.truth(self.time_s(k + 1))
        z = np.stack([tr["pos"][i] + self.SIG_SM_POS * _normal(self.b, k, 26 + i, self.seed) for i in range(3)])
        rv = np.zeros((3, self.B))
        rv[2] = self.SIG_SM_YAW * _normal(self.b, k, 29, self.seed)
        q = _quat_mul(_quat_exp_world(rv), tr["quat"])
        Rd = np.empty((4, self.B))
        Rd[0:2] = self.SIG_SM_POS ** 2
        Rd[2] = self.SIG_SM_POS ** 2
        Rd[3] = self.SIG_SM_YAW ** 2
        return z, q, Rd

    # ---------------- initial conditions ----------------
    def initial_state(self):
        """x0 (vec [n,B], quat [4,B]) and P0 [n,n,B] (rbis_initializer.cpp:66-94 block layout).
        x0: v=0, pos=0, rpy ~ U(+-0.1), biases 0; the truth starts elsewhere, P0 covers the gap (*)."""
        n, B = self.n, self.B
        vec = np.zeros((n, B))
        tr = self.truth(0.0)
        # start from the truth attitude perturbed by ~1 deg so the filter has something to correct
        pert = np.stack([np.deg2rad(1.0) * _normal(self.b, _PARAM_K, 50 + i, self.seed) for i in range(3)])
        quat = _quat_mul(tr["quat"], _quat_exp(pert))
        vec[3:6] = tr["vel_b"]
        vec[9:12] = tr["pos"]
        P0 = np.zeros((n, n, B))
        sig = np.zeros(n)
        sig[9:12] = 0.5                    # sigma0.Delta_xy/z
        sig[6:9] = np.deg2rad(3.0)         # sigma0.chi_xy/z (degrees in the param file)
        sig[3:6] = 0.15                    # sigma0.vb
        if n == 21:
            sig[15:18] = np.deg2rad(0.5)   # sigma0.gyro_bias
            sig[18:21] = 0.1               # sigma0.accel_bias
        for i in range(n):
            P0[i, i] = sig[i] ** 2
        return vec, quat, P0

    def process_noise(self):
        """(q_gyro, q_accel, q_gyro_bias, q_accel_bias) = squared, rad-converted params
        (sensor_handlers.cpp:18-25)."""
        if self.n == 21:
            return np.array([np.deg2rad(0.5) ** 2, 0.1 ** 2, np.deg2rad(0.001) ** 2, 0.0001 ** 2])
        return np.array([np.deg2rad(0.5) ** 2, 0.1 ** 2, 0.0, 0.0])

    # ---------------- whole streams ----------------
    def streams(self, k0, T):
        """imu [T,7,B], legodo [T,6,B], mask [T,B] for steps k0..k0+T-1."""
        imu = np.empty((T, 7, self.B))
        lo = np.empty((T, 6, self.B))
        mask = np.empty((T, self.B), dtype=np.uint8)
        for j in range(T):
            imu[j] = self.imu_block(k0 + j)
            lo[j], mask[j] = self.legodo_block(k0 + j)
        return imu, lo, mask


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx])


def _quat_exp(rv):
    n = np.sqrt(np.sum(rv * rv, axis=0))
    safe = np.where(n > 0, n, 1.0)
    s = np.where(n > 0, np.sin(0.5 * n) / safe, 0.5)
    return np.concatenate([np.cos(0.5 * n)[None], rv * s])


_quat_exp_world = _quat_exp
