"""Second, INDEPENDENT restatement of the RBIS EKF equations in numpy.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.  Written from the equations of SURVEY.md 8a (rows a2-a9),
not transliterated from oracle/pronto_oracle.c: rotation matrices come from the Rodrigues/Euler-Rodrigues
form, the gain from numpy.linalg.solve, log-det from slogdet, Qd from its closed form.  Two
restatements agreeing to <=1e-12 is the only substitute available for running the reference
(state-estimator/src/mav_state_est/rbis.cpp:12-227), which cannot be built here.

All functions are batched: vec [B,21], quat [B,4] (w,x,y,z), cov [B,21,21].
"""
import numpy as np

N = 21
W, V, CHI, POS, A, BG, BA = 0, 3, 6, 9, 12, 15, 18
G_VAL = 9.80665
CHI_TOL = 1e-6


def skew(v):
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def rot_of_quat(q):
    """R = I + 2 w [u]x + 2 [u]x^2 for unit q = (w, u) (rbis.cpp:18 quat.toRotationMatrix())."""
    w = q[..., 0:1, None]
    ux = skew(q[..., 1:4])
    return np.eye(3) + 2.0 * w * ux + 2.0 * ux @ ux


def quat_mul(a, b):
    aw, av = a[..., 0:1], a[..., 1:4]
    bw, bv = b[..., 0:1], b[..., 1:4]
    w = aw * bw - np.sum(av * bv, -1, keepdims=True)
    v = aw * bv + bw * av + np.cross(av, bv)
    return np.concatenate([w, v], -1)


def quat_exp(rv):
    """AngleAxis(|rv|, rv/|rv|) as a quaternion; identity where |rv| <= CHI_TOL (mask returned)."""
    n = np.linalg.norm(rv, axis=-1, keepdims=True)
    big = n[..., 0] > CHI_TOL
    safe = np.where(n > 0, n, 1.0)
    q = np.concatenate([np.cos(0.5 * n), np.sin(0.5 * n) * rv / safe], -1)
    ident = np.zeros_like(q)
    ident[..., 0] = 1.0
    return np.where(big[..., None], q, ident), big


def fold_chi(vec, quat):
    """eigen_utils chiToQuat [NOT IN TREE]: fold vec[chi] into quat when |chi| > tol."""
    dq, big = quat_exp(vec[..., CHI:CHI + 3])
    quat = quat_mul(quat, dq)
    vec = vec.copy()
    vec[..., CHI:CHI + 3] = np.where(big[..., None], 0.0, vec[..., CHI:CHI + 3])
    return vec, quat


def add_state(vec, quat, dvec, dquat):
    vec, quat = fold_chi(vec + dvec, quat)
    return vec, quat_mul(quat, dquat)


def quat_log_diff(q1, q2):
    """subtractQuats(q1, q2): rotation vector of q2^-1 * q1, angle in [-pi, pi]."""
    q2c = q2 * np.array([1.0, -1.0, -1.0, -1.0]) / np.sum(q2 * q2, -1, keepdims=True)
    r = quat_mul(q2c, q1)
    r = np.where(r[..., 0:1] < 0, -r, r)  # same rotation, w >= 0 -> angle in [0, pi]
    n = np.linalg.norm(r[..., 1:4], axis=-1, keepdims=True)
    ang = 2.0 * np.arctan2(n, r[..., 0:1])
    safe = np.where(n > 0, n, 1.0)
    return np.where(n > 0, r[..., 1:4] / safe * ang, 0.0)


def process_matrices(vec, quat, dt):
    """Ad = I + Ac dt (rbis.cpp:12-35,112-114) from block formulas; dt [B]."""
    B = vec.shape[0]
    R = rot_of_quat(quat)
    gb = np.einsum('bji,j->bi', R, np.array([0.0, 0.0, -G_VAL]))  # R^T g
    what, vhat = skew(vec[:, W:W + 3]), skew(vec[:, V:V + 3])
    Ac = np.zeros((B, N, N))
    Ac[:, V:V + 3, V:V + 3] = -what
    Ac[:, V:V + 3, CHI:CHI + 3] = skew(gb)
    Ac[:, CHI:CHI + 3, CHI:CHI + 3] = -what
    Ac[:, POS:POS + 3, V:V + 3] = R
    Ac[:, POS:POS + 3, CHI:CHI + 3] = -R @ vhat
    Ac[:, V:V + 3, BG:BG + 3] = -vhat
    Ac[:, V:V + 3, BA:BA + 3] = -np.eye(3)
    Ac[:, CHI:CHI + 3, BG:BG + 3] = -np.eye(3)
    return np.eye(N) + Ac * dt[:, None, None]


def process_noise(vec, dt, q4):
    """Closed form of Wc Qc Wc^T dt (rbis.cpp:91-116), SURVEY.md 8a row a4."""
    B = vec.shape[0]
    qg, qa, qbg, qba = q4
    vhat = skew(vec[:, V:V + 3])
    Qd = np.zeros((B, N, N))
    Qd[:, V:V + 3, V:V + 3] = qg * vhat @ np.transpose(vhat, (0, 2, 1)) + qa * np.eye(3)
    Qd[:, V:V + 3, CHI:CHI + 3] = qg * vhat
    Qd[:, CHI:CHI + 3, V:V + 3] = qg * np.transpose(vhat, (0, 2, 1))
    Qd[:, CHI:CHI + 3, CHI:CHI + 3] = qg * np.eye(3)
    Qd[:, BG:BG + 3, BG:BG + 3] = qbg * np.eye(3)
    Qd[:, BA:BA + 3, BA:BA + 3] = qba * np.eye(3)
    return Qd * dt[:, None, None]


def predict(vec, quat, cov, gyro, accel, dt, q4):
    """RBISIMUProcessStep::updateFilter (rbis_update_interface.cpp:30-52)."""
    dt = np.broadcast_to(np.asarray(dt, dtype=np.float64), (vec.shape[0],))
    # covariance first, about the PRIOR state (rbis_update_interface.cpp:39)
    Ad = process_matrices(vec, quat, dt)
    cov = Ad @ cov @ np.transpose(Ad, (0, 2, 1)) + process_noise(vec, dt, q4)
    cov[:, A:A + 3, A:A + 3] = q4[1] * np.eye(3)
    cov[:, W:W + 3, W:W + 3] = q4[0] * np.eye(3)
    # state (rbis.cpp:37-75)
    vec = vec.copy()
    R = rot_of_quat(quat)
    w = gyro - vec[:, BG:BG + 3]
    a = accel - vec[:, BA:BA + 3]
    v = vec[:, V:V + 3]
    gb = np.einsum('bji,j->bi', R, np.array([0.0, 0.0, -G_VAL]))
    vdot = -np.cross(w, v) + gb + a
    pdot = np.einsum('bij,bj->bi', R, v)
    vec[:, W:W + 3] = w
    vec[:, A:A + 3] = a
    d = np.zeros_like(vec)
    d[:, V:V + 3] = vdot * dt[:, None]
    d[:, CHI:CHI + 3] = w * dt[:, None]
    d[:, POS:POS + 3] = pdot * dt[:, None]
    ident = np.zeros_like(quat)
    ident[:, 0] = 1.0
    d, dq = fold_chi(d, ident)
    vec, quat = add_state(vec, quat, d, dq)
    return vec, quat, cov


def update(vec, quat, cov, ll, idx, z, R, quat_meas=None):
    """RBISIndexed(PlusOrientation)Measurement::updateFilter (rbis_update_interface.cpp:54-107,
    rbis.cpp:124-227).  idx [m]; z [B,m]; R [B,m,m]."""
    idx = np.asarray(idx)
    B = vec.shape[0]
    resid = z - vec[:, idx]
    if quat_meas is not None:
        dq = quat_log_diff(quat_meas, quat)
        for i, ii in enumerate(idx):
            if CHI <= ii <= CHI + 2:
                resid[:, i] = dq[:, ii - CHI]
    PCt = cov[:, :, idx]                         # P C^T
    S = R + cov[:, idx][:, :, idx]
    K = np.transpose(np.linalg.solve(S, cov[:, idx, :]), (0, 2, 1))   # (S^-1 C P)^T
    dcov = K @ cov[:, idx, :]
    sign, logdet = np.linalg.slogdet(S)
    quad = np.einsum('bi,bi->b', resid, np.linalg.solve(S, resid[..., None])[..., 0])
    dx = np.einsum('bnm,bm->bn', K, resid)
    ident = np.zeros((B, 4))
    ident[:, 0] = 1.0
    dxf, dxq = fold_chi(dx, ident)
    vec, quat = add_state(vec, quat, dxf, dxq)
    _ = PCt
    return vec, quat, cov - dcov, ll + (-logdet - quad)
