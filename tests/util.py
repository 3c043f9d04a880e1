"""Shared helpers for the tests (state packing, error norms, oracle drivers)."""
import numpy as np


def rel(a, b):
    """max |a-b| relative to the largest magnitude of the reference block."""
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def rel_elem(a, b, floor_frac=1e-4):
    """Element-wise relative error with an absolute floor: max_i |a_i-b_i| / max(|b_i|, floor_frac * max|b|).
    A wrong SMALL entry (an off-diagonal of P four orders below the largest one) fails this where the block-relative
    `rel` would let it through; entries below the floor are checked to floor_frac * tol * max|b| absolute."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(float(np.max(np.abs(b))), 1e-300)
    den = np.maximum(np.abs(b), floor_frac * scale)
    return float(np.max(np.abs(a - b) / den))


def embed21(vec, P):
    """n-state (vec [n,B], P [n,n,B]) -> the oracle's 21-state layout with zero bias rows (KAT vii)."""
    n, B = vec.shape
    v21 = np.zeros((21, B))
    v21[:n] = vec
    P21 = np.zeros((21, 21, B))
    P21[:n, :n] = P
    return v21, P21


def random_spd(n, B, scale, seed):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n, B)) * scale
    return np.einsum("ikb,jkb->ijb", A, A)


def diag_full(Rd):
    """[m,B] diagonal -> [B,m,m]."""
    m, B = Rd.shape
    R = np.zeros((B, m, m))
    for i in range(m):
        R[:, i, i] = Rd[i]
    return R


def pad_z(z, m):
    """z [k,B] -> [m,B] zero padded (entries at chi indices are ignored by the reference, rbis.cpp:203-205)."""
    out = np.zeros((m, z.shape[1]))
    out[: z.shape[0]] = z
    return np.ascontiguousarray(out)


def run_config(est_like, w, T, vo_every=0, sm_every=0, k0=0):
    """Drive `est_like` (anything with predict/update_indexed) through T steps of workload w:
    IMU predict + legodo m=3 each step, optional VO m=6 / scan-match m=4 corrections."""
    q4 = w.process_noise()
    for k in range(k0, k0 + T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est_like.predict(imu, q4)
        est_like.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
        if vo_every and k % vo_every == vo_every - 1:
            z, qm, Rd = w.vo_block(k)
            est_like.update_indexed([9, 10, 11, 6, 7, 8], pad_z(z, 6), Rd, quat_meas=np.ascontiguousarray(qm))
        if sm_every and k % sm_every == sm_every - 1:
            z, qm, Rd = w.scanmatch_block(k)
            est_like.update_indexed([9, 10, 11, 8], pad_z(z, 4), Rd, quat_meas=np.ascontiguousarray(qm))
