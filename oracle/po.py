"""ctypes binding of the CPU oracle (oracle/pronto_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see pronto_oracle.h).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from pronto_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "libpronto_oracle.so")

N = 21
VEL, CHI, POS = 3, 6, 9


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("pronto_oracle.c", "leg_odometry.c", "joint_filter.c", "pronto_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class Rbis(C.Structure):
    _fields_ = [("vec", C.c_double * N), ("quat", C.c_double * 4), ("utime", C.c_int64)]


class Rbim(C.Structure):
    _fields_ = [("m", C.c_double * (N * N))]


class Notch(C.Structure):
    _fields_ = [("b", C.c_double * 3), ("a", C.c_double * 3), ("x", C.c_double * 2), ("y", C.c_double * 2)]


def notch_cascade_run(acc, notch_freq, fs=1000.0):
    """Oracle run of InsHandler::doFilter over a packet stream acc [T,3] -> filtered [T,3] (one filter)."""
    L = lib()
    filt = (Notch * 9)()
    L.po_notch_cascade_init(filt, notch_freq, fs)
    out = np.empty_like(acc)
    v = (C.c_double * 3)()
    for t in range(acc.shape[0]):
        v[0], v[1], v[2] = acc[t]
        L.po_notch_cascade(filt, v)
        out[t] = v[0], v[1], v[2]
    return out


def noise_id_window(truth_vec, truth_quat, start_cov, dt, q_gyro, q_accel, idx):
    """Oracle sampleProcessForward + likelihood pieces for ONE window (noise_id.cpp:19-40,52-58).
    truth_vec [N+1,21], truth_quat [N+1,4], start_cov [21,21] -> (err [21], cov [21,21], logdet, maha, loglike)."""
    L = lib()
    N1 = truth_vec.shape[0]
    tr = (Rbis * N1)()
    for k in range(N1):
        tr[k].vec[:] = list(truth_vec[k])
        tr[k].quat[:] = list(truth_quat[k])
    sc = Rbim()
    sc.m[:] = list(np.ascontiguousarray(start_cov.T).ravel())
    err, cov = Rbis(), Rbim()
    L.po_noise_id_window(N1 - 1, tr, C.byref(sc), dt, q_gyro, q_accel, C.byref(err), C.byref(cov))
    ia = (C.c_int * len(idx))(*idx)
    ld, mh = C.c_double(), C.c_double()
    ll = L.po_loglike_pieces(len(idx), ia, C.byref(err), C.byref(cov), C.byref(ld), C.byref(mh))
    return np.array(err.vec[:]), np.array(cov.m[:]).reshape(21, 21).T, ld.value, mh.value, ll


class Batch(C.Structure):
    _fields_ = [("B", C.c_int), ("vec", C.c_void_p), ("quat", C.c_void_p), ("cov", C.c_void_p),
                ("ll", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.po_set_constants.argtypes = [C.c_double, C.c_double]
        L.po_get_constants.argtypes = [dp, dp]
        L.po_ins_update_state.argtypes = [dp, dp, C.c_double, C.POINTER(Rbis)]
        L.po_ins_update_covariance.argtypes = [C.c_double] * 4 + [C.POINTER(Rbis), C.POINTER(Rbim), C.c_double]
        L.po_get_imu_linearization.argtypes = [C.POINTER(Rbis), C.POINTER(Rbim)]
        L.po_imu_process_step.argtypes = [dp, dp] + [C.c_double] * 5 + [C.POINTER(Rbis), C.POINTER(Rbim), C.c_double,
                                                                         C.POINTER(Rbis), C.POINTER(Rbim), dp]
        L.po_indexed_update.argtypes = [C.c_int, ip, dp, dp, C.POINTER(Rbis), C.POINTER(Rbim), C.c_double,
                                        C.POINTER(Rbis), C.POINTER(Rbim), dp]
        L.po_indexed_orient_update.argtypes = [C.c_int, ip, dp, dp, dp, C.POINTER(Rbis), C.POINTER(Rbim),
                                               C.c_double, C.POINTER(Rbis), C.POINTER(Rbim), dp]
        L.po_subtract_quats.argtypes = [dp, dp, dp]
        L.po_quat_mul.argtypes = [dp, dp, dp]
        L.po_quat_rotate.argtypes = [dp, dp, dp]
        L.po_quat_inv_rotate.argtypes = [dp, dp, dp]
        L.po_quat_to_rot.argtypes = [dp, dp]
        L.po_chi_to_quat.argtypes = [C.POINTER(Rbis)]
        L.po_add_state.argtypes = [C.POINTER(Rbis), C.POINTER(Rbis)]
        L.po_euler_to_quat.argtypes = [C.c_double] * 3 + [dp]
        L.po_quat_to_euler.argtypes = [dp, dp]
        L.po_delta_as_velocity.argtypes = [dp, dp, C.c_int64, dp, dp]
        L.po_legodo_create_measurement.argtypes = [C.c_int, dp, dp, dp, dp, C.c_int64, C.c_int64, C.c_int,
                                                   C.c_float, ip, dp, dp]
        L.po_legodo_create_measurement.restype = C.c_int
        L.po_fovis_compose.argtypes = [dp, dp, dp, dp, dp, dp]
        L.po_ekf_smoothing_step.argtypes = [C.POINTER(Rbis), C.POINTER(Rbim), C.POINTER(Rbis), C.POINTER(Rbim),
                                            C.c_double, C.POINTER(Rbis), C.POINTER(Rbim)]
        L.po_batch_predict.argtypes = [C.POINTER(Batch), dp, dp, C.c_int]
        L.po_batch_update_indexed.argtypes = [C.POINTER(Batch), C.c_int, ip, dp, dp, dp, C.c_void_p, C.c_int]
        L.po_batch_run_legodo.argtypes = [C.POINTER(Batch), C.c_int, dp, dp, C.c_void_p, dp, C.c_int]
        L.po_batch_run_legodo.restype = C.c_double
        L.po_max_threads.restype = C.c_int
        L.po_noise_id_window.argtypes = [C.c_int, C.POINTER(Rbis), C.POINTER(Rbim), C.c_double, C.c_double, C.c_double,
                                         C.POINTER(Rbis), C.POINTER(Rbim)]
        L.po_loglike_pieces.argtypes = [C.c_int, ip, C.POINTER(Rbis), C.POINTER(Rbim), dp, dp]
        L.po_loglike_pieces.restype = C.c_double
        L.po_notch_cascade_init.argtypes = [C.POINTER(Notch), C.c_double, C.c_double]
        L.po_notch_cascade.argtypes = [C.POINTER(Notch), dp]
        L.po_notch_init.argtypes = [C.POINTER(Notch), C.c_double, C.c_double]
        L.po_notch_process.argtypes = [C.POINTER(Notch), C.c_double]
        L.po_notch_process.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def constants():
    g, t = C.c_double(), C.c_double()
    lib().po_get_constants(C.byref(g), C.byref(t))
    return g.value, t.value


class OracleBatch:
    """B filters in the oracle's 21-state SoA layout (filter index fastest)."""

    def __init__(self, vec, quat, cov, ll=None):
        # vec [21,B], quat [4,B] (w,x,y,z), cov [21,21,B] indexed [row, col, b]
        self.B = vec.shape[1]
        self.vec = np.ascontiguousarray(vec, dtype=np.float64).copy()
        self.quat = np.ascontiguousarray(quat, dtype=np.float64).copy()
        # C side wants col-major flat index c*21+r  ->  array [c, r, b]
        self.cov_cm = np.ascontiguousarray(np.transpose(cov, (1, 0, 2)), dtype=np.float64).copy()
        self.ll = np.zeros(self.B) if ll is None else np.ascontiguousarray(ll, dtype=np.float64).copy()

    def _c(self):
        return Batch(self.B, self.vec.ctypes.data, self.quat.ctypes.data, self.cov_cm.ctypes.data,
                     self.ll.ctypes.data)

    @property
    def cov(self):  # [row, col, b]
        return np.transpose(self.cov_cm, (1, 0, 2))

    def predict(self, imu_block, q4, nthreads=0):
        imu_block = np.ascontiguousarray(imu_block, dtype=np.float64)
        assert imu_block.shape == (7, self.B)
        q4 = np.ascontiguousarray(q4, dtype=np.float64)
        b = self._c()
        lib().po_batch_predict(C.byref(b), _dp(imu_block), _dp(q4), nthreads)

    def update_indexed(self, idx, z, rdiag, quat_meas=None, mask=None, nthreads=0):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        m = len(idx)
        z = np.ascontiguousarray(z, dtype=np.float64)
        rdiag = np.ascontiguousarray(rdiag, dtype=np.float64)
        assert z.shape == (m, self.B) and rdiag.shape == (m, self.B)
        qm = None
        if quat_meas is not None:
            quat_meas = np.ascontiguousarray(quat_meas, dtype=np.float64)
            qm = _dp(quat_meas)
        mk = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mk = mask.ctypes.data
        b = self._c()
        lib().po_batch_update_indexed(C.byref(b), m, _ip(idx), _dp(z), _dp(rdiag), qm, mk, nthreads)

    def run_legodo(self, imu_stream, lo_stream, mask_stream, q4, nthreads=0):
        T = imu_stream.shape[0]
        imu_stream = np.ascontiguousarray(imu_stream, dtype=np.float64)
        lo_stream = np.ascontiguousarray(lo_stream, dtype=np.float64)
        assert imu_stream.shape == (T, 7, self.B) and lo_stream.shape == (T, 6, self.B)
        mk = None
        if mask_stream is not None:
            mask_stream = np.ascontiguousarray(mask_stream, dtype=np.uint8)
            mk = mask_stream.ctypes.data
        q4 = np.ascontiguousarray(q4, dtype=np.float64)
        b = self._c()
        return lib().po_batch_run_legodo(C.byref(b), T, _dp(imu_stream), _dp(lo_stream), mk, _dp(q4), nthreads)
