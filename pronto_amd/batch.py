"""Thin Python binding of the C ABI (include/pronto_batch.h) -- test/bench plumbing, not the product.

`BatchEstimator` mirrors the reference's estimator entry points for B filters at once
(state-estimator/src/mav_state_est/mav_state_est.hpp:20-22: addUpdate / getHeadState /
getMeasurementsLogLikelihood) by forwarding to pb_* one-to-one.  Arrays may be numpy (host, staged over
PCIe) or torch CUDA tensors (HBM-resident, read in place).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PB_DEVICE, PB_HOST, PB_HOST_BROADCAST, PB_R_DIAG, PB_R_DIAG_BROADCAST, PB_R_FULL


class PbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pronto_batch error %d: %s" % (code, msg))
        self.code = code


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a, dtype=np.float64, shape=None):
    """(pointer, mem) of a numpy array or a torch tensor; None -> (NULL, None).  The C ABI takes no lengths (it reads
    rows x batch elements), so a mis-shaped array would be an out-of-bounds read: `shape` is checked here."""
    if a is None:
        return None, None
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected an array of shape %s, got %s" % (tuple(shape), tuple(a.shape)))
    if _is_torch(a):
        import torch
        want = {np.float64: torch.float64, np.uint8: torch.uint8, np.float32: torch.float32, np.int32: torch.int32}[dtype]
        if a.dtype != want or not a.is_contiguous():
            raise TypeError("expected a contiguous %s tensor" % want)
        return C.c_void_p(a.data_ptr()), (PB_DEVICE if a.is_cuda else PB_HOST)
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("expected a C-contiguous numpy array of %s" % np.dtype(dtype))
    return C.c_void_p(a.ctypes.data), PB_HOST


def _ptr_block(a, rows=None, B=None, dtype=np.float64):
    """Like _ptr for a [rows, B] data block; a 1-D numpy array of `rows` values means ONE message for every filter of the
    batch (PB_HOST_BROADCAST: expanded on the device, nothing of batch size crosses PCIe)."""
    if isinstance(a, np.ndarray) and a.ndim == 1:
        p, _ = _ptr(a, dtype, shape=None if rows is None else (rows,))
        return p, PB_HOST_BROADCAST
    return _ptr(a, dtype, shape=None if rows is None else (rows, B))


def _same_mem(*mems):
    ms = {m for m in mems if m is not None}
    if len(ms) != 1:
        raise ValueError("all buffers of one call must live in the same memory space")
    return ms.pop()


class BatchEstimator:
    def __init__(self, batch, n_states=15, device=0, n_snapshots=1):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.pb_create(C.byref(h), n_states, batch, device, n_snapshots)
        if rc:
            raise PbError(rc, (self._L.pb_last_error(None) or b"").decode())
        self._h = h
        self._pinned = []
        self.B, self.n, self.device = batch, n_states, device
        # Launch on torch's current stream of that device (usually the null stream): device tensors handed to this
        # object are produced by torch kernels/copies on that stream, so stream order makes them visible.
        try:
            import torch
            if torch.cuda.is_available():
                self.set_stream(torch.cuda.current_stream(device).cuda_stream)
        except ImportError:
            pass

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self._L.pb_host_free(self._h, p)
            self._pinned = []
            self._L.pb_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise PbError(rc, (self._L.pb_last_error(self._h) or b"").decode())

    # --- plumbing ---
    def set_stream(self, stream_ptr):
        """hipStream_t handle, taken literally (0 = the null stream = torch's default stream)."""
        self._chk(self._L.pb_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_own_stream(self):
        self._chk(self._L.pb_use_own_stream(self._h))

    def set_constants(self, g, chi_tol):
        self._chk(self._L.pb_set_constants(self._h, g, chi_tol))

    def sync(self):
        self._chk(self._L.pb_sync(self._h))

    def hot_kernel(self):
        return self._L.pb_hot_kernel(self._h).decode()

    def run_block(self):
        """filters per block of run_legodo's cache-blocked order (0: the whole batch per launch)"""
        return self._L.pb_run_block(self._h)

    # --- posterior checkpoints, RTS smoother ---
    def history_reserve(self, n_slots):
        self._chk(self._L.pb_history_reserve(self._h, n_slots))

    def pinned_empty(self, shape, dtype=np.float64):
        """A numpy array in page-locked host memory (pb_host_alloc): PB_HOST blocks copied from it move at link rate.
        Freed when the estimator is closed."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self._L.pb_host_alloc(self._h, nbytes, C.byref(p)))
        self._pinned.append(p)
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def set_output_slot(self, slot):
        """The next update writes its posterior straight into checkpoint `slot` (which becomes the head): a checkpoint
        per update without a copy.  -1 cancels."""
        self._chk(self._L.pb_set_output_slot(self._h, int(slot)))

    def state_save(self, slot):
        self._chk(self._L.pb_state_save(self._h, slot))

    def state_restore(self, slot):
        self._chk(self._L.pb_state_restore(self._h, slot))

    def smooth_step(self, slot_next_pred, slot_next, slot_cur, slot_out, dt):
        self._chk(self._L.pb_smooth_step(self._h, slot_next_pred, slot_next, slot_cur, slot_out, dt))

    def smooth_log_slots(self, n_steps, stride):
        return self._L.pb_smooth_log_slots(n_steps, stride)

    def smooth_log(self, imu_stream, lo_stream, mask_stream, q4, dt, stride, first_slot=0, sink=None, timed=False):
        """Whole-log RTS smoothing with bounded memory (checkpoint and recompute): sink(step, slot) is called newest step first; read the
        slot with get_slot before returning from the sink.  Returns device ms if timed."""
        T = imu_stream.shape[0]
        pi, m1 = _ptr(imu_stream, shape=(T, 7, self.B))
        pl, m2 = _ptr(lo_stream, shape=(T, 6, self.B))
        pm, m3 = _ptr(mask_stream, np.uint8, shape=(T, self.B))
        if _same_mem(m1, m2, m3) != PB_DEVICE:
            raise ValueError("smooth_log needs device-resident streams")
        q = (C.c_double * 4)(*q4)
        ms = C.c_float(0)
        SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)
        cb = SINK((lambda user, step, slot: sink(step, slot))) if sink is not None else None
        self._chk(self._L.pb_smooth_log(self._h, T, stride, pi, pl, pm, q, dt, first_slot, C.cast(cb, C.c_void_p) if cb else None, None,
                                        C.byref(ms) if timed else None))
        return ms.value if timed else None

    def get_slot(self, slot, first=0, count=None, want_cov=True):
        """get_head for a posterior that lives in a checkpoint slot."""
        count = self.B - first if count is None else count
        n = self.n
        vec = np.empty((n, count))
        quat = np.empty((4, count))
        ll = np.empty(count)
        cov = np.empty((n, n, count)) if want_cov else None
        self._chk(self._L.pb_get_slot(self._h, slot, first, count, C.c_void_p(vec.ctypes.data), C.c_void_p(quat.ctypes.data),
                                      C.c_void_p(cov.ctypes.data) if want_cov else None, C.c_void_p(ll.ctypes.data), PB_HOST))
        if want_cov:
            cov = np.swapaxes(cov, 0, 1)
        return vec, quat, cov, ll

    # --- noise identification ---
    def set_process_noise_block(self, q_block):
        """q_block: torch CUDA tensor [4,B] (kept alive by the caller) or None."""
        if q_block is None:
            self._chk(self._L.pb_set_process_noise_block(self._h, None))
            return
        p, m = _ptr(q_block, shape=(4, self.B))
        if m != PB_DEVICE:
            raise ValueError("the process-noise block must be device memory")
        self._chk(self._L.pb_set_process_noise_block(self._h, p))

    def window_nll(self, idx, truth_vec, truth_quat, want_err=False):
        """(logdet, maha, nll) [3,B] (+ error vector [n,B]) of head (-) truth over the active indices."""
        m = len(idx)
        ia = (C.c_int * m)(*[int(i) for i in idx])
        out = np.empty((3, self.B))
        err = np.empty((self.n, self.B)) if want_err else None
        pv, m1 = _ptr(truth_vec, shape=(self.n, self.B))
        pq, m2 = _ptr(truth_quat, shape=(4, self.B))
        if _same_mem(m1, m2) != PB_HOST:
            raise ValueError("window_nll takes host truth arrays")
        self._chk(self._L.pb_window_nll(self._h, m, ia, pv, pq, C.c_void_p(out.ctypes.data),
                                        C.c_void_p(err.ctypes.data) if want_err else None, PB_HOST))
        return (out, err) if want_err else out

    # --- leg kinematic odometry ---
    def legodo_init(self, schmitt_low, schmitt_high, low_delay_us, high_delay_us, filter_contact_events=True):
        self._chk(self._L.pb_legodo_init(self._h, schmitt_low, schmitt_high, int(low_delay_us), int(high_delay_us),
                                         int(bool(filter_contact_events))))
        self._leg_mode = 0

    def legodo_update(self, utime, feet, forces, r_vxyz, r_vxyz_uncertain, delta_out=None, status_out=None, lo_out=None,
                      mask_out=None, zero_delta=False, after_predict=None):
        """leg_estimate::updateOdometry for every filter.  feet [14,B] (or [14] broadcast), forces [2,B] (or [2]);
        outputs are torch CUDA tensors: delta [7,B], status [B] (float64), lo block [6,B] + mask [B] (uint8).
        after_predict = an IMU block ([7,B] or [7]): the odometry is slaved to the orientation the filter WILL have after
        predict(that block), which is then run fused with the measurement produced here (step_legodo)."""
        pf, m1 = _ptr_block(feet, 14, self.B)
        pz, m2 = _ptr_block(forces, 2, self.B)
        outs = []
        for a, dt, shp in ((delta_out, np.float64, (7, self.B)), (status_out, np.float64, (self.B,)),
                           (lo_out, np.float64, self._leg_shapes()[0]), (mask_out, np.uint8, self._leg_shapes()[1])):
            p, m = _ptr(a, dt, shape=shp)
            if a is not None and m != PB_DEVICE:
                raise ValueError("legodo_update outputs must be device tensors")
            outs.append(p)
        if after_predict is not None:
            pi, mi = _ptr_block(after_predict, 7, self.B)
            self._chk(self._L.pb_legodo_update_after_predict(self._h, pi, mi, int(utime), pf, pz, _same_mem(m1, m2),
                                                             int(bool(zero_delta)), r_vxyz, r_vxyz_uncertain, *outs))
            return
        self._chk(self._L.pb_legodo_update(self._h, int(utime), pf, pz, _same_mem(m1, m2), int(bool(zero_delta)), r_vxyz,
                                           r_vxyz_uncertain, *outs))

    def legodo_set_contact_mode(self, standing=False, total_force=0.0, standing_schmitt_level=0.0, use_controller_input=False):
        self._chk(self._L.pb_legodo_set_contact_mode(self._h, int(bool(standing)), total_force, standing_schmitt_level,
                                                     int(bool(use_controller_input))))

    def legodo_set_zero_initial_velocity(self, ticks):
        self._chk(self._L.pb_legodo_set_zero_initial_velocity(self._h, int(ticks)))

    def legodo_set_control_contacts(self, n_contacts):
        """[2] int32 (every filter) or [2,B] int32 (numpy or torch CUDA)."""
        p, m = _ptr_block(n_contacts, 2, self.B, dtype=np.int32)
        self._chk(self._L.pb_legodo_set_control_contacts(self._h, p, m))

    def legodo_set_chain(self, n_left, n_right, joint_type, joint_row, origin_xyz_rpy, axis, adjustment_gain=None):
        n = n_left + n_right
        ty = (C.c_int * n)(*[int(v) for v in joint_type])
        rw = (C.c_int * n)(*[int(v) for v in joint_row])
        org = np.ascontiguousarray(origin_xyz_rpy, dtype=np.float64).reshape(n, 6)
        ax = np.ascontiguousarray(axis, dtype=np.float64).reshape(n, 3)
        gain = None if adjustment_gain is None else np.ascontiguousarray(adjustment_gain, dtype=np.float32).reshape(n)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.pb_legodo_set_chain(self._h, n_left, n_right, ty, rw, org.ctypes.data_as(dp), ax.ctypes.data_as(dp),
                                              None if gain is None else gain.ctypes.data_as(C.POINTER(C.c_float))))

    def legodo_set_measurement_mode(self, mode, r_xyz=0.0, r_vang=0.0, r_vang_uncertain=0.0):
        """LegOdoCommon's mode for the odometry and pair calls: 0 lin_rate (lo [6,B], mask [B]), 1 lin_rot_rate (lo [12,B], mask [B]),
        2 pos_and_lin_rate (lo [12,B], masks [2,B])."""
        self._chk(self._L.pb_legodo_set_measurement_mode(self._h, int(mode), float(r_xyz), float(r_vang), float(r_vang_uncertain)))
        self._leg_mode = int(mode)

    def _leg_shapes(self):
        mode = getattr(self, "_leg_mode", 0)
        return ((6, self.B), (self.B,)) if mode == 0 else ((12, self.B), (self.B,) if mode == 1 else (2, self.B))

    def legodo_update_joints(self, utime, joint_position, joint_effort, forces, r_vxyz, r_vxyz_uncertain, delta_out=None,
                             status_out=None, lo_out=None, mask_out=None, zero_delta=False, after_predict=None, position_out=None,
                             position_status_out=None):
        """leg_estimate::updateOdometry from a joint state: joint_position [rows,B] float32 (or [rows] broadcast),
        joint_effort the same or None, forces [2,B] float32 (or [2]).  Outputs as legodo_update."""
        rows = joint_position.shape[0]
        pj, m1 = _ptr_block(joint_position, rows, self.B, dtype=np.float32)
        pe, m2 = (None, None) if joint_effort is None else _ptr_block(joint_effort, rows, self.B, dtype=np.float32)
        pz, m3 = _ptr_block(forces, 2, self.B, dtype=np.float32)
        outs = []
        for a, dt, shp in ((delta_out, np.float64, (7, self.B)), (status_out, np.float64, (self.B,)),
                           (lo_out, np.float64, self._leg_shapes()[0]), (mask_out, np.uint8, self._leg_shapes()[1]),
                           (position_out, np.float64, (3, self.B)), (position_status_out, np.uint8, (self.B,))):
            p, m = _ptr(a, dt, shape=shp)
            if a is not None and m != PB_DEVICE:
                raise ValueError("legodo_update_joints outputs must be device tensors")
            outs.append(p)
        pi, mi = (None, PB_DEVICE) if after_predict is None else _ptr_block(after_predict, 7, self.B)
        self._chk(self._L.pb_legodo_update_joints(self._h, pi, mi, int(utime), rows, pj, pe, pz, _same_mem(m1, m2, m3),
                                                  int(bool(zero_delta)), r_vxyz, r_vxyz_uncertain, *outs))

    def step_legodo_joints(self, imu_block, q4, utime, joint_position, joint_effort, forces, r_vxyz, r_vxyz_uncertain, lo_out=None,
                           mask_out=None):
        """One call (one kernel where the context has it) per IMU + joint-state pair: pb_step_legodo_joints."""
        pi, mi = _ptr_block(imu_block, 7, self.B)
        rows = joint_position.shape[0]
        pj, m1 = _ptr_block(joint_position, rows, self.B, dtype=np.float32)
        pe, m2 = (None, None) if joint_effort is None else _ptr_block(joint_effort, rows, self.B, dtype=np.float32)
        pz, m3 = _ptr_block(forces, 2, self.B, dtype=np.float32)
        pl, _ = _ptr(lo_out, shape=self._leg_shapes()[0])
        pm, _ = _ptr(mask_out, np.uint8, shape=self._leg_shapes()[1])
        q = (C.c_double * 4)(*q4)
        self._chk(self._L.pb_step_legodo_joints(self._h, pi, mi, q, int(utime), rows, pj, pe, pz, _same_mem(m1, m2, m3), r_vxyz,
                                                r_vxyz_uncertain, pl, pm))

    def step_legodo_feet(self, imu_block, q4, utime, feet, forces, r_vxyz, r_vxyz_uncertain, lo_out=None, mask_out=None):
        pi, mi = _ptr_block(imu_block, 7, self.B)
        pf, m1 = _ptr_block(feet, 14, self.B)
        pz, m2 = _ptr_block(forces, 2, self.B)
        pl, _ = _ptr(lo_out, shape=self._leg_shapes()[0])
        pm, _ = _ptr(mask_out, np.uint8, shape=self._leg_shapes()[1])
        q = (C.c_double * 4)(*q4)
        self._chk(self._L.pb_step_legodo_feet(self._h, pi, mi, q, int(utime), pf, pz, _same_mem(m1, m2), r_vxyz, r_vxyz_uncertain, pl, pm))

    def calib_copy_checksum(self, reps=1):
        out = (C.c_uint64 * 2)()
        self._chk(self._L.pb_calib_copy_checksum(self._h, int(reps), out))
        return int(out[0]), int(out[1])

    def legodo_fk(self, joint_position, joint_effort, feet_out):
        rows = joint_position.shape[0]
        pj, m1 = _ptr_block(joint_position, rows, self.B, dtype=np.float32)
        pe, m2 = (None, None) if joint_effort is None else _ptr_block(joint_effort, rows, self.B, dtype=np.float32)
        po, mo = _ptr(feet_out, shape=(14, self.B))
        if mo != PB_DEVICE:
            raise ValueError("legodo_fk output must be a device tensor")
        self._chk(self._L.pb_legodo_fk(self._h, rows, pj, pe, _same_mem(m1, m2), po))

    def joint_filter_init(self, mode, process_noise_pos=0.01, process_noise_vel=0.01, observation_noise=5e-4):
        """mode: "lowpass" | "kalman" (state_estimator.legodo.filter_joint_positions, leg_estimate.cpp:43-61); the noise
        defaults are SimpleKalmanFilter's constructor defaults."""
        self._chk(self._L.pb_joint_filter_init(self._h, {"lowpass": 1, "kalman": 2}[mode], process_noise_pos, process_noise_vel,
                                               observation_noise))

    def joint_filter(self, utime, joint_position, joint_velocity, joint_effort, joint_position_out):
        """One joint-state message through the joint filters.  Per-robot [rows, B] blocks (numpy or device tensors) ->
        joint_position_out must be a device tensor [rows, B]; one robot's 1-D [rows] numpy arrays -> a 1-D numpy output."""
        rows = joint_position.shape[0]
        pj, m1 = _ptr_block(joint_position, rows, self.B, dtype=np.float32)
        pv, m2 = (None, None) if joint_velocity is None else _ptr_block(joint_velocity, rows, self.B, dtype=np.float32)
        pe, m3 = (None, None) if joint_effort is None else _ptr_block(joint_effort, rows, self.B, dtype=np.float32)
        mem = _same_mem(m1, m2, m3)
        if mem == PB_HOST_BROADCAST:
            po, mo = _ptr(joint_position_out, np.float32, shape=(rows,))
            if mo != PB_HOST:
                raise ValueError("one robot's message: the output is a numpy array [rows]")
        else:
            po, mo = _ptr(joint_position_out, np.float32, shape=(rows, self.B))
            if mo != PB_DEVICE:
                raise ValueError("per-robot blocks: the output must be a device tensor [rows, B]")
        self._chk(self._L.pb_joint_filter(self._h, int(utime), rows, pj, pv, pe, mem, po))

    def legodo_get(self, b):
        pose = (C.c_double * 7)()
        info = (C.c_int64 * 4)()
        self._chk(self._L.pb_legodo_get(self._h, b, pose, info))
        return np.array(pose), [int(v) for v in info]

    # --- IMU front end ---
    def imu_notch_init(self, notch_freq, fs=1000.0):
        self._chk(self._L.pb_imu_notch_init(self._h, notch_freq, fs))

    def imu_notch(self, accel_packets, accel_out):
        """accel_packets [n_packets,3,B] oldest first -> accel_out [3,B] (newest filtered sample)."""
        if accel_packets.ndim != 3:
            raise ValueError("accel_packets must be [n_packets, 3, B]")
        pi, m1 = _ptr(accel_packets, shape=(accel_packets.shape[0], 3, self.B))
        po_, m2 = _ptr(accel_out, shape=(3, self.B))
        self._chk(self._L.pb_imu_notch(self._h, accel_packets.shape[0], pi, po_, _same_mem(m1, m2)))

    # --- update objects ---
    def reset(self, vec, quat, cov, broadcast=False):
        """RBISResetUpdate.  vec [n,B], quat [4,B], cov [n,n,B] indexed [row,col,b] (or [n],[4],[n,n] broadcast)."""
        if _is_torch(cov):
            cov_cm = cov.transpose(0, 1).contiguous()
        else:
            cov_cm = np.ascontiguousarray(np.swapaxes(cov, 0, 1))  # column-major flat index c*n+r
        n, B = self.n, self.B
        pv, m1 = _ptr(vec, shape=(n,) if broadcast else (n, B))
        pq, m2 = _ptr(quat, shape=(4,) if broadcast else (4, B))
        pc, m3 = _ptr(cov_cm, shape=(n, n) if broadcast else (n, n, B))
        self._chk(self._L.pb_reset(self._h, pv, pq, pc, int(broadcast), _same_mem(m1, m2, m3)))

    def predict(self, imu_block, q4):
        p, m = _ptr_block(imu_block, 7, self.B)
        q = (C.c_double * 4)(*q4)
        self._chk(self._L.pb_predict(self._h, p, q, m))

    def _r(self, R, m):
        if isinstance(R, (list, tuple)) or (isinstance(R, np.ndarray) and R.ndim == 1):
            arr = np.ascontiguousarray(R, dtype=np.float64)
            if arr.shape == (m * m,) and m > 1:  # one full column-major R for every filter
                return arr, C.c_void_p(arr.ctypes.data), PB_R_FULL, PB_HOST_BROADCAST
            assert arr.shape == (m,)
            return arr, C.c_void_p(arr.ctypes.data), PB_R_DIAG_BROADCAST, None
        kind = PB_R_DIAG if R.shape[0] == m and len(R.shape) == 2 else PB_R_FULL
        p, mem = _ptr(R, shape=(m, self.B) if kind == PB_R_DIAG else (m * m, self.B))
        return R, p, kind, mem

    def update_indexed(self, idx, z, R, mask=None, quat_meas=None):
        """RBISIndexedMeasurement / RBISIndexedPlusOrientationMeasurement.  z [m,B];
        R: length-m list (broadcast diag), [m,B] per-filter diag, or [m*m,B] full column-major."""
        m = len(idx)
        ia = (C.c_int * m)(*[int(i) for i in idx])
        pz, mz = _ptr_block(z, m, self.B)
        _keep, pr, kind, mr = self._r(R, m)
        pm, mm = _ptr(mask, np.uint8, shape=(self.B,))
        if quat_meas is None:
            self._chk(self._L.pb_update_indexed(self._h, m, ia, pz, pr, kind, pm, _same_mem(mz, mr, mm)))
        else:
            pq, mq = _ptr_block(quat_meas, 4, self.B)
            self._chk(self._L.pb_update_indexed_orient(self._h, m, ia, pz, pr, kind, pq, pm, _same_mem(mz, mr, mm, mq)))

    def step_legodo(self, imu_block, lo_block, mask, q4):
        """IMU block and leg-odometry block (+ mask) may live in different spaces (e.g. a broadcast [7] IMU message and a
        per-filter device measurement from legodo_update(..., after_predict=imu)): pb_step_legodo_split."""
        pi, m1 = _ptr_block(imu_block, 7, self.B)
        pl, m2 = _ptr_block(lo_block, 6, self.B)
        pm, m3 = _ptr(mask, np.uint8, shape=(self.B,))
        q = (C.c_double * 4)(*q4)
        mlo = _same_mem(m2, m3)
        if m1 == mlo:
            self._chk(self._L.pb_step_legodo(self._h, pi, pl, pm, q, m1))
        else:
            self._chk(self._L.pb_step_legodo_split(self._h, pi, m1, pl, pm, mlo, q))

    def step_legodo_correct(self, imu_block, lo_block, mask, q4, corr_kind, z2, R2, quat_meas2, mask2=None):
        """predict + leg-odometry update + one more orientation update (corr_kind: _lib.PB_CORR_POS_ORIENT m=6 idx
        9,10,11,6,7,8 / PB_CORR_POS_YAW m=4 idx 9,10,11,8) in one state round trip.  R2: length-m list or [m,B]."""
        pi, m1 = _ptr_block(imu_block, 7, self.B)
        pl, m2 = _ptr_block(lo_block, 6, self.B)
        pm, m3 = _ptr(mask, np.uint8, shape=(self.B,))
        if corr_kind not in (_lib.PB_CORR_POS_ORIENT, _lib.PB_CORR_POS_YAW):
            raise ValueError("corr_kind must be PB_CORR_POS_ORIENT or PB_CORR_POS_YAW")
        m = 6 if corr_kind == _lib.PB_CORR_POS_ORIENT else 4
        pz, m4 = _ptr_block(z2, m, self.B)
        _keep, pr, kind, m5 = self._r(R2, m)
        pq, m6 = _ptr_block(quat_meas2, 4, self.B)
        pm2, m7 = _ptr(mask2, np.uint8, shape=(self.B,))
        q = (C.c_double * 4)(*q4)
        self._chk(self._L.pb_step_legodo_correct(self._h, pi, pl, pm, q, _same_mem(m1, m2, m3), int(corr_kind), pz, pr, kind,
                                                 pq, pm2, _same_mem(m4, m5, m6, m7)))

    def run_legodo(self, imu_stream, lo_stream, mask_stream, q4, timed=False):
        """n_steps fused steps from device-resident streams [T,7,B], [T,6,B], [T,B]; returns device ms if timed."""
        T = imu_stream.shape[0]
        pi, m1 = _ptr(imu_stream, shape=(T, 7, self.B))
        pl, m2 = _ptr(lo_stream, shape=(T, 6, self.B))
        pm, m3 = _ptr(mask_stream, np.uint8, shape=(T, self.B))
        if _same_mem(m1, m2, m3) != PB_DEVICE:
            raise ValueError("run_legodo needs device-resident streams")
        q = (C.c_double * 4)(*q4)
        ms = C.c_float(0)
        self._chk(self._L.pb_run_legodo(self._h, T, pi, pl, pm, q, C.byref(ms) if timed else None))
        return ms.value if timed else None

    def replay_legodo_fused(self, imu_stream, lo_stream, mask_stream, q4, steps_per_launch, timed=False):
        """Time-fused replay (P resident in registers for steps_per_launch steps); same results as run_legodo."""
        T = imu_stream.shape[0]
        pi, m1 = _ptr(imu_stream, shape=(T, 7, self.B))
        pl, m2 = _ptr(lo_stream, shape=(T, 6, self.B))
        pm, m3 = _ptr(mask_stream, np.uint8, shape=(T, self.B))
        if _same_mem(m1, m2, m3) != PB_DEVICE:
            raise ValueError("replay_legodo_fused needs device-resident streams")
        q = (C.c_double * 4)(*q4)
        ms = C.c_float(0)
        self._chk(self._L.pb_replay_legodo_fused(self._h, T, steps_per_launch, pi, pl, pm, q, C.byref(ms) if timed else None))
        return ms.value if timed else None

    def replay_legodo_checkpointed(self, imu_stream, lo_stream, mask_stream, q4, steps_per_launch, first_slot=0, timed=False):
        """The time-fused replay as a forward pass that keeps every posterior: step t also lands in checkpoint slot first_slot + t
        (history_reserve first), the state stays in registers -- pb_replay_legodo_checkpointed."""
        T = imu_stream.shape[0]
        pi, m1 = _ptr(imu_stream, shape=(T, 7, self.B))
        pl, m2 = _ptr(lo_stream, shape=(T, 6, self.B))
        pm, m3 = _ptr(mask_stream, np.uint8, shape=(T, self.B))
        if _same_mem(m1, m2, m3) != PB_DEVICE:
            raise ValueError("replay_legodo_checkpointed needs device-resident streams")
        q = (C.c_double * 4)(*q4)
        ms = C.c_float(0)
        self._chk(self._L.pb_replay_legodo_checkpointed(self._h, T, steps_per_launch, pi, pl, pm, q, int(first_slot), C.byref(ms) if timed else None))
        return ms.value if timed else None

    # --- fovis history ---
    def snapshot(self, slot=0):
        self._chk(self._L.pb_snapshot(self._h, slot))

    def compose_delta(self, slot, t, q, z_out, quat_out):
        pt, m1 = _ptr_block(t, 3, self.B)
        pq, m2 = _ptr_block(q, 4, self.B)
        pz, m3 = _ptr(z_out, shape=(3, self.B))
        po_, m4 = _ptr(quat_out, shape=(4, self.B))
        if m3 != PB_DEVICE or m4 != PB_DEVICE:
            raise ValueError("compose_delta outputs must be device tensors")
        self._chk(self._L.pb_compose_delta(self._h, slot, pt, pq, pz, po_, _same_mem(m1, m2)))

    # --- queries ---
    def get_head(self, first=0, count=None, want_cov=True):
        """getHeadState: (vec [n,c], quat [4,c], cov [n,n,c] indexed [row,col,b] or None, ll [c]) as numpy."""
        count = self.B - first if count is None else count
        n = self.n
        vec = np.empty((n, count))
        quat = np.empty((4, count))
        ll = np.empty(count)
        cov = np.empty((n, n, count)) if want_cov else None
        self._chk(self._L.pb_get_head(self._h, first, count, C.c_void_p(vec.ctypes.data), C.c_void_p(quat.ctypes.data),
                                      C.c_void_p(cov.ctypes.data) if want_cov else None, C.c_void_p(ll.ctypes.data),
                                      PB_HOST))
        if want_cov:
            cov = np.swapaxes(cov, 0, 1)  # stored [col,row,b] -> [row,col,b]
        return vec, quat, cov, ll

    def filter_state(self, b):
        q = (C.c_double * 4)()
        s = (C.c_double * 21)()
        c = (C.c_double * 441)()
        self._chk(self._L.pb_get_filter_state(self._h, b, q, s, c))
        return np.array(q), np.array(s), np.array(c).reshape(21, 21).T  # cov col-major -> [row,col]

    def calib_copy(self, reps=10):
        """device ms for `reps` plain copies of the state array (counter calibration / copy-rate probe)."""
        ms = C.c_float(0)
        self._chk(self._L.pb_calib_copy(self._h, reps, C.byref(ms)))
        return ms.value

    def state_checksum(self, slot=-1):
        """(sum, xor) over every 64-bit word of the device state (slot < 0: the head)."""
        out = (C.c_uint64 * 2)()
        self._chk(self._L.pb_state_checksum(self._h, int(slot), out))
        return int(out[0]), int(out[1])

    def summary(self):
        out = (C.c_double * 4)()
        self._chk(self._L.pb_summary(self._h, out))
        return np.array(out)
