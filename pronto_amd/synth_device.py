"""The synthetic IMU / leg-odometry streams of synth.Workload generated ON THE DEVICE with torch ops (bench plumbing).

Same counter-based generator -- every sample is a pure function of (seed, filter, step, channel) -- so a rank's shard is the
slice of the single-GPU workload.  The 64-bit mixing runs in int64 (wrapping multiply / add have the same bits as uint64; the
logical right shifts are masked arithmetic ones); sin / cos / log / sqrt are the device's, so the values agree with the numpy
generator to rounding, not bit for bit.  Why: at N ranks on one host every rank used to generate its (W + K) x 65 536 x 13
doubles with numpy before the first barrier (15-20 s of one core each); here that is a few hundred milliseconds of GPU time.
"""
import numpy as np
import torch

from . import synth

_I64 = torch.int64


def _c(v):
    """python int (as uint64) -> the int64 with the same bits"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_M1, _M2, _GOLD, _CH = (_c(int(x)) for x in (synth._M1, synth._M2, synth._GOLD, synth._CH))


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x):
    x = x ^ _lsr(x, 30)
    x = x * _M1
    x = x ^ _lsr(x, 27)
    x = x * _M2
    return x ^ _lsr(x, 31)


class DeviceWorkload:
    """streams(k0, T) of synth.Workload(B, b0, dt_us, n_states, seed) as torch tensors on `device`."""

    def __init__(self, B, b0=0, dt_us=1000, n_states=15, seed=synth.SEED, device="cuda"):
        self.host = synth.Workload(B, b0=b0, dt_us=dt_us, n_states=n_states, seed=seed)
        self.B, self.dt_us, self.device = B, dt_us, torch.device(device)
        h = self.host
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        for name in ("rpy_c", "rpy_f", "rpy_A", "rpy_phi", "pos_f", "pos_A", "pos_phi", "bg", "ba", "gait_T", "gait_off"):
            setattr(self, name, up(getattr(h, name)))
        b = torch.arange(b0, b0 + B, dtype=_I64, device=self.device)
        self.key = _mix(b ^ _c(int(seed)))      # _mix(U64(seed) ^ b)

    def _uniform(self, k, ch):
        """k [T,1] int64 steps, ch int -> [T,B]"""
        u = _mix(self.key[None, :] + k * _GOLD + _c(ch * (_CH & ((1 << 64) - 1))))
        return (_lsr(u, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)

    def _normal(self, k, ch):
        u1, u2 = self._uniform(k, 2 * ch), self._uniform(k, 2 * ch + 1)
        return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)

    def _truth(self, t):
        """t [T,1,1] seconds -> omega, f_b, vel_b each [T,3,B]"""
        w2 = 2 * np.pi * self.rpy_f
        ang = w2 * t + self.rpy_phi
        rpy = self.rpy_c + self.rpy_A * torch.sin(ang)
        rpy_d = self.rpy_A * w2 * torch.cos(ang)
        wp = 2 * np.pi * self.pos_f
        angp = wp * t + self.pos_phi
        vel_w = self.pos_A * wp * torch.cos(angp)
        acc_w = -self.pos_A * wp * wp * torch.sin(angp)
        r, p, y = rpy[:, 0], rpy[:, 1], rpy[:, 2]
        cr, sr, cp, sp, cy, sy = torch.cos(r), torch.sin(r), torch.cos(p), torch.sin(p), torch.cos(y), torch.sin(y)
        R = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
             [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
             [-sp, cp * sr, cp * cr]]
        rd, pd, yd = rpy_d[:, 0], rpy_d[:, 1], rpy_d[:, 2]
        omega = torch.stack([rd - yd * sp, pd * cr + yd * sr * cp, -pd * sr + yd * cr * cp], dim=1)
        g = acc_w.clone()
        g[:, 2] += synth.G_VAL
        rt = lambda v: torch.stack([R[0][i] * v[:, 0] + R[1][i] * v[:, 1] + R[2][i] * v[:, 2] for i in range(3)], dim=1)  # R^T v
        return omega, rt(g), rt(vel_w)

    def streams(self, k0, T):
        """imu [T,7,B], legodo [T,6,B] (float64), mask [T,B] (uint8) for steps k0 .. k0+T-1"""
        W = synth.Workload
        k = torch.arange(k0, k0 + T, dtype=_I64, device=self.device)[:, None]
        t0 = (k.to(torch.float64) * self.dt_us * 1e-6)[:, :, None]
        t1 = ((k + 1).to(torch.float64) * self.dt_us * 1e-6)[:, :, None]
        omega, f_b, _ = self._truth(t0)
        imu = torch.empty((T, 7, self.B), dtype=torch.float64, device=self.device)
        for i in range(3):
            imu[:, i] = omega[:, i] + self.bg[i] + W.SIG_GYRO * self._normal(k, i)
            imu[:, 3 + i] = f_b[:, i] + self.ba[i] + W.SIG_ACCEL * self._normal(k, 3 + i)
        imu[:, 6] = ((k + 1) * self.dt_us - k * self.dt_us).to(torch.float64) * 1E-6
        _, _, vel_b = self._truth(t1)
        lo = torch.empty((T, 6, self.B), dtype=torch.float64, device=self.device)
        for i in range(3):
            lo[:, i] = vel_b[:, i] + W.SIG_LEGODO * self._normal(k, 10 + i)
        ph = torch.remainder(t1[:, 0] + self.gait_off, self.gait_T)
        brk = 0.5 * self.gait_T
        uncertain = (ph >= brk) & (ph < brk + W.BREAK_UNCERTAIN)
        skip = ph < W.STRIKE_BLACKOUT
        f64 = lambda v: torch.tensor(v, dtype=torch.float64, device=self.device)  # (python scalars would make a float32 result)
        r = torch.where(uncertain & ~skip, f64(W.R_VXYZ_UNCERTAIN ** 2), f64(W.R_VXYZ ** 2))
        lo[:, 3:6] = r[:, None, :]
        return imu, lo, (~skip).to(torch.uint8)
