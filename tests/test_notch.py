"""IMU front end: the 3-stage IIR notch cascade of InsHandler::doFilter
(sensor_handlers.cpp:29-42,154-162; iir_notch.cpp:3-61).  CPU: the oracle's restatement against scipy (an
independent implementation of the same published design) and analytic properties.  GPU: pb_imu_notch vs oracle."""
import numpy as np
import pytest
from scipy import signal


def test_oracle_notch_coefficients_and_response_match_scipy(oracle):
    import ctypes as C
    L = oracle.lib()
    for f0 in (87.0, 174.0, 348.0):
        nf = oracle.Notch()
        L.po_notch_init(C.byref(nf), f0, 1000.0)
        # MATLAB/scipy iirnotch with BW = Wo  <=>  Q = 1 (iir_notch.cpp:6-8)
        b, a = signal.iirnotch(f0, 1.0, fs=1000.0)
        assert np.allclose(list(nf.b), b, rtol=1e-13) and np.allclose(list(nf.a), a, rtol=1e-13)
    rng = np.random.default_rng(0)
    acc = rng.normal(size=(400, 3))
    out = oracle.notch_cascade_run(acc, 87.0)
    ref = acc.copy()
    for i in range(3):
        b, a = signal.iirnotch(87.0 * 2 ** i, 1.0, fs=1000.0)
        ref = signal.lfilter(b, a, ref, axis=0)
    assert np.max(np.abs(out - ref)) < 1e-12


def test_oracle_notch_removes_the_vibration_and_keeps_dc(oracle):
    t = np.arange(3000) * 1e-3
    acc = np.stack([9.8 + 2.0 * np.sin(2 * np.pi * 87 * t), 0.5 + np.sin(2 * np.pi * 174 * t), np.sin(2 * np.pi * 348 * t)], 1)
    out = oracle.notch_cascade_run(acc, 87.0)
    tail = out[2000:]
    assert np.max(np.abs(tail[:, 0] - 9.8)) < 1e-3      # DC passes with unit gain, the 87 Hz line is gone
    assert np.max(np.abs(tail[:, 1] - 0.5)) < 1e-3      # 174 Hz: second stage
    assert np.max(np.abs(tail[:, 2])) < 1e-3            # 348 Hz: third stage


@pytest.mark.gpu
def test_device_notch_matches_oracle():
    import torch
    from oracle import po
    from pronto_amd.batch import BatchEstimator, PbError
    B, T = 200, 240
    rng = np.random.default_rng(3)
    t = np.arange(T) * 1e-3
    acc = 9.8 * (np.arange(3) == 2)[None, :, None] + rng.normal(size=(T, 3, B)) + np.sin(2 * np.pi * 87 * t)[:, None, None]
    est = BatchEstimator(B, n_states=15)
    with pytest.raises(PbError):
        est.imu_notch(np.zeros((1, 3, B)), np.zeros((3, B)))          # before init
    est.imu_notch_init(87.0, 1000.0)
    got = np.zeros((T, 3, B))
    dev = torch.device("cuda:0")
    k, call = 0, 0
    while k < T:
        npk = min(T - k, 1 + call % 4)           # 1..4 new packets per call, filter state carried across calls
        out = np.zeros((3, B))
        if call % 2:                              # alternate host and device buffers
            d_out = torch.zeros((3, B), dtype=torch.float64, device=dev)
            est.imu_notch(torch.from_numpy(np.ascontiguousarray(acc[k:k + npk])).to(dev), d_out)
            out = d_out.cpu().numpy()
        else:
            est.imu_notch(np.ascontiguousarray(acc[k:k + npk]), out)
        got[k + npk - 1] = out                    # only the newest filtered sample is returned
        k += npk
        call += 1
    idx = [b for b in (0, 63, 64, 199)]
    for b in idx:
        ref = po.notch_cascade_run(np.ascontiguousarray(acc[:, :, b]), 87.0)
        rows = np.where(np.any(got[:, :, b] != 0, axis=1))[0]
        assert len(rows) > 50
        assert np.max(np.abs(got[rows, :, b] - ref[rows])) < 1e-12
