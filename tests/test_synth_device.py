"""The device-side stream generator of bench.py (pronto_amd/synth_device.py, torch ops) against the numpy generator it
restates (pronto_amd/synth.py): the same counter-based samples, to rounding; shards are slices of the whole workload."""
import numpy as np
import pytest

from pronto_amd.synth import Workload


@pytest.mark.parametrize("n", [15, 21])
def test_device_generator_matches_numpy_generator_on_cpu(n):
    import torch
    from pronto_amd.synth_device import DeviceWorkload
    B, b0, k0, T = 96, 1000, 37, 9
    dw = DeviceWorkload(B, b0=b0, n_states=n, device="cpu")
    imu, lo, mask = dw.streams(k0, T)
    himu, hlo, hmask = Workload(B, b0=b0, n_states=n).streams(k0, T)
    assert np.array_equal(mask.numpy(), hmask) and 0 < hmask.mean() < 1
    assert np.max(np.abs(imu.numpy() - himu)) < 1e-12 and np.array_equal(imu.numpy()[:, 6], himu[:, 6])
    assert np.max(np.abs(lo.numpy() - hlo)) < 1e-12 and len(np.unique(hlo[:, 3])) == 2
    # a shard is a slice: filters [b0+32, b0+64) generated on their own
    simu, slo, smask = DeviceWorkload(32, b0=b0 + 32, n_states=n, device="cpu").streams(k0, T)
    assert torch.equal(simu, imu[:, :, 32:64]) and torch.equal(slo, lo[:, :, 32:64]) and torch.equal(smask, mask[:, 32:64])


@pytest.mark.gpu
def test_device_generator_on_gpu():
    import torch
    from pronto_amd.synth_device import DeviceWorkload
    B, T = 4096, 12
    imu, lo, mask = DeviceWorkload(B, n_states=21, device="cuda:0").streams(5, T)
    himu, hlo, hmask = Workload(B, n_states=21).streams(5, T)
    assert np.array_equal(mask.cpu().numpy(), hmask)
    assert np.max(np.abs(imu.cpu().numpy() - himu)) < 1e-11 and np.max(np.abs(lo.cpu().numpy() - hlo)) < 1e-11
