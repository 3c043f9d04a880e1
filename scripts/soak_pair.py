"""Soak of the one-kernel IMU + joint-state pair (pb_step_legodo_joints) at 64k filters over thousands of ticks: masks against
the two-call sequence every tick, summaries to rounding at the end, and the pair kernel replayed from the same start must give
the same bits (pb_state_checksum).  tests/test_leg_odometry.py::test_pair_kernel_at_full_batch_size_on_gpu is the 40-tick
version of this that runs in the suite.   usage: python scripts/soak_pair.py [ticks=2000] [n_states=15] [filters=65536] [mode=0]
mode 1 / 2: LegOdoCommon's lin_rot_rate / pos_and_lin_rate inside the pair kernel against predict + six-row update(s) (the masks
bit-identical, the summaries to rounding: two 3-row blocks against one six-row update)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import legs  # noqa: E402
from pronto_amd import batch as pa  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
MODE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
SCHMITT = (475.0, 525.0, 7000, 7000)
R_VXYZ = (5.0, 10.0)
dev = torch.device("cuda:0")
chain = legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, legs.ATLAS_ROWS)
w = Workload(B, n_states=n, dt_us=2000)
vec, quat, P0 = w.initial_state()
q4 = w.process_noise()
msgs = legs.joint_gait(1, T, seed=33)
imus = [np.ascontiguousarray(w.imu_block(k % 64)[:, 0]) for k in range(T)]


def run(one_call):
    e = pa.BatchEstimator(B, n_states=n)
    e.reset(vec, quat, P0)
    e.legodo_init(*SCHMITT, True)
    e.legodo_set_chain(*chain)
    if MODE:
        e.legodo_set_measurement_mode(MODE, 0.05, 0.4, 0.9)
    lo = torch.zeros((12 if MODE else 6, B), dtype=torch.float64, device=dev)
    mk = torch.zeros((2, B) if MODE == 2 else (B,), dtype=torch.uint8, device=dev)
    acc = torch.zeros(T, dtype=torch.int64, device=dev)
    idx = [3, 4, 5, 0, 1, 2] if MODE == 1 else [9, 10, 11, 3, 4, 5]
    for k, (utime, jp, je, forces, _) in enumerate(msgs):
        a = (np.ascontiguousarray(jp[:, 0]), None, np.ascontiguousarray(forces[:, 0]))
        if one_call:
            e.step_legodo_joints(imus[k], q4, utime, *a, *R_VXYZ, lo, mk)
        elif MODE == 0:
            e.legodo_update_joints(utime, *a, *R_VXYZ, None, None, lo, mk, after_predict=imus[k])
            e.step_legodo(imus[k], lo, mk, q4)
        else:
            e.legodo_update_joints(utime, *a, *R_VXYZ, None, None, lo, mk, after_predict=imus[k])
            e.predict(imus[k], q4)
            e.update_indexed(idx, lo[0:6], lo[6:12], mask=mk if MODE == 1 else mk[0])
            if MODE == 2:
                e.update_indexed([3, 4, 5], lo[3:6].contiguous(), lo[9:12].contiguous(), mask=mk[1])
        acc[k] = mk.sum()
    out = (acc.cpu().numpy(), e.summary(), e.state_checksum())
    e.close()
    return out


m1, s1, c1 = run(True)
m2, s2, _ = run(False)
m3, s3, c3 = run(True)
ok = (np.array_equal(m1, m2) and np.array_equal(m1, m3) and s1[3] == 0 and s2[3] == 0 and c1 == c3 and np.array_equal(s1, s3) and
      abs(s1[0] - s2[0]) <= 1e-9 * abs(s2[0]) and abs(s1[1] - s2[1]) <= 1e-9 * abs(s2[1]))
print("n=%d: %d ticks x %d filters, %d updates applied, masks equal to the two-call sequence: %s, replay bit-identical: %s, "
      "sum loglik one call / two calls %.12e / %.12e, non-finite %d" % (n, T, B, int(m1.sum()), np.array_equal(m1, m2), c1 == c3 and np.array_equal(s1, s3),
                                                                         s1[0], s2[0], int(s1[3] + s2[3])))
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
