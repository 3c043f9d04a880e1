"""The two six-row updates of a 21-state batch at 64k filters, for counter runs: k_update_quad<CorrPosOrient> (compile-time list,
VO position_orient) and k_update_quad_list (LegOdoCommon's lin_rot_rate; PRONTO_BATCH_GENERIC_UPDATE=1: the run-time-list kernel k_update_quad_rt<6>); plus the m = 3 / 4 ones for scale."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
B, n = 65536, 21
w = Workload(B, n_states=n)
vec, quat, P0 = w.initial_state()
est = BatchEstimator(B, n_states=n)
est.reset(vec, quat, P0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lo, mask = w.legodo_block(0)
d_mask = up(mask)
z, qm, Rd = w.vo_block(0)
z6, d_qm, d_Rd = up(np.vstack([z, np.zeros((3, B))])), up(qm), up(Rd)
z6r, r6r = up(np.vstack([lo[0:3], 0.01 * np.ones((3, B))])), up(np.vstack([lo[3:6], 0.09 * np.ones((3, B))]))
z3, r3 = up(lo[0:3]), up(lo[3:6])
z3f = up(0.1 * np.ones((3, B)))
cases = [("k_update_quad m=6 orient (compile-time list 9,10,11,6,7,8)", lambda: est.update_indexed([9, 10, 11, 6, 7, 8], z6, d_Rd, quat_meas=d_qm)),
         ("k_update_quad_list m=6 (lin_rot_rate 3,4,5,0,1,2)", lambda: est.update_indexed([3, 4, 5, 0, 1, 2], z6r, r6r, mask=d_mask)),
         ("k_update_quad m=3 (3,4,5)", lambda: est.update_indexed([3, 4, 5], z3, r3, mask=d_mask)),
         ("k_update_quad_rt m=3 (2,9,20)", lambda: est.update_indexed([2, 9, 20], z3f, r3, mask=d_mask))]
for name, fn in cases:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    print("%-62s %6.1f us" % (name, (time.perf_counter() - t0) / 100 * 1e6))
