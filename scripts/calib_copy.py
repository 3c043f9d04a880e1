"""Copy-rate probe + PMC calibration target: plain copies of a 1M-filter state array (1.17 GB, beyond the
256 MiB Infinity Cache) with k_step's access pattern; prints bytes moved per launch and the achieved GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402

B = 1 << 20
est = BatchEstimator(B, n_states=15)
q = np.zeros(4); q[0] = 1
est.reset(np.zeros(15), q, np.eye(15), broadcast=True)
est.calib_copy(2)
reps = 10
ms = est.calib_copy(reps)
nbytes = 140 * B * 8
print("calib_copy: %d bytes read + %d bytes written per launch, %.1f us/launch, %.1f GB/s (read+write)"
      % (nbytes, nbytes, ms * 1e3 / reps, 2 * nbytes * reps / (ms * 1e-3) / 1e9))
