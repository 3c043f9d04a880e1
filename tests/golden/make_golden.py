"""Generates the golden trajectory fixtures from the CPU oracle (oracle/pronto_oracle.c).

The reference itself ships no vectors for this path and cannot be built here, so these
fixtures pin the ORACLE (and, through it, the HIP path) across machines and rounds -- "parity unpinned"
still applies to the oracle-vs-reference link.  Inputs are not stored: they are regenerated from
pronto_amd/synth.py (counter-based, seed in the file).  Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import po  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402
from util import embed21, run_config  # noqa: E402

CASES = {
    # name: (n_states, B, T, stride, vo_every, sm_every)   -- BASELINE.json configs 2, 3, 5 in miniature
    "n15_legodo": (15, 64, 1000, 100, 0, 0),
    "n15_legodo_vo": (15, 64, 1000, 100, 32, 0),
    "n21_legodo_scanmatch": (21, 64, 1000, 100, 0, 25),
    "n15_long": (15, 8, 10000, 1000, 0, 0),
}


def generate(name):
    n, B, T, stride, vo, sm = CASES[name]
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    v21, P21 = embed21(vec, P0)
    ob = po.OracleBatch(v21, quat, P21)
    snaps = dict(vec=[], quat=[], pdiag=[], ll=[], pvv=[])
    for s in range(T // stride):
        run_config(ob, w, stride, vo_every=vo, sm_every=sm, k0=s * stride)
        snaps["vec"].append(ob.vec.copy())
        snaps["quat"].append(ob.quat.copy())
        snaps["pdiag"].append(np.stack([ob.cov[i, i] for i in range(21)]))
        snaps["pvv"].append(ob.cov[3:6, 6:12].copy())  # a few off-diagonal blocks: [v, chi|pos]
        snaps["ll"].append(ob.ll.copy())
    out = {k: np.stack(v) for k, v in snaps.items()}
    g, tol = po.constants()
    out["meta"] = np.array([n, B, T, stride, vo, sm, g, tol])
    return out


SMOOTHER_CASES = {"smoother_n15": (15, 8, 24, 1e-3), "smoother_n21": (21, 8, 24, 1e-3)}  # n, B, T, dt


def generate_smoother(name):
    """Smoothed posterior at the first and the middle step of a T-step forward pass + backward recursion."""
    from smoother_ref import oracle_backward_pass
    n, B, T, dt = SMOOTHER_CASES[name]
    w = Workload(B, n_states=n)
    res = oracle_backward_pass(po, w, n, T, B, dt, keep=(0, T // 2))
    g, tol = po.constants()
    out = {"meta": np.array([n, B, T, dt, g, tol])}
    for tag, k in (("first", 0), ("mid", T // 2)):
        out["vec_" + tag], out["quat_" + tag], out["cov_" + tag] = res[k]
    return out


if __name__ == "__main__":
    for name in SMOOTHER_CASES:
        out = generate_smoother(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
    for name in CASES:
        out = generate(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
