"""RTS smoother (rbis.cpp:234-266 + the backward pass of mav_state_est.cpp:98-189): the device
kernels behind pb_smooth_step -- k_smooth_wide (default for 15 states: persistent workgroups, one wave per SIMD, rows through LDS),
k_smooth_lane (default for 21 states; PRONTO_SMOOTH_KERNEL=lane for 15: one lane per filter, role waves, unpivoted LDL^T), k_smooth_reg
(PRONTO_SMOOTH_KERNEL=reg) and its pivoted variant (PRONTO_SMOOTH_PIVOT=1, Eigen's diagonal pivoting like the reference's
.ldlt()) -- on posterior checkpoints against the oracle's dense restatement po_ekf_smoothing_step.

Tolerance: TOL = 1e-9 relative (block-wise).  The step inverts P^-_{k+1} (condition number up to ~1e7 here: variances from
1e-8 to 0.25); observed over 23 backward steps: 2e-15 .. 4e-15 for all three kernels, the worst error is printed."""
import os

import numpy as np
import pytest

from smoother_ref import oracle_backward_pass, oracle_forward, oracle_smooth_step, start_of
from util import rel

from pronto_amd.synth import Workload

pytestmark = pytest.mark.gpu
TOL = 1e-9   # observed: 2e-15 .. 4e-15 (block-relative) for all three kernels over 23 backward steps


@pytest.mark.parametrize("n,zero_bias", [(15, False), (21, False), (21, True)])
def test_backward_pass_matches_oracle(oracle, n, zero_bias):
    """Forward pass on the GPU with a checkpoint after every update (INS and legodo), then the backward recursion of
    EKFSmoothBackwardsPass with pb_smooth_step; the same recursion with the oracle on the oracle's forward pass."""
    from pronto_amd.batch import BatchEstimator
    B, T, dt = 37, 24, 1e-3
    w = Workload(B, n_states=n)
    hist = oracle_forward(oracle, w, n, T, B, zero_bias)
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    vec, quat, P0, q4 = start_of(w, zero_bias)  # zero_bias: P^-'s bias blocks are exactly 0 -> replaced by I in the solve
    est.reset(vec, quat, P0)
    est.history_reserve(2 * T + 2)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est.predict(imu, q4)
        est.state_save(2 * k)            # posterior of the INS update (the "pred" of step k)
        est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
        est.state_save(2 * k + 1)        # filtered posterior of step k
    S = [2 * T, 2 * T + 1]               # ping-pong slots for the smoothed posterior
    # backward recursion: next = filtered(T-1), next_pred = pred(T-1)
    nxt_slot, nxt_o = 2 * (T - 1) + 1, hist[T - 1][1]
    worst = 0.0
    for k in range(T - 2, -1, -1):
        out_slot = S[k % 2]
        est.smooth_step(2 * (k + 1), nxt_slot, 2 * k + 1, out_slot, dt)
        nxt_o = oracle_smooth_step(oracle, hist[k + 1][0], nxt_o, hist[k][1], dt)
        nxt_slot = out_slot
        est.state_restore(out_slot)
        v, q, P, ll = est.get_head()
        assert rel(v, nxt_o[0][:n]) < TOL and rel(q, nxt_o[1]) < TOL, (k, rel(v, nxt_o[0][:n]))
        assert rel(P, nxt_o[2][:n, :n]) < TOL, (k, rel(P, nxt_o[2][:n, :n]))
        worst = max(worst, rel(v, nxt_o[0][:n]), rel(q, nxt_o[1]), rel(P, nxt_o[2][:n, :n]))
        # smoothing never increases the uncertainty: P_filtered - P_smoothed is PSD
        d = hist[k][1][2][:n, :n, 0] - P[:, :, 0]
        assert np.linalg.eigvalsh(0.5 * (d + d.T)).min() > -1e-9 * np.abs(hist[k][1][2]).max()
    print("smoother backward pass n=%d zero_bias=%s: worst relative error against the oracle over %d steps: %.2e" % (n, zero_bias, T - 1, worst))
    est.close()


@pytest.mark.parametrize("blk,tells", [(15, False), (18, True)])
def test_bias_block_fix_applies_to_the_factorised_matrix_only(oracle, blk, tells):
    """rbis.cpp:244-251 replaces a bias block of P^-_{k+1} by the identity when ONE of its variances is below 1e-11 -- in
    the matrix it factorises; D = P^s_{k+1} - P^-_{k+1} keeps the uncorrected P^-.  With the other bias variances and the
    cross-covariances non-zero the two readings differ at the 4e-4 level for the accel-bias block of this workload (2e-8
    for the gyro-bias block: checked for parity only), so this pins the kernel to the reference's."""
    from pronto_amd.batch import BatchEstimator
    n, B, T, dt = 21, 37, 3, 1e-3
    w = Workload(B, n_states=n)
    hist = oracle_forward(oracle, w, n, T, B)
    nxt_pred = tuple(a.copy() for a in hist[1][0])
    nxt, cur = hist[1][1], hist[0][1]
    nxt_pred[2][blk + 1, blk + 1, :] = 1e-13          # one variance of the block below the threshold ...
    assert np.all(nxt_pred[2][blk, blk, :] > 1e-9) and np.abs(nxt_pred[2][blk, :blk, :]).max() > 1e-12  # ... the rest alive
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    est.history_reserve(4)
    for slot, (v, q, P) in enumerate((nxt_pred, nxt, cur)):
        est.reset(np.ascontiguousarray(v[:n]), np.ascontiguousarray(q), np.ascontiguousarray(P[:n, :n]))
        est.state_save(slot)
    est.smooth_step(0, 1, 2, 3, dt)
    est.state_restore(3)
    v, q, P, _ = est.get_head()
    ov, oq, oP = oracle_smooth_step(oracle, nxt_pred, nxt, cur, dt)
    assert rel(v, ov[:n]) < TOL and rel(q, oq) < TOL and rel(P, oP[:n, :n]) < TOL
    # the other reading (D from the corrected matrix) is far outside the tolerance: the test can tell them apart
    fixed = tuple(a.copy() for a in nxt_pred)
    fixed[2][blk:blk + 3, blk:blk + 3, :] = np.eye(3)[:, :, None]
    _, _, wrongP = oracle_smooth_step(oracle, fixed, nxt, cur, dt)
    assert not tells or rel(wrongP[:n, :n], oP[:n, :n]) > 1e3 * TOL
    est.close()


def test_smooth_step_identity_when_next_equals_prediction(oracle):
    """If the smoothed next state equals its prediction there is nothing to propagate back: out == cur."""
    from pronto_amd.batch import BatchEstimator
    B, n = 70, 15
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    est.history_reserve(3)
    est.state_save(0)
    est.predict(w.imu_block(0), w.process_noise())
    est.state_save(1)
    est.smooth_step(1, 1, 0, 2, 1e-3)
    est.state_restore(0)
    v0, q0, P0_, _ = est.get_head()
    est.state_restore(2)
    v2, q2, P2, _ = est.get_head()
    assert rel(v2, v0) < 1e-12 and rel(q2, q0) < 1e-12 and rel(P2, P0_) < 1e-12
    with pytest.raises(Exception):
        est.smooth_step(1, 1, 0, 1, 1e-3)    # out must not alias a k+1 slot


@pytest.mark.parametrize("n", [15, 21])
def test_backward_pass_matches_golden_fixture(oracle, n):
    """The committed smoother fixtures (tests/golden/smoother_n*.npz, written by make_golden.py from the oracle): forward
    pass + full backward recursion on the GPU, compared at the two stored steps."""
    from pronto_amd.batch import BatchEstimator
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smoother_n%d.npz" % n))
    nn, B, T = (int(v) for v in gold["meta"][:3])
    dt = float(gold["meta"][3])
    assert nn == n and (gold["meta"][4], gold["meta"][5]) == oracle.constants()
    w = Workload(B, n_states=n)
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    vec, quat, P0 = w.initial_state()
    est.reset(vec, quat, P0)
    est.history_reserve(2 * T + 2)
    q4 = w.process_noise()
    for k in range(T):
        lo, mask = w.legodo_block(k)
        est.predict(w.imu_block(k), q4)
        est.state_save(2 * k)
        est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
        est.state_save(2 * k + 1)
    nxt_slot = 2 * (T - 1) + 1
    for k in range(T - 2, -1, -1):
        out_slot = 2 * T + (k % 2)
        est.smooth_step(2 * (k + 1), nxt_slot, 2 * k + 1, out_slot, dt)
        nxt_slot = out_slot
        if k in (T // 2, 0):
            tag = "mid" if k == T // 2 and k != 0 else "first"
            est.state_restore(out_slot)
            v, q, P, ll = est.get_head()
            assert rel(v, gold["vec_" + tag][:n]) < TOL and rel(q, gold["quat_" + tag]) < TOL
            assert rel(P, gold["cov_" + tag][:n, :n]) < TOL
    est.close()


@pytest.mark.parametrize("n", [15, 21])
def test_smooth_step_in_place_equals_out_of_place(oracle, n):
    """pb_smooth_step allows slot_out == slot_cur (pronto_batch.h).  k_smooth_lane reads the filtered checkpoint in three places (the
    right-hand sides, the state, P_k(r, c) inside the products) while its role waves store P^s(r, c) as they go: every entry has to be
    read by the thread that overwrites it, before it does.  In place == out of place, bit for bit, over a ragged batch."""
    from pronto_amd.batch import BatchEstimator
    B, dt = 200, 1e-3
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    est.history_reserve(5)
    q4 = w.process_noise()
    for k in range(3):
        lo, mask = w.legodo_block(k)
        est.step_legodo(w.imu_block(k), lo, mask, q4)
    est.state_save(0)
    est.state_save(4)                       # a second copy of the filtered posterior: smoothed in place below
    est.predict(w.imu_block(3), q4)
    est.state_save(1)
    lo, mask = w.legodo_block(3)
    est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
    est.state_save(2)
    est.smooth_step(1, 2, 0, 3, dt)
    est.smooth_step(1, 2, 4, 4, dt)
    est.state_restore(3)
    a = est.get_head()
    est.state_restore(4)
    b = est.get_head()
    est.state_restore(0)
    f = est.get_head()
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[2], f[2])   # (and the step did something)
    est.close()


@pytest.mark.parametrize("n", [15, 21])
def test_full_size_smooth_step_equals_its_shard(oracle, n):
    """BASELINE's batch size (65 536 filters = 1 024 tiles of 64 filters, one lane per filter, 4 / 8 role waves per tile): the
    smoothed posterior of a window of filters must be bit-identical to the same filters smoothed alone in a small batch
    (another lane, another tile), stay finite everywhere, and never increase the uncertainty.  (Parity with the oracle on
    small batches: the tests above.)"""
    from pronto_amd.batch import BatchEstimator
    dt = 1e-3

    def run(B, b0, first, count):
        w = Workload(B, b0=b0, n_states=n)
        est = BatchEstimator(B, n_states=n)
        est.set_constants(*oracle.constants())
        vec, quat, P0 = w.initial_state()
        est.reset(vec, quat, P0)
        est.history_reserve(4)
        q4 = w.process_noise()
        for k in range(4):
            lo, mask = w.legodo_block(k)
            est.step_legodo(w.imu_block(k), lo, mask, q4)
        est.state_save(0)
        est.predict(w.imu_block(4), q4)
        est.state_save(1)
        lo, mask = w.legodo_block(4)
        est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
        est.state_save(2)
        est.smooth_step(1, 2, 0, 3, dt)
        est.state_restore(0)
        filt = est.get_head(first, count)
        est.state_restore(3)
        sm = est.get_head(first, count)
        summ = est.summary()
        est.close()
        return filt, sm, summ

    B, b0, Bs = 65536, 40007, 45
    filt, whole, summ = run(B, 0, b0, Bs)
    assert summ[3] == 0                      # no non-finite entry anywhere in the 64k smoothed states
    _, shard, _ = run(Bs, b0, 0, Bs)
    for a, b in zip(whole[:3], shard[:3]):
        assert np.array_equal(a, b)
    for j in range(Bs):
        d = filt[2][:, :, j] - whole[2][:, :, j]
        assert np.linalg.eigvalsh(0.5 * (d + d.T)).min() > -1e-9 * np.abs(filt[2][:, :, j]).max()


def test_mfma_f64_lane_maps(tmp_path):
    """The smoother's products rely on the operand / result lane maps of v_mfma_f64_16x16x4_f64 (which differ from the f32
    forms): scripts/mfma_f64_probe.hip checks them with random data on this box."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mfma_f64_probe")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-Wno-unused-value",
                    os.path.join(root, "scripts", "mfma_f64_probe.hip"), "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], check=True, timeout=120, capture_output=True, text=True).stdout
    assert "lane maps" in out and "(OK)" in out, out


@pytest.mark.parametrize("n", [15, 21])
def test_smooth_log_equals_the_all_checkpoints_pass_bit_for_bit(oracle, n):
    """pb_smooth_log -- EKFSmoothBackwardsPass over a whole log with bounded memory (checkpoint and recompute: the forward pass keeps
    the posterior in front of every 7th step, the backward pass re-runs the log stretch by stretch into a window of 14 slots) --
    against the all-checkpoints pass (a slot per update, 2 T slots): the same smoothed posterior at every step, BIT FOR BIT, and the
    oracle's backward recursion within TOL.  50 steps in stretches of 7: a last stretch of one step, boundaries inside the log."""
    import torch
    from pronto_amd.batch import BatchEstimator
    B, T, K, dt = 37, 50, 7, 1e-3
    dev = torch.device("cuda:0")
    w = Workload(B, n_states=n)
    vec, quat, P0, q4 = start_of(w)
    imu, lo, mask = w.streams(0, T)
    # the all-checkpoints pass: every process-step posterior and every update posterior in a slot of its own
    ref = BatchEstimator(B, n_states=n)
    ref.set_constants(*oracle.constants())
    ref.reset(vec, quat, P0)
    ref.history_reserve(2 * T + 2)
    for k in range(T):
        ref.set_output_slot(2 * k)
        ref.predict(imu[k], q4)
        ref.set_output_slot(2 * k + 1)
        ref.update_indexed([3, 4, 5], np.ascontiguousarray(lo[k][0:3]), np.ascontiguousarray(lo[k][3:6]), mask=mask[k])
    want = {}
    nxt = 2 * (T - 1) + 1
    for k in range(T - 2, -1, -1):
        out = 2 * T + (k % 2)
        ref.smooth_step(2 * (k + 1), nxt, 2 * k + 1, out, dt)
        want[k] = ref.get_slot(out)
        nxt = out
    final = ref.get_slot(2 * (T - 1) + 1)
    ref.close()
    # checkpoint and recompute
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    need = est.smooth_log_slots(T, K)
    assert need == (T + K - 1) // K + 2 * K + 4 and need < 2 * T
    est.history_reserve(need + 1)
    got, order = {}, []

    def sink(step, slot):
        order.append(step)
        got[step] = est.get_slot(slot)
    est.smooth_log(*(torch.from_numpy(a).to(dev) for a in (imu, lo, mask)), q4, dt, K, first_slot=1, sink=sink)
    assert order == list(range(T - 2, -1, -1))
    for k in range(T - 1):
        for a, b in zip(got[k], want[k]):
            assert np.array_equal(a, b), k
    for a, b in zip(est.get_head(), final):      # the head afterwards: the newest filtered posterior
        assert np.array_equal(a, b)
    # ... and the oracle's recursion
    keep = {0, 1, K - 1, K, 3 * K, T - 3, T - 2}
    ora = oracle_backward_pass(oracle, w, n, T, B, dt, keep)
    worst = 0.0
    for k in keep:
        v, q, P, _ = got[k]
        worst = max(worst, rel(v, ora[k][0][:n]), rel(q, ora[k][1]), rel(P, ora[k][2][:n, :n]))
    assert worst < TOL, worst
    est.close()


def test_whole_log_smoothing_of_10000_steps_with_bounded_memory(oracle):
    """The reference's "-S" smooths the ENTIRE log (fusion.cpp:268-269, lcm_front_end.cpp:168-203).  10 000 steps x 4 096 15-state
    filters would need 20 000 posterior slots (92 GB) with a slot per update; pb_smooth_log runs it in 157 + 128 + 4 = 289 slots
    (1.3 GB).  Sampled filters at sampled steps -- the ends of the log, both sides of stretch boundaries -- against the oracle's
    forward + backward recursion over the same 10 000 steps of the same inputs (copied back from the device streams)."""
    import torch
    from pronto_amd.batch import BatchEstimator
    from pronto_amd.synth_device import DeviceWorkload
    from util import embed21
    n, B, T, K, dt = 15, 4096, 10000, 64, 1e-3
    dw = DeviceWorkload(B, n_states=n, device="cuda:0")
    w = dw.host
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    d_imu, d_lo, d_mask = dw.streams(0, T)
    est = BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    need = est.smooth_log_slots(T, K)
    assert need == 289
    est.history_reserve(need)
    filters = [0, 2049, B - 1]
    steps = {0, 1, K - 1, K, K + 1, 5000, T - K - 1, T - K, T - 3, T - 2}
    got = {}

    def sink(step, slot):
        if step in steps:
            got[step] = [est.get_slot(slot, first=b, count=1) for b in filters]
    ms = est.smooth_log(d_imu, d_lo, d_mask, q4, dt, K, sink=sink, timed=True)
    print("pb_smooth_log: %d steps x %d filters, stride %d, %d slots: %.0f ms = %.3f us per smoothed filter-step (forward + recompute + smoother)"
          % (T, B, K, need, ms, ms * 1e3 / (T * B)))
    assert sorted(got) == sorted(steps)
    # the oracle on the sampled filters: forward pass keeping (pred, filtered), then the backward recursion
    sel = torch.tensor(filters, device=d_imu.device)
    imu, lo, mask = (t.index_select(t.dim() - 1, sel).cpu().numpy() for t in (d_imu, d_lo, d_mask))
    v21, P21 = embed21(np.ascontiguousarray(vec[:, filters]), np.ascontiguousarray(P0[:, :, filters]))
    ob = oracle.OracleBatch(v21, np.ascontiguousarray(quat[:, filters]), P21)
    hist = []
    for k in range(T):
        ob.predict(np.ascontiguousarray(imu[k]), q4)
        pred = (ob.vec.copy(), ob.quat.copy(), ob.cov.copy())
        ob.update_indexed([3, 4, 5], np.ascontiguousarray(lo[k][0:3]), np.ascontiguousarray(lo[k][3:6]), mask=np.ascontiguousarray(mask[k]))
        hist.append((pred, (ob.vec.copy(), ob.quat.copy(), ob.cov.copy())))
    nxt = hist[T - 1][1]
    worst = 0.0
    for k in range(T - 2, -1, -1):
        nxt = oracle_smooth_step(oracle, hist[k + 1][0], nxt, hist[k][1], dt)
        if k in steps:
            for i in range(len(filters)):
                v, q, P, _ = got[k][i]
                worst = max(worst, rel(v[:, 0], nxt[0][:n, i]), rel(q[:, 0], nxt[1][:, i]), rel(P[:, :, 0], nxt[2][:n, :n, i]))
    print("worst relative error against the oracle at %d sampled (step, filter) pairs: %.2e" % (len(steps) * len(filters), worst))
    assert worst < TOL, worst
    est.close()


_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from pronto_amd.batch import BatchEstimator
from pronto_amd.synth import Workload
B = %(B)d
w = Workload(B, n_states=15)
est = BatchEstimator(B, n_states=15)
vec, quat, P0 = w.initial_state()
est.reset(vec, quat, P0)
est.history_reserve(6)
q4 = w.process_noise()
for k in range(2):
    est.predict(w.imu_block(k), q4)
    est.state_save(2 * k)
    lo, mask = w.legodo_block(k)
    est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
    est.state_save(2 * k + 1)
est.smooth_step(2, 3, 1, 4, 1e-3)
est.smooth_step(2, 3, 1, 1, 1e-3)     # in place: slot_out == slot_cur
out = []
for slot in (4, 1):
    est.state_restore(slot)
    v, q, P, ll = est.get_head()
    out += [v, q, P, ll]
np.savez(%(path)r, *out)
est.close()
"""


@pytest.mark.parametrize("B", [1, 100, 16384 + 64 * 3 + 5, 65536 + 37])
def test_wide_and_lane_kernels_agree_at_ragged_batch_sizes(tmp_path, B):
    """k_smooth_wide<15> is persistent (a workgroup walks tiles blockIdx.x, + gridDim.x, ...) and moves whole rows of a tile: partly filled
    last tiles, fewer tiles than CUs, workgroups with an uneven number of tiles, and the in-place call (slot_out == slot_cur) must give what
    k_smooth_lane<15> gives (PRONTO_SMOOTH_KERNEL=lane; the switch is read once per process, hence the child processes).  The two kernels
    round differently (orders of the sums): 1e-10 block-relative, observed ~1e-14."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for kern in ("wide", "lane"):
        env = dict(os.environ)
        env.pop("PRONTO_SMOOTH_KERNEL", None)
        if kern == "lane":
            env["PRONTO_SMOOTH_KERNEL"] = "lane"
        path = str(tmp_path / (kern + ".npz"))
        r = subprocess.run([sys.executable, "-c", _SNIPPET % {"root": root, "B": B, "path": path}], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        d = np.load(path)
        res[kern] = [d["arr_%d" % i] for i in range(8)]
    worst = 0.0
    for a, b in zip(res["wide"], res["lane"]):
        assert np.all(np.isfinite(a)) and a.shape == b.shape
        worst = max(worst, rel(a, b))
    print("wide vs lane, B = %d: worst block-relative difference %.2e" % (B, worst))
    assert worst < 1e-10
    for i in range(4):   # the in-place call gives the same bits as the call into a free slot
        assert np.array_equal(res["wide"][i], res["wide"][4 + i])


def _exact_smoothed_cov(Ad, Pc, Pp, Ps):
    """P_k + G (P^s - P^-) G^T, G = P_k Ad^T (P^-)^-1 in exact rational arithmetic on the float inputs (one filter)."""
    from fractions import Fraction
    n = Pc.shape[0]
    F = lambda M: [[Fraction(float(M[i, j])) for j in range(M.shape[1])] for i in range(M.shape[0])]
    A, C, P, S = F(Ad), F(Pc), F(Pp), F(Ps)
    # X = (P^-)^-1 (Ad P_k): Gaussian elimination with partial pivoting on [P | Ad P_k] (exact: the pivoting only avoids zero pivots)
    rhs = [[sum(A[i][k] * C[k][j] for k in range(n)) for j in range(n)] for i in range(n)]
    M = [P[i][:] + rhs[i][:] for i in range(n)]
    for c in range(n):
        piv = max(range(c, n), key=lambda r: abs(M[r][c]))
        M[c], M[piv] = M[piv], M[c]
        inv = 1 / M[c][c]
        M[c] = [v * inv for v in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    X = [row[n:] for row in M]                       # X = (P^-)^-1 Ad P_k, G = X^T
    D = [[S[i][j] - P[i][j] for j in range(n)] for i in range(n)]
    GD = [[sum(X[k][i] * D[k][j] for k in range(n)) for j in range(n)] for i in range(n)]
    out = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            out[i, j] = float(C[i][j] + sum(GD[i][k] * X[k][j] for k in range(n)))
    return out


@pytest.mark.parametrize("n", [15, 21])
def test_unpivoted_factorisation_is_as_accurate_as_the_pivoted_one_on_ill_conditioned_covariances(oracle, n):
    """the default kernels factorise P^-_{k+1} WITHOUT the reference's diagonal pivoting (.ldlt()).  For an SPD matrix the
    two differ by rounding; the question is what happens as P^- approaches a positive-SEMI-definite matrix.  One smoother step on crafted
    checkpoints whose P^- = Q diag(lambda) Q^T has condition numbers 1e6 .. 1e10 (dense Q), judged against the EXACT smoothed covariance
    (rational arithmetic on the same float inputs): the GPU's unpivoted result must be finite and no further from the truth than a small
    multiple of the oracle's pivoted result -- the step itself loses digits like cond^2 (G is as large as cond), pivoting does not buy
    them back.  An exactly singular P^- is outside what either factorisation defines (the reference divides by whatever rounding leaves
    of the zero pivot as well: Eigen's solve zeroes only |d| <= DBL_MIN)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import numpy_restatement as nr
    from pronto_amd.batch import BatchEstimator
    from util import embed21
    B = 64
    NX = 3          # filters checked against exact arithmetic (seconds each)
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    rng = np.random.default_rng(1234 + n)
    dt = 1e-3
    for cond in (1e6, 1e8, 1e10):
        Pp = np.zeros((n, n, B))
        for b in range(B):
            Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            lam = 1e-2 * np.logspace(0.0, -np.log10(cond), n)
            Pp[:, :, b] = (Q * lam) @ Q.T
            Pp[:, :, b] = 0.5 * (Pp[:, :, b] + Pp[:, :, b].T)
        Ps = 0.7 * Pp
        Pc = P0.copy()
        for i in range(n):
            Pc[i, i] += 1e-4     # (every state has some variance in the filtered checkpoint)
        vp = vec + 1e-3 * rng.standard_normal(vec.shape)
        vs = vp + 1e-4 * rng.standard_normal(vec.shape)
        est = BatchEstimator(B, n_states=n)
        est.set_constants(*oracle.constants())
        est.history_reserve(4)
        for slot, (v, P) in enumerate(((vec, Pc), (vp, Pp), (vs, Ps))):
            est.reset(v, quat, P)
            est.state_save(slot)
        est.smooth_step(1, 2, 0, 3, dt)
        est.state_restore(3)
        gv, gq, gP, _ = est.get_head()
        est.close()
        e = lambda v, P: (embed21(v, P)[0], quat, embed21(v, P)[1])
        ov, oq, oP = oracle_smooth_step(oracle, e(vp, Pp), e(vs, Ps), e(vec, Pc), dt)
        assert np.all(np.isfinite(gv)) and np.all(np.isfinite(gq)) and np.all(np.isfinite(gP))
        v21 = embed21(vec, Pc)[0]
        Ad = nr.process_matrices(np.ascontiguousarray(v21.T), np.ascontiguousarray(quat.T), np.full(B, dt))
        eg = eo = 0.0
        for b in range(NX):
            truth = _exact_smoothed_cov(Ad[b][:n, :n], Pc[:, :, b], Pp[:, :, b], Ps[:, :, b])
            sc = np.abs(truth).max()
            eg = max(eg, np.abs(gP[:, :, b] - truth).max() / sc)
            eo = max(eo, np.abs(oP[:n, :n, b] - truth).max() / sc)
        print("n=%d cond %.0e: error against exact arithmetic: GPU (unpivoted) %.2e, oracle (pivoted) %.2e; GPU against oracle %.2e"
              % (n, cond, eg, eo, rel(gP, oP[:n, :n])))
        assert eg < 10.0 * eo + 1e-12, (cond, eg, eo)
