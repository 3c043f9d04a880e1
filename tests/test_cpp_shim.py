"""The C++ mirror of the reference API (pronto_amd/csrc/mav_state_est_batch.hpp): it compiles and links against the
C ABI everywhere; on a GPU the miniature se-fusion in tests/cpp/test_shim.cpp and the delayed-measurement replay in
tests/cpp/test_history.cpp must agree with the oracle.  Every handler test runs for the 15-state filter and for the
21-state one (biases estimated online: BASELINE config 5's filter) -- "n21" on the executable's command line
(tests/cpp/test_n.hpp)."""
import os
import subprocess

import pytest

from pronto_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NARG = {15: [], 21: ["n21"]}  # tests/cpp/test_n.hpp


def build_exe(oracle, name="test_shim"):
    _lib.build()
    exe = os.path.join(ROOT, "tests", "build", name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    deps = [src, os.path.join(ROOT, "tests", "cpp", "test_n.hpp"), os.path.join(ROOT, "pronto_amd", "csrc", "mav_state_est_batch.hpp"),
            os.path.join(ROOT, "pronto_amd", "csrc", "segment_batcher.hpp"), os.path.join(ROOT, "pronto_amd", "csrc", "segment_stream.hpp"),
            os.path.join(ROOT, "pronto_amd", "csrc", "lcm_schema.hpp"),
            os.path.join(ROOT, "pronto_amd", "csrc", "pronto_wire.hpp"),
            os.path.join(ROOT, "include", "pronto_batch.h"), _lib.LIB_PATH]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    cmd = ["g++", "-O1", "-std=c++17", "-fopenmp", "-Wall", "-Werror=return-type", "-o", exe, src,
           "-L" + os.path.dirname(_lib.LIB_PATH), "-lpronto_batch", "-L" + os.path.join(ROOT, "oracle", "build"),
           "-lpronto_oracle", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle", "build"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
def test_atlas_imu_front_end_on_gpu(oracle, tmp_path, n):
    """KVH batch de-dup + device notch cascade + process step through InsHandler::processMessageAtlas vs the oracle."""
    exe = build_exe(oracle, "test_atlas_imu")
    r = subprocess.run([exe, str(tmp_path)] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("kernel", ["default", "lane", "reg", "reg_pivoted"])
def test_smooth_backwards_pass_on_gpu(oracle, n, kernel):
    """EKFSmoothBackwardsPass (mav_state_est.cpp:98-189) through the shim vs the oracle's backward recursion, steps with
    and without a measurement after the INS update.  `kernel`: the default (15 states: k_smooth_wide, rbis_smooth_wide.hpp; 21 states:
    k_smooth_lane, rbis_smooth_lane.hpp), k_smooth_lane for 15 states as well (PRONTO_SMOOTH_KERNEL=lane), and the two kernels they
    replaced, which stay selectable (PRONTO_SMOOTH_KERNEL=reg: 16 / 32 lanes per filter without the pivot search; PRONTO_SMOOTH_PIVOT=1:
    with Eigen's diagonal pivoting) -- the switch is read once per process, hence the executable."""
    exe = build_exe(oracle, "test_smooth_pass")
    env = dict(os.environ)
    env.pop("PRONTO_SMOOTH_KERNEL", None)
    env.pop("PRONTO_SMOOTH_PIVOT", None)
    if kernel in ("reg", "lane"):
        env["PRONTO_SMOOTH_KERNEL"] = kernel
    elif kernel == "reg_pivoted":
        env["PRONTO_SMOOTH_PIVOT"] = "1"
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("every", [3, 7])
def test_smooth_backwards_pass_with_sparse_checkpoints_on_gpu(oracle, n, every):
    """EKFSmoothBackwardsPass when the history does not fit a posterior per update (what the reference keeps by value): with
    state_estimator.history_checkpoint_every = K only every K-th update has a checkpoint and the pass re-derives the others stretch
    by stretch, newest first, by re-applying the updates between two checkpoints into a window of free slots (checkpoint and
    recompute).  60 steps (100 updates) in 2 T / K + K + 8 slots; every smoothed step against the oracle's recursion."""
    exe = build_exe(oracle, "test_smooth_pass")
    r = subprocess.run([exe, str(n), str(every)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
def test_lcm_log_replay_and_filter_state_publishing_on_gpu(oracle, tmp_path, n):
    """A recorded LCM event log (pronto::indexed_measurement_t, pronto::update_t, raw IMU ticks, foreign channels)
    replayed into the batch through LogPlayer + the reference's handlers vs the oracle; the head published as
    pronto::filter_state_t and read back bit-exactly (pronto_wire.hpp)."""
    exe = build_exe(oracle, "test_log_replay")
    r = subprocess.run([exe, str(tmp_path)] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("variant", ["", "derived", "fuse", "empty"])
def test_fovis_keyframe_lookup_in_history_on_gpu(oracle, variant, n):
    """FovisHandler with posterior checkpoints on: T0 comes from history.updateMap.lower_bound(prev_timestamp) like in the
    reference (25 ms rule, cached per keyframe, 'at the end' rejection), no manual keyframe marking.  "derived": only
    utime_history_span is configured (sparse default checkpoints), "fuse": the look-up lands on the INS half of a fused
    pair -- in both the posterior has no checkpoint and is re-derived from the nearest earlier one.  "empty": updates whose
    device-resident mask lets no filter through (the reference's handler returns NULL for such a message, so its history never
    holds them) sit right behind the keyframe instants: the look-up steps over them (pb_mask_count, asked lazily)."""
    exe = build_exe(oracle, "test_fovis_history")
    r = subprocess.run([exe] + ([variant] if variant else []) + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
def test_fovis_updates_own_their_measurement_across_a_replay(oracle, n):
    """Two FovisHandler updates in the window, then a measurement older than both: the replay re-applies each VO update
    with the z / quaternion it was built with (the handler is even destroyed before the estimator)."""
    exe = build_exe(oracle, "test_fovis_replay")
    r = subprocess.run([exe] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sm_position", "sm_velocity", "sm_yaw", "sm_position_yaw", "sm_velocity_yaw", "fovis_velocity",
                                  "fovis_position", "legodo_zero3", "legodo_ft"])
@pytest.mark.parametrize("n", [15, 21])
def test_every_handler_mode_on_gpu(oracle, mode, n):
    """The handler modes the miniature se-fusion does not reach: all five ScanMatcherHandler modes
    (sensor_handlers.cpp:612-724), FovisHandler velocity / position (rbis_fovis_update.cpp:93-117), LegOdoHandler's
    zero_initial_velocity = 3 and its force/torque gate (rbis_legodo_update.cpp:208-211,264-268), each against the oracle's
    restatement of the same handler arithmetic."""
    exe = build_exe(oracle, "test_handler_modes")
    r = subprocess.run([exe, mode] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,fuse", [("lin_rate", ""), ("lin_rot_rate", ""), ("lin_rate", "fuse"), ("lin_rot_rate", "fuse")])
@pytest.mark.parametrize("n", [15, 21])
def test_leg_odometry_from_foot_transforms_through_the_handler_on_gpu(oracle, mode, fuse, n):
    """LegOdoHandler::processMessageFeet: leg_estimate::updateOdometry + contact classification + createMeasurement on the
    device for every filter (its world_to_body_ is the filter's own head orientation), against the oracle's restatement.
    "fuse": with state_estimator.fuse_ins_legodo the INS step stays pending, the odometry is slaved to the orientation after it
    and the pair runs as one fused kernel -- same oracle sequence (predict, odometry, update)."""
    exe = build_exe(oracle, "test_leg_feet")
    r = subprocess.run([exe, mode] + ([fuse] if fuse else []) + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_ins_gravity_initialisation_host_only(oracle):
    """InsHandler::processMessageInit (sensor_handlers.cpp:254-364) is host arithmetic: it runs here, without a GPU,
    against the oracle's po_ins_init."""
    exe = build_exe(oracle, "test_ins_init")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


def test_urdf_chains_and_leg_handler_configuration_host_only(oracle):
    """ModelClient::fromURDFString (URDF text -> the two chains the forward kinematics evaluates) and LegOdoHandler's
    constructor keys / message handlers: host logic, runs here without a GPU."""
    exe = build_exe(oracle, "test_urdf_chain")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode,slots,fuse", [("pos_and_lin_rate", 0, ""), ("pos_and_lin_rate", 6, ""), ("lin_rot_rate", 0, ""),
                                             ("lin_rate", 0, "fuse"), ("lin_rate", 4, "fuse"), ("pos_and_lin_rate", 0, "fuse")])
@pytest.mark.parametrize("n", [15, 21])
def test_legodo_modes_on_gpu(oracle, mode, slots, fuse, n):
    """LegOdoCommon's other modes, incl. the per-filter pos_and_lin_rate -> lin_rate fall-back (two complementary masked
    updates), with and without posterior checkpoints, vs the oracle's createMeasurement + indexed update.  "fuse": the
    opt-in state_estimator.fuse_ins_legodo -- every INS step followed by a lin_rate measurement runs as one fused kernel,
    with checkpoints too (the pair is checkpointed behind its second half); updates that are not fusible run one by one."""
    exe = build_exe(oracle, "test_legodo_modes")
    r = subprocess.run([exe, mode, str(slots), fuse] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name", ["test_shim", "test_history", "test_atlas_imu", "test_smooth_pass", "test_log_replay",
                                  "test_legodo_modes", "test_fovis_history", "test_fovis_replay",
                                  "test_handler_modes", "test_leg_feet", "test_leg_joints", "test_segments", "test_host_update"])
def test_shim_compiles_and_links(oracle, name):
    exe = build_exe(oracle, name)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpronto_batch.so" in out and "not found" not in out.split("libpronto_batch.so")[1].split("\n")[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("fuse", ["", "fuse3"])
def test_shim_matches_oracle_on_gpu(oracle, n, fuse):
    """The miniature se-fusion (INS + leg odometry + VO position_orient + scan-match position_yaw through the handlers)
    against the oracle.  "fuse3": with state_estimator.fuse_ins_legodo / fuse_corrections every INS + leg-odometry pair runs
    as one kernel and the pairs followed by a VO / scan-match message as ONE kernel with all three updates
    (pb_step_legodo_correct); the executable checks the launch counts."""
    exe = build_exe(oracle)
    r = subprocess.run([exe, str(n)] + ([fuse] if fuse else []), capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert "discarding update" in r.stderr  # the late update was rejected like update_history.cpp:28-39


@pytest.mark.gpu
@pytest.mark.parametrize("checkpoint_every,delay,fuse", [(1, 0, ""), (1, 7, ""), (4, 13, ""), (0, 7, ""), (1, 7, "fuse"), (4, 13, "fuse"),
                                                         (0, 7, "fuse")])
@pytest.mark.parametrize("n", [15, 21])
def test_delayed_measurements_replay_equals_in_order(oracle, checkpoint_every, delay, fuse, n):
    """measurements arriving `delay` steps late are inserted at their timestamp and everything
    after them is re-applied from the nearest posterior checkpoint (mav_state_est.cpp:28-80); the head must equal an
    in-order pass (the oracle).  Dense and sparse checkpointing agree; checkpoint_every = 0 sets ONLY utime_history_span,
    like a reference .cfg: the estimator derives its checkpoint pool and cadence from the span.  "fuse": the same with
    state_estimator.fuse_ins_legodo -- INS + leg-odometry pairs run as one kernel and are checkpointed as one update."""
    exe = build_exe(oracle, "test_history")
    r = subprocess.run([exe, str(checkpoint_every), str(delay)] + ([fuse] if fuse else []) + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert "before the first in history" in r.stderr  # the too-old fix was discarded (update_history.cpp:28-39)


def build_multi_gpu_sweep():
    """examples/multi_gpu_sweep.cpp exactly as its header comment says (plain g++, the C ABI, RCCL called directly)."""
    _lib.build()
    exe = os.path.join(ROOT, "tests", "build", "multi_gpu_sweep")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "examples", "multi_gpu_sweep.cpp")
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in (src, _lib.LIB_PATH)):
        return exe
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", src, "-L" + os.path.dirname(_lib.LIB_PATH), "-lpronto_batch",
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_multi_gpu_sweep_example_compiles_and_links():
    assert os.path.exists(build_multi_gpu_sweep())


@pytest.mark.gpu
def test_multi_gpu_sweep_example_on_the_visible_devices():
    """The C++ filter-range split: one context and one host thread per shard (two shards share the one device of the test
    box), no data-path exchange, one RCCL all-reduce of the summary; its built-in check re-runs the job as ONE context and
    demands the same summary."""
    exe = build_multi_gpu_sweep()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "4096", "40", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL all-reduce" in r.stdout and "x 2 shard(s)" in r.stdout, r.stdout


def build_shim_sweep_rate():
    """examples/shim_sweep_rate.cpp as its header comment says."""
    _lib.build()
    exe = os.path.join(ROOT, "tests", "build", "shim_sweep_rate")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "examples", "shim_sweep_rate.cpp")
    deps = (src, _lib.LIB_PATH, os.path.join(ROOT, "pronto_amd", "csrc", "mav_state_est_batch.hpp"))
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "pronto_amd", "csrc"), src, "-L" + os.path.dirname(_lib.LIB_PATH),
                           "-lpronto_batch", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_shim_sweep_rate_example_compiles_and_links():
    assert os.path.exists(build_shim_sweep_rate())


@pytest.mark.gpu
@pytest.mark.parametrize("n,slots", [(15, "0"), (21, "0"), (15, "8")])
def test_shim_sweep_rate_example_runs(n, slots):
    """The drop-in path end to end (handler API -> estimator -> kernels) with one robot's log broadcast into a sweep of
    filters: runs, stays finite, drops nothing."""
    r = subprocess.run([build_shim_sweep_rate(), "2048", "300", str(n), slots], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropped 0" in r.stdout and "finite" in r.stdout and "NON-FINITE" not in r.stdout, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("lin_rate", "alt"), ("lin_rate", "alt", "fuse"), ("lin_rate", "standing", "fuse", "bcast"),
                                  ("lin_rot_rate", "ctrl"), ("pos_and_lin_rate", "alt"), ("lin_rate", "ctrl", "nofuse", "bcast"),
                                  ("lin_rate", "alt", "fuse", "blocks", "lowpass"), ("lin_rate", "alt", "fuse", "bcast", "kalman"),
                                  ("lin_rot_rate", "standing", "nofuse", "blocks", "kalman"), ("lin_rate", "alt", "nofuse", "bcast", "lowpass"),
                                  ("lin_rate", "alt", "fuse", "device", "none"), ("lin_rate", "ctrl", "fuse", "device", "lowpass"),
                                  ("lin_rate", "alt", "fuse3", "device", "none"), ("pos_and_lin_rate", "alt", "nofuse", "bcast"),
                                  ("lin_rot_rate", "alt", "nofuse", "bcast"),
                                  # the six-row modes inside the pair kernel (fuse_ins_legodo: SIX, rbis_legstep.hpp)
                                  ("lin_rot_rate", "alt", "fuse", "device", "none"), ("pos_and_lin_rate", "alt", "fuse", "device", "none"),
                                  ("lin_rot_rate", "ctrl", "fuse", "bcast"), ("pos_and_lin_rate", "alt", "fuse", "bcast"),
                                  ("pos_and_lin_rate", "standing", "fuse", "blocks", "lowpass"), ("pos_and_lin_rate", "alt", "fuse3", "device", "none"),
                                  # "late": legodo.roll_forward_on_receive = false, and every 10th joint state is OLDER than the held INS step
                                  # and than a pose that has been applied, its odometry still deferred (replay from the
                                  # checkpoint in front of it -- neither applied on top of the head nor the pose applied twice)
                                  ("lin_rate", "alt", "fuse", "device", "none", "late"), ("pos_and_lin_rate", "alt", "fuse", "bcast", "none", "late")])
@pytest.mark.parametrize("n", [15, 21])
def test_joint_state_handler_on_gpu(oracle, args, n):
    """LegOdoHandler::processMessage(joint_state_t) -- the reference's handler signature -- from a synthetic 6-DoF-per-leg gait:
    URDF text -> chains, force/torque + controller messages, torque adjustment, forward kinematics, contact logic, odometry and
    LegOdoCommon's measurement on the device, against the oracle chain (tests/cpp/test_leg_joints.cpp)."""
    exe = build_exe(oracle, "test_leg_joints")
    r = subprocess.run([exe, *args] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("n,fuse", [(15, "fuse"), (21, "fuse"), (15, "nofuse")])
def test_independent_log_segments_as_one_batch_on_gpu(oracle, tmp_path, n, fuse):
    """SegmentBatcher: 64 DIFFERENT recorded segments (another gait, IMU stream, absolute time base, time-stamp jitter, length; some
    opened at a start_timestamp) written with LogWriter and replayed as ONE batch through InsHandler / LegOdoHandler::processMessage
    (joint states -> forward kinematics -> odometry, per-filter message times) / ScanMatcherHandler, against 64 single-segment
    oracle runs -- the reference's se-batch-process.sh workload (one se-fusion run per log) as one batch."""
    exe = build_exe(oracle, "test_segments")
    r = subprocess.run([exe, str(tmp_path)] + ([fuse] if fuse == "nofuse" else []) + NARG[n], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("variant", [(), ("slots",), ("mask",), ("slots", "mask")])
def test_user_defined_update_with_the_reference_signature_on_gpu(oracle, n, variant):
    """RBISHostUpdate: a user-defined update written against the REFERENCE's contract -- updateFilter(const RBIS &prior_state,
    const RBIM &prior_cov, double prior_loglikelihood), rbis_update_interface.hpp:14-35 -- handed to MavStateEstimator::addUpdate
    between built-in updates (slow path: pb_get_head -> user code per filter -> pb_set_head).  The user's scalar altimeter update is
    written from the textbook equations; the expected head is the oracle's po_indexed_update in the same place.  "slots": with
    posterior checkpoints and one user update that arrives late and is replayed; "mask": applied to every other filter only."""
    exe = build_exe(oracle, "test_host_update")
    r = subprocess.run([exe, *variant] + NARG[n], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("variant", [("stream",), ("stream", "chunk7"), ("stream", "nofuse"), ("kvh",), ("kvh", "chunk7"), ("kvhbatch",)])
def test_independent_log_segments_streamed_on_gpu(oracle, tmp_path, n, variant):
    """SegmentStreamer (segment_stream.hpp): the same 64 DIFFERENT recorded segments as test_independent_log_segments_as_one_batch_on_gpu,
    replayed as a pipeline -- memory-mapped logs decoded ahead in chunks, one upload per chunk on the copy stream, the handlers called
    with PB_DEVICE messages (frame rotation, per-filter message times and validity on the device) -- against the same 64
    single-segment oracle runs, <= 1e-9.  "chunk7": chunks of 7 batched messages, so that every chunk-boundary case occurs.
    "kvh": the IMU arrives as the reference's own Atlas channel, bot_core::kvh_raw_imu_batch_t on ATLAS_IMU_BATCH (fusion.cpp:161-163
    -> InsHandler::processMessageAtlas, sensor_handlers.cpp:165-252): 64 different KVH logs with repeated packets, shorter first
    messages and messages with NO new packet; one IMUStream de-duplication per segment (imu_stream.cpp:62-98), the notch cascade on
    the device with per-filter packet counts, raw_dt and message-time dt per filter.  "kvhbatch": the same 64 KVH logs through the
    per-message SegmentBatcher::subscribeKvhBatch (host blocks, staged by the handler)."""
    exe = build_exe(oracle, "test_segments")
    r = subprocess.run([exe, str(tmp_path), *variant] + NARG[n], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
