"""Build / load libpronto_batch.so (the C ABI of include/pronto_batch.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible, the
product path raises.  The library is kept IN-TREE (pronto_amd/lib/) so it travels with the repo
snapshot and shows up as a loaded native module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("PRONTO_BATCH_LIB") or os.path.join(_HERE, "lib", "libpronto_batch.so")  # env: A/B builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pronto_batch.h")

PB_OK, PB_ERR_ARG, PB_ERR_HIP, PB_ERR_NO_DEVICE, PB_ERR_STATE = range(5)
PB_HOST, PB_DEVICE, PB_HOST_BROADCAST = 0, 1, 2
PB_R_DIAG_BROADCAST, PB_R_DIAG, PB_R_FULL = 0, 1, 2
PB_CORR_POS_ORIENT, PB_CORR_POS_YAW = 0, 1


def sources():
    return [os.path.join(CSRC, f) for f in ("pronto_batch.hip", "pb_step.hip", "pb_update.hip", "pb_update_rt21.hip", "pb_update_ct.hip", "pb_smooth.hip", "pb_ctx.hpp",
                                            "rbis_kernels.hpp", "rbis_frontend.hpp", "rbis_legodo.hpp", "rbis_legstep.hpp", "rbis_jointfilt.hpp", "rbis_coop.hpp", "rbis_quad.hpp", "rbis_quad_rt.hpp", "rbis_smooth.hpp",
                                            "rbis_device.hpp", "Makefile")] + [HEADER]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False):
    """hipcc --offload-arch=gfx950 -shared (cross-compiles without a GPU)."""
    if force or is_stale():
        # (the translation units are independent: a serial from-scratch build takes ~10 minutes, -j8 ~2.5)
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j%d" % max(1, min(8, os.cpu_count() or 1))] + (["-B"] if force else []))
    return LIB_PATH


_lib = None

_dp = C.POINTER(C.c_double)
_SIGS = {
    "pb_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "pb_destroy": (C.c_int, [C.c_void_p]),
    "pb_last_error": (C.c_char_p, [C.c_void_p]),
    "pb_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pb_use_own_stream": (C.c_int, [C.c_void_p]),
    "pb_set_constants": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "pb_sync": (C.c_int, [C.c_void_p]),
    "pb_hot_kernel": (C.c_char_p, [C.c_void_p]),
    "pb_batch": (C.c_int, [C.c_void_p]),
    "pb_run_block": (C.c_int, [C.c_void_p]),
    "pb_n_states": (C.c_int, [C.c_void_p]),
    "pb_malloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "pb_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pb_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pb_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pb_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pb_fence_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pb_fence_record": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_fence_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "pb_upload_join": (C.c_int, [C.c_void_p]),
    "pb_upload_sync": (C.c_int, [C.c_void_p]),
    "pb_imu_notch_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_ins_body_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, _dp, _dp,
                                    C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pb_set_imu_valid": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pb_ins_body_reset": (C.c_int, [C.c_void_p]),
    "pb_get_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_smooth_log_slots": (C.c_int, [C.c_int, C.c_int]),
    "pb_smooth_log": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_float)]),
    "pb_set_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_predict": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_int]),
    "pb_update_indexed": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int]),
    "pb_update_indexed_orient": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int]),
    "pb_step_legodo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_int]),
    "pb_step_legodo_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _dp]),
    "pb_step_legodo_correct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_run_legodo": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.POINTER(C.c_float)]),
    "pb_replay_legodo_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp,
                                         C.POINTER(C.c_float)]),
    "pb_replay_legodo_checkpointed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_int,
                                                C.POINTER(C.c_float)]),
    "pb_snapshot": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_compose_delta": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_set_process_noise_block": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pb_window_nll": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int]),
    "pb_legodo_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int]),
    "pb_legodo_update": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pb_legodo_update_after_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pb_legodo_set_contact_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]),
    "pb_legodo_set_control_contacts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "pb_legodo_set_chain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp, C.POINTER(C.c_float)]),
    "pb_legodo_update_joints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "pb_legodo_set_zero_initial_velocity": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_legodo_set_message_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_legodo_set_measurement_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]),
    "pb_step_legodo_joints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "pb_step_legodo_feet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                      C.c_double, C.c_void_p, C.c_void_p]),
    "pb_calib_copy_checksum": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "pb_legodo_fk": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pb_joint_filter_init": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]),
    "pb_joint_filter": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pb_legodo_get": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64)]),
    "pb_imu_notch_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "pb_imu_notch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_history_reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_state_save": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_set_output_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_head_slot": (C.c_int, [C.c_void_p]),
    "pb_snapshot_from_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pb_host_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "pb_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pb_state_restore": (C.c_int, [C.c_void_p, C.c_int]),
    "pb_smooth_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "pb_get_head": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pb_get_filter_state": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "pb_summary": (C.c_int, [C.c_void_p, _dp]),
    "pb_mask_count": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "pb_state_checksum": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "pb_calib_copy": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "pb_set_utime": (C.c_int, [C.c_void_p, C.c_int64]),
    "pb_get_utime": (C.c_int64, [C.c_void_p]),
    "pb_version": (C.c_char_p, []),
}


def exported_names():
    return sorted(_SIGS)


def load():
    """dlopen the library and bind every symbol the header declares; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pronto_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    try:  # share torch's HIP runtime when torch is present (same soname, libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        if not hasattr(lib, name) and os.environ.get("PRONTO_BATCH_LIB"):
            continue  # an older A/B build named by PRONTO_BATCH_LIB may lack the newest entry points (timing runs only)
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
