"""Static guard of the 16-byte-store hazard workaround: the ISA hipcc emits for the step kernels must keep every
`buffer_store_dwordx4`'s data registers untouched until the `s_nop` behind the store (rbis_kernels.hpp stg2).  Runs on the
CPU tier (hipcc cross-compiles); the GPU tier repeats thousands of launches and compares bits
(tests/test_gpu_edge_cases.py::test_store_hazard_regression_thousands_of_launches)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pronto_amd", "csrc")


# every assembly dump this file looks at: name -> (source, defines).  They are made together, in parallel, the first time one is
# asked for (four hipcc runs of 15-70 s each: one after the other they were half of the CPU tier's time)
DUMPS = {"pb_step.s": ("pb_step.hip", ()), "pb_step_leg15.s": ("pb_step_leg.hip", ("-DPB_LEG_NS=15",)),
         "pb_step_leg21.s": ("pb_step_leg.hip", ("-DPB_LEG_NS=21",)), "pb_smooth.s": ("pb_smooth.hip", ()),
         "pb_smooth_wide.s": ("pb_smooth_wide.hip", ("-mllvm", "-disable-machine-licm"))}   # (the Makefile's flags for that object)


def _stale(out):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))]
    return not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps)


def _asm(name, src=None, *defs):
    bdir = os.path.join(ROOT, "tests", "build")
    os.makedirs(bdir, exist_ok=True)
    todo = {n: v for n, v in DUMPS.items() if _stale(os.path.join(bdir, n))}
    if name not in DUMPS and _stale(os.path.join(bdir, name)):
        todo[name] = (src, defs)
    procs = [(n, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o",
                                   os.path.join(bdir, n + ".tmp"), *d, os.path.join(CSRC, f)], stderr=subprocess.DEVNULL))
             for n, (f, d) in todo.items()]
    for n, pr in procs:
        assert pr.wait() == 0, n
        os.replace(os.path.join(bdir, n + ".tmp"), os.path.join(bdir, n))
    return os.path.join(bdir, name)


def test_no_write_to_a_16_byte_stores_data_registers_before_its_nop():
    out = _asm("pb_step.s")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chk_store_hazard.py"), out], capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0, r.stdout[-3000:]
    n_stores = int(last.split()[1])
    assert n_stores > 1000, last     # the step kernels really were in that file


def _kernel_metadata(path):
    """name -> (vgprs, agprs, scratch bytes per lane) from the code object metadata hipcc appends to the assembly (one entry per
    kernel, opened by its `- .agpr_count:` line, members in alphabetical order)."""
    import re
    meta, cur = {}, None
    for ln in open(path, errors="replace"):
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", ln)
        if m:
            cur = {"agpr_count": int(m.group(1))}
            continue
        if cur is None:
            continue
        m = re.match(r"\s+\.(vgpr_count|private_segment_fixed_size):\s+(\d+)", ln)
        if m:
            cur[m.group(1)] = int(m.group(2))
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            cur["name"] = m.group(1)
        if all(k in cur for k in ("name", "vgpr_count", "private_segment_fixed_size")):
            meta[cur["name"]] = (cur["vgpr_count"], cur["agpr_count"], cur["private_segment_fixed_size"])
            cur = None
    return meta


def test_pair_kernels_keep_their_registers_and_the_store_guard():
    """The pair kernels (pb_step_leg.hip: lin_rate and LegOdoCommon's six-row modes inside the step kernel).  (1) The 16-byte-store
    guard holds in them too.  (2) Two waves per SIMD: <= 256 registers, no AGPRs, and NO scratch for 15 states (a few dozen bytes for
    21) -- when the compiler sinks role C's covariance propagation behind barrier L (what `pb_pin` prevents, rbis_coop.hpp) the
    six-row variants carry 456-528 bytes of scratch and run 8-10 us slower at 64k filters."""
    for ns in (15, 21):
        path = _asm("pb_step_leg%d.s" % ns, "pb_step_leg.hip", "-DPB_LEG_NS=%d" % ns)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chk_store_hazard.py"), path], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        meta = {k: v for k, v in _kernel_metadata(path).items() if "k_step_leg" in k or "k_step_quad_leg" in k}
        assert len(meta) == 9, sorted(meta)   # three modes x three cache policies
        for name, (vgpr, agpr, scratch) in meta.items():
            assert vgpr <= 256 and agpr == 0, (name, vgpr, agpr)
            # (15 states: lin_rate and lin_rot_rate none at all; pos_and_lin_rate, whose odometry wave holds the whole state vector
            # for the world constraint, 68 bytes today)
            six2 = name.endswith("ELi2EEEvPKdPdiS2_ddddNS_6ConstsENS_9StepBcastENS_6LegParENS_5LegInEPKNS_8LegChainENS_11LegStepArgsE") and "ILi15E" in name
            assert scratch <= (64 if ns == 21 else 80 if six2 else 0), (name, scratch)


def test_step_kernels_fit_two_waves_per_simd():
    """The hot kernels sit a few registers under the 256 that two waves per SIMD allow (k_step_coop<15>: 250; its launch bounds
    permit 512): an innocent-looking reordering in a role body -- e.g. the state update in front of the covariance downdate -- made
    the compiler take 256 + 58 AGPRs, one wave per SIMD, and the 64k-filter step went from 21.3 to 24.9 us with every test green.
    This holds the register files of the fused steps."""
    path = _asm("pb_step.s", "pb_step.hip")
    meta = _kernel_metadata(path)
    hot = {k: v for k, v in meta.items() if k.startswith("_ZN2pb11k_step_coopILi15ELb1E") and "NS_4CorrILb0EJEEELb1E" in k}
    assert len(hot) == 3, sorted(meta)     # the plain 15-state fused step, three cache policies
    for name, (vgpr, agpr, scratch) in hot.items():
        assert vgpr <= 256 and agpr == 0 and scratch == 0, (name, vgpr, agpr, scratch)
    quad = {k: v for k, v in meta.items() if k.startswith("_ZN2pb11k_step_quadILb1E")}
    assert len(quad) == 3, sorted(meta)
    for name, (vgpr, agpr, scratch) in quad.items():
        # (round 5 tried keeping the angular-velocity / acceleration entries of a filter without an IMU message INSIDE the step kernels:
        # 24 bytes of scratch here and 39.7 instead of 37.0-37.9 us at 64k filters -- done in front of the kernel instead, rbis_frontend.hpp)
        assert vgpr <= 256 and agpr == 0 and scratch <= 16, (name, vgpr, agpr, scratch)


def test_smoother_lane_kernel_keeps_its_values_in_front_of_the_barriers():
    """k_smooth_lane (rbis_smooth_lane.hpp): two waves per SIMD, no AGPRs and (next to) no scratch.  Without `lane_pin` the backend sinks a
    role's arithmetic below the next barrier to its first use, keeps the LDS operands it was computed from alive across the barrier
    and spills them: 200-750 bytes of scratch per lane, 428 instead of 313 us per 21-state step at 64k filters."""
    path = _asm("pb_smooth.s", "pb_smooth.hip")
    meta = _kernel_metadata(path)
    lane = {k: v for k, v in meta.items() if k.startswith("_ZN2pb13k_smooth_laneILi")}
    assert len(lane) == 2, sorted(meta)
    for name, (vgpr, agpr, scratch) in lane.items():
        assert vgpr <= 256 and agpr == 0 and scratch <= (0 if "ILi15E" in name else 16), (name, vgpr, agpr, scratch)


def test_smoother_wide_kernel_has_no_scratch_and_guards_its_row_stores():
    """k_smooth_wide<15> (rbis_smooth_wide.hpp): one wave per SIMD, architectural + accumulation registers, NO scratch -- with the
    backend's loop-invariant code motion left on, ~60 constants of the attitude arithmetic are lifted in front of the loop over the tiles
    and spilled (372 bytes per lane, 162 instead of 139 us per step when that was measured), hence the object's own flags (Makefile).
    Its posterior leaves by 16-byte rows: the store hazard workaround of stg2 must hold here as well."""
    path = _asm("pb_smooth_wide.s")
    meta = _kernel_metadata(path)
    wide = {k: v for k, v in meta.items() if k.startswith("_ZN2pb13k_smooth_wideILi15E")}
    assert len(wide) == 1, sorted(meta)
    for name, (vgpr, agpr, scratch) in wide.items():
        assert vgpr <= 512 and 0 < agpr <= 256 and scratch == 0, (name, vgpr, agpr, scratch)   # (.vgpr_count is the unified total here)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chk_store_hazard.py"), path], capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0, r.stdout[-3000:]
    assert int(last.split()[1]) >= 18, last     # the role's rows
