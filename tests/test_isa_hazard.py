"""Static guard of the 16-byte-store hazard workaround: the ISA hipcc emits for the step kernels must keep every
`buffer_store_dwordx4`'s data registers untouched until the `s_nop` behind the store (rbis_kernels.hpp stg2).  Runs on the
CPU tier (hipcc cross-compiles); the GPU tier repeats thousands of launches and compares bits
(tests/test_gpu_edge_cases.py::test_store_hazard_regression_thousands_of_launches)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pronto_amd", "csrc")


def test_no_write_to_a_16_byte_stores_data_registers_before_its_nop():
    out = os.path.join(ROOT, "tests", "build", "pb_step.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
                               os.path.join(CSRC, "pb_step.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chk_store_hazard.py"), out], capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0, r.stdout[-3000:]
    n_stores = int(last.split()[1])
    assert n_stores > 1000, last     # the step kernels really were in that file


def _asm(name, src, *defs):
    out = os.path.join(ROOT, "tests", "build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, *defs,
                               os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return out


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
