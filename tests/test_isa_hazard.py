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
