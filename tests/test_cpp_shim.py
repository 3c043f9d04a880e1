"""The C++ mirror of the reference API (pronto_amd/csrc/mav_state_est_batch.hpp): it compiles and links against the
C ABI everywhere; on a GPU the miniature se-fusion in tests/cpp/test_shim.cpp must agree with the oracle."""
import os
import subprocess

import pytest

from pronto_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "build", "test_shim")


def build_exe(oracle):
    _lib.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_shim.cpp")
    deps = [src, os.path.join(ROOT, "pronto_amd", "csrc", "mav_state_est_batch.hpp"),
            os.path.join(ROOT, "include", "pronto_batch.h"), _lib.LIB_PATH]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Werror=return-type", "-o", EXE, src,
           "-L" + os.path.dirname(_lib.LIB_PATH), "-lpronto_batch", "-L" + os.path.join(ROOT, "oracle", "build"),
           "-lpronto_oracle", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle", "build"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_shim_compiles_and_links(oracle):
    exe = build_exe(oracle)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpronto_batch.so" in out and "not found" not in out.split("libpronto_batch.so")[1].split("\n")[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [15, 21])
def test_shim_matches_oracle_on_gpu(oracle, n):
    exe = build_exe(oracle)
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert "discarding update" in r.stderr  # the late update was rejected like update_history.cpp:28-39
