import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; PARITY UNPINNED -- see oracle/pronto_oracle.h)."""
    from oracle import po
    po.build()
    po.lib()
    return po


@pytest.fixture(scope="session")
def harness():
    """g++ build of the kernels' per-lane arithmetic (tests/host_harness.cpp), test-only."""
    import ctypes as C
    out = os.path.join(ROOT, "tests", "build", "libhost_harness.so")
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    deps = [src] + [os.path.join(ROOT, "pronto_amd", "csrc", h)
                    for h in ("rbis_device.hpp", "rbis_coop.hpp", "rbis_quad.hpp", "rbis_legodo.hpp", "rbis_jointfilt.hpp")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-pthread", "-o", out, src])
    return C.CDLL(out)
