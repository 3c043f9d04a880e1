"""The C ABI is usable from plain C: include/pronto_batch.h compiles as C99, and examples/param_sweep.c (the reference's
param_sweep.py as one batched context) builds with gcc and, on a GPU, finds the leg-odometry noise the data was made with;
examples/leg_sweep.c drives the joint-state path (chain table, joint filters, one kernel per IMU + joint-state pair) from C."""
import os
import subprocess

import pytest

from pronto_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(name="param_sweep"):
    _lib.build()
    exe = os.path.join(ROOT, "tests", "build", name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L" + os.path.dirname(_lib.LIB_PATH),
                           "-lpronto_batch", "-lm", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_header_is_valid_c99_and_example_links():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "pronto_batch.h")])
    build()
    build("leg_sweep")


@pytest.mark.gpu
def test_param_sweep_example_runs_on_gpu():
    exe = build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_leg_sweep_example_runs_on_gpu():
    exe = build("leg_sweep")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
