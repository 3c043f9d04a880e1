"""Wire formats at the seam: pronto_amd/csrc/pronto_wire.hpp against the independent Python
restatement tests/lcm_ref.py, byte for byte, plus the one public known answer available without lcm-gen: the base hash
of the LCM tutorial's exlcm::example_t.  CPU only."""
import os
import struct
import subprocess

import pytest

import lcm_ref as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool():
    exe = os.path.join(ROOT, "tests", "build", "wire_tool")
    src = os.path.join(ROOT, "tests", "cpp", "wire_tool.cpp")
    hdr = os.path.join(ROOT, "pronto_amd", "csrc", "pronto_wire.hpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", exe, src])
    return exe


def val(i):
    return (-1.0 if i % 2 else 1.0) * (0.1 + i) / 7.0


def expected_events():
    """The messages `wire_tool write` produces, encoded by the Python restatement."""
    ev = []
    for k in range(5):
        t = 1000000 + 1000 * k
        ev.append((t, "STATE_ESTIMATOR_STATE", L.enc_filter_state(t, [val(k + i) for i in range(4)],
                                                                   [val(3 * k + i) for i in range(21)],
                                                                   [val(k + 2 * i) for i in range(441)])))
        m = 1 + k
        ev.append((t + 1, "GPF_MEASUREMENT", L.enc_indexed_measurement(t + 1, t - 7, [val(10 + i + k) for i in range(m)],
                                                                        [(3 + 2 * i + k) % 21 for i in range(m)],
                                                                        [val(20 + i) for i in range(m * m)])))
        if k == 2:
            ev.append((t + 2, "SOMETHING_ELSE", bytes([1, 2, 3])))
        ev.append((t + 2, "KINECT_REL_ODOMETRY", L.enc_update(t + 2, t - 33333, [val(30 + i + k) for i in range(3)],
                                                              [val(40 + i + k) for i in range(4)],
                                                              [[val(50 + 6 * i + j + k) for j in range(6)] for i in range(6)],
                                                              k % 5)))
    return ev


def test_fingerprint_algorithm_known_answer(tool):
    """0x1baa9e29b0fbaa8b is the constant lcm-gen emits for the tutorial type exlcm::example_t (its generated
    `__exlcm_example_t_hash_recursive` / `_get_hash_recursive`): pins the hash recurrence, the string and dimension
    handling in both restatements."""
    assert L.base_hash(L.EXAMPLE_T) == 0x1BAA9E29B0FBAA8B
    out = dict(l.split() for l in subprocess.check_output([tool, "hashes"], text=True).splitlines())
    assert int(out["example_t_base"], 16) == 0x1BAA9E29B0FBAA8B
    assert int(out["filter_state_t"], 16) == L.fingerprint(L.FILTER_STATE_T)
    assert int(out["indexed_measurement_t"], 16) == L.fingerprint(L.INDEXED_MEASUREMENT_T)
    assert int(out["update_t"], 16) == L.fingerprint(L.UPDATE_T)
    assert len({out["filter_state_t"], out["indexed_measurement_t"], out["update_t"]}) == 3


def test_cpp_writer_equals_python_encoder(tool, tmp_path):
    path = str(tmp_path / "cpp.lcmlog")
    subprocess.check_call([tool, "write", path])
    got = L.read_log(open(path, "rb").read())
    exp = expected_events()
    assert len(got) == len(exp) == 16
    for i, ((num, ts, ch, data), (ets, ech, edata)) in enumerate(zip(got, exp)):
        assert (num, ts, ch) == (i, ets, ech)
        assert data == edata, (i, ch)
    # wire sizes follow from the type definitions: 8 + 8 + 32 + 4 + 21*8 + 4 + 441*8, and 8 + 16 + 56 + 288 + 1
    assert len(exp[0][2]) == 3752 and len(exp[-1][2]) == 369


def test_cpp_reader_decodes_python_log_and_resyncs(tool, tmp_path):
    """A log written by the Python restatement, with garbage spliced between two events and a truncated last event:
    the C++ reader must deliver every intact event with every field bit-exact, and skip the damage."""
    exp = expected_events()
    blob = b""
    for i, (ts, ch, data) in enumerate(exp):
        blob += L.log_event(i, ts, ch, data)
        if i == 4:
            blob += bytes(range(7, 200)) + struct.pack(">I", 0xEDA1DA00)  # no sync word inside
    blob += L.log_event(99, 5, "GPF_MEASUREMENT", exp[1][2])[:-5]  # truncated tail
    path = str(tmp_path / "py.lcmlog")
    open(path, "wb").write(blob)
    lines = subprocess.check_output([tool, "dump", path], text=True).splitlines()
    assert len(lines) == len(exp)
    for i, (line, (ts, ch, data)) in enumerate(zip(lines, exp)):
        f = line.split()
        assert f[0] == "event" and int(f[1]) == i and int(f[2]) == ts and f[3] == ch and int(f[4]) == len(data)
        k = [e[1] for e in exp[:i + 1]].count(ch) - 1  # k-th message of this channel
        t = 1000000 + 1000 * k
        if ch == "STATE_ESTIMATOR_STATE":
            want = [t] + [val(k + j) for j in range(4)] + [21] + [val(3 * k + j) for j in range(21)] + [441] + \
                   [val(k + 2 * j) for j in range(441)]
            assert f[5] == "filter_state_t"
        elif ch == "GPF_MEASUREMENT":
            m = 1 + k
            want = [t + 1, t - 7, m] + [val(10 + j + k) for j in range(m)] + [(3 + 2 * j + k) % 21 for j in range(m)] + \
                   [m * m] + [val(20 + j) for j in range(m * m)]
            assert f[5] == "indexed_measurement_t"
        elif ch == "KINECT_REL_ODOMETRY":
            want = [t + 2, t - 33333] + [val(30 + j + k) for j in range(3)] + [val(40 + j + k) for j in range(4)] + \
                   [val(50 + j + k) for j in range(36)] + [k % 5]
            assert f[5] == "update_t"
        else:
            assert f[5] == "unknown"
            continue
        got = [float(x) for x in f[6:]]
        assert got == [float(x) for x in want], (i, ch)


def test_decode_rejects_wrong_type_and_short_buffers(tool, tmp_path):
    """Fingerprint mismatch and truncation are errors, never a partial message: feed every type's bytes to the dump
    tool under a truncated length and expect 'unknown'."""
    exp = expected_events()
    blob = b""
    for i, (ts, ch, data) in enumerate(exp[:4]):
        blob += L.log_event(i, ts, ch, data[:-1])  # one byte short: no decoder may accept it
    path = str(tmp_path / "short.lcmlog")
    open(path, "wb").write(blob)
    lines = subprocess.check_output([tool, "dump", path], text=True).splitlines()
    assert len(lines) == 4 and all(l.split()[5] == "unknown" for l in lines)
