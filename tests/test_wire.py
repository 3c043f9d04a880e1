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
    hdr2 = os.path.join(ROOT, "pronto_amd", "csrc", "lcm_schema.hpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
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


DEMO_LCM = """
// a made-up package exercising every construct of the LCM type language (not a reference file)
package demo;
struct inner_t {
    int16_t id;      /* block
                        comment */
    float w[2];
    string label;
}
struct outer_t
{
    int64_t utime;
    int32_t n, m;
    const int32_t K = 3, L = 0x10;
    inner_t items[n];
    double grid[2][m];
    boolean ok;
    byte raw[4];
    demo.inner_t single;
    const double PI = 3.14;
}
"""
DEMO_TYPES = {
    "demo.inner_t": [("id", "int16_t", []), ("w", "float", [(0, "2")]), ("label", "string", [])],
    "demo.outer_t": [("utime", "int64_t", []), ("n", "int32_t", []), ("m", "int32_t", []),
                     ("items", "demo.inner_t", [(1, "n")]), ("grid", "double", [(0, "2"), (1, "m")]), ("ok", "boolean", []),
                     ("raw", "byte", [(0, "4")]), ("single", "demo.inner_t", [])],
}
EXAMPLE_LCM = """package exlcm;
struct example_t
{
    int64_t  timestamp;
    double   position[3];
    double   orientation[4];
    int32_t  num_ranges;
    int16_t  ranges[num_ranges];
    string   name;
    boolean  enabled;
}
"""


def test_runtime_schema_parse_fingerprint_decode(tool, tmp_path):
    """lcm_schema.hpp: a .lcm text parsed at run time (comments, consts, several declarators, nested and qualified types,
    variable and multi-dimensional arrays, strings), its recursive fingerprint and a decoded message, against the Python
    restatement; and the tutorial type's known constant through the same path."""
    demo = tmp_path / "demo.lcm"
    demo.write_text(DEMO_LCM)
    ex = tmp_path / "example_t.lcm"
    ex.write_text(EXAMPLE_LCM)
    value = {"utime": 123456789012, "n": 2, "m": 3,
             "items": [{"id": -7, "w": [0.5, -2.25], "label": "left foot"}, {"id": 300, "w": [1.0, 8.0], "label": ""}],
             "grid": [[0.1, 0.2, 0.3], [-1.5, 2.5e-7, 3e9]], "ok": 1, "raw": [0, 127, 128, 255],
             "single": {"id": 1, "w": [3.0, 4.0], "label": "x"}}
    msg = tmp_path / "outer.bin"
    msg.write_bytes(L.encode_message(DEMO_TYPES, "demo.outer_t", value))
    out = subprocess.check_output([tool, "schema", "%s,%s" % (demo, ex), "demo.outer_t", str(msg)], text=True).splitlines()
    assert int(out[0].split()[1], 16) == L.fingerprint_nested(DEMO_TYPES, "demo.outer_t")
    assert out[1] == ('{utime:123456789012,n:2,m:3,items:[{id:-7,w:[0.5,-2.25],label:"left foot"},{id:300,w:[1,8],label:""}],'
                      'grid:[[0.10000000000000001,0.20000000000000001,0.29999999999999999],[-1.5,2.4999999999999999e-07,3000000000]],'
                      'ok:1,raw:[0,127,128,255],single:{id:1,w:[3,4],label:"x"}}')
    # the nested types enter the fingerprint: it is not the rotated base hash alone
    hb = L.base_hash_nested(DEMO_TYPES["demo.outer_t"])
    assert L.fingerprint_nested(DEMO_TYPES, "demo.outer_t") != ((hb << 1) + (hb >> 63)) & L.M64
    # tutorial type, bare name: rot1(0x1baa9e29b0fbaa8b)
    ex_types = {"exlcm.example_t": L.EXAMPLE_T}
    ex_val = {"timestamp": 5, "position": [1.0, 2.0, 3.0], "orientation": [1.0, 0.0, 0.0, 0.0], "num_ranges": 3,
              "ranges": [10, -20, 30], "name": "example string", "enabled": 1}
    emsg = tmp_path / "example.bin"
    emsg.write_bytes(L.encode_message(ex_types, "exlcm.example_t", ex_val))
    out = subprocess.check_output([tool, "schema", "%s,%s" % (demo, ex), "example_t", str(emsg)], text=True).splitlines()
    base = 0x1BAA9E29B0FBAA8B
    assert int(out[0].split()[1], 16) == ((base << 1) + (base >> 63)) & L.M64
    assert out[1].startswith("{timestamp:5,position:[1,2,3],orientation:[1,0,0,0],num_ranges:3,ranges:[10,-20,30],name:\"example string\"")
    # damage: truncated message and a wrong type are refused
    (tmp_path / "short.bin").write_bytes(msg.read_bytes()[:-3])
    out = subprocess.check_output([tool, "schema", str(demo), "demo.outer_t", str(tmp_path / "short.bin")], text=True)
    assert "decode error" in out
    out = subprocess.check_output([tool, "schema", "%s,%s" % (demo, ex), "demo.inner_t", str(msg)], text=True)
    assert "fingerprint mismatch" in out


def test_compiled_extraction_plans_agree_with_the_value_tree(tool, tmp_path):
    """lcm_schema.hpp Schema::compile / Plan::run -- the replay fast path (SegmentBatcher): a few named members out of a message
    without building the value tree.  Top-level scalars, fixed and variable arrays, strings, a member of every element of an
    array of structs, a member behind variable-length members; fixed-size runs that are stepped over in one jump; refusals."""
    demo = tmp_path / "demo.lcm"
    demo.write_text(DEMO_LCM)
    value = {"utime": 123456789012, "n": 2, "m": 3,
             "items": [{"id": -7, "w": [0.5, -2.25], "label": "left foot"}, {"id": 300, "w": [1.0, 8.0], "label": ""}],
             "grid": [[0.1, 0.2, 0.3], [-1.5, 2.5e-7, 3e9]], "ok": 1, "raw": [0, 127, 128, 255],
             "single": {"id": 1, "w": [3.0, 4.0], "label": "x"}}
    msg = tmp_path / "outer.bin"
    msg.write_bytes(L.encode_message(DEMO_TYPES, "demo.outer_t", value))
    run = lambda members, path=msg, typ="demo.outer_t": subprocess.check_output([tool, "plan", str(demo), typ, str(path), members], text=True).splitlines()
    out = run("utime,grid,items.w,items.label,single.id,raw,ok,items.id")
    assert out[0] == "utime: 123456789012"
    assert [float(v) for v in out[1].split()[1:]] == [0.1, 0.2, 0.3, -1.5, 2.5e-7, 3e9]
    assert [float(v) for v in out[2].split()[1:]] == [0.5, -2.25, 1.0, 8.0]          # w of items[0], then of items[1]
    assert out[3] == 'items.label: "left foot" ""'
    assert out[4] == "single.id: 1" and out[5] == "raw: 0 127 128 255" and out[6] == "ok: 1" and out[7] == "items.id: -7 300"
    # only members behind the variable-length ones: everything in front is stepped over
    assert run("single.label,ok") == ['single.label: "x"', "ok: 1"]
    # a STREAM of messages through one Plan::Shape: the same layout (lengths, strings) -> the numbers by offset, no strings
    # rebuilt; another string, another array length, a broken message -> decoded the long way, the stream carries on
    v2 = dict(value, utime=5, grid=[[9.0, 8.0, 7.0], [6.0, 5.0, 4.0]], raw=[1, 2, 3, 4],
              items=[{"id": 1, "w": [10.0, 20.0], "label": "left foot"}, {"id": 2, "w": [30.0, 40.0], "label": ""}])
    v3 = dict(v2, items=[{"id": 1, "w": [10.0, 20.0], "label": "left feet"}, {"id": 2, "w": [30.0, 40.0], "label": ""}])   # same length, other bytes
    v4 = dict(v2, n=1, m=2, items=[{"id": 1, "w": [10.0, 20.0], "label": "left foot"}], grid=[[9.0, 8.0], [7.0, 6.0]])
    files = []
    for k, v in enumerate((v2, v3, v4, v4)):
        f = tmp_path / ("stream%d.bin" % k)
        f.write_bytes(L.encode_message(DEMO_TYPES, "demo.outer_t", v))
        files.append(str(f))
    (tmp_path / "cut.bin").write_bytes(msg.read_bytes()[:-5])
    out = subprocess.check_output([tool, "plan", str(demo), "demo.outer_t", str(msg), "utime,items.w,items.label,raw,grid", files[0],
                                   str(tmp_path / "cut.bin"), files[0], files[1], files[2], files[3]], text=True).splitlines()
    blocks, cur = [], None
    for ln in out:
        if ln.startswith("shape:") or ln == "run refused":
            cur = [ln]
            blocks.append(cur)
        else:
            cur.append(ln)
    assert [b[0] for b in blocks] == ["shape: new", "shape: same", "run refused", "shape: new", "shape: new", "shape: new", "shape: same"]
    assert blocks[1][1:] == ["utime: 5", "items.w: 10 20 30 40", "items.label:", "raw: 1 2 3 4", "grid: 9 8 7 6 5 4"]
    assert blocks[3][1:] == ["utime: 5", "items.w: 10 20 30 40", 'items.label: "left foot" ""', "raw: 1 2 3 4", "grid: 9 8 7 6 5 4"]
    assert blocks[4][3] == 'items.label: "left feet" ""'
    assert blocks[5][1:] == ["utime: 5", "items.w: 10 20", 'items.label: "left foot"', "raw: 1 2 3 4", "grid: 9 8 7 6"]
    assert blocks[6][1:] == ["utime: 5", "items.w: 10 20", "items.label:", "raw: 1 2 3 4", "grid: 9 8 7 6"]
    # a member that does not exist: no plan; a truncated message, a message of another type: refused
    assert run("utime,nonsense") == ["no plan"]
    (tmp_path / "short.bin").write_bytes(msg.read_bytes()[:-3])
    assert run("utime", tmp_path / "short.bin") == ["run refused"]
    assert run("id", msg, "demo.inner_t") == ["run refused"]
    # the bot_core shapes the SegmentBatcher reads (definitions as its test writes them)
    bot = tmp_path / "bot.lcm"
    bot.write_text("package bot_core;\n"
                   "struct six_axis_force_torque_t { int64_t utime; double force[3]; double moment[3]; }\n"
                   "struct six_axis_force_torque_array_t { int64_t utime; int32_t num_sensors; string names[num_sensors]; "
                   "six_axis_force_torque_t sensors[num_sensors]; }\n")
    types = {"bot_core.six_axis_force_torque_t": [("utime", "int64_t", []), ("force", "double", [(0, "3")]), ("moment", "double", [(0, "3")])],
             "bot_core.six_axis_force_torque_array_t": [("utime", "int64_t", []), ("num_sensors", "int32_t", []), ("names", "string", [(1, "num_sensors")]),
                                                        ("sensors", "bot_core.six_axis_force_torque_t", [(1, "num_sensors")])]}
    ft = {"utime": 77, "num_sensors": 2, "names": ["l_foot", "r_foot"],
          "sensors": [{"utime": 77, "force": [1.0, -2.0, 812.5], "moment": [0.1, 0.2, 0.3]}, {"utime": 77, "force": [3.0, 4.0, -90.25], "moment": [0.0, 0.0, 0.0]}]}
    fmsg = tmp_path / "ft.bin"
    fmsg.write_bytes(L.encode_message(types, "bot_core.six_axis_force_torque_array_t", ft))
    out = subprocess.check_output([tool, "plan", str(bot), "bot_core.six_axis_force_torque_array_t", str(fmsg), "utime,sensors.force"], text=True).splitlines()
    assert out == ["utime: 77", "sensors.force: 1 -2 812.5 3 4 -90.25"]
