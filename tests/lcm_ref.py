"""TEST-ONLY second restatement of the LCM wire rules, written independently of pronto_amd/csrc/pronto_wire.hpp (Python
`struct`, no shared code) so that tests/test_wire.py can compare the two byte for byte.

Sources restated: LCM's type-specification document (big-endian fields in declaration order, arrays without length
prefix, 8-byte fingerprint first), lcm-gen's hash (base 0x12345678, v = ((v<<8) ^ (v>>55)) + c in signed 64-bit, one-bit
left rotation at the end) and lcm/eventlog.c's event framing.  Field lists follow
pronto-lcmtypes/lcmtypes/pronto_{filter_state,indexed_measurement,update}_t.lcm.
"""
import struct

M64 = (1 << 64) - 1
SYNC = 0xEDA1DA01


def _s64(v):
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def _upd(v, c):
    v = _s64(v)
    return _s64(((v << 8) ^ (v >> 55)) + c)  # python's >> on a negative int is arithmetic, like C's on int64_t


def _upd_str(v, s):
    v = _upd(v, len(s))
    for ch in s.encode():
        v = _upd(v, ch)
    return v


def base_hash(members):
    """members: list of (name, primitive type name, [(mode, size string), ...]); mode 0 = constant, 1 = variable."""
    v = 0x12345678
    for name, typ, dims in members:
        v = _upd_str(v, name)
        v = _upd_str(v, typ)
        v = _upd(v, len(dims))
        for mode, size in dims:
            v = _upd(v, mode)
            v = _upd_str(v, size)
    return v & M64


def fingerprint(members):
    h = base_hash(members)
    return ((h << 1) + (h >> 63)) & M64


EXAMPLE_T = [("timestamp", "int64_t", []), ("position", "double", [(0, "3")]), ("orientation", "double", [(0, "4")]),
             ("num_ranges", "int32_t", []), ("ranges", "int16_t", [(1, "num_ranges")]), ("name", "string", []),
             ("enabled", "boolean", [])]
FILTER_STATE_T = [("utime", "int64_t", []), ("quat", "double", [(0, "4")]), ("num_states", "int32_t", []),
                  ("state", "double", [(1, "num_states")]), ("num_cov_elements", "int32_t", []),
                  ("cov", "double", [(1, "num_cov_elements")])]
INDEXED_MEASUREMENT_T = [("utime", "int64_t", []), ("state_utime", "int64_t", []), ("measured_dim", "int32_t", []),
                         ("z_effective", "double", [(1, "measured_dim")]), ("z_indices", "int32_t", [(1, "measured_dim")]),
                         ("measured_cov_dim", "int32_t", []), ("R_effective", "double", [(1, "measured_cov_dim")])]
UPDATE_T = [("timestamp", "int64_t", []), ("prev_timestamp", "int64_t", []), ("translation", "double", [(0, "3")]),
            ("rotation", "double", [(0, "4")]), ("covariance", "double", [(0, "6"), (0, "6")]),
            ("estimate_status", "int8_t", [])]


def enc_filter_state(utime, quat, state, cov):
    return (struct.pack(">Qq4di", fingerprint(FILTER_STATE_T), utime, *quat, len(state)) +
            struct.pack(">%dd" % len(state), *state) + struct.pack(">i", len(cov)) + struct.pack(">%dd" % len(cov), *cov))


def enc_indexed_measurement(utime, state_utime, z, idx, R):
    m = len(z)
    return (struct.pack(">Qqqi", fingerprint(INDEXED_MEASUREMENT_T), utime, state_utime, m) + struct.pack(">%dd" % m, *z) +
            struct.pack(">%di" % m, *idx) + struct.pack(">i", len(R)) + struct.pack(">%dd" % len(R), *R))


def enc_update(timestamp, prev_timestamp, translation, rotation, cov6x6, status):
    flat = [v for row in cov6x6 for v in row]
    return struct.pack(">Qqq3d4d36db", fingerprint(UPDATE_T), timestamp, prev_timestamp, *translation, *rotation, *flat, status)


def log_event(eventnum, timestamp, channel, data):
    ch = channel.encode()
    return struct.pack(">Iqqii", SYNC, eventnum, timestamp, len(ch), len(data)) + ch + data


def read_log(blob):
    """-> [(eventnum, timestamp, channel, data)], scanning for the sync word like lcm_eventlog_read_next_event."""
    out, pos = [], 0
    while True:
        pos = blob.find(struct.pack(">I", SYNC), pos)
        if pos < 0 or pos + 28 > len(blob):
            return out
        num, ts, clen, dlen = struct.unpack(">qqii", blob[pos + 4:pos + 28])
        end = pos + 28 + clen + dlen
        if clen < 0 or dlen < 0 or end > len(blob):
            return out
        out.append((num, ts, blob[pos + 28:pos + 28 + clen].decode(), blob[pos + 28 + clen:end]))
        pos = end


# ---- generic types (tests of pronto_amd/csrc/lcm_schema.hpp) --------------------------------------------------------------
PRIMS = {"int8_t": ">b", "int16_t": ">h", "int32_t": ">i", "int64_t": ">q", "byte": ">B", "float": ">f", "double": ">d",
         "boolean": ">b"}


def base_hash_nested(fields):
    """Like base_hash, for fields whose type may be a struct: (name, type or None for a struct member, dims)."""
    v = 0x12345678
    for name, typ, dims in fields:
        v = _upd_str(v, name)
        if typ in PRIMS or typ == "string":
            v = _upd_str(v, typ)
        v = _upd(v, len(dims))
        for mode, size in dims:
            v = _upd(v, mode)
            v = _upd_str(v, size)
    return v & M64


def fingerprint_nested(types, name, parents=()):
    """lcm-gen's recursive hash: base + the nested members' hashes (0 for a type already among the parents), rotated."""
    if name in parents:
        return 0
    h = base_hash_nested(types[name])
    for _, typ, _ in types[name]:
        if typ not in PRIMS and typ != "string":
            h = (h + fingerprint_nested(types, typ, parents + (name,))) & M64
    return ((h << 1) + (h >> 63)) & M64


def _enc_value(types, typ, dims, value, scope):
    if dims:
        mode, size = dims[0]
        n = int(size) if mode == 0 else scope[size]
        assert len(value) == n, (typ, size, len(value), n)
        return b"".join(_enc_value(types, typ, dims[1:], v, scope) for v in value)
    if typ == "string":
        b = value.encode()
        return struct.pack(">i", len(b) + 1) + b + b"\0"
    if typ in PRIMS:
        return struct.pack(PRIMS[typ], value)
    return encode_struct(types, typ, value)


def encode_struct(types, name, value):
    return b"".join(_enc_value(types, typ, dims, value[fname], value) for fname, typ, dims in types[name])


def encode_message(types, name, value):
    return struct.pack(">Q", fingerprint_nested(types, name)) + encode_struct(types, name, value)
