// pronto_wire.hpp -- wire formats at the seam of the hot path (SURVEY.md 8b "wire types", 8f rank 4): the three
// pronto LCM types the estimator publishes / consumes, and the LCM event-log container recorded segments come in.
// Host-only, header-only C++17; no lcm / lcm-gen dependency (neither exists in this image).
//
//   pronto::filter_state_t         pronto-lcmtypes/lcmtypes/pronto_filter_state_t.lcm:3-11      (published head state)
//   pronto::indexed_measurement_t  pronto-lcmtypes/lcmtypes/pronto_indexed_measurement_t.lcm:3-13
//   pronto::update_t               pronto-lcmtypes/lcmtypes/pronto_update_t.lcm:6-30            (fovis VO delta)
//
// Encoding rules restated from LCM's published type specification ("LCM Type Specification Language": every field
// big-endian, in declaration order; arrays are their elements back to back with no length prefix -- a variable length
// lives in the integer field that names it; the message starts with the 8-byte fingerprint of the type).
// The fingerprint is lcm-gen's: a base hash over (member name, primitive type name, dimensions) starting from
// 0x12345678 with  v = ((v << 8) ^ (v >> 55)) + c  per character (signed 64-bit, arithmetic shift), then rotated left
// by one bit.  KAT (tests/test_wire.py): the LCM tutorial's exlcm::example_t has base hash 0x1baa9e29b0fbaa8b.
// The bot_core types (ins_t, joint_state_t, pose_t ...) are NOT handled: their .lcm schemas are not in the reference
// tree, so their fingerprints and layouts cannot be derived here.
//
// Log container (lcm/eventlog.c): per event  u32 0xEDA1DA01 | i64 event number | i64 timestamp [us] | i32 channel
// length | i32 data length | channel bytes | data bytes, all big-endian; a reader that loses sync scans byte-wise for
// the next sync word.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace pronto_wire {

// ---------------------------------------------------------------------------------------------------------------
// big-endian primitives
// ---------------------------------------------------------------------------------------------------------------
class Writer {
public:
  std::vector<uint8_t> buf;
  void u8(uint8_t v) { buf.push_back(v); }
  void i8(int8_t v) { buf.push_back((uint8_t) v); }
  void u32(uint32_t v) { for (int s = 24; s >= 0; s -= 8) buf.push_back((uint8_t) (v >> s)); }
  void i32(int32_t v) { u32((uint32_t) v); }
  void u64(uint64_t v) { for (int s = 56; s >= 0; s -= 8) buf.push_back((uint8_t) (v >> s)); }
  void i64(int64_t v) { u64((uint64_t) v); }
  void i16(int16_t v) { buf.push_back((uint8_t) ((uint16_t) v >> 8)); buf.push_back((uint8_t) ((uint16_t) v & 0xFF)); }
  void f32(float v) { uint32_t u; memcpy(&u, &v, 4); u32(u); }
  void str(const std::string &s) { i32((int32_t) s.size() + 1); bytes(s.data(), s.size()); u8(0); }  // LCM string: length incl. NUL
  void f64(double v) { uint64_t u; memcpy(&u, &v, 8); u64(u); }
  void f64s(const double *v, size_t n) { for (size_t i = 0; i < n; i++) f64(v[i]); }
  void bytes(const void *p, size_t n) { const uint8_t *b = (const uint8_t *) p; buf.insert(buf.end(), b, b + n); }
};

class Reader {
public:
  const uint8_t *p;
  size_t n, pos = 0;
  bool ok = true;
  Reader(const void *data, size_t len) : p((const uint8_t *) data), n(len) {}
  bool need(size_t k) { if (!ok || n - pos < k) ok = false; return ok; }
  int8_t i8() { if (!need(1)) return 0; return (int8_t) p[pos++]; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[pos++]; return v; }
  int32_t i32() { return (int32_t) u32(); }
  uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[pos++]; return v; }
  int64_t i64() { return (int64_t) u64(); }
  double f64() { uint64_t u = u64(); double v; memcpy(&v, &u, 8); return v; }
  void f64s(double *v, size_t k) { for (size_t i = 0; i < k; i++) v[i] = f64(); }
};

// ---------------------------------------------------------------------------------------------------------------
// lcm-gen's type fingerprint
// ---------------------------------------------------------------------------------------------------------------
enum { LCM_CONST = 0, LCM_VAR = 1 };  // dimension modes

struct Dim {
  int mode;
  const char *size;  // the size exactly as written in the .lcm file ("3", "num_states", ...)
};
struct Member {
  const char *name;
  const char *prim;  // primitive type name as written in the .lcm file; nullptr for a nested struct member
  std::vector<Dim> dims;
};

inline int64_t hash_update(int64_t v, int c)
{
  // signed arithmetic exactly as lcm-gen does it; the left shift is done unsigned to stay defined in C++
  return (int64_t) (((uint64_t) v << 8) ^ (uint64_t) (v >> 55)) + c;
}
inline int64_t hash_string_update(int64_t v, const char *s)
{
  v = hash_update(v, (int) strlen(s));
  for (; *s; s++) v = hash_update(v, (unsigned char) *s);
  return v;
}
inline int64_t lcm_base_hash(const std::vector<Member> &members)
{
  int64_t v = 0x12345678;
  for (const Member &m : members) {
    v = hash_string_update(v, m.name);
    if (m.prim) v = hash_string_update(v, m.prim);
    v = hash_update(v, (int) m.dims.size());
    for (const Dim &d : m.dims) {
      v = hash_update(v, d.mode);
      v = hash_string_update(v, d.size);
    }
  }
  return v;
}
// fingerprint of a struct whose members are all primitive (true for the three pronto types): rotate the base hash
inline uint64_t lcm_fingerprint(const std::vector<Member> &members)
{
  const uint64_t h = (uint64_t) lcm_base_hash(members);
  return (h << 1) + ((h >> 63) & 1u);
}

enum { WIRE_OK = 0, WIRE_ERR_SHORT = -1, WIRE_ERR_FINGERPRINT = -2, WIRE_ERR_LENGTH = -3 };

// ---------------------------------------------------------------------------------------------------------------
// pronto::filter_state_t  (rbis.cpp:287-304 fills it: quat w,x,y,z; state[21]; cov[441] column-major)
// ---------------------------------------------------------------------------------------------------------------
struct filter_state_t {
  int64_t utime = 0;
  double quat[4] = { 1, 0, 0, 0 };
  int32_t num_states = 0;
  std::vector<double> state;
  int32_t num_cov_elements = 0;
  std::vector<double> cov;

  static const std::vector<Member> &members()
  {
    static const std::vector<Member> m = { { "utime", "int64_t", {} },
                                           { "quat", "double", { { LCM_CONST, "4" } } },
                                           { "num_states", "int32_t", {} },
                                           { "state", "double", { { LCM_VAR, "num_states" } } },
                                           { "num_cov_elements", "int32_t", {} },
                                           { "cov", "double", { { LCM_VAR, "num_cov_elements" } } } };
    return m;
  }
  static uint64_t fingerprint() { static const uint64_t f = lcm_fingerprint(members()); return f; }
  void encode(std::vector<uint8_t> &out) const
  {
    Writer w;
    w.u64(fingerprint());
    w.i64(utime);
    w.f64s(quat, 4);
    w.i32(num_states);
    w.f64s(state.data(), (size_t) num_states);
    w.i32(num_cov_elements);
    w.f64s(cov.data(), (size_t) num_cov_elements);
    out.swap(w.buf);
  }
  int decode(const void *data, size_t len)
  {
    Reader r(data, len);
    if (r.u64() != fingerprint()) return r.ok ? WIRE_ERR_FINGERPRINT : WIRE_ERR_SHORT;
    utime = r.i64();
    r.f64s(quat, 4);
    num_states = r.i32();
    if (!r.ok || num_states < 0 || (size_t) num_states > (len - r.pos) / 8) return r.ok ? WIRE_ERR_LENGTH : WIRE_ERR_SHORT;
    state.resize((size_t) num_states);
    r.f64s(state.data(), state.size());
    num_cov_elements = r.i32();
    if (!r.ok || num_cov_elements < 0 || (size_t) num_cov_elements > (len - r.pos) / 8)
      return r.ok ? WIRE_ERR_LENGTH : WIRE_ERR_SHORT;
    cov.resize((size_t) num_cov_elements);
    r.f64s(cov.data(), cov.size());
    return r.ok ? (int) r.pos : WIRE_ERR_SHORT;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// pronto::indexed_measurement_t  (consumed by IndexedMeasurementHandler::processMessage, sensor_handlers.cpp:576-582;
// R_effective is mapped as a column-major measured_dim x measured_dim matrix)
// ---------------------------------------------------------------------------------------------------------------
struct indexed_measurement_t {
  int64_t utime = 0, state_utime = 0;
  int32_t measured_dim = 0;
  std::vector<double> z_effective;
  std::vector<int32_t> z_indices;
  int32_t measured_cov_dim = 0;
  std::vector<double> R_effective;

  static const std::vector<Member> &members()
  {
    static const std::vector<Member> m = { { "utime", "int64_t", {} },
                                           { "state_utime", "int64_t", {} },
                                           { "measured_dim", "int32_t", {} },
                                           { "z_effective", "double", { { LCM_VAR, "measured_dim" } } },
                                           { "z_indices", "int32_t", { { LCM_VAR, "measured_dim" } } },
                                           { "measured_cov_dim", "int32_t", {} },
                                           { "R_effective", "double", { { LCM_VAR, "measured_cov_dim" } } } };
    return m;
  }
  static uint64_t fingerprint() { static const uint64_t f = lcm_fingerprint(members()); return f; }
  void encode(std::vector<uint8_t> &out) const
  {
    Writer w;
    w.u64(fingerprint());
    w.i64(utime);
    w.i64(state_utime);
    w.i32(measured_dim);
    w.f64s(z_effective.data(), (size_t) measured_dim);
    for (int i = 0; i < measured_dim; i++) w.i32(z_indices[(size_t) i]);
    w.i32(measured_cov_dim);
    w.f64s(R_effective.data(), (size_t) measured_cov_dim);
    out.swap(w.buf);
  }
  int decode(const void *data, size_t len)
  {
    Reader r(data, len);
    if (r.u64() != fingerprint()) return r.ok ? WIRE_ERR_FINGERPRINT : WIRE_ERR_SHORT;
    utime = r.i64();
    state_utime = r.i64();
    measured_dim = r.i32();
    if (!r.ok || measured_dim < 0 || (size_t) measured_dim > (len - r.pos) / 12) return r.ok ? WIRE_ERR_LENGTH : WIRE_ERR_SHORT;
    z_effective.resize((size_t) measured_dim);
    r.f64s(z_effective.data(), z_effective.size());
    z_indices.resize((size_t) measured_dim);
    for (auto &v : z_indices) v = r.i32();
    measured_cov_dim = r.i32();
    if (!r.ok || measured_cov_dim < 0 || (size_t) measured_cov_dim > (len - r.pos) / 8)
      return r.ok ? WIRE_ERR_LENGTH : WIRE_ERR_SHORT;
    R_effective.resize((size_t) measured_cov_dim);
    r.f64s(R_effective.data(), R_effective.size());
    return r.ok ? (int) r.pos : WIRE_ERR_SHORT;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// pronto::update_t  (fovis delta; consumed by FovisHandler::processMessage, rbis_fovis_update.cpp:158-312).
// The int8 constants of the .lcm file do not enter the fingerprint or the wire image.
// ---------------------------------------------------------------------------------------------------------------
struct update_t {
  enum { NO_DATA = 0, ESTIMATE_VALID = 1, ESTIMATE_INSUFFICIENT_FEATURES = 2, ESTIMATE_DEGENERATE = 3,
         ESTIMATE_REPROJECTION_ERROR = 4 };
  int64_t timestamp = 0, prev_timestamp = 0;
  double translation[3] = { 0, 0, 0 };
  double rotation[4] = { 1, 0, 0, 0 };  // w, x, y, z
  double covariance[6][6] = {};
  int8_t estimate_status = NO_DATA;

  static const std::vector<Member> &members()
  {
    static const std::vector<Member> m = { { "timestamp", "int64_t", {} },
                                           { "prev_timestamp", "int64_t", {} },
                                           { "translation", "double", { { LCM_CONST, "3" } } },
                                           { "rotation", "double", { { LCM_CONST, "4" } } },
                                           { "covariance", "double", { { LCM_CONST, "6" }, { LCM_CONST, "6" } } },
                                           { "estimate_status", "int8_t", {} } };
    return m;
  }
  static uint64_t fingerprint() { static const uint64_t f = lcm_fingerprint(members()); return f; }
  void encode(std::vector<uint8_t> &out) const
  {
    Writer w;
    w.u64(fingerprint());
    w.i64(timestamp);
    w.i64(prev_timestamp);
    w.f64s(translation, 3);
    w.f64s(rotation, 4);
    for (int i = 0; i < 6; i++) w.f64s(covariance[i], 6);
    w.i8(estimate_status);
    out.swap(w.buf);
  }
  int decode(const void *data, size_t len)
  {
    Reader r(data, len);
    if (r.u64() != fingerprint()) return r.ok ? WIRE_ERR_FINGERPRINT : WIRE_ERR_SHORT;
    timestamp = r.i64();
    prev_timestamp = r.i64();
    r.f64s(translation, 3);
    r.f64s(rotation, 4);
    for (int i = 0; i < 6; i++) r.f64s(covariance[i], 6);
    estimate_status = r.i8();
    return r.ok ? (int) r.pos : WIRE_ERR_SHORT;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// LCM event log
// ---------------------------------------------------------------------------------------------------------------
struct LogEvent {
  int64_t eventnum = 0, timestamp = 0;
  std::string channel;
  std::vector<uint8_t> data;
};

static const uint32_t LOG_SYNC = 0xEDA1DA01u;

class LogWriter {
public:
  explicit LogWriter(const std::string &path) : f_(fopen(path.c_str(), "wb")) {}
  ~LogWriter() { if (f_) fclose(f_); }
  LogWriter(const LogWriter &) = delete;
  LogWriter &operator=(const LogWriter &) = delete;
  bool good() const { return f_ != nullptr; }
  bool write(int64_t timestamp, const std::string &channel, const std::vector<uint8_t> &data)
  {
    if (!f_) return false;
    Writer w;
    w.u32(LOG_SYNC);
    w.i64(next_++);
    w.i64(timestamp);
    w.i32((int32_t) channel.size());
    w.i32((int32_t) data.size());
    w.bytes(channel.data(), channel.size());
    w.bytes(data.data(), data.size());
    return fwrite(w.buf.data(), 1, w.buf.size(), f_) == w.buf.size();
  }
private:
  FILE *f_;
  int64_t next_ = 0;
};

class LogReader {
public:
  explicit LogReader(const std::string &path) : f_(fopen(path.c_str(), "rb"))
  {
    if (f_ && fseek(f_, 0, SEEK_END) == 0) {
      size_ = ftell(f_);
      fseek(f_, 0, SEEK_SET);
    }
    if (f_) setvbuf(f_, nullptr, _IOFBF, 32768);   // (a replay of thousands of segments reads a few messages per refill)
  }
  ~LogReader() { if (f_) fclose(f_); }
  LogReader(const LogReader &) = delete;
  LogReader &operator=(const LogReader &) = delete;
  bool good() const { return f_ != nullptr; }
  // next event, or false at end of file / on a truncated tail.  A corrupt stretch is skipped byte-wise up to the next
  // sync word, like lcm_eventlog_read_next_event.
  bool next(LogEvent &ev)
  {
    if (!f_) return false;
    uint32_t magic = 0;
    int c;
    int have = 0;
    while ((c = fgetc(f_)) != EOF) {
      pos_++;
      magic = (magic << 8) | (uint32_t) c;
      if (++have >= 4 && magic == LOG_SYNC) break;
    }
    if (c == EOF) return false;
    uint8_t hdr[24];
    if (fread(hdr, 1, 24, f_) != 24) return false;
    pos_ += 24;
    Reader r(hdr, 24);
    ev.eventnum = r.i64();
    ev.timestamp = r.i64();
    const int32_t clen = r.i32(), dlen = r.i32();
    if (clen < 0 || clen > 1000 || dlen < 0) return false;  // eventlog.c rejects channel names > 1000
    // a damaged length field must not turn into a multi-gigabyte allocation: the payload cannot outrun the file
    // (the position is counted here: ftell() is a system call per event)
    if (size_ >= 0 && (long) clen + (long) dlen > size_ - pos_) return false;
    ev.channel.resize((size_t) clen);
    if (clen && fread(&ev.channel[0], 1, (size_t) clen, f_) != (size_t) clen) return false;
    ev.data.resize((size_t) dlen);
    if (dlen && fread(ev.data.data(), 1, (size_t) dlen, f_) != (size_t) dlen) return false;
    pos_ += (long) clen + (long) dlen;
    return true;
  }
private:
  FILE *f_;
  long size_ = -1, pos_ = 0;
};

}  // namespace pronto_wire
