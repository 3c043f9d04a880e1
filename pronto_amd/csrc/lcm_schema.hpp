// lcm_schema.hpp -- run-time LCM types: parse .lcm definitions, compute lcm-gen's fingerprints (nested types included),
// decode messages into a generic value tree.  Host-only, header-only C++17, no lcm / lcm-gen dependency.
//
// Why: the bot_core messages the reference's handlers consume (ins_t, kvh_raw_imu_batch_t, joint_state_t, pose_t,
// rigid_transform_t, gps_data_t ...) are defined in libbot's bot_core .lcm files, which are NOT in the reference tree, so
// their layouts and fingerprints cannot be compiled in (pronto_wire.hpp only carries the three pronto types that are).
// A user replaying a recorded log has those .lcm files: hand their text to Schema::parse and the log's events decode by
// field NAME -- the names the reference's handlers read (msg->gyro, msg->accel, msg->pos, msg->orientation,
// msg->raw_imu[i].delta_rotation ...: sensor_handlers.cpp:96-131,165-252,374-381,476-536,689-724, imu_stream.cpp:62-98).
//
// Rules restated from LCM's type-specification document and lcm-gen's emitted code:
//  * grammar: [package P;] struct N { member* };  member: TYPE name[dim]... {, name[dim]...};  |  const TYPE n = v {, ...};
//    dim = integer constant or the name of an earlier integer member; TYPE = primitive | [package.]struct name
//  * primitives: int8_t int16_t int32_t int64_t byte float double string boolean; all big-endian; string = int32 length
//    (with the terminating NUL) + bytes + NUL; boolean and byte one octet; arrays element after element, outer dimension
//    first, no length prefix; a nested struct is encoded inline WITHOUT a fingerprint; only the message starts with one
//  * fingerprint(S) = rot1(base(S) + sum over members of non-primitive type of fingerprint(member type | parents + S)),
//    a type already among its parents contributing 0; base(S) as in pronto_wire.hpp (member names, primitive type names,
//    dimensions; compound type names are NOT hashed); rot1(h) = (h << 1) + (h >> 63).
// The all-primitive case is pinned by the LCM tutorial's example_t constant (tests/test_wire.py); the nested rule is
// restated twice (here and in tests/lcm_ref.py) but has no known answer available in this image.
#pragma once

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "pronto_wire.hpp"

namespace pronto_wire {

struct SchemaDim {
  int mode;          // LCM_CONST / LCM_VAR
  std::string size;  // as written
};
struct SchemaField {
  std::string name, type;  // type as written ("double", "inner_t", "bot_core.kvh_raw_imu_t")
  std::vector<SchemaDim> dims;
};
struct SchemaType {
  std::string package, name;
  std::vector<SchemaField> fields;
  std::string full() const { return package.empty() ? name : package + "." + name; }
};

// one decoded value: a number, a string, a struct (named fields) or an array (items)
struct Value {
  enum Kind { INT, FLOAT, STRING, STRUCT, ARRAY } kind = INT;
  int64_t i = 0;
  double f = 0.0;
  std::string s;
  std::vector<Value> items;
  std::vector<std::pair<std::string, Value>> fields;

  const Value *get(const std::string &name) const
  {
    for (const auto &kv : fields)
      if (kv.first == name) return &kv.second;
    return nullptr;
  }
  double number() const { return kind == FLOAT ? f : (double) i; }
  // numeric array member -> out[0..n); false if the member is missing, not an array or of another length
  bool numbers(const std::string &name, double *out, size_t n) const
  {
    const Value *v = get(name);
    if (v == nullptr || v->kind != ARRAY || v->items.size() != n) return false;
    for (size_t k = 0; k < n; k++) out[k] = v->items[k].number();
    return true;
  }
  bool integer(const std::string &name, int64_t &out) const
  {
    const Value *v = get(name);
    if (v == nullptr || v->kind != INT) return false;
    out = v->i;
    return true;
  }
};

class Schema {
public:
  // Adds every struct of `text` (one .lcm file's contents; several files = several calls).  false + *err on a syntax error.
  bool parse(const std::string &text, std::string *err = nullptr)
  {
    Lexer lx(text);
    std::string package;
    for (;;) {
      std::string t = lx.next();
      if (t.empty()) return true;
      if (t == "package") {
        package = lx.next();
        if (lx.next() != ";") return fail(err, "expected ';' after package");
      } else if (t == "struct") {
        SchemaType st;
        st.package = package;
        st.name = lx.next();
        if (st.name.empty() || lx.next() != "{") return fail(err, "expected '{' after struct " + st.name);
        for (;;) {
          std::string type = lx.next();
          if (type.empty()) return fail(err, "unterminated struct " + st.name);
          if (type == "}") break;
          const bool is_const = (type == "const");
          if (is_const) type = lx.next();
          for (;;) {
            SchemaField f;
            f.type = type;
            f.name = lx.next();
            if (f.name.empty()) return fail(err, "member name expected in " + st.name);
            std::string t2 = lx.next();
            while (t2 == "[") {
              SchemaDim d;
              d.size = lx.next();
              d.mode = (!d.size.empty() && isdigit((unsigned char) d.size[0])) ? LCM_CONST : LCM_VAR;
              if (lx.next() != "]") return fail(err, "expected ']' in " + st.name + "." + f.name);
              f.dims.push_back(d);
              t2 = lx.next();
            }
            if (is_const) {  // const TYPE name = value: not on the wire, not in the fingerprint
              if (t2 != "=") return fail(err, "expected '=' in const " + f.name);
              lx.next();
              t2 = lx.next();
            } else {
              st.fields.push_back(f);
            }
            if (t2 == ",") continue;
            if (t2 == ";") break;
            return fail(err, "expected ',' or ';' after " + st.name + "." + f.name);
          }
        }
        types_[st.full()] = st;
      } else if (t != ";") {
        return fail(err, "unexpected token '" + t + "'");
      }
    }
  }

  // "package.name", or a bare name if only one package defines it
  const SchemaType *find(const std::string &name) const
  {
    auto it = types_.find(name);
    if (it != types_.end()) return &it->second;
    const SchemaType *hit = nullptr;
    for (const auto &kv : types_)
      if (kv.second.name == name) {
        if (hit != nullptr) return nullptr;
        hit = &kv.second;
      }
    return hit;
  }

  static bool is_primitive(const std::string &t)
  {
    static const char *prim[] = { "int8_t", "int16_t", "int32_t", "int64_t", "byte", "float", "double", "string", "boolean" };
    for (const char *p : prim)
      if (t == p) return true;
    return false;
  }

  // 0 if the type (or one it nests) is unknown
  uint64_t fingerprint(const std::string &name) const
  {
    const SchemaType *t = find(name);
    if (t == nullptr) return 0;
    std::vector<const SchemaType *> parents;
    bool ok = true;
    const uint64_t h = hash_recursive(*t, parents, ok);
    return ok ? h : 0;
  }

  // decodes one message (fingerprint + body) of type `name`; false + *err if it does not fit
  bool decode(const std::string &name, const void *data, size_t len, Value &out, std::string *err = nullptr) const
  {
    const SchemaType *t = find(name);
    if (t == nullptr) return fail(err, "unknown type " + name);
    Reader r(data, len);
    const uint64_t fp = r.u64();
    if (!r.ok) return fail(err, "short message");
    if (fp != fingerprint(name)) return fail(err, "fingerprint mismatch for " + name);
    if (!decode_struct(*t, r, out, err, 0)) return false;
    if (r.pos != len) return fail(err, "trailing bytes after " + name);
    return true;
  }

  // ---- fast path for replays: a few named members out of a message, without the value tree ----
  // A plan is compiled once per (type, wanted members) and run per message: it walks the encoding in member order, copies the
  // wanted members into flat arrays and steps over everything else (fixed-size runs in one jump).  wanted[k]: a top-level member
  // ("utime", "gyro", "joint_name"), or "array.member" for one member of every element of an array of structs
  // ("sensors.force").  Numbers of any primitive type come out as doubles, strings as strings.
  struct Extracted {
    std::vector<double> num;
    std::vector<std::string> str;
  };
  class Plan {
  public:
    bool ok() const { return root_ >= 0; }
    // false: fingerprint mismatch, short or over-long message
    bool run(const void *data, size_t len, std::vector<Extracted> &out) const
    {
      if (root_ < 0) return false;
      out.resize(n_slots_);
      for (Extracted &e : out) { e.num.clear(); e.str.clear(); }
      Reader r(data, len);
      if (r.u64() != fp_ || !r.ok) return false;
      return walk(root_, r, out, nullptr) && r.pos == len;
    }
    // The byte layout of ONE message, kept by the caller per message stream (one log's channel): a recorded stream repeats its
    // shape -- the same array lengths, the same strings (joint names) -- message after message, and then every wanted number sits
    // at the offset it had last time.  run() with a Shape checks exactly that (the message length, the length members, the string
    // length fields and the wanted strings' bytes against its copy: a handful of memcmp) and copies the numbers out by offset,
    // without walking the structure or building a string; anything different is decoded the long way and becomes the new shape.
    struct Shape {
      bool valid = false;
      size_t len = 0;
      struct Cmp { uint32_t off, n, at; };                       // bytes [off, off + n) must equal expect[at, at + n)
      struct Op { uint32_t off, count; uint8_t kind; int16_t slot; };
      std::vector<Cmp> cmps;
      std::vector<Op> ops;
      std::vector<uint8_t> expect;
      void clear() { valid = false; cmps.clear(); ops.clear(); expect.clear(); }
    };
    // The streaming decoders' entry (segment_stream.hpp): true = `sh` IS the layout of this message -- the cached one, verified
    // (length, length members, string lengths, wanted strings: a handful of memcmp), or rebuilt the long way (*rebuilt; `out` then
    // holds the decoded members incl. the wanted strings).  The wanted numbers are then read in place with gather_f64 / _f32 /
    // _i64 from sh.ops: no value vectors, no conversions through double for float members.
    bool layout(const void *data, size_t len, Shape &sh, std::vector<Extracted> &out, bool *rebuilt) const
    {
      if (rebuilt) *rebuilt = false;
      if (root_ < 0) return false;
      if (sh.valid && len == sh.len) {
        const uint8_t *p = (const uint8_t *) data;
        bool same = true;
        for (const Shape::Cmp &c : sh.cmps)
          if (memcmp(p + c.off, sh.expect.data() + c.at, c.n) != 0) { same = false; break; }
        if (same) return true;
      }
      sh.valid = false;
      bool same_strings = false;
      if (!run(data, len, out, sh, &same_strings)) return false;
      if (rebuilt) *rebuilt = true;
      return true;
    }
    // the numbers of wanted member `slot` of a message whose layout is `sh`, in encoding order; returns how many (at most cap)
    static size_t gather_f64(const Shape &sh, const void *data, int slot, double *out, size_t cap)
    {
      const uint8_t *p = (const uint8_t *) data;
      size_t n = 0;
      for (const Shape::Op &o : sh.ops) {
        if (o.slot != slot) continue;
        const uint8_t *q = p + o.off;
        for (uint32_t k = 0; k < o.count && n < cap; k++, n++) {
          switch ((Kind) o.kind) {
          case I8: out[n] = (double) (int8_t) q[k]; break;
          case U8: out[n] = (double) q[k]; break;
          case I16: out[n] = (double) (int16_t) ((q[2 * k] << 8) | q[2 * k + 1]); break;
          case I32: { uint32_t u; memcpy(&u, q + 4 * k, 4); out[n] = (double) (int32_t) __builtin_bswap32(u); break; }
          case I64: { uint64_t u; memcpy(&u, q + 8 * k, 8); out[n] = (double) (int64_t) __builtin_bswap64(u); break; }
          case F32: { uint32_t u; memcpy(&u, q + 4 * k, 4); u = __builtin_bswap32(u); float x; memcpy(&x, &u, 4); out[n] = x; break; }
          default: { uint64_t u; memcpy(&u, q + 8 * k, 8); u = __builtin_bswap64(u); memcpy(&out[n], &u, 8); break; }
          }
        }
      }
      return n;
    }
    static size_t gather_f32(const Shape &sh, const void *data, int slot, float *out, size_t cap)
    {
      const uint8_t *p = (const uint8_t *) data;
      size_t n = 0;
      for (const Shape::Op &o : sh.ops) {
        if (o.slot != slot) continue;
        const uint8_t *q = p + o.off;
        if ((Kind) o.kind == F32) {
          for (uint32_t k = 0; k < o.count && n < cap; k++, n++) { uint32_t u; memcpy(&u, q + 4 * k, 4); u = __builtin_bswap32(u); memcpy(&out[n], &u, 4); }
        } else {
          double v;
          Shape one = Shape();
          one.ops.push_back(o);
          for (uint32_t k = 0; k < o.count && n < cap; k++, n++) {
            one.ops[0].off = o.off + k * (uint32_t) prim_size((Kind) o.kind);
            one.ops[0].count = 1;
            gather_f64(one, data, slot, &v, 1);
            out[n] = (float) v;
          }
        }
      }
      return n;
    }
    static size_t gather_i64(const Shape &sh, const void *data, int slot, int64_t *out, size_t cap)
    {
      const uint8_t *p = (const uint8_t *) data;
      size_t n = 0;
      for (const Shape::Op &o : sh.ops) {
        if (o.slot != slot) continue;
        const uint8_t *q = p + o.off;
        for (uint32_t k = 0; k < o.count && n < cap; k++, n++) {
          switch ((Kind) o.kind) {
          case I8: out[n] = (int8_t) q[k]; break;
          case U8: out[n] = q[k]; break;
          case I16: out[n] = (int16_t) ((q[2 * k] << 8) | q[2 * k + 1]); break;
          case I32: { uint32_t u; memcpy(&u, q + 4 * k, 4); out[n] = (int32_t) __builtin_bswap32(u); break; }
          case I64: { uint64_t u; memcpy(&u, q + 8 * k, 8); out[n] = (int64_t) __builtin_bswap64(u); break; }
          default: { double v; Shape one = Shape(); one.ops.push_back(o); one.ops[0].off = o.off + k * (uint32_t) prim_size((Kind) o.kind); one.ops[0].count = 1; gather_f64(one, data, slot, &v, 1); out[n] = (int64_t) v; break; }
          }
        }
      }
      return n;
    }
    // as run(); *same_strings (may be NULL) = the message had the cached shape: out[k].str is then left EMPTY -- the strings are
    // the ones the call that made the shape returned
    bool run(const void *data, size_t len, std::vector<Extracted> &out, Shape &sh, bool *same_strings) const
    {
      if (same_strings) *same_strings = false;
      if (root_ < 0) return false;
      const uint8_t *p = (const uint8_t *) data;
      if (sh.valid && len == sh.len) {
        bool same = true;
        for (const Shape::Cmp &c : sh.cmps)
          if (memcmp(p + c.off, sh.expect.data() + c.at, c.n) != 0) { same = false; break; }
        if (same) {
          out.resize(n_slots_);
          for (Extracted &e : out) { e.num.clear(); e.str.clear(); }
          for (const Shape::Op &o : sh.ops) {
            std::vector<double> &v = out[(size_t) o.slot].num;
            const uint8_t *q = p + o.off;
            const size_t base = v.size();
            v.resize(base + o.count);
            double *dst = v.data() + base;
            switch ((Kind) o.kind) {
            case I8: for (uint32_t k = 0; k < o.count; k++) dst[k] = (double) (int8_t) q[k]; break;
            case U8: for (uint32_t k = 0; k < o.count; k++) dst[k] = (double) q[k]; break;
            case I16: for (uint32_t k = 0; k < o.count; k++) dst[k] = (double) (int16_t) ((q[2 * k] << 8) | q[2 * k + 1]); break;
            case I32: for (uint32_t k = 0; k < o.count; k++) { uint32_t u; memcpy(&u, q + 4 * k, 4); dst[k] = (double) (int32_t) __builtin_bswap32(u); } break;
            case I64: for (uint32_t k = 0; k < o.count; k++) { uint64_t u; memcpy(&u, q + 8 * k, 8); dst[k] = (double) (int64_t) __builtin_bswap64(u); } break;
            case F32: for (uint32_t k = 0; k < o.count; k++) { uint32_t u; memcpy(&u, q + 4 * k, 4); u = __builtin_bswap32(u); float x; memcpy(&x, &u, 4); dst[k] = x; } break;
            default: for (uint32_t k = 0; k < o.count; k++) { uint64_t u; memcpy(&u, q + 8 * k, 8); u = __builtin_bswap64(u); memcpy(&dst[k], &u, 8); } break;
            }
          }
          if (same_strings) *same_strings = true;
          return true;
        }
      }
      sh.clear();
      out.resize(n_slots_);
      for (Extracted &e : out) { e.num.clear(); e.str.clear(); }
      Reader r(data, len);
      if (r.u64() != fp_ || !r.ok) return false;
      sh.cmps.push_back(Shape::Cmp{ 0u, 8u, 0u });     // the fingerprint
      if (!walk(root_, r, out, &sh) || r.pos != len || len > 0xffffffffull) {
        sh.clear();
        return false;
      }
      // the expected bytes, adjacent ranges merged
      std::vector<Shape::Cmp> merged;
      for (const Shape::Cmp &c : sh.cmps) {
        if (!merged.empty() && merged.back().off + merged.back().n == c.off) merged.back().n += c.n;
        else merged.push_back(c);
      }
      sh.expect.clear();
      for (Shape::Cmp &c : merged) {
        c.at = (uint32_t) sh.expect.size();
        sh.expect.insert(sh.expect.end(), p + c.off, p + c.off + c.n);
      }
      sh.cmps = std::move(merged);
      sh.len = len;
      sh.valid = true;
      return true;
    }
  private:
    friend class Schema;
    enum Kind { I8, U8, I16, I32, I64, F32, F64, STR, STRUCT };
    struct Dim { int64_t n; int from; };              // constant length, or the value of integer member `from` of the same struct
    struct Field { Kind kind; int slot, node; std::vector<Dim> dims; bool is_len; };
    struct Node { std::vector<Field> f; int fixed_bytes; };   // fixed_bytes >= 0: no strings, no variable arrays, nothing wanted inside
    std::vector<Node> nodes_;
    int root_ = -1;
    size_t n_slots_ = 0;
    uint64_t fp_ = 0;
    static int prim_size(Kind k) { return (k == I8 || k == U8) ? 1 : k == I16 ? 2 : (k == I32 || k == F32) ? 4 : 8; }
    // sh != NULL: record where the wanted numbers were, and what fixes the layout (length members, string lengths, wanted strings)
    bool walk(int ni, Reader &r, std::vector<Extracted> &out, Shape *sh) const
    {
      const Node &nd = nodes_[(size_t) ni];
      int64_t ints[16];
      int n_ints = 0;
      for (const Field &f : nd.f) {
        int64_t count = 1;
        for (const Dim &d : f.dims) count *= d.from >= 0 ? ints[d.from] : d.n;
        if (count < 0 || (size_t) count > r.n - r.pos + 1) return false;
        if (f.kind == STRUCT) {
          const Node &sub = nodes_[(size_t) f.node];
          if (sub.fixed_bytes >= 0) {
            if (!r.need((size_t) count * (size_t) sub.fixed_bytes)) return false;
            r.pos += (size_t) count * (size_t) sub.fixed_bytes;
          } else {
            for (int64_t k = 0; k < count; k++)
              if (!walk(f.node, r, out, sh)) return false;
          }
          continue;
        }
        if (f.kind == STR) {
          for (int64_t k = 0; k < count; k++) {
            const size_t at = r.pos;
            const int32_t n = r.i32();
            if (!r.ok || n < 1 || !r.need((size_t) n)) return false;
            if (f.slot >= 0) out[(size_t) f.slot].str.emplace_back((const char *) r.p + r.pos, (size_t) n - 1);
            if (sh) sh->cmps.push_back(Shape::Cmp{ (uint32_t) at, (uint32_t) (f.slot >= 0 ? 4 + n : 4), 0u });
            r.pos += (size_t) n;
          }
          continue;
        }
        const int sz = prim_size(f.kind);
        if (!r.need((size_t) count * (size_t) sz)) return false;
        if (f.slot < 0 && !f.is_len) {
          r.pos += (size_t) count * (size_t) sz;
          continue;
        }
        if (sh) {
          if (f.is_len) sh->cmps.push_back(Shape::Cmp{ (uint32_t) r.pos, (uint32_t) (count * sz), 0u });
          if (f.slot >= 0 && count > 0) {
            // (consecutive elements of one wanted member: one run; a member inside an array of structs adds a run per element)
            if (!sh->ops.empty() && sh->ops.back().slot == f.slot && sh->ops.back().kind == (uint8_t) f.kind &&
                sh->ops.back().off + sh->ops.back().count * (uint32_t) sz == (uint32_t) r.pos)
              sh->ops.back().count += (uint32_t) count;
            else sh->ops.push_back(Shape::Op{ (uint32_t) r.pos, (uint32_t) count, (uint8_t) f.kind, (int16_t) f.slot });
          }
        }
        for (int64_t k = 0; k < count; k++) {
          double v = 0.0;
          int64_t iv = 0;
          switch (f.kind) {
          case I8: iv = r.i8(); v = (double) iv; break;
          case U8: iv = (uint8_t) r.i8(); v = (double) iv; break;   // `byte` is unsigned
          case I16: iv = (int16_t) ((r.p[r.pos] << 8) | r.p[r.pos + 1]); r.pos += 2; v = (double) iv; break;
          case I32: iv = r.i32(); v = (double) iv; break;
          case I64: iv = r.i64(); v = (double) iv; break;
          case F32: { const uint32_t u = r.u32(); float x; memcpy(&x, &u, 4); v = x; break; }
          default: v = r.f64(); break;
          }
          if (f.slot >= 0) out[(size_t) f.slot].num.push_back(f.kind == I64 ? (double) iv : v);
          if (f.is_len && n_ints < 16) ints[n_ints] = iv;
        }
        if (f.is_len) n_ints++;
      }
      return r.ok;
    }
  };
  // (an int64 member comes out as a double: exact below 2^53 -- microsecond time stamps are)
  Plan compile(const std::string &type, const std::vector<std::string> &wanted) const
  {
    Plan p;
    const SchemaType *t = find(type);
    if (t == nullptr) return p;
    p.fp_ = fingerprint(type);
    p.n_slots_ = wanted.size();
    std::vector<bool> used(wanted.size(), false);
    const int root = compile_node(*t, "", wanted, used, p, 0);
    for (bool u : used)
      if (!u) return p;                      // a wanted member does not exist
    p.root_ = root;
    return p;
  }

private:
  std::map<std::string, SchemaType> types_;

  int compile_node(const SchemaType &t, const std::string &prefix, const std::vector<std::string> &wanted, std::vector<bool> &used, Plan &p,
                   int depth) const
  {
    if (depth > 32) return -1;
    const int me = (int) p.nodes_.size();
    p.nodes_.emplace_back();
    Plan::Node nd;
    nd.fixed_bytes = 0;
    std::vector<std::string> int_names;   // integer members other members take their lengths from, in order
    for (const SchemaField &f : t.fields)
      for (const SchemaDim &d : f.dims)
        if (d.mode == LCM_VAR && std::find(int_names.begin(), int_names.end(), d.size) == int_names.end()) int_names.push_back(d.size);
    // ints[] is filled in member order: index = position among the length members AS THEY APPEAR in the struct
    std::vector<std::string> seen_len;
    for (const SchemaField &f : t.fields) {
      Plan::Field pf;
      pf.slot = -1;
      pf.node = -1;
      pf.is_len = std::find(int_names.begin(), int_names.end(), f.name) != int_names.end();
      const std::string path = prefix + f.name;
      for (size_t k = 0; k < wanted.size(); k++)
        if (wanted[k] == path) { pf.slot = (int) k; used[k] = true; }
      bool variable = false;
      for (const SchemaDim &d : f.dims) {
        Plan::Dim pd{ 0, -1 };
        if (d.mode == LCM_CONST) pd.n = atoll(d.size.c_str());
        else {
          const auto it = std::find(seen_len.begin(), seen_len.end(), d.size);
          if (it == seen_len.end()) return -1;          // a length member must precede its array
          pd.from = (int) (it - seen_len.begin());
          variable = true;
        }
        pf.dims.push_back(pd);
      }
      if (is_primitive(f.type)) {
        pf.kind = f.type == "double" ? Plan::F64 : f.type == "float" ? Plan::F32 : f.type == "int64_t" ? Plan::I64 : f.type == "int32_t" ? Plan::I32
                  : f.type == "int16_t" ? Plan::I16 : f.type == "string" ? Plan::STR : f.type == "byte" ? Plan::U8 : Plan::I8;
        if (pf.kind == Plan::STR || variable || pf.slot >= 0 || pf.is_len) nd.fixed_bytes = -1;
        else if (nd.fixed_bytes >= 0) {
          int64_t c = 1;
          for (const Plan::Dim &d : pf.dims) c *= d.n;
          nd.fixed_bytes += (int) c * Plan::prim_size(pf.kind);
        }
      } else {
        const SchemaType *nested = resolve(t, f.type);
        if (nested == nullptr) return -1;
        pf.kind = Plan::STRUCT;
        pf.node = compile_node(*nested, path + ".", wanted, used, p, depth + 1);
        if (pf.node < 0) return -1;
        const int sub = p.nodes_[(size_t) pf.node].fixed_bytes;
        if (sub < 0 || variable) nd.fixed_bytes = -1;
        else if (nd.fixed_bytes >= 0) {
          int64_t c = 1;
          for (const Plan::Dim &d : pf.dims) c *= d.n;
          nd.fixed_bytes += (int) c * sub;
        }
      }
      if (pf.is_len) {
        if (seen_len.size() >= 16) return -1;
        seen_len.push_back(f.name);
      }
      nd.f.push_back(pf);
    }
    p.nodes_[(size_t) me] = std::move(nd);
    return me;
  }

  static bool fail(std::string *err, const std::string &what)
  {
    if (err) *err = what;
    return false;
  }

  const SchemaType *resolve(const SchemaType &ctx, const std::string &type) const
  {
    if (type.find('.') != std::string::npos) {
      auto it = types_.find(type);
      return it == types_.end() ? nullptr : &it->second;
    }
    auto it = types_.find(ctx.package.empty() ? type : ctx.package + "." + type);
    return it == types_.end() ? nullptr : &it->second;
  }

  uint64_t hash_recursive(const SchemaType &t, std::vector<const SchemaType *> &parents, bool &ok) const
  {
    for (const SchemaType *p : parents)
      if (p == &t) return 0;
    std::vector<Member> members;
    for (const SchemaField &f : t.fields) {
      Member m;
      m.name = f.name.c_str();
      m.prim = is_primitive(f.type) ? f.type.c_str() : nullptr;
      for (const SchemaDim &d : f.dims) m.dims.push_back(Dim{ d.mode, d.size.c_str() });
      members.push_back(m);
    }
    uint64_t h = (uint64_t) lcm_base_hash(members);
    parents.push_back(&t);
    for (const SchemaField &f : t.fields) {
      if (is_primitive(f.type)) continue;
      const SchemaType *nested = resolve(t, f.type);
      if (nested == nullptr) {
        ok = false;
        continue;
      }
      h += hash_recursive(*nested, parents, ok);
    }
    parents.pop_back();
    return (h << 1) + ((h >> 63) & 1u);
  }

  bool decode_primitive(const std::string &type, Reader &r, Value &v, std::string *err) const
  {
    if (type == "double") { v.kind = Value::FLOAT; v.f = r.f64(); }
    else if (type == "float") {
      const uint32_t u = r.u32();
      float x;
      memcpy(&x, &u, 4);
      v.kind = Value::FLOAT;
      v.f = x;
    }
    else if (type == "int64_t") { v.kind = Value::INT; v.i = r.i64(); }
    else if (type == "int32_t") { v.kind = Value::INT; v.i = r.i32(); }
    else if (type == "int16_t") {
      if (!r.need(2)) return fail(err, "short message");
      v.kind = Value::INT;
      v.i = (int16_t) ((r.p[r.pos] << 8) | r.p[r.pos + 1]);
      r.pos += 2;
    }
    else if (type == "int8_t" || type == "boolean") { v.kind = Value::INT; v.i = r.i8(); }
    else if (type == "byte") { v.kind = Value::INT; v.i = (uint8_t) r.i8(); }
    else if (type == "string") {
      const int32_t n = r.i32();
      if (!r.ok || n < 1 || !r.need((size_t) n)) return fail(err, "bad string length");
      v.kind = Value::STRING;
      v.s.assign((const char *) r.p + r.pos, (size_t) n - 1);
      r.pos += (size_t) n;
    }
    else return fail(err, "not a primitive: " + type);
    return r.ok ? true : fail(err, "short message");
  }

  bool decode_elements(const SchemaType &ctx, const SchemaField &f, size_t dim, const Value &self, Reader &r, Value &out,
                       std::string *err, int depth) const
  {
    if (dim == f.dims.size()) {
      if (is_primitive(f.type)) return decode_primitive(f.type, r, out, err);
      const SchemaType *nested = resolve(ctx, f.type);
      if (nested == nullptr) return fail(err, "unknown type " + f.type);
      return decode_struct(*nested, r, out, err, depth + 1);
    }
    int64_t n = 0;
    if (f.dims[dim].mode == LCM_CONST) n = atoll(f.dims[dim].size.c_str());
    else if (!self.integer(f.dims[dim].size, n)) return fail(err, "array length member " + f.dims[dim].size + " not found");
    if (n < 0 || (size_t) n > r.n - r.pos) return fail(err, "array length of " + f.name + " exceeds the message");
    out.kind = Value::ARRAY;
    out.items.resize((size_t) n);
    for (int64_t k = 0; k < n; k++)
      if (!decode_elements(ctx, f, dim + 1, self, r, out.items[(size_t) k], err, depth)) return false;
    return true;
  }

  bool decode_struct(const SchemaType &t, Reader &r, Value &out, std::string *err, int depth) const
  {
    if (depth > 32) return fail(err, "types nest deeper than 32 levels (a struct containing itself?)");
    out = Value();
    out.kind = Value::STRUCT;
    for (const SchemaField &f : t.fields) {
      Value v;
      if (!decode_elements(t, f, 0, out, r, v, err, depth)) return false;
      out.fields.emplace_back(f.name, std::move(v));
    }
    return true;
  }

  class Lexer {
  public:
    explicit Lexer(const std::string &s) : s_(s) {}
    std::string next()
    {
      for (;;) {  // white space and comments
        while (i_ < s_.size() && isspace((unsigned char) s_[i_])) i_++;
        if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '/') {
          while (i_ < s_.size() && s_[i_] != '\n') i_++;
        } else if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '*') {
          i_ += 2;
          while (i_ + 1 < s_.size() && !(s_[i_] == '*' && s_[i_ + 1] == '/')) i_++;
          i_ = (i_ + 2 <= s_.size()) ? i_ + 2 : s_.size();
        } else {
          break;
        }
      }
      if (i_ >= s_.size()) return "";
      const char c = s_[i_];
      if (isalnum((unsigned char) c) || c == '_' || c == '.' || c == '-' || c == '+') {
        const size_t b = i_;
        while (i_ < s_.size() && (isalnum((unsigned char) s_[i_]) || s_[i_] == '_' || s_[i_] == '.' || s_[i_] == '-' || s_[i_] == '+')) i_++;
        return s_.substr(b, i_ - b);
      }
      i_++;
      return std::string(1, c);
    }
  private:
    const std::string &s_;
    size_t i_ = 0;
  };
};

}  // namespace pronto_wire
