// wire_tool.cpp -- CPU-only test driver of pronto_amd/csrc/pronto_wire.hpp (tests/test_wire.py builds and runs it and
// cross-checks every byte against the independent Python restatement in tests/lcm_ref.py).
//   wire_tool hashes          print the exlcm::example_t base hash (LCM tutorial KAT) and the three fingerprints
//   wire_tool write <path>    write a log of deterministic messages of the three types plus one foreign event
//   wire_tool dump <path>     decode every event of a log, one text line per event (%.17g, so doubles round-trip)
#include <cinttypes>
#include <cstdio>
#include <cstdlib>

#include "../../pronto_amd/csrc/lcm_schema.hpp"

using namespace pronto_wire;

static double val(int i) { return (i % 2 ? -1.0 : 1.0) * (0.1 + i) / 7.0; }

int main(int argc, char **argv)
{
  const std::string mode = argc > 1 ? argv[1] : "";
  if (mode == "hashes") {
    const std::vector<Member> example = { { "timestamp", "int64_t", {} },
                                          { "position", "double", { { LCM_CONST, "3" } } },
                                          { "orientation", "double", { { LCM_CONST, "4" } } },
                                          { "num_ranges", "int32_t", {} },
                                          { "ranges", "int16_t", { { LCM_VAR, "num_ranges" } } },
                                          { "name", "string", {} },
                                          { "enabled", "boolean", {} } };
    printf("example_t_base %016" PRIx64 "\n", (uint64_t) lcm_base_hash(example));
    printf("filter_state_t %016" PRIx64 "\n", filter_state_t::fingerprint());
    printf("indexed_measurement_t %016" PRIx64 "\n", indexed_measurement_t::fingerprint());
    printf("update_t %016" PRIx64 "\n", update_t::fingerprint());
    return 0;
  }
  if (mode == "write" && argc > 2) {
    LogWriter log(argv[2]);
    if (!log.good()) return 2;
    std::vector<uint8_t> buf;
    for (int k = 0; k < 5; k++) {
      filter_state_t fs;
      fs.utime = 1000000 + 1000 * k;
      for (int i = 0; i < 4; i++) fs.quat[i] = val(k + i);
      fs.num_states = 21;
      fs.num_cov_elements = 441;
      for (int i = 0; i < 21; i++) fs.state.push_back(val(3 * k + i));
      for (int i = 0; i < 441; i++) fs.cov.push_back(val(k + 2 * i));
      fs.encode(buf);
      log.write(fs.utime, "STATE_ESTIMATOR_STATE", buf);

      indexed_measurement_t im;
      im.utime = fs.utime + 1;
      im.state_utime = fs.utime - 7;
      im.measured_dim = 1 + k;
      for (int i = 0; i < im.measured_dim; i++) {
        im.z_effective.push_back(val(10 + i + k));
        im.z_indices.push_back((3 + 2 * i + k) % 21);
      }
      im.measured_cov_dim = im.measured_dim * im.measured_dim;
      for (int i = 0; i < im.measured_cov_dim; i++) im.R_effective.push_back(val(20 + i));
      im.encode(buf);
      log.write(im.utime, "GPF_MEASUREMENT", buf);

      if (k == 2) log.write(im.utime + 1, "SOMETHING_ELSE", std::vector<uint8_t>{ 1, 2, 3 });  // not one of the types

      update_t up;
      up.timestamp = fs.utime + 2;
      up.prev_timestamp = fs.utime - 33333;
      for (int i = 0; i < 3; i++) up.translation[i] = val(30 + i + k);
      for (int i = 0; i < 4; i++) up.rotation[i] = val(40 + i + k);
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) up.covariance[i][j] = val(50 + 6 * i + j + k);
      up.estimate_status = (int8_t) (k % 5);
      up.encode(buf);
      log.write(up.timestamp, "KINECT_REL_ODOMETRY", buf);
    }
    return 0;
  }
  if (mode == "dump" && argc > 2) {
    LogReader rd(argv[2]);
    if (!rd.good()) return 2;
    LogEvent ev;
    while (rd.next(ev)) {
      printf("event %" PRId64 " %" PRId64 " %s %zu", ev.eventnum, ev.timestamp, ev.channel.c_str(), ev.data.size());
      filter_state_t fs;
      indexed_measurement_t im;
      update_t up;
      if (fs.decode(ev.data.data(), ev.data.size()) == (int) ev.data.size()) {
        printf(" filter_state_t %" PRId64, fs.utime);
        for (double v : fs.quat) printf(" %.17g", v);
        printf(" %d", fs.num_states);
        for (double v : fs.state) printf(" %.17g", v);
        printf(" %d", fs.num_cov_elements);
        for (double v : fs.cov) printf(" %.17g", v);
      } else if (im.decode(ev.data.data(), ev.data.size()) == (int) ev.data.size()) {
        printf(" indexed_measurement_t %" PRId64 " %" PRId64 " %d", im.utime, im.state_utime, im.measured_dim);
        for (double v : im.z_effective) printf(" %.17g", v);
        for (int v : im.z_indices) printf(" %d", v);
        printf(" %d", im.measured_cov_dim);
        for (double v : im.R_effective) printf(" %.17g", v);
      } else if (up.decode(ev.data.data(), ev.data.size()) == (int) ev.data.size()) {
        printf(" update_t %" PRId64 " %" PRId64, up.timestamp, up.prev_timestamp);
        for (double v : up.translation) printf(" %.17g", v);
        for (double v : up.rotation) printf(" %.17g", v);
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) printf(" %.17g", up.covariance[i][j]);
        printf(" %d", (int) up.estimate_status);
      } else {
        printf(" unknown");
      }
      printf("\n");
    }
    return 0;
  }
  if (mode == "schema" && argc > 4) {  // schema <lcm files, comma separated> <type> <message file>
    Schema sc;
    std::string files = argv[2], err;
    size_t a = 0;
    while (a <= files.size()) {
      const size_t b = files.find(',', a);
      const std::string path = files.substr(a, b == std::string::npos ? std::string::npos : b - a);
      FILE *f = fopen(path.c_str(), "rb");
      if (!f) return 2;
      std::string text;
      char buf[4096];
      size_t n;
      while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
      fclose(f);
      if (!sc.parse(text, &err)) { printf("parse error: %s\n", err.c_str()); return 3; }
      if (b == std::string::npos) break;
      a = b + 1;
    }
    printf("fingerprint %016" PRIx64 "\n", sc.fingerprint(argv[3]));
    FILE *f = fopen(argv[4], "rb");
    if (!f) return 2;
    std::vector<uint8_t> msg;
    int c;
    while ((c = fgetc(f)) != EOF) msg.push_back((uint8_t) c);
    fclose(f);
    Value v;
    if (!sc.decode(argv[3], msg.data(), msg.size(), v, &err)) { printf("decode error: %s\n", err.c_str()); return 0; }
    struct Printer {
      static void print(const Value &v)
      {
        switch (v.kind) {
          case Value::INT: printf("%lld", (long long) v.i); break;
          case Value::FLOAT: printf("%.17g", v.f); break;
          case Value::STRING: printf("\"%s\"", v.s.c_str()); break;
          case Value::ARRAY:
            printf("[");
            for (size_t k = 0; k < v.items.size(); k++) { if (k) printf(","); print(v.items[k]); }
            printf("]");
            break;
          case Value::STRUCT:
            printf("{");
            for (size_t k = 0; k < v.fields.size(); k++) {
              if (k) printf(",");
              printf("%s:", v.fields[k].first.c_str());
              print(v.fields[k].second);
            }
            printf("}");
            break;
        }
      }
    };
    Printer::print(v);
    printf("\n");
    return 0;
  }
  if (mode == "plan" && argc > 5) {  // plan <lcm file> <type> <message file> <wanted,members,...>: Schema::compile + Plan::run
    Schema sc;
    std::string err, text;
    {
      FILE *f = fopen(argv[2], "rb");
      if (!f) return 2;
      char buf[4096];
      size_t n;
      while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
      fclose(f);
    }
    if (!sc.parse(text, &err)) { printf("parse error: %s\n", err.c_str()); return 3; }
    std::vector<std::string> wanted;
    {
      std::string w = argv[5];
      size_t a = 0;
      while (a <= w.size()) {
        const size_t b = w.find(',', a);
        wanted.push_back(w.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
      }
    }
    const Schema::Plan plan = sc.compile(argv[3], wanted);
    if (!plan.ok()) { printf("no plan\n"); return 0; }
    FILE *f = fopen(argv[4], "rb");
    if (!f) return 2;
    std::vector<uint8_t> msg;
    int c;
    while ((c = fgetc(f)) != EOF) msg.push_back((uint8_t) c);
    fclose(f);
    std::vector<Schema::Extracted> out;
    auto print = [&]() {
      for (size_t k = 0; k < out.size(); k++) {
        printf("%s:", wanted[k].c_str());
        for (double v : out[k].num) printf(" %.17g", v);
        for (const std::string &t : out[k].str) printf(" \"%s\"", t.c_str());
        printf("\n");
      }
    };
    if (argc > 6) {  // ... <more message files>: one stream through ONE Plan::Shape -- which messages had the cached layout
      Schema::Plan::Shape shape;
      for (int a = 4; a < argc; a++) {
        if (a == 5) continue;
        FILE *g = fopen(argv[a], "rb");
        if (!g) return 2;
        std::vector<uint8_t> m2;
        while ((c = fgetc(g)) != EOF) m2.push_back((uint8_t) c);
        fclose(g);
        bool same = false;
        if (!plan.run(m2.data(), m2.size(), out, shape, &same)) { printf("run refused\n"); continue; }
        printf("shape: %s\n", same ? "same" : "new");
        print();
      }
      return 0;
    }
    if (!plan.run(msg.data(), msg.size(), out)) { printf("run refused\n"); return 0; }
    print();
    return 0;
  }
  fprintf(stderr, "usage: wire_tool hashes | write <path> | dump <path> | schema <lcm files> <type> <message file> | plan <lcm file> <type> <message file> <members>\n");
  return 1;
}
