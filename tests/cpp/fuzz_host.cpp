// fuzz_host.cpp -- the host-side parsers under AddressSanitizer + UndefinedBehaviorSanitizer, CPU only (tests/test_host_sanitizers.py).
// Everything that reads bytes a user hands over -- recorded LCM event logs, .lcm schema text, encoded messages, URDF text -- is run
// over deterministic mutations of valid inputs: truncations, bit flips, byte splices and length-field edits.  Nothing may read or
// write out of bounds, overflow a signed integer, or loop forever; results are not checked (the parity tests do that).
//   * pronto_wire::LogReader::next, MappedLog::next / seek                     (pronto_wire.hpp, segment_stream.hpp)
//   * pronto_wire::Schema::parse / decode, Schema::Plan::run and ::layout with a WARM Shape (lcm_schema.hpp)
//   * pronto_wire::{filter_state_t, indexed_measurement_t, update_t}::decode   (pronto_wire.hpp)
//   * SegmentBatcher::run and SegmentStreamer::run on damaged log files        (segment_batcher.hpp, segment_stream.hpp)
//   * ModelClient::fromURDFString / chainTo                                     (mav_state_est_batch.hpp)
// The C ABI is replaced by HOST stubs generated from include/pronto_batch.h by the test (device blocks are plain heap blocks, so the
// sanitizer also sees every byte the replayers write into their chunks and every size they pass to an upload): this program is test
// infrastructure and never runs on a GPU.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../pronto_amd/csrc/segment_batcher.hpp"
#include "../../pronto_amd/csrc/segment_stream.hpp"

using namespace MavStateEst;

static uint64_t rng_state = 0x46555A5AULL;
static uint64_t rnd()
{
  rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return rng_state >> 17;
}

// one deterministic mutation of `v` (never empties it completely unless it was a truncation to 0)
static void mutate(std::vector<uint8_t> &v)
{
  if (v.empty()) return;
  const int kind = (int) (rnd() % 7);
  if (kind == 0) {   // truncation
    v.resize((size_t) (rnd() % (v.size() + 1)));
  } else if (kind == 1) {   // bit flips
    const int n = 1 + (int) (rnd() % 8);
    for (int i = 0; i < n; i++) v[(size_t) (rnd() % v.size())] ^= (uint8_t) (1u << (rnd() % 8));
  } else if (kind == 2) {   // a 32-bit big-endian field gets an extreme value (length fields live on 4-byte boundaries mostly)
    if (v.size() >= 4) {
      const size_t at = (size_t) (rnd() % (v.size() - 3));
      static const uint32_t vals[] = { 0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, 1000u, 1001u, 65536u };
      const uint32_t x = vals[rnd() % (sizeof vals / sizeof vals[0])];
      for (int i = 0; i < 4; i++) v[at + (size_t) i] = (uint8_t) (x >> (24 - 8 * i));
    }
  } else if (kind == 3) {   // a byte range is overwritten with another range of the same buffer
    const size_t n = 1 + (size_t) (rnd() % 64);
    if (v.size() > n) {
      const size_t a = (size_t) (rnd() % (v.size() - n)), b = (size_t) (rnd() % (v.size() - n));
      memmove(v.data() + a, v.data() + b, n);
    }
  } else if (kind == 4) {   // bytes removed from the middle
    const size_t n = 1 + (size_t) (rnd() % 32);
    if (v.size() > n) {
      const size_t a = (size_t) (rnd() % (v.size() - n));
      v.erase(v.begin() + (long) a, v.begin() + (long) (a + n));
    }
  } else if (kind == 5) {   // random bytes inserted
    const size_t n = 1 + (size_t) (rnd() % 32), a = (size_t) (rnd() % (v.size() + 1));
    std::vector<uint8_t> ins(n);
    for (auto &b : ins) b = (uint8_t) rnd();
    v.insert(v.begin() + (long) a, ins.begin(), ins.end());
  } else {   // a small signed / 16-bit field
    const size_t at = (size_t) (rnd() % v.size());
    v[at] = (uint8_t) (rnd() % 2 ? 0xff : 0x80);
  }
}

static const char *BOT_CORE_LCM = R"(package bot_core;
struct ins_t { int64_t utime; int64_t device_time; double gyro[3]; double mag[3]; double accel[3]; double quat[4]; double pressure; double rel_alt; }
struct joint_state_t { int64_t utime; int16_t num_joints; string joint_name[num_joints]; float joint_position[num_joints];
  float joint_velocity[num_joints]; float joint_effort[num_joints]; }
struct six_axis_force_torque_t { int64_t utime; double force[3]; double moment[3]; }
struct six_axis_force_torque_array_t { int64_t utime; int32_t num_sensors; string names[num_sensors]; six_axis_force_torque_t sensors[num_sensors]; }
struct pose_t { int64_t utime; double pos[3]; double vel[3]; double orientation[4]; double rotation_rate[3]; double accel[3]; }
struct kvh_raw_imu_t { int64_t utime; int64_t packet_count; double delta_rotation[3]; double linear_acceleration[3]; }
struct kvh_raw_imu_batch_t { int64_t utime; int32_t num_packets; kvh_raw_imu_t raw_imu[num_packets]; }
)";

static const char *URDF = R"(<?xml version="1.0"?>
<robot name="biped">
  <link name="pelvis"/><link name="l_uglut"/>
  <joint name="l_leg_hpz" type="revolute"><origin xyz="0 0.089 0" rpy="0 0 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="l_uglut"/></joint>
  <joint name="l_leg_hpx" type="revolute"><origin xyz="0 0 0"/><axis xyz="1 0 0"/><parent link="l_uglut"/><child link="l_lglut"/></joint>
  <joint name="l_leg_kny" type="continuous"><origin rpy="0 0.02 0" xyz="-0.05 0 -0.374"/><parent link="l_lglut"/><child link="l_foot"/></joint>
  <joint name="r_leg_hpz" type="revolute"><origin xyz="0 -0.089 0"/><axis xyz="0 0 1"/><parent link="pelvis"/><child link="r_uglut"/></joint>
  <joint name="r_leg_kny" type="fixed"><origin xyz="-0.05 0 -0.374"/><parent link="r_uglut"/><child link="r_foot"/></joint>
</robot>)";

// a small valid log: T ticks of KVH batch / ins_t, force-torque, joint state, now and then a pose and a pronto::update_t
static std::vector<uint8_t> make_log(const pronto_wire::Schema &schema, const std::string &path, int T, int64_t base)
{
  const std::vector<std::string> names = { "back_bkz", "l_leg_hpz", "l_leg_hpx", "l_leg_kny", "r_leg_hpz", "r_leg_kny" };
  {
    pronto_wire::LogWriter log(path);
    for (int k = 0; k < T; k++) {
      const int64_t t = base + (int64_t) (k + 1) * 2000;
      pronto_wire::Writer w;
      w.u64(schema.fingerprint("bot_core.ins_t"));
      w.i64(t); w.i64(t + 17);
      for (int i = 0; i < 15; i++) w.f64(0.01 * i + k);
      log.write(t, "IMU", w.buf);
      pronto_wire::Writer q;
      q.u64(schema.fingerprint("bot_core.kvh_raw_imu_batch_t"));
      const int np = 1 + k % 4;
      q.i64(t); q.i32(np);
      for (int j = 0; j < np; j++) { q.i64(t - 1000 * j); q.i64(100 + 2 * k - j); for (int i = 0; i < 6; i++) q.f64(0.1 * i); }
      log.write(t + 10, "ATLAS_IMU_BATCH", q.buf);
      pronto_wire::Writer f;
      f.u64(schema.fingerprint("bot_core.six_axis_force_torque_array_t"));
      f.i64(t + 100); f.i32(2); f.str("l_foot"); f.str("r_foot");
      for (int s = 0; s < 2; s++) { f.i64(t + 100); for (int i = 0; i < 6; i++) f.f64(100.0 * i); }
      log.write(t + 100, "FORCE_TORQUE", f.buf);
      pronto_wire::Writer j;
      j.u64(schema.fingerprint("bot_core.joint_state_t"));
      j.i64(t + 300); j.i16((int16_t) names.size());
      for (const auto &nm : names) j.str(nm);
      for (int r = 0; r < 3; r++)
        for (size_t i = 0; i < names.size(); i++) j.f32(0.1f * (float) i);
      log.write(t + 300, "JOINT_STATE", j.buf);
      if (k % 5 == 4) {
        pronto_wire::Writer p;
        p.u64(schema.fingerprint("bot_core.pose_t"));
        p.i64(t + 500);
        for (int i = 0; i < 16; i++) p.f64(i == 6 ? 1.0 : 0.0);
        log.write(t + 500, "POSE_SCAN", p.buf);
        pronto_wire::update_t u;
        u.timestamp = t + 600; u.prev_timestamp = t - 9400;
        u.rotation[0] = 1;
        u.estimate_status = pronto_wire::update_t::ESTIMATE_VALID;
        std::vector<uint8_t> ub;
        u.encode(ub);
        log.write(t + 600, "VO_UPDATE", ub);
      }
      if (k % 7 == 3) log.write(t + 50, "SOMETHING_ELSE", std::vector<uint8_t>(13, 0x5A));
    }
  }
  std::vector<uint8_t> bytes;
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) { printf("cannot read back %s\n", path.c_str()); exit(2); }
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) bytes.insert(bytes.end(), buf, buf + n);
  fclose(fp);
  return bytes;
}
static void write_file(const std::string &path, const std::vector<uint8_t> &v)
{
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) { printf("cannot write %s\n", path.c_str()); exit(2); }
  if (!v.empty()) fwrite(v.data(), 1, v.size(), fp);
  fclose(fp);
}

static uint64_t touched = 0;   // (a checksum of what the callbacks read, so that nothing is optimised away)
template <class T>
static void touch(const T *p, size_t rows, size_t B)
{
  if (p == nullptr || rows == 0 || B == 0) return;
  touched += (uint64_t) (long long) p[0] + (uint64_t) (long long) p[rows * B - 1];
}

int main(int argc, char **argv)
{
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const int iters = argc > 2 ? atoi(argv[2]) : 300;
  pronto_wire::Schema schema;
  std::string err;
  if (!schema.parse(BOT_CORE_LCM, &err)) { printf("schema: %s\n", err.c_str()); return 2; }
  const std::string good_path = dir + "/fuzz_good.lcmlog", bad_path = dir + "/fuzz_bad.lcmlog", bad2_path = dir + "/fuzz_bad2.lcmlog";
  const std::vector<uint8_t> good = make_log(schema, good_path, 40, 1000000000LL);

  // ---- (1) the two log readers on damaged containers ----
  long long events = 0;
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> v = good;
    const int nm = 1 + (int) (rnd() % 3);
    for (int m = 0; m < nm; m++) mutate(v);
    write_file(bad_path, v);
    {
      pronto_wire::LogReader rd(bad_path);
      pronto_wire::LogEvent ev;
      int guard = 0;
      while (rd.good() && rd.next(ev) && guard++ < 100000) events += (long long) ev.data.size() + (long long) ev.channel.size();
    }
    {
      auto ml = MappedLog::open(bad_path);
      if (ml) {
        size_t pos = 0;
        MappedLog::Event ev;
        int guard = 0;
        while (ml->next(pos, ev) && guard++ < 100000) {
          events += ev.dlen + ev.clen;
          if (ev.dlen) touched += ev.data[0] + ev.data[ev.dlen - 1];
          if (ev.clen) touched += ev.channel[0] + ev.channel[ev.clen - 1];
        }
        for (int q = 0; q < 4; q++) touched += (uint64_t) ml->seek(1000000000LL + (int64_t) (rnd() % 200000) - 50000);
      }
    }
  }
  printf("log containers: %d damaged logs read twice, %lld event bytes\n", iters, events);

  // ---- (2) schema text, value-tree decoder, plans with a warm shape, the fixed pronto types ----
  {
    long long ok_parse = 0, ok_decode = 0, ok_plan = 0;
    const std::string text = BOT_CORE_LCM;
    for (int it = 0; it < iters; it++) {
      std::vector<uint8_t> t(text.begin(), text.end());
      mutate(t);
      pronto_wire::Schema s2;
      std::string e2;
      if (s2.parse(std::string(t.begin(), t.end()), &e2)) {
        ok_parse++;
        for (const char *ty : { "bot_core.ins_t", "bot_core.joint_state_t", "bot_core.kvh_raw_imu_batch_t" }) touched += (uint64_t) s2.fingerprint(ty);
      }
    }
    // messages of every type out of the good log, mutated; the plans are warmed with the valid message first
    struct Want { const char *channel, *type; std::vector<std::string> members; };
    const std::vector<Want> wants = {
      { "IMU", "bot_core.ins_t", { "utime", "gyro", "accel" } },
      { "JOINT_STATE", "bot_core.joint_state_t", { "utime", "joint_name", "joint_position", "joint_velocity", "joint_effort" } },
      { "FORCE_TORQUE", "bot_core.six_axis_force_torque_array_t", { "utime", "sensors.force" } },
      { "ATLAS_IMU_BATCH", "bot_core.kvh_raw_imu_batch_t", { "utime", "raw_imu.utime", "raw_imu.packet_count", "raw_imu.delta_rotation", "raw_imu.linear_acceleration" } },
      { "POSE_SCAN", "bot_core.pose_t", { "utime", "pos", "vel", "orientation" } },
    };
    pronto_wire::LogReader rd(good_path);
    pronto_wire::LogEvent ev;
    std::map<std::string, std::vector<uint8_t>> sample;
    while (rd.next(ev))
      if (!sample.count(ev.channel)) sample[ev.channel] = ev.data;
    for (const Want &w : wants) {
      const std::vector<uint8_t> &msg = sample[w.channel];
      pronto_wire::Schema::Plan plan = schema.compile(w.type, w.members);
      if (!plan.ok() || msg.empty()) { printf("FAIL: no plan / sample for %s\n", w.type); return 1; }
      for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> m = msg;
        mutate(m);
        pronto_wire::Value v;
        if (schema.decode(w.type, m.data(), m.size(), v)) ok_decode++;
        std::vector<pronto_wire::Schema::Extracted> x;
        pronto_wire::Schema::Plan::Shape sh;
        bool same = false;
        if (!plan.run(msg.data(), msg.size(), x, sh, &same)) { printf("FAIL: the valid %s does not decode\n", w.type); return 1; }
        if (plan.run(m.data(), m.size(), x, sh, &same)) ok_plan++;            // warm shape, damaged message
        pronto_wire::Schema::Plan::Shape sh2;
        bool rebuilt = false;
        plan.layout(msg.data(), msg.size(), sh2, x, &rebuilt);
        if (plan.layout(m.data(), m.size(), sh2, x, &rebuilt)) {               // the streaming decoders' path
          double d[64]; float f[64]; int64_t i64[64];
          for (int slot = 0; slot < (int) w.members.size(); slot++) {
            touched += (uint64_t) pronto_wire::Schema::Plan::gather_f64(sh2, m.data(), slot, d, 64);
            touched += (uint64_t) pronto_wire::Schema::Plan::gather_f32(sh2, m.data(), slot, f, 64);
            touched += (uint64_t) pronto_wire::Schema::Plan::gather_i64(sh2, m.data(), slot, i64, 64);
          }
        }
        if (plan.run(m.data(), m.size(), x)) ok_plan++;
      }
    }
    // the compiled-in pronto types
    pronto_wire::update_t u;
    u.rotation[0] = 1;
    std::vector<uint8_t> ub;
    u.encode(ub);
    pronto_wire::indexed_measurement_t im;
    im.measured_dim = 3; im.z_effective = { 1, 2, 3 }; im.z_indices = { 3, 4, 5 }; im.measured_cov_dim = 9; im.R_effective.assign(9, 0.5);
    std::vector<uint8_t> ib;
    im.encode(ib);
    for (int it = 0; it < iters; it++) {
      std::vector<uint8_t> a = ub, b = ib;
      mutate(a);
      mutate(b);
      pronto_wire::update_t u2;
      pronto_wire::indexed_measurement_t i2;
      pronto_wire::filter_state_t f2;
      touched += (uint64_t) (u2.decode(a.data(), a.size()) + i2.decode(b.data(), b.size()) + f2.decode(b.data(), b.size()));
    }
    printf("schema / messages: %lld mutated schema texts parsed, %lld mutated messages decoded into trees, %lld through plans\n", ok_parse, ok_decode, ok_plan);
  }

  // ---- (3) URDF text ----
  {
    long long ok = 0;
    const std::string text = URDF;
    for (int it = 0; it < iters; it++) {
      std::vector<uint8_t> t(text.begin(), text.end());
      const int nm = 1 + (int) (rnd() % 3);
      for (int m = 0; m < nm; m++) mutate(t);
      const std::string s(t.begin(), t.end());
      ModelClient model;
      if (model.fromURDFString(s, "l_foot", "r_foot")) ok++;
      std::vector<ModelClient::Joint> chain;
      if (ModelClient::chainTo(s, "l_foot", chain)) touched += (uint64_t) chain.size();
    }
    printf("URDF: %lld of %d mutated texts still gave two chains\n", ok, iters);
  }

  // ---- (4) the two replayers on damaged logs (three segments: a good log, two damaged ones), with callbacks that read what they get ----
  {
    const int B = 5;
    BotParam param;
    param.set("state_estimator.utime_history_span", "1000000");
    param.set("state_estimator.history_slots", "0");
    RBIS x0(15, B);
    RBIM P0(15, B);
    long long batches = 0;
    for (int it = 0; it < iters / 4 + 1; it++) {
      std::vector<uint8_t> v = good, v2 = good;
      const int nm = 1 + (int) (rnd() % 3);
      for (int m = 0; m < nm; m++) { mutate(v); mutate(v2); }
      write_file(bad_path, v);
      write_file(bad2_path, v2);
      MavStateEstimator est(new RBISResetUpdate(x0, P0, RBISUpdateInterface::reset, 0), &param, 0);
      {
        SegmentBatcher batch(&est);
        batch.addSegment(bad_path);
        batch.addSegment(good_path, 1000000000LL + 20000);
        batch.addSegment(bad2_path);
        if (it % 2)
          batch.subscribeIns("IMU", &schema, "bot_core.ins_t", [&](const msgs::ins_t *m) { touch(m->gyro.p, 3, B); touch(m->accel.p, 3, B); touch(m->valid, 1, B); });
        else
          batch.subscribeKvhBatch("ATLAS_IMU_BATCH", &schema, "bot_core.kvh_raw_imu_batch_t", it % 4 == 0, 3, [&](const msgs::kvh_raw_imu_segments_t *m) {
            touch(m->new_accel, 3 * (size_t) m->max_new, B); touch(m->delta_rotation, 3, B); touch(m->raw_dt, 1, B); touch(m->utimes, 1, B);
            touch(m->n_new, 1, B); touch(m->valid, 1, B);
          });
        batch.subscribeForceTorque("FORCE_TORQUE", &schema, "bot_core.six_axis_force_torque_array_t",
                                   [&](const msgs::six_axis_force_torque_array_t *m) { touch(m->force_z.p, 2, B); });
        batch.subscribeJointState("JOINT_STATE", &schema, "bot_core.joint_state_t", [&](const msgs::joint_state_t *m) {
          touched += (uint64_t) m->joint_name.size();
          touch(m->utimes, 1, B);
          touch(m->valid, 1, B);
          (void) m->joint_position;   // (device memory in the real replay: the stub's heap block, checked where it was written)
        });
        batch.subscribePose("POSE_SCAN", &schema, "bot_core.pose_t", [&](const msgs::pose_t *m) { touch(m->pos.p, 3, B); touch(m->orientation.p, 4, B); });
        batch.subscribeUpdate("VO_UPDATE", [&](const msgs::update_t *m) { touch(m->translation.p, 3, B); touch(m->rotation.p, 4, B); touch(m->estimate_valid, 1, B); });
        const int64_t nb = batch.run();
        if (nb > 0) batches += nb;
      }
      {
        SegmentStreamer stream(&est);
        stream.max_slots = 1 + (int) (rnd() % 9);
        stream.readahead_cap = 3 + (size_t) (rnd() % 20);
        stream.addSegment(bad_path);
        stream.addSegment(good_path, 1000000000LL + 20000, it % 2 ? 1000000000LL + 60000 : 0);
        stream.addSegment(bad2_path, (int64_t) (rnd() % 2) * (1000000000LL + 30000));
        const size_t Bz = (size_t) B;
        if (it % 2)
          stream.subscribeIns("IMU", &schema, "bot_core.ins_t", [&](const msgs::ins_t *m) { touch(m->gyro.p, 3, Bz); touch(m->accel.p, 3, Bz); touch(m->valid, 1, Bz); });
        else
          stream.subscribeKvhBatch("ATLAS_IMU_BATCH", &schema, "bot_core.kvh_raw_imu_batch_t", it % 4 == 0, 3, [&](const msgs::kvh_raw_imu_segments_t *m) {
            touch(m->new_accel, 3 * (size_t) m->max_new, Bz); touch(m->delta_rotation, 3, Bz); touch(m->raw_dt, 1, Bz); touch(m->utimes, 1, Bz);
            touch(m->n_new, 1, Bz); touch(m->valid, 1, Bz);
          });
        stream.subscribeForceTorque("FORCE_TORQUE", &schema, "bot_core.six_axis_force_torque_array_t", [&](const float *p) { touch(p, 2, Bz); });
        stream.subscribeJointState("JOINT_STATE", &schema, "bot_core.joint_state_t", [&](const msgs::joint_state_t *m) {
          const size_t n = m->joint_name.size();
          touch(m->joint_position, n, Bz); touch(m->joint_velocity, n, Bz); touch(m->joint_effort, n, Bz); touch(m->utimes, 1, Bz); touch(m->valid, 1, Bz);
        });
        stream.subscribePose("POSE_SCAN", &schema, "bot_core.pose_t", [&](const msgs::pose_t *m) { touch(m->pos.p, 3, Bz); touch(m->orientation.p, 4, Bz); touch(m->valid, 1, Bz); });
        stream.subscribeUpdate("VO_UPDATE", [&](const msgs::update_t *m) { touch(m->translation.p, 3, Bz); touch(m->rotation.p, 4, Bz); touch(m->estimate_valid, 1, Bz); });
        const int64_t nb = stream.run();
        if (nb > 0) batches += nb;
      }
    }
    printf("replayers: %d rounds of three segments (two damaged) through SegmentBatcher and SegmentStreamer, %lld batched messages\n", iters / 4 + 1, batches);
  }
  remove(good_path.c_str());
  remove(bad_path.c_str());
  remove(bad2_path.c_str());
  printf("checksum %llu\nPASS\n", (unsigned long long) touched);
  return 0;
}
