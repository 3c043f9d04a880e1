// segment_stream.hpp -- N independent recorded logs replayed as ONE batch at the speed of the device: SegmentBatcher's job
// (segment_batcher.hpp: segment s feeds filter s, the reference's se-batch-process.sh workload -- one se-fusion run per recorded
// log, motion_estimate/scripts/se-batch-process.sh:17-26,58-74, each opened at a start_timestamp, lcm_front_end.cpp:21-33) as a
// PIPELINE instead of a per-message loop:
//
//   decode-ahead threads    every segment's log is memory-mapped and decoded in place (run-time .lcm schema, lcm_schema.hpp: a stream
//                           repeats its byte layout, so the wanted numbers are read by offset); a team of host threads decodes CHUNKS
//                           of up to `max_slots` batched messages ahead of the device -- groups of 16 segments into a thread-local
//                           block, then whole cache lines into
//   a page-locked ring      of chunk buffers: per batched message ("slot") one block [rows][B] per member, filter index fastest --
//                           the layout the kernels read;
//   one upload per chunk    on the context's copy stream (pb_upload_async), overlapped with the kernels of the previous chunk;
//   the handlers            are then called slot by slot with PB_DEVICE messages -- InsHandler (frame rotation, KVH notch cascade and
//                           time steps on the device: pb_ins_body_block / pb_imu_notch_counts), LegOdoHandler::forceTorqueDevice /
//                           processMessage (per-filter message times read in place) -- so a tick costs the host a few enqueues and
//                           no O(B) pass, no staging copy and no synchronisation.  Sparse channels (scan-match poses, VO updates)
//                           are handed over as PB_HOST blocks that point into the page-locked chunk.
//
// Semantics are SegmentBatcher's (the tests run both against the same single-segment oracle runs): messages are aligned BY INDEX
// per channel and dispatched in the file order of the lead segment (the lowest-numbered segment that still has events); a batched
// message carries the lead's time on the batch's time base, every filter's own stamp travels beside it; a segment that has run out
// idles (valid = 0) and its RESULT is its filter's head at the end of ITS log (finalState()).  What is counted instead of assumed:
// order_violations, max_skew_us, ragged columns, undecodable events, read-ahead that hit its bound.
//
// New here: the reference's own IMU channel for Atlas logs -- bot_core::kvh_raw_imu_batch_t on ATLAS_IMU_BATCH
// (motion_estimate/src/fusion/fusion.cpp:161-163 -> InsHandler::processMessageAtlas, sensor_handlers.cpp:165-252): one IMUStream
// de-duplication state per SEGMENT (imu_stream.cpp:62-98) where the message is decoded, the notch cascade and the per-filter time
// steps on the device (subscribeKvhBatch -> InsHandler::processMessageAtlasSegments).
//
// Host side only.  Header-only, POSIX (mmap).
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "mav_state_est_batch.hpp"

namespace MavStateEst {

// A recorded LCM event log, mapped read-only and walked in place (lcm/eventlog.c's container, pronto_wire.hpp).  Segments that are
// start offsets into the same file share one mapping.
class MappedLog {
public:
  struct Event {
    int64_t timestamp = 0;
    const char *channel = nullptr;
    uint32_t clen = 0, dlen = 0;
    const uint8_t *data = nullptr;
    size_t at = 0;   // offset of the event's sync word
  };
  static std::shared_ptr<MappedLog> open(const std::string &path)
  {
    static std::mutex mu;
    static std::map<std::string, std::weak_ptr<MappedLog>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(path);
    if (it != cache.end())
      if (auto sp = it->second.lock()) return sp;
    std::shared_ptr<MappedLog> m(new MappedLog());
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); return nullptr; }
    m->size_ = (size_t) st.st_size;
    if (m->size_ > 0) {
      void *p = mmap(nullptr, m->size_, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) { ::close(fd); return nullptr; }
      m->data_ = (const uint8_t *) p;
      madvise(p, m->size_, MADV_SEQUENTIAL);
    }
    ::close(fd);
    cache[path] = m;
    return m;
  }
  ~MappedLog() { if (data_) munmap((void *) data_, size_); }
  MappedLog(const MappedLog &) = delete;
  MappedLog &operator=(const MappedLog &) = delete;
  size_t size() const { return size_; }
  // The event at or after `pos` (a reader that has lost sync scans byte-wise for the next sync word, like lcm_eventlog_read_next_event
  // and LogReader); pos then is the offset behind it.  false: end of the log, a truncated tail, or a damaged length field.
  bool next(size_t &pos, Event &ev) const
  {
    while (pos + 28 <= size_ && be32(data_ + pos) != pronto_wire::LOG_SYNC) pos++;
    if (pos + 28 > size_) return false;
    const int32_t clen = (int32_t) be32(data_ + pos + 20), dlen = (int32_t) be32(data_ + pos + 24);
    if (clen < 0 || clen > 1000 || dlen < 0) return false;   // eventlog.c rejects channel names > 1000
    if ((size_t) clen + (size_t) dlen > size_ - pos - 28) return false;
    ev.at = pos;
    ev.timestamp = (int64_t) be64(data_ + pos + 12);
    ev.channel = (const char *) data_ + pos + 28;
    ev.clen = (uint32_t) clen;
    ev.data = data_ + pos + 28 + clen;
    ev.dlen = (uint32_t) dlen;
    pos += 28 + (size_t) clen + (size_t) dlen;
    return true;
  }
  // An offset at or in front of the first event whose timestamp is >= ts, found by bisection on the file offset (event timestamps
  // of a recorded log do not decrease: lcm's own lcm_eventlog_seek_to_timestamp makes the same assumption); walking on from there
  // with next() and skipping what is still older gives exactly the events a linear scan from the start would.
  size_t seek(int64_t ts) const
  {
    size_t lo = 0, hi = size_;
    while (hi - lo > 65536) {
      const size_t mid = lo + (hi - lo) / 2;
      size_t p = mid;
      Event ev;
      bool found = false;
      while (p + 28 <= hi) {   // a sync word that starts a plausible event FOLLOWED by another sync word (or the end of the file)
        while (p + 28 <= hi && be32(data_ + p) != pronto_wire::LOG_SYNC) p++;
        if (p + 28 > hi) break;
        size_t q = p;
        if (next(q, ev) && ev.at == p && (q == size_ || (q + 4 <= size_ && be32(data_ + q) == pronto_wire::LOG_SYNC))) { found = true; break; }
        p++;
      }
      if (!found) { hi = mid; continue; }
      if (ev.timestamp < ts) lo = p;
      else hi = mid;
    }
    return lo;
  }
private:
  MappedLog() {}
  static uint32_t be32(const uint8_t *p) { uint32_t u; memcpy(&u, p, 4); return __builtin_bswap32(u); }
  static uint64_t be64(const uint8_t *p) { uint64_t u; memcpy(&u, p, 8); return __builtin_bswap64(u); }
  const uint8_t *data_ = nullptr;
  size_t size_ = 0;
};

class SegmentStreamer {
public:
  struct Stats {
    int64_t chunks = 0, batches = 0, segment_messages = 0, ragged = 0;
    int64_t order_violations = 0, undecodable = 0, max_skew_us = 0, readahead_capped = 0;
    std::map<std::string, int64_t> per_channel;
    uint64_t chunk_bytes = 0;           // size of one chunk buffer (host page-locked and device, `ring` of each)
    uint64_t uploaded_bytes = 0;        // what crossed PCIe in chunk uploads
    // wall-clock seconds: the decode-ahead thread at work (lead pass + the parallel decode / assembly of the chunks), the dispatching
    // thread waiting for a chunk, issuing uploads, inside the handlers' callbacks (enqueues), reading finished runs' heads
    double t_decode = 0, t_lead = 0, t_wait_chunk = 0, t_upload = 0, t_handlers = 0, t_final = 0, t_wait_free = 0;
  };
  Stats stats;
  int max_slots = 96;                  // batched messages per chunk, at most ...
  uint64_t chunk_budget_bytes = 48ull << 20;   // ... and as many as fit this many bytes
  int ring = 3;                        // chunk buffers in flight (one being decoded, one uploading, one being consumed)
  int group = 16;                      // segments a host thread decodes together (16 floats = one cache line of a row)
  size_t readahead_cap = 4096;         // events a segment may read past its quota looking for a channel that has stopped

  explicit SegmentStreamer(MavStateEstimator *est)
      : est_(est), B_(est->B), final_vec_((size_t) est->n * est->B, 0.0), final_quat_((size_t) 4 * est->B, 0.0),
        final_cov_((size_t) est->n * est->n * est->B, 0.0), final_ll_((size_t) est->B, 0.0), final_utime_((size_t) est->B, 0),
        finished_((size_t) est->B, 0) {}
  ~SegmentStreamer()
  {
    est_->flushPending();   // (an update the estimator is holding back may still read the device chunks)
    pb_sync(est_->ctx);
    for (Buf &b : bufs_) {
      if (b.host) pb_host_free(est_->ctx, b.host);
      if (b.dev) pb_free(est_->ctx, b.dev);
    }
  }
  SegmentStreamer(const SegmentStreamer &) = delete;
  SegmentStreamer &operator=(const SegmentStreamer &) = delete;

  // segment s = filter s, in the order added; false: the file cannot be opened or the batch is full.  start_timestamp: the log
  // provider's "?start_timestamp=" (lcm_front_end.cpp:21-33) -- found by bisection, not by reading the log from its start
  // end_timestamp > 0: the run stops in front of the first event at or after it (a WINDOW of a long recording; this build's addition)
  bool addSegment(const std::string &path, int64_t start_timestamp = 0, int64_t end_timestamp = 0)
  {
    if ((int) segs_.size() >= B_) return false;
    auto log = MappedLog::open(path);
    if (!log) return false;
    std::unique_ptr<Seg> sg(new Seg());
    sg->log = log;
    sg->start_timestamp = start_timestamp;
    sg->end_timestamp = end_timestamp;
    sg->pos = start_timestamp > 0 ? log->seek(start_timestamp) : 0;
    segs_.push_back(std::move(sg));
    return true;
  }
  int segments() const { return (int) segs_.size(); }

  // ---- typed subscriptions: channel -> what FrontEnd::addSensor returned (or a handler method) ----
  // bot_core::ins_t (utime, gyro[3], accel[3]) -> InsHandler::processMessage with DEVICE arrays (frame rotation on the device)
  void subscribeIns(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                    std::function<void(const msgs::ins_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "gyro", "accel" }));
    if (!plan->ok()) fprintf(stderr, "SegmentStreamer: %s has no utime / gyro / accel members\n", type.c_str());
    // record: f64 gyro[3] accel[3] | u8 valid
    c.planes = { { 8, 6 }, { 1, 1 } };
    c.decode = [plan](const MappedLog::Event &ev, SegChan &st, uint8_t *rec, int64_t &utime) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->layout(ev.data, ev.dlen, st.shape, x, nullptr)) return false;
      double v[6];
      if (pronto_wire::Schema::Plan::gather_i64(st.shape, ev.data, 0, &utime, 1) != 1 || pronto_wire::Schema::Plan::gather_f64(st.shape, ev.data, 1, v, 3) != 3 ||
          pronto_wire::Schema::Plan::gather_f64(st.shape, ev.data, 2, v + 3, 3) != 3)
        return false;
      memcpy(rec, v, sizeof v);
      rec[48] = 1;
      return true;
    };
    c.blank = [](uint8_t *rec) { rec[48] = 0; };   // (a filter without a message idles on its own last sample)
    c.dispatch = [this, cb](const SlotView &v) {
      msgs::ins_t m{ v.utime, BatchArray((const double *) v.dev(0), PB_DEVICE), BatchArray((const double *) v.dev(24), PB_DEVICE) };
      m.valid = (const uint8_t *) v.dev(48);
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::kvh_raw_imu_batch_t (utime, raw_imu[]{utime, packet_count, delta_rotation[3], linear_acceleration[3]}; raw_imu[0] is the
  // NEWEST packet) -> InsHandler::processMessageAtlasSegments.  One IMUStream state per segment; atlas_filter = the handler's
  // (true: de-duplicate, every new packet goes to the device notch cascade; false: newest packet, raw_dt from the two newest).
  // max_packets: the most packets a message can carry (rows of the new-packet block).
  void subscribeKvhBatch(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type, bool atlas_filter, int max_packets,
                         std::function<void(const msgs::kvh_raw_imu_segments_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(
        schema->compile(type, { "utime", "raw_imu.utime", "raw_imu.packet_count", "raw_imu.delta_rotation", "raw_imu.linear_acceleration" }));
    if (!plan->ok()) fprintf(stderr, "SegmentStreamer: %s is not a kvh_raw_imu_batch_t\n", type.c_str());
    const int MP = max_packets < 1 ? 1 : max_packets;
    // record: f64 new_accel[MP][3] | f64 delta_rotation[3] | f64 raw_dt | i64 utime | i32 n_new | u8 valid
    const size_t o_dr = (size_t) 24 * MP, o_rd = o_dr + 24, o_ut = o_rd + 8, o_nn = o_ut + 8, o_va = o_nn + 4;
    c.planes = { { 8, 3 * MP + 5 }, { 4, 1 }, { 1, 1 } };
    c.decode = [plan, MP, atlas_filter, o_dr, o_rd, o_ut, o_nn, o_va](const MappedLog::Event &ev, SegChan &st, uint8_t *rec, int64_t &utime) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      static thread_local std::vector<int64_t> put, pcnt;
      static thread_local std::vector<double> dr, la;
      if (!plan->layout(ev.data, ev.dlen, st.shape, x, nullptr)) return false;
      using P = pronto_wire::Schema::Plan;
      if (P::gather_i64(st.shape, ev.data, 0, &utime, 1) != 1) return false;
      size_t np = 0;
      for (const auto &o : st.shape.ops)
        if (o.slot == 1) np += o.count;
      if (np < 1) return false;
      put.resize(np); pcnt.resize(np); dr.resize(3 * np); la.resize(3 * np);
      if (P::gather_i64(st.shape, ev.data, 1, put.data(), np) != np || P::gather_i64(st.shape, ev.data, 2, pcnt.data(), np) != np ||
          P::gather_f64(st.shape, ev.data, 3, dr.data(), 3 * np) != 3 * np || P::gather_f64(st.shape, ev.data, 4, la.data(), 3 * np) != 3 * np)
        return false;
      double *acc = (double *) rec, *drot = (double *) (rec + o_dr), *raw_dt = (double *) (rec + o_rd);
      int32_t n_new = 0;
      if (!atlas_filter) {   // sensor_handlers.cpp:199-204: newest packet, raw_dt from the two newest
        if (np < 2) return false;
        for (int i = 0; i < 3; i++) { acc[i] = la[(size_t) i]; drot[i] = dr[(size_t) i]; }
        *raw_dt = (double) (put[0] - put[1]) * 1E-6;
        n_new = 1;
      } else {               // IMUStream::convertFromLCMBatch (imu_stream.cpp:62-98), this segment's own state
        if (pcnt[0] < st.last_packet) {   // "Detected time skip, resetting IMUStream" (:63-68)
          st.last_packet = -1;
          st.last_packet_utime = 0;
        }
        for (size_t i = np; i-- > 0;) {   // oldest first
          if (pcnt[i] <= st.last_packet) continue;
          const int64_t utime_delta = put[i] - st.last_packet_utime;
          if (n_new < MP)
            for (int a = 0; a < 3; a++) acc[(size_t) 3 * n_new + a] = la[3 * i + a];
          n_new++;
          for (int a = 0; a < 3; a++) drot[a] = dr[3 * i + a];   // (of the newest new packet in the end)
          *raw_dt = (double) utime_delta * 1E-6;
          st.last_packet = pcnt[i];
          st.last_packet_utime = put[i];
        }
        if (n_new > MP) { st.overflow++; n_new = MP; }   // (more new packets than rows: the oldest ones were filtered... not at all -- counted)
      }
      memcpy(rec + o_ut, &utime, 8);
      memcpy(rec + o_nn, &n_new, 4);
      rec[o_va] = n_new > 0;
      return true;
    };
    c.blank = [o_nn, o_va](uint8_t *rec) { const int32_t z = 0; memcpy(rec + o_nn, &z, 4); rec[o_va] = 0; };
    c.dispatch = [this, cb, MP, o_dr, o_rd, o_ut, o_nn, o_va](const SlotView &v) {
      msgs::kvh_raw_imu_segments_t m;
      m.utime = v.utime;
      m.max_new = MP;
      m.new_accel = (const double *) v.dev(0);
      m.delta_rotation = (const double *) v.dev(o_dr);
      m.raw_dt = (const double *) v.dev(o_rd);
      m.utimes = (const int64_t *) v.dev(o_ut);
      m.n_new = (const int32_t *) v.dev(o_nn);
      m.valid = (const uint8_t *) v.dev(o_va);
      m.mem = PB_DEVICE;
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::joint_state_t -> LegOdoHandler::processMessage with DEVICE blocks and per-filter message times read in place
  void subscribeJointState(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                           std::function<void(const msgs::joint_state_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(
        schema->compile(type, { "utime", "joint_name", "joint_position", "joint_velocity", "joint_effort" }));
    if (!plan->ok()) fprintf(stderr, "SegmentStreamer: %s is not a joint_state_t\n", type.c_str());
    auto js = std::make_shared<JointInfo>();
    // record (after configure): i64 utime | f32 position[n] velocity[n] effort[n] | u8 valid
    c.configure = [plan, js](const MappedLog::Event &ev, Chan &self) {
      std::vector<pronto_wire::Schema::Extracted> x;
      pronto_wire::Schema::Plan::Shape sh;
      if (!plan->run(ev.data, ev.dlen, x, sh, nullptr) || x[1].str.empty()) return false;
      js->names = x[1].str;
      js->n = js->names.size();
      js->hash = names_hash(js->names);
      self.planes = { { 8, 1 }, { 4, (int) (3 * js->n) }, { 1, 1 } };
      return true;
    };
    c.decode = [plan, js](const MappedLog::Event &ev, SegChan &st, uint8_t *rec, int64_t &utime) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      bool rebuilt = false;
      if (!plan->layout(ev.data, ev.dlen, st.shape, x, &rebuilt)) return false;
      if (rebuilt) st.names_hash = names_hash(x[1].str);   // a new joint list (normally: the first message of the segment)
      // one robot model for the batch: the same joints in the same order as the lead's
      if (st.names_hash != js->hash) return false;
      using P = pronto_wire::Schema::Plan;
      const size_t n = js->n;
      float *f = (float *) (rec + 8);
      if (P::gather_i64(st.shape, ev.data, 0, &utime, 1) != 1 || P::gather_f32(st.shape, ev.data, 2, f, n) != n ||
          P::gather_f32(st.shape, ev.data, 3, f + n, n) != n || P::gather_f32(st.shape, ev.data, 4, f + 2 * n, n) != n)
        return false;
      memcpy(rec, &utime, 8);
      rec[8 + 12 * n] = 1;
      return true;
    };
    c.blank = [js](uint8_t *rec) { rec[8 + 12 * js->n] = 0; };   // (the block keeps this filter's last message; it is masked)
    c.dispatch = [this, cb, js](const SlotView &v) {
      msgs::joint_state_t m;
      m.utime = v.utime;
      m.joint_name = js->names;
      const size_t n = js->n;
      m.joint_position = (const float *) v.dev(8);
      m.joint_velocity = (const float *) v.dev(8 + 4 * n);
      m.joint_effort = (const float *) v.dev(8 + 8 * n);
      m.mem = PB_DEVICE;
      m.utimes = (const int64_t *) v.dev(0);
      m.valid = (const uint8_t *) v.dev(8 + 12 * n);
      m.times_mem = PB_DEVICE;
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::six_axis_force_torque_array_t (sensors[0 / 1].force[2] = left / right foot) -> LegOdoHandler::forceTorqueDevice: what
  // the handler keeps of the message is |force z| of the two feet as floats (rbis_legodo_update.cpp:195-204,234-235) -- [2][B] in HBM
  void subscribeForceTorque(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                            std::function<void(const float *abs_force_z_dev)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "sensors.force" }));
    if (!plan->ok()) fprintf(stderr, "SegmentStreamer: %s is not a six_axis_force_torque_array_t\n", type.c_str());
    c.planes = { { 4, 2 } };
    c.decode = [plan](const MappedLog::Event &ev, SegChan &st, uint8_t *rec, int64_t &utime) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->layout(ev.data, ev.dlen, st.shape, x, nullptr)) return false;
      double f[6];
      if (pronto_wire::Schema::Plan::gather_i64(st.shape, ev.data, 0, &utime, 1) != 1 || pronto_wire::Schema::Plan::gather_f64(st.shape, ev.data, 1, f, 6) != 6) return false;
      const float a[2] = { (float) fabs(f[2]), (float) fabs(f[5]) };   // FootSensing(fabs(...)), float members
      memcpy(rec, a, 8);
      return true;
    };
    c.blank = [](uint8_t *) {};   // (the handler keeps the LAST force/torque message, per filter: a missing one changes nothing)
    c.dispatch = [cb](const SlotView &v) { cb((const float *) v.dev(0)); };
    chans_[channel] = std::move(c);
  }
  // bot_core::pose_t (utime, pos[3], vel[3], orientation[4]) -> ScanMatcherHandler::processMessage.  A sparse channel: handed over
  // as PB_HOST blocks inside the page-locked chunk (the handler copies what it keeps).
  void subscribePose(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                     std::function<void(const msgs::pose_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "pos", "vel", "orientation" }));
    if (!plan->ok()) fprintf(stderr, "SegmentStreamer: %s is not a pose_t\n", type.c_str());
    c.planes = { { 8, 10 }, { 1, 1 } };
    c.carry = false;
    c.decode = [plan](const MappedLog::Event &ev, SegChan &st, uint8_t *rec, int64_t &utime) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->layout(ev.data, ev.dlen, st.shape, x, nullptr)) return false;
      using P = pronto_wire::Schema::Plan;
      double v[10];
      if (P::gather_i64(st.shape, ev.data, 0, &utime, 1) != 1 || P::gather_f64(st.shape, ev.data, 1, v, 3) != 3 || P::gather_f64(st.shape, ev.data, 2, v + 3, 3) != 3 ||
          P::gather_f64(st.shape, ev.data, 3, v + 6, 4) != 4)
        return false;
      memcpy(rec, v, sizeof v);
      rec[80] = 1;
      return true;
    };
    c.blank = [](uint8_t *rec) {
      const double none[10] = { 0, 0, 0, 0, 0, 0, 1, 0, 0, 0 };
      memcpy(rec, none, sizeof none);
      rec[80] = 0;
    };
    c.dispatch = [cb](const SlotView &v) {
      msgs::pose_t m{ v.utime, BatchArray((const double *) v.host(0), PB_HOST), BatchArray((const double *) v.host(24), PB_HOST),
                      BatchArray((const double *) v.host(48), PB_HOST) };
      m.valid = (const uint8_t *) v.host(80);
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }
  // pronto::update_t (pronto_wire.hpp) -> FovisHandler::processMessage.  timestamp / prev_timestamp are the lead's.  Sparse: PB_HOST.
  void subscribeUpdate(const std::string &channel, std::function<void(const msgs::update_t *)> cb)
  {
    Chan c;
    c.planes = { { 8, 7 }, { 1, 1 } };
    c.carry = false;
    c.decode = [](const MappedLog::Event &ev, SegChan &, uint8_t *rec, int64_t &utime) {
      pronto_wire::update_t w;
      if (w.decode(ev.data, ev.dlen) < 0) return false;
      utime = w.timestamp;
      memcpy(rec, w.translation, 24);
      memcpy(rec + 24, w.rotation, 32);
      rec[56] = w.estimate_status == pronto_wire::update_t::ESTIMATE_VALID;
      return true;
    };
    c.aux = [](const MappedLog::Event &ev) {   // the lead's prev_timestamp behind its timestamp
      pronto_wire::update_t w;
      return w.decode(ev.data, ev.dlen) < 0 ? (int64_t) 0 : w.prev_timestamp - w.timestamp;
    };
    c.blank = [](uint8_t *rec) {
      const double none[7] = { 0, 0, 0, 1, 0, 0, 0 };
      memcpy(rec, none, sizeof none);
      rec[56] = 0;
    };
    c.dispatch = [cb](const SlotView &v) {
      msgs::update_t m{ v.utime, v.utime + v.aux, (const uint8_t *) v.host(56), BatchArray((const double *) v.host(0), PB_HOST),
                        BatchArray((const double *) v.host(24), PB_HOST) };
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }

  // Replays every segment to its end.  Returns the number of batched messages dispatched, or -1 when no segment was added.
  int64_t run()
  {
    if (segs_.empty()) return -1;
    chan_list_.clear();
    for (auto &kv : chans_) {
      kv.second.id = (int) chan_list_.size();
      kv.second.name = kv.first;
      chan_list_.push_back(&kv.second);
    }
    const size_t NC = chan_list_.size();
    for (auto &sg : segs_)
      if (sg->chan.size() != NC) sg->chan.resize(NC);
    if (ring < 3) ring = 3;   // (a block is read by its own chunk's kernels and by the first ones of the next)
    if (ring > 8) ring = 8;
    if (group < 1) group = 1;
    if (max_slots < 1) max_slots = 1;
    bufs_.reserve(8);   // (the decode-ahead thread appends ring buffers while this thread reads the ones in use)
    for (int i = 0; i < ring; i++) {
      int f = -1;
      if (pb_fence_create(est_->ctx, &f) != PB_OK) { fprintf(stderr, "SegmentStreamer: %s\n", pb_last_error(est_->ctx)); exit(1); }
      fences_.push_back(f);
    }
    // ---- the decode-ahead thread ----
    std::thread producer([this]() { produce(); });
    // ---- this thread: upload, dispatch, finalize ----
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto since = [&now](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    for (;;) {
      auto t0 = now();
      Chunk *ck = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_ready_.wait(lk, [this]() { return !ready_.empty() || done_; });
        if (ready_.empty()) break;
        ck = ready_.front();
        ready_.pop_front();
      }
      stats.t_wait_chunk += since(t0);
      if (ck->buf < 0) {   // nothing but finished runs
        finalize_after(*ck, -1);
        delete ck;
        continue;
      }
      t0 = now();
      Buf &bf = bufs_[(size_t) ck->buf];
      // the device block of this ring slot was last read by the kernels behind fences_[buf]: the copy waits for them, the kernels of
      // the chunk in front of this one run meanwhile
      int rc = pb_upload_async(est_->ctx, bf.dev, bf.host, ck->bytes, fences_[(size_t) ck->buf]);
      if (rc == PB_OK) rc = pb_upload_join(est_->ctx);
      if (rc != PB_OK) { fprintf(stderr, "SegmentStreamer: %s\n", pb_last_error(est_->ctx)); est_->last_status = rc; }
      stats.uploaded_bytes += ck->bytes;
      stats.t_upload += since(t0);
      finalize_after(*ck, -1);
      for (int k = 0; k < (int) ck->slots.size(); k++) {
        const Slot &sl = ck->slots[(size_t) k];
        t0 = now();
        SlotView v{ bf.host + sl.off, bf.dev + sl.off, (size_t) B_, sl.utime, sl.aux };
        chan_list_[(size_t) sl.chan]->dispatch(v);
        stats.batches++;
        stats.per_channel[chan_list_[(size_t) sl.chan]->name]++;
        stats.t_handlers += since(t0);
        finalize_after(*ck, k);
      }
      // Everything that reads this chunk's device block has been enqueued -- except what crosses into the NEXT chunk: a force/torque
      // block kept by the handler for the joint state that follows it, an update the estimator holds back.  So the marker of the
      // PREVIOUS chunk's block is set (again) here, behind this chunk's kernels: the block of chunk i is written again by the upload
      // of chunk i + ring, issued after chunk i + ring - 1 >= i + 2 has been dispatched.
      pb_fence_record(est_->ctx, fences_[(size_t) ck->buf]);
      if (prev_buf_ >= 0 && prev_buf_ != ck->buf) pb_fence_record(est_->ctx, fences_[(size_t) prev_buf_]);
      prev_buf_ = ck->buf;
      t0 = now();
      pb_upload_sync(est_->ctx);   // the page-locked block may be refilled (the sparse channels' PB_HOST blocks were copied by their handlers)
      stats.t_upload += since(t0);
      stats.chunks++;
      {
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(ck->buf);
        delete ck;
      }
      cv_free_.notify_one();
    }
    producer.join();
    for (const auto &sg : segs_) {
      stats.undecodable += sg->undecodable;
      stats.order_violations += sg->order_violations;
      stats.max_skew_us = std::max(stats.max_skew_us, sg->max_skew);
      stats.segment_messages += sg->messages;
      stats.ragged += sg->ragged;
      stats.readahead_capped += sg->capped;
      for (const SegChan &sc : sg->chan) stats.undecodable += sc.overflow;
      sg->undecodable = sg->order_violations = sg->max_skew = sg->messages = sg->ragged = sg->capped = 0;
    }
    return stats.batches;
  }

  // the head of every segment's filter at the end of ITS log (valid after run(); columns of segments never added are zero)
  void finalState(RBIS &state, RBIM &cov) const
  {
    state = RBIS(est_->n, B_);
    cov = RBIM(est_->n, B_);
    state.vec = final_vec_;
    state.quat = final_quat_;
    cov.m = final_cov_;
  }
  const std::vector<double> &finalLogLikelihood() const { return final_ll_; }
  int64_t finalUtime(int s) const { return final_utime_[(size_t) s]; }

private:
  struct Plane { int elem, count; };
  struct SegChan {   // what one segment remembers about one channel
    pronto_wire::Schema::Plan::Shape shape;
    uint64_t names_hash = 0;
    std::vector<uint8_t> last;        // its last record on this channel (what a slot without a message repeats)
    int64_t last_packet = -1, last_packet_utime = 0, overflow = 0;   // IMUStream (imu_stream.hpp:10-37)
  };
  struct SlotView {
    uint8_t *host_base, *dev_base;
    size_t B;
    int64_t utime, aux;
    const void *dev(size_t rec_off) const { return dev_base + rec_off * B; }    // element at record offset o of filter s: base + o * B + s * elem
    const void *host(size_t rec_off) const { return host_base + rec_off * B; }
  };
  struct Chan {
    std::vector<Plane> planes;        // record layout: the planes back to back, 8-byte members first (natural alignment)
    size_t rec_bytes = 0, rec_stride = 0;   // exact size (a chunk block is rec_bytes * B), and rounded up to 8 for the host-side copies
    bool carry = true, configured = false;
    std::function<bool(const MappedLog::Event &, Chan &)> configure;   // sizes that only the first message knows (joint count)
    std::function<bool(const MappedLog::Event &, SegChan &, uint8_t *, int64_t &)> decode;
    std::function<int64_t(const MappedLog::Event &)> aux;
    std::function<void(uint8_t *)> blank;
    std::function<void(const SlotView &)> dispatch;
    int id = -1;
    std::string name;
    std::vector<std::pair<uint32_t, uint8_t>> elems;   // (record offset, element size) of every element, for the transposition
  };
  struct JointInfo {
    std::vector<std::string> names;
    size_t n = 0;
    uint64_t hash = 0;
  };
  struct Seg {
    std::shared_ptr<MappedLog> log;
    int64_t start_timestamp = 0, end_timestamp = 0, t0 = INT64_MIN;
    size_t pos = 0;
    bool eof = false, ended = false;
    std::vector<size_t> carry;                // offsets of events read past a quota: consumed first by the next chunk
    std::vector<SegChan> chan;
    int64_t undecodable = 0, order_violations = 0, max_skew = 0, messages = 0, ragged = 0, capped = 0;
  };
  struct Slot {
    int chan, j;            // channel and index of this message among the chunk's messages of that channel
    size_t off;             // offset of its block in the chunk buffer
    int64_t utime, aux, lead_rel;
  };
  struct Chunk {
    int buf = 0;
    size_t bytes = 0;
    std::vector<Slot> slots;
    std::vector<std::pair<int, int>> ends;   // (slot after which it ended, segment); slot -1 = before the first
  };
  struct Buf {
    uint8_t *host = nullptr, *dev = nullptr;
  };

  static uint64_t names_hash(const std::vector<std::string> &names)
  {
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the names and their boundaries
    for (const std::string &nm : names) {
      for (unsigned char ch : nm) h = (h ^ ch) * 1099511628211ull;
      h = (h ^ 0xffu) * 1099511628211ull;
    }
    return h;
  }
  int channel_of(const MappedLog::Event &ev) const
  {
    for (const Chan *c : chan_list_)
      if (c->name.size() == ev.clen && memcmp(c->name.data(), ev.channel, ev.clen) == 0) return c->id;
    return -1;
  }
  // the segment's next event at or after its start_timestamp: first what an earlier chunk read past, then the log
  bool next_event(Seg &sg, size_t &carry_at, MappedLog::Event &ev)
  {
    while (carry_at < sg.carry.size()) {
      size_t p = sg.carry[carry_at++];
      if (sg.log->next(p, ev)) return true;
    }
    while (!sg.eof) {
      if (!sg.log->next(sg.pos, ev) || (sg.end_timestamp > 0 && ev.timestamp >= sg.end_timestamp)) { sg.eof = true; return false; }
      if (ev.timestamp < sg.start_timestamp) continue;   // "?start_timestamp=": lcm_front_end.cpp:21-33
      return true;
    }
    return false;
  }
  void finish_layout(Chan &c)
  {
    c.rec_bytes = 0;
    c.elems.clear();
    for (const Plane &p : c.planes) {
      for (int k = 0; k < p.count; k++) c.elems.push_back({ (uint32_t) (c.rec_bytes + (size_t) k * p.elem), (uint8_t) p.elem });
      c.rec_bytes += (size_t) p.elem * p.count;
    }
    c.rec_stride = (c.rec_bytes + 7) / 8 * 8;
    c.configured = true;
  }

  // ---- decode-ahead: one chunk after the other until every segment has ended ----
  void produce()
  {
    auto now = []() { return std::chrono::steady_clock::now(); };
    const int nseg = (int) segs_.size();
    const size_t NC = chan_list_.size();
    std::vector<std::vector<int>> slot_of(NC);
    std::vector<uint8_t> lead_recs;
    std::vector<size_t> lead_rec_at;
    std::vector<std::pair<int, int>> pending_ends;
    for (;;) {
      const auto t_start = now();
      // the lead: the first segment that still has events
      while (first_alive_ < nseg && segs_[(size_t) first_alive_]->ended) first_alive_++;
      if (first_alive_ >= nseg) break;
      Seg &lead = *segs_[(size_t) first_alive_];
      std::unique_ptr<Chunk> ck(new Chunk());
      for (auto &v : slot_of) v.clear();
      lead_recs.clear();
      lead_rec_at.clear();
      // -- serial pass: the lead's next events make the chunk's schedule (and are decoded on the way) --
      size_t off = 0, carry_at = 0;
      std::vector<size_t> lead_carry_rest;
      int last_k = -1;
      while ((int) ck->slots.size() < max_slots) {
        MappedLog::Event ev;
        if (!next_event(lead, carry_at, ev)) break;
        const int c = channel_of(ev);
        if (c < 0) continue;
        Chan &ch = *chan_list_[(size_t) c];
        if (!ch.configured) {
          if (ch.configure && !ch.configure(ev, ch)) { lead.undecodable++; continue; }
          finish_layout(ch);
        }
        const size_t blk = (ch.rec_bytes * (size_t) B_ + 255) / 256 * 256;
        if (!ck->slots.empty() && off + blk > chunk_budget_bytes) {   // does not fit any more: it opens the next chunk
          lead_carry_rest.push_back(ev.at);
          break;
        }
        const size_t at = lead_recs.size();
        lead_recs.resize(at + ch.rec_stride);
        int64_t utime = 0;
        SegChan &sc = lead.chan[(size_t) c];
        if (sc.last.size() != ch.rec_stride) sc.last.assign(ch.rec_stride, 0);
        memcpy(lead_recs.data() + at, sc.last.data(), ch.rec_bytes);   // (members the decoder leaves alone keep their last value)
        if (!ch.decode(ev, sc, lead_recs.data() + at, utime)) {
          lead_recs.resize(at);
          lead.undecodable++;
          continue;
        }
        memcpy(sc.last.data(), lead_recs.data() + at, ch.rec_bytes);
        if (lead.t0 == INT64_MIN) lead.t0 = utime;
        if (base_ == INT64_MIN) base_ = lead.t0;   // the batch's time base: the first lead's first message
        const int64_t rel = utime - lead.t0;
        Slot sl{ c, (int) slot_of[(size_t) c].size(), off, base_ + rel, ch.aux ? ch.aux(ev) : 0, rel };
        slot_of[(size_t) c].push_back((int) ck->slots.size());
        lead_rec_at.push_back(at);
        ck->slots.push_back(sl);
        off += blk;
        lead.messages++;
        last_k = (int) ck->slots.size() - 1;
      }
      // what the lead had read past in earlier chunks and did not consume now stays in front of the event that did not fit
      {
        std::vector<size_t> rest(lead.carry.begin() + (long) std::min(carry_at, lead.carry.size()), lead.carry.end());
        rest.insert(rest.end(), lead_carry_rest.begin(), lead_carry_rest.end());
        std::sort(rest.begin(), rest.end());   // (file order: the event that did not fit may itself have come from the list)
        lead.carry.swap(rest);
      }
      peek_end(lead, last_k, *ck, first_alive_);
      if (ck->slots.empty()) {   // the lead had nothing decodable left: the next segment leads; its end is finalized in front of the next chunk
        for (const auto &e : ck->ends) pending_ends.push_back({ -1, e.second });
        continue;
      }
      ck->ends.insert(ck->ends.end(), pending_ends.begin(), pending_ends.end());
      pending_ends.clear();
      ck->bytes = off;
      stats.t_lead += std::chrono::duration<double>(now() - t_start).count();
      // -- a free ring slot --
      {
        const auto tw = now();
        std::unique_lock<std::mutex> lk(mu_);
        if ((int) bufs_.size() < ring && free_.empty()) {
          lk.unlock();
          Buf b;
          const uint64_t cap = chunk_budget_bytes + ((uint64_t) largest_block() + 255) / 256 * 256;
          void *h = nullptr, *d = nullptr;
          if (pb_host_alloc(est_->ctx, cap, &h) != PB_OK || pb_malloc(est_->ctx, cap, &d) != PB_OK) {
            fprintf(stderr, "SegmentStreamer: %s\n", pb_last_error(est_->ctx));
            exit(1);
          }
          memset(h, 0, cap);   // (columns of filters without a segment stay what they are here: zero, valid = 0)
          b.host = (uint8_t *) h;
          b.dev = (uint8_t *) d;
          stats.chunk_bytes = cap;
          lk.lock();
          bufs_.push_back(b);
          free_.push_back((int) bufs_.size() - 1);
        }
        cv_free_.wait(lk, [this]() { return !free_.empty(); });
        ck->buf = free_.front();
        free_.pop_front();
        stats.t_wait_free += std::chrono::duration<double>(now() - tw).count();
      }
      const auto t_dec = now();
      uint8_t *host = bufs_[(size_t) ck->buf].host;
      const int G = group, ngroups = (nseg + G - 1) / G;
      const int n_slots = (int) ck->slots.size();
      std::vector<std::vector<std::pair<int, int>>> ends_of((size_t) ngroups);
      Chunk *ckp = ck.get();
      const int lead_s = first_alive_;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(pb_shim_threads())
#endif
      for (int g = 0; g < ngroups; g++) {
        static thread_local std::vector<uint8_t> local;    // [slot][segment of the group][record]
        static thread_local std::vector<size_t> slot_at;
        static thread_local std::vector<int> taken;
        static thread_local std::vector<uint8_t> filled;
        const int s0 = g * G, ns = std::min(G, nseg - s0);
        slot_at.resize((size_t) n_slots);
        size_t tot = 0;
        for (int k = 0; k < n_slots; k++) {
          slot_at[(size_t) k] = tot;
          tot += chan_list_[(size_t) ckp->slots[(size_t) k].chan]->rec_stride * (size_t) G;
        }
        local.resize(tot);
        for (int i = 0; i < ns; i++) {
          const int s = s0 + i;
          Seg &sg = *segs_[(size_t) s];
          filled.assign((size_t) n_slots, 0);
          auto rec_of = [&](int k) { return local.data() + slot_at[(size_t) k] + (size_t) i * chan_list_[(size_t) ckp->slots[(size_t) k].chan]->rec_stride; };
          if (s == lead_s) {
            for (int k = 0; k < n_slots; k++) {
              memcpy(rec_of(k), lead_recs.data() + lead_rec_at[(size_t) k], chan_list_[(size_t) ckp->slots[(size_t) k].chan]->rec_bytes);
              filled[(size_t) k] = 1;
            }
          } else if (!sg.ended) {
            taken.assign(NC, 0);
            int total = 0, last = -1;
            size_t cat = 0;
            std::vector<size_t> keep;
            MappedLog::Event ev;
            while (total < n_slots && next_event(sg, cat, ev)) {
              const int c = channel_of(ev);
              if (c < 0) continue;
              if (taken[(size_t) c] >= (int) slot_of[(size_t) c].size()) {   // past this chunk's quota for the channel: the next chunk's
                keep.push_back(ev.at);
                if (keep.size() > readahead_cap) { sg.capped++; break; }
                continue;
              }
              const int k = slot_of[(size_t) c][(size_t) taken[(size_t) c]];
              Chan &ch = *chan_list_[(size_t) c];
              SegChan &sc = sg.chan[(size_t) c];
              if (sc.last.size() != ch.rec_stride) sc.last.assign(ch.rec_stride, 0);
              uint8_t *rec = rec_of(k);
              memcpy(rec, sc.last.data(), ch.rec_bytes);
              int64_t utime = 0;
              if (!ch.decode(ev, sc, rec, utime)) { sg.undecodable++; continue; }
              memcpy(sc.last.data(), rec, ch.rec_bytes);
              taken[(size_t) c]++;
              total++;
              filled[(size_t) k] = 1;
              if (k < last) sg.order_violations++;   // this segment's own log had them in another order than the lead's
              last = std::max(last, k);
              if (sg.t0 == INT64_MIN) sg.t0 = utime;
              sg.max_skew = std::max<int64_t>(sg.max_skew, std::llabs((utime - sg.t0) - ckp->slots[(size_t) k].lead_rel));
              sg.messages++;
            }
            {
              std::vector<size_t> rest(sg.carry.begin() + (long) std::min(cat, sg.carry.size()), sg.carry.end());
              // (events read past a quota were read in file order: those from the old carry list come first)
              keep.insert(keep.end(), rest.begin(), rest.end());
              std::sort(keep.begin(), keep.end());
              sg.carry.swap(keep);
            }
            peek_end_into(sg, last, ends_of[(size_t) g], s);
          }
          // slots without a message of this segment: its last record on that channel, marked "no message"
          for (int k = 0; k < n_slots; k++) {
            if (filled[(size_t) k]) continue;
            Chan &ch = *chan_list_[(size_t) ckp->slots[(size_t) k].chan];
            SegChan &sc = sg.chan[(size_t) ch.id];
            if (sc.last.size() != ch.rec_stride) sc.last.assign(ch.rec_stride, 0);
            uint8_t *rec = rec_of(k);
            memcpy(rec, sc.last.data(), ch.rec_bytes);
            if (!ch.carry) memset(rec, 0, ch.rec_bytes);
            ch.blank(rec);
            sg.ragged++;
          }
        }
        // the group's records -> the chunk's [rows][B] blocks, whole runs of `ns` neighbouring filters at a time
        for (int k = 0; k < n_slots; k++) {
          const Chan &ch = *chan_list_[(size_t) ckp->slots[(size_t) k].chan];
          uint8_t *blk = host + ckp->slots[(size_t) k].off;
          const uint8_t *src = local.data() + slot_at[(size_t) k];
          const size_t rb = ch.rec_stride;
          for (const auto &el : ch.elems) {
            uint8_t *dst = blk + (size_t) el.first * (size_t) B_ + (size_t) s0 * el.second;
            const uint8_t *sp = src + el.first;
            if (el.second == 8)
              for (int i = 0; i < ns; i++) memcpy(dst + 8 * (size_t) i, sp + (size_t) i * rb, 8);
            else if (el.second == 4)
              for (int i = 0; i < ns; i++) memcpy(dst + 4 * (size_t) i, sp + (size_t) i * rb, 4);
            else
              for (int i = 0; i < ns; i++) dst[i] = sp[(size_t) i * rb];
          }
        }
      }
      for (auto &v : ends_of) ck->ends.insert(ck->ends.end(), v.begin(), v.end());
      std::sort(ck->ends.begin(), ck->ends.end());
      stats.t_decode += std::chrono::duration<double>(now() - t_dec).count();
      {
        std::lock_guard<std::mutex> lk(mu_);
        ready_.push_back(ck.release());
      }
      cv_ready_.notify_one();
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!pending_ends.empty()) {   // runs that ended behind the last chunk: an empty chunk carries their finalization
        Chunk *ck = new Chunk();
        ck->buf = -1;
        ck->ends = pending_ends;
        std::sort(ck->ends.begin(), ck->ends.end());
        ready_.push_back(ck);
      }
      done_ = true;
    }
    cv_ready_.notify_one();
  }
  size_t largest_block() const
  {
    size_t m = 0;
    for (const Chan *c : chan_list_) m = std::max(m, c->rec_bytes * (size_t) B_);
    return m;
  }
  // Does the segment have a subscribed event left?  Reads ahead to the next one (kept for the next chunk) like SegmentBatcher's
  // fill(); at the end of the log the run is complete: it is finalized right behind the last slot it had a message in.
  void peek_end_into(Seg &sg, int last_slot, std::vector<std::pair<int, int>> &ends, int s)
  {
    if (sg.ended || !sg.carry.empty()) return;
    MappedLog::Event ev;
    while (next_event_from_log(sg, ev)) {
      if (channel_of(ev) < 0) continue;
      sg.carry.push_back(ev.at);
      return;
    }
    sg.ended = true;
    ends.push_back({ last_slot, s });
  }
  void peek_end(Seg &sg, int last_slot, Chunk &ck, int s) { peek_end_into(sg, last_slot, ck.ends, s); }
  bool next_event_from_log(Seg &sg, MappedLog::Event &ev)
  {
    while (!sg.eof) {
      if (!sg.log->next(sg.pos, ev) || (sg.end_timestamp > 0 && ev.timestamp >= sg.end_timestamp)) { sg.eof = true; return false; }
      if (ev.timestamp < sg.start_timestamp) continue;
      return true;
    }
    return false;
  }

  // the runs that ended right behind slot k of this chunk: read their heads (applies whatever the estimator holds back first)
  void finalize_after(const Chunk &ck, int k)
  {
    auto lo = std::lower_bound(ck.ends.begin(), ck.ends.end(), std::make_pair(k, -1));
    if (lo == ck.ends.end() || lo->first != k) return;
    const auto t0 = std::chrono::steady_clock::now();
    auto it = lo;
    while (it != ck.ends.end() && it->first == k) {
      int first = it->second, count = 1;
      ++it;
      while (it != ck.ends.end() && it->first == k && it->second == first + count) { count++; ++it; }
      finalize(first, count);
    }
    stats.t_final += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  void finalize(int first, int count)
  {
    const int n = est_->n;
    est_->flushPending();
    std::vector<double> v((size_t) n * count), q((size_t) 4 * count), c((size_t) n * n * count), l((size_t) count);
    if (pb_get_head(est_->ctx, first, count, v.data(), q.data(), c.data(), l.data(), PB_HOST) != PB_OK) {
      fprintf(stderr, "SegmentStreamer: %s\n", pb_last_error(est_->ctx));
      return;
    }
    for (int k = 0; k < count; k++) {
      const size_t s = (size_t) (first + k);
      for (int i = 0; i < n; i++) final_vec_[(size_t) i * B_ + s] = v[(size_t) i * count + k];
      for (int i = 0; i < 4; i++) final_quat_[(size_t) i * B_ + s] = q[(size_t) i * count + k];
      for (int i = 0; i < n * n; i++) final_cov_[(size_t) i * B_ + s] = c[(size_t) i * count + k];
      final_ll_[s] = l[(size_t) k];
      final_utime_[s] = est_->head_utime;
      finished_[s] = 1;
    }
  }

  MavStateEstimator *est_;
  int B_;
  int64_t base_ = INT64_MIN;
  std::vector<double> final_vec_, final_quat_, final_cov_, final_ll_;
  std::vector<int64_t> final_utime_;
  std::vector<uint8_t> finished_;
  std::vector<std::unique_ptr<Seg>> segs_;
  std::map<std::string, Chan> chans_;
  std::vector<Chan *> chan_list_;
  int first_alive_ = 0;
  std::vector<Buf> bufs_;
  std::vector<int> fences_;
  int prev_buf_ = -1;
  std::mutex mu_;
  std::condition_variable cv_ready_, cv_free_;
  std::deque<Chunk *> ready_;
  std::deque<int> free_;
  bool done_ = false;
};

}  // namespace MavStateEst
