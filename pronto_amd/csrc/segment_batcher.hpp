// segment_batcher.hpp -- N independent log segments replayed as ONE batch: segment s feeds filter s.
//
// The reference's many-runs workload is a loop over recorded logs, one se-fusion process after the other
// (motion_estimate/scripts/se-batch-process.sh:17-26,58-59,70-74), each opened as "file://<log>?speed=..&start_timestamp=<t>"
// (state-estimator/src/mav_state_est/lcm_front_end.cpp:21-33: the log provider skips to the first event at or after t).  Here
// the N runs share every kernel launch: the k-th message of a channel in each segment becomes column s of ONE batched message
// ([rows][B] host blocks, filter index fastest) that is handed to the callback LCMFrontEnd::addSensor returns -- the same
// handlers, the same estimator, B = N filters.  LogPlayer (mav_state_est_batch.hpp) is the other case: one log for every filter.
//
// Alignment.  Messages are aligned BY INDEX per channel, and the channels are dispatched in the file order of the LEAD segment
// (the lowest-numbered segment that still has events).  That is exact for segments recorded by the same robot software -- the
// same channels at the same rates in the same interleaving -- which is what a set of runs of one robot, or N start offsets into
// one long log, are.  What is checked instead of assumed:
//   * order_violations   a segment's own file order differs from the lead's (its k-th message of channel C was not the next
//                        subscribed event in its log): the filter then sees its updates in the lead's order, not its own;
//   * max_skew_us        largest difference between a segment's message time (relative to its first dispatched message) and the
//                        lead's for the same batched message.
// Time.  A batched message carries the LEAD's time, relative to the lead's first message, on the time base of the first segment (it
// orders the update in the estimator's history and stays continuous when the lead changes).  Every filter's own time
// stamp travels beside it where the arithmetic reads one: the leg odometry (msgs::joint_state_t::utimes -> per-filter elapsed
// time, 30 ms reset, contact-classifier clocks: pb_legodo_set_message_times).  InsHandler::processMessage integrates with the
// configured timestep_dt like the reference (sensor_handlers.cpp:96-131), so it needs none.
// Ragged ends.  A segment that runs out of messages on a channel gets valid = 0 in that channel's batched messages from then on:
// its measurements are masked and its IMU step is taken with dt = 0 on its own last sample.  The RESULT of a run is its head at the
// end of ITS log: the moment a segment has no event left, its filter's head is read into finalState() / finalLogLikelihood()
// (one pb_get_head per run of segments that end together: with equal-length segments one call for the whole batch).  The
// filter idles in the batch afterwards; an idle step keeps pose, velocity, biases and covariance but re-derives the
// angular-velocity / acceleration entries from its last sample, so the estimator's own head is NOT the result of a finished run.
//
// Host side only; every batched array lives in page-locked memory (pb_host_alloc) so that the library's double-buffered staging
// copies run as DMA.  Decoding goes through the run-time .lcm schema (lcm_schema.hpp): the bot_core definitions are the caller's.
#pragma once

#include <chrono>
#include <deque>

#include "mav_state_est_batch.hpp"

namespace MavStateEst {

class SegmentBatcher {
public:
  struct Stats {
    int64_t batches = 0;            // batched messages dispatched
    int64_t segment_messages = 0;   // per-segment messages they carried
    int64_t ragged = 0;             // columns without a message (segment ended)
    int64_t order_violations = 0, undecodable = 0, max_skew_us = 0;
    int64_t readahead_capped = 0;   // ticks in which a segment read `readahead_cap` events ahead without finding the channel's next message
    std::map<std::string, int64_t> per_channel;
    // where run() spent its wall-clock time, seconds: finding the lead (serial; includes its read-ahead), the segments' reads +
    // decodes (parallel), book-keeping (serial), assembling the batched arrays + the handler call (parallel loop + one enqueue),
    // the read-ahead of the next messages (parallel), reading finished runs' heads
    double t_lead = 0, t_pull = 0, t_book = 0, t_dispatch = 0, t_fill = 0, t_final = 0;
    double t_handler = 0, t_upload = 0;   // of t_dispatch: inside the handlers' callbacks (host passes, enqueues), the joint blocks' DMA

  };
  Stats stats;
  size_t readahead_cap = 4096;

  explicit SegmentBatcher(MavStateEstimator *est)
      : est_(est), B_(est->B), final_vec_((size_t) est->n * est->B, 0.0), final_quat_((size_t) 4 * est->B, 0.0),
        final_cov_((size_t) est->n * est->n * est->B, 0.0), final_ll_((size_t) est->B, 0.0), final_utime_((size_t) est->B, 0),
        finished_((size_t) est->B, 0) {}
  ~SegmentBatcher()
  {
    est_->flushPending();   // (an update the estimator is holding back may still read the device blocks)
    pb_sync(est_->ctx);
    for (void *p : pinned_) pb_host_free(est_->ctx, p);
    for (void *p : device_) pb_free(est_->ctx, p);
  }
  SegmentBatcher(const SegmentBatcher &) = delete;
  SegmentBatcher &operator=(const SegmentBatcher &) = delete;

  template <class F>
  void timed(double &acc, F &&f)
  {
    const auto t0 = std::chrono::steady_clock::now();
    f();
    acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // segment s = filter s, in the order added; false: the file cannot be opened or the batch is full
  bool addSegment(const std::string &path, int64_t start_timestamp = 0)
  {
    if ((int) segs_.size() >= B_) return false;
    auto sg = std::unique_ptr<Seg>(new Seg(path, start_timestamp));
    if (!sg->rd.good()) return false;
    segs_.push_back(std::move(sg));
    return true;
  }
  int segments() const { return (int) segs_.size(); }

  // ---- typed subscriptions: channel -> the callback FrontEnd::addSensor returned ----
  // bot_core::ins_t by field name (utime, gyro[3], accel[3]): InsHandler::processMessage
  void subscribeIns(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                    std::function<void(const msgs::ins_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "gyro", "accel" }));
    if (!plan->ok()) fprintf(stderr, "SegmentBatcher: %s has no utime / gyro / accel members\n", type.c_str());
    c.decode = [plan](const pronto_wire::LogEvent &ev, Rec &r, Stream &st) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->run(ev.data.data(), ev.data.size(), x, st.shape, nullptr) || x[0].num.size() != 1 || x[1].num.size() != 3 || x[2].num.size() != 3) return false;
      r.utime = (int64_t) x[0].num[0];
      r.d.assign(x[1].num.begin(), x[1].num.end());
      r.d.insert(r.d.end(), x[2].num.begin(), x[2].num.end());
      return true;
    };
    double *blk = pinned<double>((size_t) 6 * B_);
    uint8_t *valid = pinned<uint8_t>((size_t) B_);
    auto last = std::make_shared<std::vector<double>>((size_t) 6 * B_, 0.0);
    c.dispatch = [this, cb, blk, valid, last](const std::vector<const Rec *> &col, int64_t utime) {
      PB_SHIM_PARALLEL_FOR
      for (int s = 0; s < B_; s++) {
        valid[s] = col[(size_t) s] != nullptr;
        for (int i = 0; i < 6; i++) {
          if (col[(size_t) s]) (*last)[(size_t) i * B_ + s] = col[(size_t) s]->d[(size_t) i];
          blk[(size_t) i * B_ + s] = (*last)[(size_t) i * B_ + s];   // (a filter without a message idles on its own last sample)
        }
      }
      msgs::ins_t m{ utime, BatchArray(blk, PB_HOST), BatchArray(blk + (size_t) 3 * B_, PB_HOST) };
      m.valid = valid;
      timed(stats.t_handler, [&]() { cb(&m); });
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::kvh_raw_imu_batch_t (utime, raw_imu[]{utime, packet_count, delta_rotation[3], linear_acceleration[3]}; raw_imu[0] is the
  // NEWEST packet): the reference's Atlas IMU channel (fusion.cpp:161-163 -> InsHandler::processMessageAtlas, sensor_handlers.cpp:165-252)
  // for N independent logs -> InsHandler::processMessageAtlasSegments.  Every segment has its OWN IMUStream de-duplication state
  // (imu_stream.cpp:62-98); atlas_filter = the handler's (true: the new packets go to the device notch cascade with per-filter
  // counts; false: newest packet, raw_dt from the two newest); max_packets = the most packets a message can carry.
  // (SegmentStreamer has the same subscription with the blocks already in HBM.)
  void subscribeKvhBatch(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type, bool atlas_filter, int max_packets,
                         std::function<void(const msgs::kvh_raw_imu_segments_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(
        schema->compile(type, { "utime", "raw_imu.utime", "raw_imu.packet_count", "raw_imu.delta_rotation", "raw_imu.linear_acceleration" }));
    if (!plan->ok()) fprintf(stderr, "SegmentBatcher: %s is not a kvh_raw_imu_batch_t\n", type.c_str());
    const int MP = max_packets < 1 ? 1 : max_packets;
    // Rec: utime; aux = number of new packets; d = new accelerations [3 * n_new] (oldest first, at most MP), delta_rotation [3], raw_dt
    c.decode = [plan, MP, atlas_filter](const pronto_wire::LogEvent &ev, Rec &r, Stream &st) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->run(ev.data.data(), ev.data.size(), x, st.shape, nullptr) || x[0].num.size() != 1) return false;
      const size_t np = x[1].num.size();
      if (np < 1 || x[2].num.size() != np || x[3].num.size() != 3 * np || x[4].num.size() != 3 * np) return false;
      r.utime = (int64_t) x[0].num[0];
      r.d.clear();
      double drot[3] = { 0, 0, 0 }, raw_dt = 0;
      int n_new = 0;
      if (!atlas_filter) {   // sensor_handlers.cpp:199-204
        if (np < 2) return false;
        for (int i = 0; i < 3; i++) { r.d.push_back(x[4].num[(size_t) i]); drot[i] = x[3].num[(size_t) i]; }
        raw_dt = (double) ((int64_t) x[1].num[0] - (int64_t) x[1].num[1]) * 1E-6;
        n_new = 1;
      } else {               // IMUStream::convertFromLCMBatch, this segment's own state
        if ((int64_t) x[2].num[0] < st.last_packet) { st.last_packet = -1; st.last_packet_utime = 0; }
        for (size_t i = np; i-- > 0;) {
          const int64_t cnt = (int64_t) x[2].num[i], put = (int64_t) x[1].num[i];
          if (cnt <= st.last_packet) continue;
          if (n_new < MP)
            for (int a = 0; a < 3; a++) r.d.push_back(x[4].num[3 * i + (size_t) a]);
          n_new++;
          for (int a = 0; a < 3; a++) drot[a] = x[3].num[3 * i + (size_t) a];
          raw_dt = (double) (put - st.last_packet_utime) * 1E-6;
          st.last_packet = cnt;
          st.last_packet_utime = put;
        }
        if (n_new > MP) n_new = MP;
      }
      r.aux = n_new;
      r.d.resize((size_t) 3 * n_new);
      r.d.insert(r.d.end(), drot, drot + 3);
      r.d.push_back(raw_dt);
      return true;
    };
    double *acc = pinned<double>((size_t) 3 * MP * B_), *drot = pinned<double>((size_t) 3 * B_), *raw_dt = pinned<double>((size_t) B_);
    int64_t *ut = pinned<int64_t>((size_t) B_);
    int32_t *n_new = pinned<int32_t>((size_t) B_);
    uint8_t *valid = pinned<uint8_t>((size_t) B_);
    std::fill_n(acc, (size_t) 3 * MP * B_, 0.0);
    std::fill_n(drot, (size_t) 3 * B_, 0.0);
    std::fill_n(raw_dt, (size_t) B_, 1.0);
    c.dispatch = [this, cb, MP, acc, drot, raw_dt, ut, n_new, valid](const std::vector<const Rec *> &col, int64_t utime) {
      PB_SHIM_PARALLEL_FOR
      for (int s = 0; s < B_; s++) {
        const Rec *r = col[(size_t) s];
        const int nn = r ? (int) r->aux : 0;
        n_new[s] = nn;
        valid[s] = nn > 0;
        ut[s] = r ? r->utime : 0;
        if (nn <= 0) continue;   // (no message, or no new packet: the filter idles; its arrays keep their last values)
        for (int p = 0; p < nn; p++)
          for (int a = 0; a < 3; a++) acc[((size_t) p * 3 + (size_t) a) * B_ + s] = r->d[(size_t) 3 * p + (size_t) a];
        for (int a = 0; a < 3; a++) drot[(size_t) a * B_ + s] = r->d[(size_t) 3 * nn + (size_t) a];
        raw_dt[s] = r->d[(size_t) 3 * nn + 3];
      }
      msgs::kvh_raw_imu_segments_t m;
      m.utime = utime;
      m.max_new = MP;
      m.n_new = n_new;
      m.valid = valid;
      m.new_accel = acc;
      m.delta_rotation = drot;
      m.raw_dt = raw_dt;
      m.utimes = ut;
      m.mem = PB_HOST;
      timed(stats.t_handler, [&]() { cb(&m); });
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::joint_state_t (utime, joint_name[], joint_position[], joint_velocity[], joint_effort[]): LegOdoHandler::processMessage
  void subscribeJointState(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                           std::function<void(const msgs::joint_state_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(
        schema->compile(type, { "utime", "joint_name", "joint_position", "joint_velocity", "joint_effort" }));
    if (!plan->ok()) fprintf(stderr, "SegmentBatcher: %s is not a joint_state_t\n", type.c_str());
    c.decode = [plan](const pronto_wire::LogEvent &ev, Rec &r, Stream &st) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      bool same = false;
      if (!plan->run(ev.data.data(), ev.data.size(), x, st.shape, &same) || x[0].num.size() != 1) return false;
      if (!same || !st.names) {   // a new joint list (normally: the first message of the segment)
        st.names = std::make_shared<const std::vector<std::string>>(x[1].str);
        uint64_t h = 1469598103934665603ull;   // FNV-1a over the names and their boundaries
        for (const std::string &nm : *st.names) {
          for (unsigned char ch : nm) h = (h ^ ch) * 1099511628211ull;
          h = (h ^ 0xffu) * 1099511628211ull;
        }
        st.names_hash = h;
      }
      const size_t n = st.names->size();
      if (x[2].num.size() != n || x[3].num.size() != n || x[4].num.size() != n) return false;
      r.utime = (int64_t) x[0].num[0];
      r.names = st.names;
      r.names_hash = st.names_hash;
      r.f.resize(3 * n);
      for (size_t j = 0; j < n; j++) {
        r.f[j] = (float) x[2].num[j];
        r.f[n + j] = (float) x[3].num[j];
        r.f[2 * n + j] = (float) x[4].num[j];
      }
      return true;
    };
    auto st = std::make_shared<JointBlocks>();
    c.dispatch = [this, cb, st](const std::vector<const Rec *> &col, int64_t utime) {
      const Rec *lead = nullptr;
      for (const Rec *r : col)
        if (r) { lead = r; break; }
      if (lead == nullptr) return;
      const size_t n = lead->names->size();
      if (st->rows != n) {   // first message (or a new joint list): size the page-locked blocks
        st->rows = n;
        st->jp = pinned<float>(3 * n * (size_t) B_);
        st->ut = pinned<int64_t>((size_t) B_);
        st->valid = pinned<uint8_t>((size_t) B_);
        std::fill_n(st->jp, 3 * n * (size_t) B_, 0.0f);
        void *d = nullptr;
        if (pb_malloc(est_->ctx, sizeof(float) * 3 * n * (size_t) B_, &d) != PB_OK) {
          fprintf(stderr, "SegmentBatcher: %s\n", pb_last_error(est_->ctx));
          exit(1);
        }
        st->d_jp = (float *) d;
        device_.push_back(d);
      }
      float *jp = st->jp, *jv = jp + n * (size_t) B_, *je = jv + n * (size_t) B_;
      int64_t mismatched = 0;
      PB_SHIM_PARALLEL_FOR_SUM(mismatched)
      for (int s = 0; s < B_; s++) {
        const Rec *r = col[(size_t) s];
        // one robot model for the batch: the same joints in the same order (compared by the hash of the name list, made where the
        // list was decoded: 30 string compares per segment and message were a third of this loop)
        const bool ok = r != nullptr && r->names_hash == lead->names_hash && r->names->size() == n;
        if (r != nullptr && !ok) mismatched++;
        st->valid[s] = ok;
        st->ut[s] = ok ? r->utime : 0;
        if (!ok) continue;                                         // (the block keeps this filter's last message; it is masked)
        for (size_t j = 0; j < n; j++) {
          jp[j * B_ + s] = r->f[j];
          jv[j * B_ + s] = r->f[n + j];
          je[j * B_ + s] = r->f[2 * n + j];
        }
      }
      stats.undecodable += mismatched;
      // The joint block goes to HBM here (one DMA from page-locked memory) and the handler gets a DEVICE message: with the IMU
      // block coming from the host, the IMU + joint-state pair then runs as ONE kernel (pb_step_legodo_joints takes at most one
      // of its two input groups from the host).  The previous message's pair kernel has been enqueued by now and the copy is
      // stream-ordered behind it, so one device block is enough.  (The call takes ~300 us at 4096 segments where the DMA of a busy
      // device takes 25: a copy on a second stream into alternating blocks, waiting for nothing but itself, took the same -- the
      // device idles between ticks in this host-bound replay and every burst pays its wake-up.)
      int urc = PB_OK;
      timed(stats.t_upload, [&]() { urc = pb_memcpy_h2d(est_->ctx, st->d_jp, jp, sizeof(float) * 3 * n * (size_t) B_); });
      if (urc != PB_OK) {
        fprintf(stderr, "SegmentBatcher: %s\n", pb_last_error(est_->ctx));
        return;
      }
      msgs::joint_state_t m;
      m.utime = utime;
      m.joint_name = *lead->names;
      m.joint_position = st->d_jp;
      m.joint_velocity = st->d_jp + n * (size_t) B_;
      m.joint_effort = st->d_jp + 2 * n * (size_t) B_;
      m.mem = PB_DEVICE;
      m.utimes = st->ut;
      m.valid = st->valid;
      timed(stats.t_handler, [&]() { cb(&m); });
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::six_axis_force_torque_array_t (sensors[].force[3]; sensors 0 / 1 = left / right foot): LegOdoHandler::forceTorqueHandler
  void subscribeForceTorque(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                            std::function<void(const msgs::six_axis_force_torque_array_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "sensors.force" }));
    if (!plan->ok()) fprintf(stderr, "SegmentBatcher: %s is not a six_axis_force_torque_array_t\n", type.c_str());
    c.decode = [plan](const pronto_wire::LogEvent &ev, Rec &r, Stream &st) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->run(ev.data.data(), ev.data.size(), x, st.shape, nullptr) || x[0].num.size() != 1 || x[1].num.size() < 6) return false;
      r.utime = (int64_t) x[0].num[0];
      r.d = { x[1].num[2], x[1].num[5] };   // sensors[0].force[2], sensors[1].force[2]
      return true;
    };
    double *fz = pinned<double>((size_t) 2 * B_);
    std::fill_n(fz, (size_t) 2 * B_, 0.0);
    c.dispatch = [this, cb, fz](const std::vector<const Rec *> &col, int64_t utime) {
      PB_SHIM_PARALLEL_FOR
      for (int s = 0; s < B_; s++)
        if (col[(size_t) s]) {   // (the handler keeps the LAST force/torque message, per filter: a missing one changes nothing)
          fz[s] = col[(size_t) s]->d[0];
          fz[(size_t) B_ + s] = col[(size_t) s]->d[1];
        }
      msgs::six_axis_force_torque_array_t m{ utime, BatchArray(fz, PB_HOST) };
      timed(stats.t_handler, [&]() { cb(&m); });
    };
    chans_[channel] = std::move(c);
  }
  // pronto::update_t (pronto_wire.hpp): FovisHandler::processMessage.  timestamp / prev_timestamp are the lead's.
  void subscribeUpdate(const std::string &channel, std::function<void(const msgs::update_t *)> cb)
  {
    Chan c;
    c.decode = [](const pronto_wire::LogEvent &ev, Rec &r, Stream &) {
      pronto_wire::update_t w;
      if (w.decode(ev.data.data(), ev.data.size()) < 0) return false;
      r.utime = w.timestamp;
      r.aux = w.prev_timestamp;
      r.flag = w.estimate_status == pronto_wire::update_t::ESTIMATE_VALID;
      r.d.assign(w.translation, w.translation + 3);
      r.d.insert(r.d.end(), w.rotation, w.rotation + 4);
      return true;
    };
    double *blk = pinned<double>((size_t) 7 * B_);
    uint8_t *valid = pinned<uint8_t>((size_t) B_);
    c.dispatch = [this, cb, blk, valid](const std::vector<const Rec *> &col, int64_t utime) {
      const Rec *lead = nullptr;
      for (int s = 0; s < B_; s++) {
        const Rec *r = col[(size_t) s];
        if (r && !lead) lead = r;
        valid[s] = r != nullptr && r->flag;
        for (int i = 0; i < 7; i++) blk[(size_t) i * B_ + s] = r ? r->d[(size_t) i] : (i == 3 ? 1.0 : 0.0);
      }
      if (lead == nullptr) return;
      // (prev_timestamp on the batch's time base, like timestamp)
      msgs::update_t m{ utime, lead->aux + (utime - lead->utime), valid, BatchArray(blk, PB_HOST), BatchArray(blk + (size_t) 3 * B_, PB_HOST) };
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }
  // bot_core::pose_t (utime, pos[3], vel[3], orientation[4]): ScanMatcherHandler::processMessage
  void subscribePose(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                     std::function<void(const msgs::pose_t *)> cb)
  {
    Chan c;
    auto plan = std::make_shared<pronto_wire::Schema::Plan>(schema->compile(type, { "utime", "pos", "vel", "orientation" }));
    if (!plan->ok()) fprintf(stderr, "SegmentBatcher: %s is not a pose_t\n", type.c_str());
    c.decode = [plan](const pronto_wire::LogEvent &ev, Rec &r, Stream &st) {
      static thread_local std::vector<pronto_wire::Schema::Extracted> x;
      if (!plan->run(ev.data.data(), ev.data.size(), x, st.shape, nullptr) || x[0].num.size() != 1 || x[1].num.size() != 3 || x[2].num.size() != 3 ||
          x[3].num.size() != 4)
        return false;
      r.utime = (int64_t) x[0].num[0];
      r.d.assign(x[1].num.begin(), x[1].num.end());
      r.d.insert(r.d.end(), x[2].num.begin(), x[2].num.end());
      r.d.insert(r.d.end(), x[3].num.begin(), x[3].num.end());
      return true;
    };
    double *blk = pinned<double>((size_t) 10 * B_);
    uint8_t *valid = pinned<uint8_t>((size_t) B_);
    c.dispatch = [this, cb, blk, valid](const std::vector<const Rec *> &col, int64_t utime) {
      PB_SHIM_PARALLEL_FOR
      for (int s = 0; s < B_; s++) {
        const Rec *r = col[(size_t) s];
        valid[s] = r != nullptr;
        for (int i = 0; i < 10; i++) blk[(size_t) i * B_ + s] = r ? r->d[(size_t) i] : (i == 6 ? 1.0 : 0.0);
      }
      msgs::pose_t m{ utime, BatchArray(blk, PB_HOST), BatchArray(blk + (size_t) 3 * B_, PB_HOST), BatchArray(blk + (size_t) 6 * B_, PB_HOST) };
      m.valid = valid;
      cb(&m);
    };
    chans_[channel] = std::move(c);
  }

  // Replays every segment to its end.  Returns the number of batched messages dispatched, or -1 when no segment was added.
  int64_t run()
  {
    if (segs_.empty()) return -1;
    // channels by index from here on: the per-event work is one name look-up, then deques of ints
    chan_list_.clear();
    chan_names_.clear();
    for (auto &kv : chans_) {
      kv.second.id = (int) chan_list_.size();
      chan_list_.push_back(&kv.second);
      chan_names_.push_back(kv.first);
    }
    for (auto &sg : segs_)
      if (sg->queue.size() != chan_list_.size()) {
        sg->queue.resize(chan_list_.size());
        sg->stream.resize(chan_list_.size());
      }
    std::vector<const Rec *> col((size_t) B_, nullptr);
    std::vector<Rec> held((size_t) B_);
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto lap = [&now](std::chrono::steady_clock::time_point &t, double &acc) {
      const auto t1 = now();
      acc += std::chrono::duration<double>(t1 - t).count();
      t = t1;
    };
    for (;;) {
      auto tick = now();
      // the lead: the first segment that still has a subscribed event; its next one names the channel of this batched message
      int lead = -1, channel = -1;
      for (int s = first_alive_; s < (int) segs_.size() && lead < 0; s++) {
        if (fill(*segs_[(size_t) s]) && !segs_[(size_t) s]->order.empty()) {
          lead = s;
          channel = segs_[(size_t) s]->order.front();
        } else if (s == first_alive_) {
          first_alive_++;   // (ended for good: not asked again)
        }
      }
      if (lead < 0) break;
      Chan &ch = *chan_list_[(size_t) channel];
      int64_t lead_rel = 0;
      std::fill(col.begin(), col.end(), nullptr);
      const int nseg = (int) segs_.size();
      lap(tick, stats.t_lead);
      // every segment reads and decodes its own log: independent work, one host thread per range of segments (-fopenmp)
      PB_SHIM_PARALLEL_FOR
      for (int s = lead; s < nseg; s++) got_[(size_t) s] = pull(*segs_[(size_t) s], channel, held[(size_t) s]);
      lap(tick, stats.t_pull);
      for (int s = lead; s < nseg; s++) {
        Seg &sg = *segs_[(size_t) s];
        if (!got_[(size_t) s]) {
          stats.ragged++;
          continue;
        }
        col[(size_t) s] = &held[(size_t) s];
        if (sg.t0 == INT64_MIN) sg.t0 = held[(size_t) s].utime;
        const int64_t rel = held[(size_t) s].utime - sg.t0;
        if (s == lead) {
          lead_rel = rel;
          if (base_ == INT64_MIN) base_ = sg.t0;   // the batch's time base: the first lead's first message
        } else {
          stats.max_skew_us = std::max<int64_t>(stats.max_skew_us, std::llabs(rel - lead_rel));
        }
        stats.segment_messages++;
      }
      stats.ragged += lead;   // (the segments in front of the lead have ended)
      lap(tick, stats.t_book);
      // batch-level time of this message: the lead's, relative to ITS first message, on the batch's base -- it stays continuous when
      // the lead changes (segments come from different recordings: their absolute times have nothing to do with each other)
      ch.dispatch(col, base_ + lead_rel);
      stats.batches++;
      stats.per_channel[chan_names_[(size_t) channel]]++;
      lap(tick, stats.t_dispatch);
      // segments whose log has no subscribed event left: their run is complete, keep its result.  (fill() is also the read-ahead
      // of the next message: in parallel over the segments)
      PB_SHIM_PARALLEL_FOR
      for (int s = 0; s < nseg; s++) got_[(size_t) s] = finished_[(size_t) s] || fill(*segs_[(size_t) s]);
      lap(tick, stats.t_fill);
      int first = -1;
      for (int s = 0; s <= nseg; s++) {
        const bool ended = s < nseg && !finished_[(size_t) s] && !got_[(size_t) s];
        if (ended && first < 0) first = s;
        if (!ended && first >= 0) {
          finalize(first, s - first);
          first = -1;
        }
      }
      lap(tick, stats.t_final);
    }
    for (const auto &sg : segs_) {
      stats.undecodable += sg->undecodable;
      stats.order_violations += sg->order_violations;
      stats.readahead_capped += sg->capped;
      sg->undecodable = sg->order_violations = sg->capped = 0;
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
  int64_t finalUtime(int s) const { return final_utime_[(size_t) s]; }   // batch-level time at which segment s ended

private:
  // one decoded message, in the compact form its channel's assembler reads
  struct Rec {
    int64_t utime = 0, aux = 0;
    bool flag = false;
    std::vector<double> d;
    std::vector<float> f;
    std::shared_ptr<const std::vector<std::string>> names;   // (shared by the messages of one stream: a joint list is decoded once)
    uint64_t names_hash = 0;
  };
  // what one segment remembers about one channel between messages: the byte layout of its last message (Schema::Plan::Shape: a
  // recorded stream repeats it, and the wanted numbers are then read by offset) and the strings that came with it
  struct Stream {
    pronto_wire::Schema::Plan::Shape shape;
    std::shared_ptr<const std::vector<std::string>> names;
    uint64_t names_hash = 0;
    int64_t last_packet = -1, last_packet_utime = 0;   // IMUStream (imu_stream.hpp:10-37), one per segment
  };
  struct Chan {
    std::function<bool(const pronto_wire::LogEvent &, Rec &, Stream &)> decode;
    std::function<void(const std::vector<const Rec *> &, int64_t)> dispatch;
    int id = -1;
  };
  struct JointBlocks {
    size_t rows = 0;
    float *jp = nullptr, *d_jp = nullptr;   // page-locked assembly block and its copy in HBM
    int64_t *ut = nullptr;
    uint8_t *valid = nullptr;
  };
  struct Seg {
    pronto_wire::LogReader rd;
    int64_t start_timestamp, t0 = INT64_MIN;
    bool eof = false;
    std::deque<int> order;                              // channels (index) of the decoded, not yet consumed events, in file order
    std::vector<std::deque<Rec>> queue;                 // ... and the events themselves, per channel
    std::vector<Stream> stream;
    int64_t undecodable = 0, order_violations = 0, capped = 0;      // (per segment: the segments are read by different host threads)
    Seg(const std::string &path, int64_t start) : rd(path), start_timestamp(start) {}
  };

  // reads ahead in the segment's log until one subscribed event is waiting (false: the log has ended and nothing is waiting)
  bool fill(Seg &sg)
  {
    while (sg.order.empty() && !sg.eof) read_one(sg);
    return !sg.order.empty();
  }
  void read_one(Seg &sg)
  {
    static thread_local pronto_wire::LogEvent ev;       // (its buffers are re-used from event to event)
    if (!sg.rd.next(ev)) {
      sg.eof = true;
      return;
    }
    if (ev.timestamp < sg.start_timestamp) return;      // "?start_timestamp=": lcm_front_end.cpp:21-33
    auto it = chans_.find(ev.channel);
    if (it == chans_.end()) return;
    const int id = it->second.id;
    Rec r;
    if (!it->second.decode(ev, r, sg.stream[(size_t) id])) {
      sg.undecodable++;
      return;
    }
    sg.order.push_back(id);
    sg.queue[(size_t) id].push_back(std::move(r));
  }
  // the segment's next message on `channel` (reading ahead past other channels' events if need be); false: none left
  bool pull(Seg &sg, int channel, Rec &out)
  {
    std::deque<Rec> &q = sg.queue[(size_t) channel];
    // (bounded: a live segment whose channel has stopped while its other channels go on would otherwise decode the whole rest of
    // its log into its queues in one tick -- the channel then simply has no message for this tick; SegmentStreamer keeps offsets
    // instead of decoded events and has the same bound)
    size_t ahead = 0;
    while (q.empty() && !sg.eof && ahead < readahead_cap) { read_one(sg); ahead++; }
    if (q.empty()) {
      if (!sg.eof) sg.capped++;
      return false;
    }
    out = std::move(q.front());
    q.pop_front();
    auto it = std::find(sg.order.begin(), sg.order.end(), channel);
    if (it != sg.order.begin()) sg.order_violations++;      // this segment's own log had something else first
    sg.order.erase(it);
    return true;
  }
  // the runs of segments [first, first + count) are complete: read their heads (applies whatever the estimator holds back first)
  void finalize(int first, int count)
  {
    const int n = est_->n;
    est_->flushPending();
    std::vector<double> v((size_t) n * count), q((size_t) 4 * count), c((size_t) n * n * count), l((size_t) count);
    if (pb_get_head(est_->ctx, first, count, v.data(), q.data(), c.data(), l.data(), PB_HOST) != PB_OK) {
      fprintf(stderr, "SegmentBatcher: %s\n", pb_last_error(est_->ctx));
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
  template <class T>
  T *pinned(size_t n)
  {
    void *p = nullptr;
    if (pb_host_alloc(est_->ctx, sizeof(T) * (n ? n : 1), &p) != PB_OK) {
      fprintf(stderr, "SegmentBatcher: %s\n", pb_last_error(est_->ctx));
      exit(1);
    }
    pinned_.push_back(p);
    return (T *) p;
  }

  MavStateEstimator *est_;
  int B_;
  int64_t base_ = INT64_MIN;
  std::vector<double> final_vec_, final_quat_, final_cov_, final_ll_;
  std::vector<int64_t> final_utime_;
  std::vector<uint8_t> finished_;
  std::vector<uint8_t> got_ = std::vector<uint8_t>((size_t) B_, 0);
  std::vector<std::unique_ptr<Seg>> segs_;
  std::map<std::string, Chan> chans_;
  std::vector<Chan *> chan_list_;
  std::vector<std::string> chan_names_;
  int first_alive_ = 0;
  std::vector<void *> pinned_, device_;
};

}  // namespace MavStateEst
