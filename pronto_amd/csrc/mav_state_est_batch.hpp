// mav_state_est_batch.hpp -- host-side C++ mirror of Pronto's plugin / update-object API for B filters at once,
// header-only on top of the C ABI (include/pronto_batch.h).  Same namespace, class and method names, argument
// meaning and error conventions as the reference for THIS path, so handler code reads like the reference's:
//
//   RBISUpdateInterface + RBISResetUpdate / RBISIMUProcessStep / RBISIndexedMeasurement /
//   RBISIndexedPlusOrientationMeasurement          state-estimator/src/mav_state_est/rbis_update_interface.hpp:8-120
//   MavStateEstimator::addUpdate / getHeadState / getMeasurementsLogLikelihood   mav_state_est.hpp:10-25, .cpp:28-96
//   InsHandler, ScanMatcherHandler, IndexedMeasurementHandler, GpsHandler        sensor_handlers.hpp:22-167
//   LegOdoCommon (+ a LegOdoHandler fed by leg-odometry deltas)                  mav_est_legodo/rbis_legodo_common.hpp
//   FovisHandler                                                                 mav_est_fovis/rbis_fovis_update.cpp:6-312
//   SensorHandler<Msg,Handler> dispatch (downsample gate, utime_offset, roll_forward)   lcm_front_end.hpp:139-203
//
// What changes, and why: every message type carries B filters' worth of data (SoA, filter index fastest, host or
// device memory); RBIS/RBIM are batched host containers; updateFilter() applies itself to the estimator's
// device-resident posterior instead of taking prior/posterior by value (the posterior of 65 536 filters is 73 MB --
// it stays in HBM).  BotParam is a key/value map with the reference's key names; *_or_fail keep libbot's behaviour
// (message on stderr, exit(1)).  LCM, BotFrames and the leg kinematics are outside this path.
//
// Throughput note: a handler that transforms inputs on the host (frame rotation in InsHandler, delta->velocity in
// LegOdoCommon) costs O(B) host work per message; bulk replays pre-stage body-frame blocks in HBM and use
// pb_run_legodo (see bench.py).  The handlers skip the host pass entirely when the transform is the identity.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pronto_batch.h"
#include "lcm_schema.hpp"  // includes pronto_wire.hpp

// The handlers' per-filter host loops (frame rotation, velocity from deltas ...) are independent per filter: compile the
// host program with -fopenmp and they run on all cores; without it the pragma disappears.
#ifdef _OPENMP
#include <omp.h>
// At most PRONTO_SHIM_THREADS host threads (default 16: the CPU share a one-GPU job gets on a shared box, whatever the machine's
// core count says -- a team of 128 spinning threads on a 16-CPU quota is slower than one thread by orders of magnitude).
static inline int pb_shim_threads()
{
  static const int n = [] {
    const char *e = getenv("PRONTO_SHIM_THREADS");
    const int want = e ? atoi(e) : 16;
    return std::max(1, std::min(want, omp_get_max_threads()));
  }();
  return n;
}
#define PB_SHIM_PARALLEL_FOR _Pragma("omp parallel for schedule(static) num_threads(pb_shim_threads())")
#define PB_SHIM_PRAGMA(x) _Pragma(#x)
#define PB_SHIM_PARALLEL_FOR_SUM(v) PB_SHIM_PRAGMA(omp parallel for schedule(static) num_threads(pb_shim_threads()) reduction(+ : v))
#else
#define PB_SHIM_PARALLEL_FOR
#define PB_SHIM_PARALLEL_FOR_SUM(v)
#endif

namespace MavStateEst {

// ---------------------------------------------------------------------------------------------------------------
// libbot stand-ins (bot_param, bot_core math) -- same names and semantics as the calls in the reference
// ---------------------------------------------------------------------------------------------------------------
struct BotParam {
  std::map<std::string, std::string> kv;
  void set(const std::string &k, const std::string &v) { kv[k] = v; }
  void set(const std::string &k, double v)
  {
    char b[64];
    snprintf(b, sizeof b, "%.17g", v);
    kv[k] = b;
  }
  // "-O key=value|key=value" overrides (fusion.cpp:98-99, lcm_front_end.cpp:62-68)
  void applyOverrides(const std::string &s)
  {
    size_t p = 0;
    while (p < s.size()) {
      size_t e = s.find('|', p);
      if (e == std::string::npos) e = s.size();
      std::string kvp = s.substr(p, e - p);
      size_t eq = kvp.find('=');
      if (eq != std::string::npos) kv[kvp.substr(0, eq)] = kvp.substr(eq + 1);
      p = e + 1;
    }
  }
};

inline const std::string &bot_param_get_raw_or_fail(BotParam *p, const char *key)
{
  auto it = p->kv.find(key);
  if (it == p->kv.end()) {
    fprintf(stderr, "ERROR: BotParam: could not get param value for key '%s'\n", key);
    exit(1);  // libbot bot_param_get_*_or_fail behaviour
  }
  return it->second;
}
inline double bot_param_get_double_or_fail(BotParam *p, const char *key) { return atof(bot_param_get_raw_or_fail(p, key).c_str()); }
inline int64_t bot_param_get_int_or_fail(BotParam *p, const char *key) { return atoll(bot_param_get_raw_or_fail(p, key).c_str()); }
inline bool bot_param_get_boolean_or_fail(BotParam *p, const char *key)
{
  const std::string &v = bot_param_get_raw_or_fail(p, key);
  return v == "true" || v == "1" || v == "True";
}
inline std::string bot_param_get_str_or_fail(BotParam *p, const char *key) { return bot_param_get_raw_or_fail(p, key); }
// "x, y, z" or "[x, y, z]": libbot's bot_param_get_double_array_or_fail
inline void bot_param_get_double_array_or_fail(BotParam *p, const char *key, double *out, int len)
{
  const std::string &v = bot_param_get_raw_or_fail(p, key);
  const char *c = v.c_str();
  for (int i = 0; i < len; i++) {
    while (*c == ' ' || *c == '[' || *c == ',') c++;
    char *end = nullptr;
    out[i] = strtod(c, &end);
    if (end == c) {
      fprintf(stderr, "ERROR: BotParam: '%s' does not hold %d numbers\n", key, len);
      exit(1);
    }
    c = end;
  }
}
inline double bot_sq(double a) { return a * a; }
inline double bot_to_radians(double d) { return d * (M_PI / 180.0); }

struct BotTrans {
  double rot_quat[4] = { 1, 0, 0, 0 };
  double trans_vec[3] = { 0, 0, 0 };
  bool isIdentityRotation() const { return rot_quat[0] == 1 && rot_quat[1] == 0 && rot_quat[2] == 0 && rot_quat[3] == 0; }
};
// libbot bot_quat_rotate_to (restated in-tree at pronto-utils/src/pronto_complementary/complementary_test.cpp:49-59)
inline void bot_quat_rotate_to(const double rot[4], const double v[3], double r[3])
{
  double ab = rot[0] * rot[1], ac = rot[0] * rot[2], ad = rot[0] * rot[3];
  double nbb = -rot[1] * rot[1], bc = rot[1] * rot[2], bd = rot[1] * rot[3];
  double ncc = -rot[2] * rot[2], cd = rot[2] * rot[3], ndd = -rot[3] * rot[3];
  r[0] = 2 * ((ncc + ndd) * v[0] + (bc - ad) * v[1] + (ac + bd) * v[2]) + v[0];
  r[1] = 2 * ((ad + bc) * v[0] + (nbb + ndd) * v[1] + (cd - ab) * v[2]) + v[1];
  r[2] = 2 * ((bd - ac) * v[0] + (ab + cd) * v[1] + (nbb + ncc) * v[2]) + v[2];
}
inline void bot_trans_apply_vec(const BotTrans *t, const double v[3], double r[3])
{
  bot_quat_rotate_to(t->rot_quat, v, r);
  r[0] += t->trans_vec[0]; r[1] += t->trans_vec[1]; r[2] += t->trans_vec[2];
}

// ---------------------------------------------------------------------------------------------------------------
// batched RBIS / RBIM (rbis.hpp:19-123): host containers, SoA with the filter index fastest
// ---------------------------------------------------------------------------------------------------------------
struct RBIS {
  enum { angular_velocity_ind = 0, velocity_ind = 3, chi_ind = 6, position_ind = 9, acceleration_ind = 12,
         basic_num_states = 15, gyro_bias_ind = 15, accel_bias_ind = 18, rbis_num_states = 21 };
  int n = 0, B = 0;
  int64_t utime = 0;
  std::vector<double> vec;   // [n][B]
  std::vector<double> quat;  // [4][B]  (w,x,y,z)
  RBIS() {}
  RBIS(int n_states, int batch) : n(n_states), B(batch), vec((size_t) n_states * batch, 0.0), quat((size_t) 4 * batch, 0.0)
  {
    for (int b = 0; b < B; b++) quat[b] = 1.0;
  }
  double &operator()(int i, int b) { return vec[(size_t) i * B + b]; }
  double operator()(int i, int b) const { return vec[(size_t) i * B + b]; }
  double &q(int i, int b) { return quat[(size_t) i * B + b]; }
  double q(int i, int b) const { return quat[(size_t) i * B + b]; }
  static std::vector<int> positionInds() { return { 9, 10, 11 }; }
  static std::vector<int> velocityInds() { return { 3, 4, 5 }; }
  static std::vector<int> chiInds() { return { 6, 7, 8 }; }
  static std::vector<int> angularVelocityInds() { return { 0, 1, 2 }; }
};

struct RBIM {
  int n = 0, B = 0;
  std::vector<double> m;  // [n*n][B], column-major per filter (Map<RBIM>, rbis.cpp:300)
  RBIM() {}
  RBIM(int n_states, int batch) : n(n_states), B(batch), m((size_t) n_states * n_states * batch, 0.0) {}
  double &operator()(int r, int c, int b) { return m[((size_t) c * n + r) * B + b]; }
  double operator()(int r, int c, int b) const { return m[((size_t) c * n + r) * B + b]; }
};

// a batched array handed over by a message: pointer + where it lives
struct BatchArray {
  const double *p = nullptr;
  int mem = PB_HOST;
  BatchArray() {}
  BatchArray(const double *ptr, int m) : p(ptr), mem(m) {}
};

class MavStateEstimator;

// Device memory that an update object OWNS (the reference's update objects own their measurement; here a measurement that
// was formed on the device -- FovisHandler's T1 = T0 * delta -- must stay what it was when the history re-applies the
// update after a late arrival).  Blocks of one size are recycled through a pool shared by the handler and its updates;
// `alive` is the estimator's lifetime token: device memory is only released while the context still exists.
struct DevicePool {
  pb_ctx *ctx;
  std::shared_ptr<bool> alive;
  size_t bytes;
  std::vector<void *> free_;
  DevicePool(pb_ctx *c, std::shared_ptr<bool> a, size_t b) : ctx(c), alive(std::move(a)), bytes(b) {}
  ~DevicePool()
  {
    if (alive && *alive)
      for (void *p : free_) pb_free(ctx, p);
  }
  // One hipMalloc per NEW block, on purpose: while the history window is still filling (the first utime_history_span of a run, ~500
  // blocks for the reference's 1 s) a few-MB hipMalloc measures ~10 us and hides behind the step it feeds, whereas slabs of many
  // blocks stall the message that asks for them (a 1 GiB hipMalloc: up to 30 ms, profiles/r04_shim_sweep.txt).
  void *get(bool &fresh)
  {
    fresh = free_.empty();
    void *p = nullptr;
    if (!fresh) {
      p = free_.back();
      free_.pop_back();
    } else if (pb_malloc(ctx, bytes, &p) != PB_OK) {
      p = nullptr;
    }
    return p;
  }
};
struct DeviceBlock {
  std::shared_ptr<DevicePool> pool;
  void *p;
  DeviceBlock(std::shared_ptr<DevicePool> pl, void *ptr) : pool(std::move(pl)), p(ptr) {}
  ~DeviceBlock() { if (p) pool->free_.push_back(p); }  // stream order makes the reuse safe: every consumer was enqueued before
  DeviceBlock(const DeviceBlock &) = delete;
  DeviceBlock &operator=(const DeviceBlock &) = delete;
};

// ---------------------------------------------------------------------------------------------------------------
// update objects (rbis_update_interface.hpp)
// ---------------------------------------------------------------------------------------------------------------
class RBISUpdateInterface {
public:
  typedef enum {
    ins, gps, vicon, laser, laser_gpf, scan_matcher, optical_flow, reset, invalid, rgbd, fovis, legodo, pose_meas,
    altimeter, airspeed, sideslip, init_message, viewer, yawlock
  } sensor_enum;
  int64_t utime;
  sensor_enum sensor_id;
  RBISUpdateInterface(sensor_enum sensor_id_, int64_t utime_) : utime(utime_), sensor_id(sensor_id_) {}
  virtual ~RBISUpdateInterface() {}
  // Applies this update to the estimator's device-resident posterior (prior = previous posterior,
  // mav_state_est.cpp:55-57).  Returns a pb_status; the posterior/loglikelihood stay on the device.
  virtual int updateFilter(pb_ctx *ctx) = 0;
  // true: this update changes NO filter of the batch -- the message for which the reference's handler returns NULL, so that
  // nothing enters the history (rbis_legodo_update.cpp:242-255).  A batched handler whose per-filter validity is decided on the
  // device cannot know that when it returns; the question is asked lazily (pb_mask_count synchronises), only where the difference
  // is observable: FovisHandler's history.updateMap.lower_bound look-up skips such updates.
  virtual bool appliesToNoFilter(pb_ctx * /*ctx*/, int /*B*/) { return false; }
  static const char *sensor_enum_string(sensor_enum s)
  {
    static const char *names[] = { "ins", "gps", "vicon", "laser", "laser_gpf", "scan_matcher", "optic_flow", "reset",
                                   "invalid", "rgbd", "fovis", "legodo", "pose_meas", "altimeter", "airspeed",
                                   "sideslip", "init_message", "viewer", "yawlock" };
    return names[(int) s];
  }
};

class RBISResetUpdate : public RBISUpdateInterface {
public:
  RBIS reset_state;
  RBIM reset_cov;
  RBISResetUpdate(const RBIS &state, const RBIM &cov, sensor_enum sensor_id_, int64_t utime)
      : RBISUpdateInterface(sensor_id_, utime), reset_state(state), reset_cov(cov) {}
  int updateFilter(pb_ctx *ctx) override
  {
    return pb_reset(ctx, reset_state.vec.data(), reset_state.quat.data(), reset_cov.m.data(), 0, PB_HOST);
  }
};

class RBISIMUProcessStep : public RBISUpdateInterface {
public:
  // gyro xyz | accelerometer xyz | dt as ONE block [7][B] (body frame), owned when built on the host
  std::vector<double> owned;
  std::shared_ptr<DeviceBlock> owned_dev;  // a block made on the device (InsHandler's device path: pb_ins_body_block)
  // [B] DEVICE or NULL: 0 = this filter has NO IMU message in this step (independent log segments; the reference's handler returned
  // NULL for it).  Whoever takes this step hands the mask to the library first (announce): the step is then a no-op for it.
  const uint8_t *valid_dev = nullptr;
  void announce(pb_ctx *ctx) const { if (valid_dev) pb_set_imu_valid(ctx, valid_dev); }
  BatchArray imu_block;
  double q_gyro, q_accel, q_gyro_bias, q_accel_bias;
  RBISIMUProcessStep(BatchArray imu_block_, double q_gyro_, double q_accel_, double q_gyro_bias_, double q_accel_bias_,
                     int64_t utime)
      : RBISUpdateInterface(ins, utime), imu_block(imu_block_), q_gyro(q_gyro_), q_accel(q_accel_),
        q_gyro_bias(q_gyro_bias_), q_accel_bias(q_accel_bias_) {}
  RBISIMUProcessStep(std::vector<double> &&block, double q_gyro_, double q_accel_, double q_gyro_bias_,
                     double q_accel_bias_, int64_t utime, int mem = PB_HOST)  // PB_HOST_BROADCAST: block is [7]
      : RBISUpdateInterface(ins, utime), owned(std::move(block)), imu_block(owned.data(), mem), q_gyro(q_gyro_),
        q_accel(q_accel_), q_gyro_bias(q_gyro_bias_), q_accel_bias(q_accel_bias_) {}
  int updateFilter(pb_ctx *ctx) override
  {
    const double q[4] = { q_gyro, q_accel, q_gyro_bias, q_accel_bias };
    announce(ctx);
    return pb_predict(ctx, imu_block.p, q, imu_block.mem);
  }
};

class RBISIndexedMeasurement : public RBISUpdateInterface {
public:
  std::vector<int> index;
  std::vector<double> owned_z, owned_R;
  std::vector<uint8_t> owned_mask;
  std::shared_ptr<DeviceBlock> owned_dev;  // device-resident z / quat / mask this update owns (FovisHandler)
  BatchArray measurement;           // [m][B]
  const double *measurement_cov;    // per r_kind
  int r_kind, cov_mem;
  const uint8_t *mask = nullptr;    // [B]; 0 = this filter's handler returned NULL (lcm_front_end.hpp:156)
  // A measurement that is still to be MADE on the device from the head state (leg kinematic odometry, LegOdoHandler): either
  // together with the INS step in front of it -- pair_kernel: that step, the odometry slaved to the state after it and this
  // update in ONE kernel (pb_step_legodo_joints / _feet; keep = write the measurement block out for later re-applications) --
  // or on its own right before this update is applied (make_measurement).  Whichever runs first clears both: the odometry
  // advances once, a history replay re-applies the measurement it left in `measurement` / `mask`.
  // make_measurement(ctx, ahead): ahead != NULL = slaved to the state AFTER that (still pending) INS step -- what the estimator
  // calls when it has to hold the pair back or was told not to roll forward, so that the handler's inputs are consumed before
  // control returns to the caller (they are only valid until the next message).
  std::function<int(pb_ctx *, const RBISIMUProcessStep *, bool keep)> pair_kernel;
  std::function<int(pb_ctx *, const RBISIMUProcessStep *ahead)> make_measurement;
  bool deferred() const { return (bool) make_measurement; }
  // run the deferred odometry now (no-op when there is none); afterwards `measurement` / `mask` hold its result
  int resolve(pb_ctx *ctx, const RBISIMUProcessStep *ahead)
  {
    if (!make_measurement) return PB_OK;
    const int rc = make_measurement(ctx, ahead);
    make_measurement = nullptr;
    pair_kernel = nullptr;
    return rc;
  }
  RBISIndexedMeasurement(const std::vector<int> &index_, BatchArray measurement_, const double *measurement_cov_,
                         int r_kind_, const uint8_t *mask_, sensor_enum sensor_id_, int64_t utime)
      : RBISUpdateInterface(sensor_id_, utime), index(index_), measurement(measurement_),
        measurement_cov(measurement_cov_), r_kind(r_kind_), cov_mem(measurement_.mem), mask(mask_) {}
  // host-built measurement: takes ownership of z [m][B], R (per r_kind) and mask
  RBISIndexedMeasurement(const std::vector<int> &index_, std::vector<double> &&z, std::vector<double> &&R, int r_kind_,
                         std::vector<uint8_t> &&mask_, sensor_enum sensor_id_, int64_t utime)
      : RBISUpdateInterface(sensor_id_, utime), index(index_), owned_z(std::move(z)), owned_R(std::move(R)),
        owned_mask(std::move(mask_)), measurement(owned_z.data(), PB_HOST), measurement_cov(owned_R.data()),
        r_kind(r_kind_), cov_mem(PB_HOST), mask(owned_mask.empty() ? nullptr : owned_mask.data()) {}
  int updateFilter(pb_ctx *ctx) override
  {
    const int rc = resolve(ctx, nullptr);
    if (rc != PB_OK) return rc;
    return pb_update_indexed(ctx, (int) index.size(), index.data(), measurement.p, measurement_cov, r_kind, mask,
                             measurement.mem);
  }
  bool appliesToNoFilter(pb_ctx *ctx, int B) override
  {
    if (mask == nullptr || deferred()) return false;  // (a deferred measurement has no mask yet: not applied, not empty)
    if (empty_known_) return empty_;
    int n = 1;
    if (measurement.mem == PB_DEVICE) {
      if (pb_mask_count(ctx, mask, &n) != PB_OK) return false;
    } else {
      n = 0;
      for (int b = 0; b < B; b++) n += mask[b] != 0;
    }
    empty_known_ = true;
    return empty_ = (n == 0);
  }
private:
  bool empty_known_ = false, empty_ = false;
};

class RBISIndexedPlusOrientationMeasurement : public RBISIndexedMeasurement {
public:
  std::vector<double> owned_q;
  BatchArray orientation;  // [4][B]
  RBISIndexedPlusOrientationMeasurement(const std::vector<int> &index_, BatchArray measurement_,
                                        const double *measurement_cov_, int r_kind_, BatchArray orientation_,
                                        const uint8_t *mask_, sensor_enum sensor_id_, int64_t utime)
      : RBISIndexedMeasurement(index_, measurement_, measurement_cov_, r_kind_, mask_, sensor_id_, utime),
        orientation(orientation_) {}
  RBISIndexedPlusOrientationMeasurement(const std::vector<int> &index_, std::vector<double> &&z, std::vector<double> &&R,
                                        int r_kind_, std::vector<double> &&quat, std::vector<uint8_t> &&mask_,
                                        sensor_enum sensor_id_, int64_t utime)
      : RBISIndexedMeasurement(index_, std::move(z), std::move(R), r_kind_, std::move(mask_), sensor_id_, utime),
        owned_q(std::move(quat)), orientation(owned_q.data(), PB_HOST) {}
  int updateFilter(pb_ctx *ctx) override
  {
    return pb_update_indexed_orient(ctx, (int) index.size(), index.data(), measurement.p, measurement_cov, r_kind,
                                    orientation.p, mask, measurement.mem);
  }
};

// An update written against the REFERENCE's contract (rbis_update_interface.hpp:14-35):
//     virtual void updateFilter(const RBIS & prior_state, const RBIM & prior_cov, double prior_loglikelihood) = 0;
// "must fill posterior_state, posterior_covariance, loglikelihood".  This is what a third-party RBISUpdateInterface subclass looks
// like (RBISOpticalFlowMeasurement, rbis_update_interface.hpp:128-154; RBISLaserGPFMeasurement::updateFilter,
// gpf/rbis_gpf_update.cpp:28-76): user arithmetic on ONE filter's state.  It runs here on the SLOW path, documented as such: the
// batch's head comes to the host (pb_get_head), the user's updateFilter is called once per filter with single-filter RBIS / RBIM
// (B = 1 containers: prior_state(i, 0), prior_state.q(i, 0), prior_cov(r, c, 0)), and the posteriors go back (pb_set_head) --
// two PCIe crossings of the whole state per update, 2 x 126 MB at 64k 15-state filters.  The built-in updates never take it.
// apply[b] = false (optional mask, like a handler's NULL return for filter b) leaves filter b's head as it is.
class RBISHostUpdate : public RBISUpdateInterface {
public:
  RBIS posterior_state;          // single-filter containers, filled by the user's updateFilter
  RBIM posterior_covariance;
  double loglikelihood = 0;
  std::vector<uint8_t> apply;    // [B] or empty (= every filter)
  RBISHostUpdate(sensor_enum sensor_id_, int64_t utime_) : RBISUpdateInterface(sensor_id_, utime_) {}
  virtual void updateFilter(const RBIS &prior_state, const RBIM &prior_cov, double prior_loglikelihood) = 0;
  int updateFilter(pb_ctx *ctx) final
  {
    const int n = pb_n_states(ctx), B = pb_batch(ctx);
    std::vector<double> vec((size_t) n * B), quat((size_t) 4 * B), cov((size_t) n * n * B), ll((size_t) B);
    int rc = pb_get_head(ctx, 0, B, vec.data(), quat.data(), cov.data(), ll.data(), PB_HOST);
    if (rc != PB_OK) return rc;
    RBIS prior(n, 1);
    RBIM prior_cov(n, 1);
    for (int b = 0; b < B; b++) {
      if (!apply.empty() && !apply[(size_t) b]) continue;
      for (int i = 0; i < n; i++) prior.vec[(size_t) i] = vec[(size_t) i * B + b];
      for (int i = 0; i < 4; i++) prior.quat[(size_t) i] = quat[(size_t) i * B + b];
      for (int i = 0; i < n * n; i++) prior_cov.m[(size_t) i] = cov[(size_t) i * B + b];
      prior.utime = utime;
      posterior_state = prior;               // (a subclass that forgets a member leaves the prior there, not garbage)
      posterior_covariance = prior_cov;
      loglikelihood = ll[(size_t) b];
      updateFilter(prior, prior_cov, ll[(size_t) b]);
      if (posterior_state.n != n || posterior_state.B != 1 || posterior_covariance.n != n || posterior_covariance.B != 1) return PB_ERR_ARG;
      for (int i = 0; i < n; i++) vec[(size_t) i * B + b] = posterior_state.vec[(size_t) i];
      for (int i = 0; i < 4; i++) quat[(size_t) i * B + b] = posterior_state.quat[(size_t) i];
      for (int i = 0; i < n * n; i++) cov[(size_t) i * B + b] = posterior_covariance.m[(size_t) i];
      ll[(size_t) b] = loglikelihood;
    }
    return pb_set_head(ctx, vec.data(), quat.data(), cov.data(), ll.data(), PB_HOST);
  }
};

// Two updates with complementary per-filter masks that together are ONE update of the reference (each filter takes
// exactly one branch): used where the reference changes the measurement dimension per message (LegOdoCommon's
// pos_and_lin_rate -> lin_rate fallback, rbis_legodo_common.cpp:118-122).  If the first half wrote its posterior into a
// checkpoint slot, the second half works in place on that slot.
class RBISEitherUpdate : public RBISUpdateInterface {
public:
  RBISUpdateInterface *first, *second;
  RBISEitherUpdate(RBISUpdateInterface *a, RBISUpdateInterface *b) : RBISUpdateInterface(a->sensor_id, a->utime), first(a), second(b) {}
  ~RBISEitherUpdate() override { delete first; delete second; }
  int updateFilter(pb_ctx *ctx) override
  {
    int rc = first->updateFilter(ctx);
    if (rc != PB_OK) return rc;
    const int slot = pb_head_slot(ctx);
    if (slot >= 0) pb_set_output_slot(ctx, slot);
    return second->updateFilter(ctx);
  }
  bool appliesToNoFilter(pb_ctx *ctx, int B) override { return first->appliesToNoFilter(ctx, B) && second->appliesToNoFilter(ctx, B); }
};

// ---------------------------------------------------------------------------------------------------------------
// updateHistory + MavStateEstimator (update_history.hpp:12-36, mav_state_est.hpp / .cpp:12-96)
//
// The reference stores every update's posterior inside the update object; a delayed measurement is inserted at
// its timestamp and everything from there on is re-applied (mav_state_est.cpp:28-80).  Here the posterior of the
// whole batch lives on the device, so the history keeps, per update, an optional CHECKPOINT slot
// (pb_state_save) instead; a replay restores the newest checkpoint at or before the insertion point and re-applies
// the updates after it.  checkpoint_every = 1 is the reference's "posterior per update"; larger values trade HBM
// (73 MB per slot for 64k 15-state filters) for longer replays.  With history_slots = 0 the estimator is in-order
// only: an update older than the head is discarded like one older than the history (update_history.cpp:28-39).
// Updates in the history own their host payloads; device payloads they point to must outlive the window.
// ---------------------------------------------------------------------------------------------------------------
class updateHistory {
public:
  typedef std::multimap<int64_t, RBISUpdateInterface *> historyMap;
  typedef historyMap::iterator historyMapIterator;
  typedef std::pair<int64_t, RBISUpdateInterface *> historyPair;
  historyMap updateMap;
  ~updateHistory()
  {
    for (auto &kv : updateMap) delete kv.second;  // update_history.cpp:9-14
  }
};

class MavStateEstimator {
public:
  int64_t utime_history_span;
  pb_ctx *ctx = nullptr;
  std::shared_ptr<bool> ctx_alive = std::make_shared<bool>(false);  // lifetime token for device memory owned elsewhere
  int n = 0, B = 0;
  int64_t head_utime = 0;
  int last_status = PB_OK;
  updateHistory history;
  updateHistory::historyMapIterator unprocessed_updates_start;
  // checkpoint bookkeeping (this build's addition; keys state_estimator.history_slots / history_checkpoint_every)
  int history_slots = 0, checkpoint_every = 1, since_checkpoint = 0;
  std::map<RBISUpdateInterface *, int> checkpoint_of;
  std::vector<int> free_slots;
  RBISUpdateInterface *device_head = nullptr;  // the update whose posterior the device currently holds
  int64_t replayed_updates = 0;                // statistics: updates re-applied because of late arrivals
  int64_t dropped_updates = 0;                 // updates discarded as too old (update_history.cpp:28-39)
  bool derived_history_ = false;               // history_slots / checkpoint cadence were derived from utime_history_span
  // state_estimator.fuse_ins_legodo = true (this build's addition, off by default; works with and without checkpoints): an INS process step is held back
  // until the next update arrives; if that is a velocity measurement on {3,4,5} with a diagonal R (LegOdoCommon's
  // lin_rate) both run as ONE fused kernel (pb_step_legodo: one state round trip instead of two -- 21.7 us instead of
  // 20.2 + 23.6 us at 64k filters).  Only the posterior after the pair exists then, so this is for replays nobody
  // observes between the two messages; with posterior checkpoints (history_slots > 0) the pair is checkpointed as one
  // update behind its second half (the backward smoother needs every INS posterior and refuses to run with it).
  // Anything that reads the device (getHeadState, FovisHandler) flushes the held step first.
  bool fuse_ins_legodo = false;
  int64_t fused_pairs = 0;
  // state_estimator.fuse_corrections = true (with fuse_ins_legodo): the fused pair is held back one more message; if
  // that is a FovisHandler position_orient (idx 9,10,11,6,7,8) or ScanMatcherHandler position_yaw (idx 9,10,11,8)
  // measurement with a diagonal R, all THREE updates run as one kernel and one state round trip
  // (pb_step_legodo_correct; reference seam: rbis_fovis_update.cpp:299-305, sensor_handlers.cpp:709-722).
  bool fuse_corrections = false;
  int64_t fused_triples = 0;
  int64_t leg_kernel_pairs = 0;  // fused pairs whose leg odometry ran inside the step kernel (pb_step_legodo_joints / _feet)

  MavStateEstimator(RBISResetUpdate *init_state, BotParam *param, int device = 0, int n_snapshots = 2)
  {
    utime_history_span = bot_param_get_int_or_fail(param, "state_estimator.utime_history_span");
    auto opt = [&](const char *k, int dflt) {
      auto it = param->kv.find(k);
      return it == param->kv.end() ? dflt : atoi(it->second.c_str());
    };
    // The reference re-orders and re-applies ANY update inside utime_history_span (update_history.cpp:16-42,
    // mav_state_est.cpp:28-80).  Here that needs posterior checkpoints on the device, so a configuration that only sets
    // utime_history_span (every reference .cfg) gets a default pool: 32 slots, spaced so that they cover the span at an
    // assumed two updates per millisecond (1 kHz IMU + leg odometry).  state_estimator.history_slots = 0 asks explicitly
    // for the in-order-only estimator (no checkpoints; an update older than the head is counted in dropped_updates and
    // discarded) -- the throughput configuration of the benchmarks.
    if (param->kv.find("state_estimator.history_slots") == param->kv.end() && utime_history_span > 0) {
      const int64_t expected = std::min<int64_t>(utime_history_span / 500 + 1, 1 << 20);
      history_slots = (int) std::min<int64_t>(32, expected + 2);
      checkpoint_every = (int) std::max<int64_t>(1, (expected + history_slots - 3) / std::max(1, history_slots - 2));
      derived_history_ = true;
    } else {
      history_slots = opt("state_estimator.history_slots", 0);
      checkpoint_every = 1;
    }
    checkpoint_every = opt("state_estimator.history_checkpoint_every", checkpoint_every);
    if (checkpoint_every < 1) checkpoint_every = 1;
    {
      auto it = param->kv.find("state_estimator.fuse_ins_legodo");
      fuse_ins_legodo = it != param->kv.end() && (it->second == "true" || it->second == "1");
      // (with posterior checkpoints a fused PAIR is checkpointed as one update, behind its second half; the three-message
      // fusion is for the in-order-only estimator)
      it = param->kv.find("state_estimator.fuse_corrections");
      fuse_corrections = fuse_ins_legodo && history_slots == 0 && it != param->kv.end() && (it->second == "true" || it->second == "1");
    }
    n = init_state->reset_state.n;
    B = init_state->reset_state.B;
    int rc = pb_create(&ctx, n, B, device, n_snapshots);
    if (rc == PB_OK && history_slots > 0) rc = pb_history_reserve(ctx, history_slots);
    if (rc != PB_OK) {
      fprintf(stderr, "MavStateEstimator: %s\n", pb_last_error(rc == PB_OK ? nullptr : ctx));
      exit(1);  // the reference's constructor cannot fail softly either (bot_param_get_int_or_fail)
    }
    *ctx_alive = true;
    for (int i = history_slots - 1; i >= 0; i--) free_slots.push_back(i);
    last_status = init_state->updateFilter(ctx);  // "apply update from zero... should reset the state" (:16)
    head_utime = init_state->utime;
    pb_set_utime(ctx, head_utime);
    history.updateMap.insert(updateHistory::historyPair(init_state->utime, init_state));  // update_history.cpp:5-8
    device_head = init_state;
    if (history_slots > 0) {
      save_checkpoint(init_state);
      // A replay never re-applies the first element of the history (it starts from that element's checkpoint), so the [n][B] and
      // [n][n][B] host arrays of the initial reset are given back here: freeing them when the window first slides past the
      // initial state (126 MB at 64k x 15 states, 243 MB at 21) stalls that one message for 16-41 ms.
      if (checkpoint_of.count(init_state)) {
        init_state->reset_state = RBIS();
        init_state->reset_cov = RBIM();
      }
    }
    unprocessed_updates_start = history.updateMap.end();
  }
  ~MavStateEstimator()
  {
    // updates may own device memory of this context: release them first, then the context
    for (auto &kv : history.updateMap) delete kv.second;
    history.updateMap.clear();
    *ctx_alive = false;
    pb_destroy(ctx);
  }
  MavStateEstimator(const MavStateEstimator &) = delete;
  MavStateEstimator &operator=(const MavStateEstimator &) = delete;

  // Takes ownership of `update` (the reference's history deletes it, update_history.cpp:12,36,52).
  void addUpdate(RBISUpdateInterface *update, bool roll_forward)
  {
    if (update == nullptr) return;
    auto &map = history.updateMap;
    // update_history.cpp:16-42: insert by time (equal keys keep arrival order); too old -> discard
    const int64_t oldest = (history_slots > 0) ? map.begin()->first : head_utime;
    if (update->utime < oldest) {
      fprintf(stderr, "error: update type %s had timestamp %jd, which was before the first in history (%jd)\ndiscarding update!\n",
              RBISUpdateInterface::sensor_enum_string(update->sensor_id), (intmax_t) update->utime, (intmax_t) oldest);
      delete update;
      dropped_updates++;
      return;
    }
    const auto old_start = unprocessed_updates_start;
    auto added_it = map.insert(map.end(), updateHistory::historyPair(update->utime, update));
    if (unprocessed_updates_start == map.end() || added_it->first < unprocessed_updates_start->first)
      unprocessed_updates_start = added_it;                                   // mav_state_est.cpp:33-40
    if (!roll_forward) {
      // a measurement that is still to be made from the handler's inputs is made NOW (those inputs are the caller's and only
      // valid until its next message): slaved to the state after the INS step in front of it when that is the one held back.
      // The update ITSELF stays unapplied (roll_forward = false).
      auto *m = deferredMeasurement(update);
      if (m == nullptr || !m->deferred()) return;
      // a late arrival in front of an update that HAS been applied: the state its odometry reads is the one at its place in the
      // history, so it takes the restore-checkpoint + replay path below like a rolled-forward update
      auto after = std::next(added_it);
      bool next_applied = after != map.end();
      for (auto it = old_start; next_applied && it != map.end(); ++it)
        if (it == after) next_applied = false;
      if (!next_applied) {
        RBISIMUProcessStep *ahead = nullptr;
        if (holding_ == 1 && added_it != map.begin()) {
          auto prev = std::prev(added_it);
          if (prev == unprocessed_updates_start) ahead = dynamic_cast<RBISIMUProcessStep *>(prev->second);
        }
        if (ahead == nullptr) flushPendingBefore(added_it);   // (what is pending IN FRONT of it, never the new element)
        const int rc = m->resolve(ctx, ahead);
        if (rc != PB_OK) last_status = rc;
        return;
      }
    }

    // The prior of the first unprocessed update is the posterior of the update before it (:45-57).  If the device
    // does not hold that posterior (late arrival), restore the newest checkpoint at or before it and replay.
    auto prev_it = unprocessed_updates_start;
    --prev_it;
    auto current_it = unprocessed_updates_start;
    if (prev_it->second != device_head) {
      auto origin = prev_it;
      while (checkpoint_of.find(origin->second) == checkpoint_of.end()) --origin;  // begin() always has one
      last_status = pb_state_restore(ctx, checkpoint_of[origin->second]);
      current_it = origin;
      ++current_it;
      for (auto it = current_it; it != unprocessed_updates_start; ++it) replayed_updates++;
      for (auto it = unprocessed_updates_start; it != map.end(); ++it)
        if (it != added_it) replayed_updates++;
      // checkpoints after the insertion point are stale now
      for (auto it = current_it; it != map.end(); ++it) drop_checkpoint(it->second);
      since_checkpoint = 0;
    }
    int held = 0;
    while (current_it != map.end()) {
      RBISUpdateInterface *u = current_it->second;
      if (fuse_ins_legodo) {
        if (auto *imu = dynamic_cast<RBISIMUProcessStep *>(u)) {
          auto nxt = current_it;
          ++nxt;
          if (nxt == map.end() && !flushing_) {  // newest element: hold it back until the next update shows up
            held = 1;
            break;
          }
          if (nxt != map.end()) {
            int rc = PB_OK;
            if (fuse_corrections && fusible_pair(imu, nxt->second)) {
              auto third = nxt;
              ++third;
              if (third == map.end() && !flushing_) {  // the pair is complete: wait for what follows it
                // (its measurement is made now, slaved to the state after the held INS step: the handler's inputs do not
                // outlive this call)
                if (auto *m2 = deferredMeasurement(nxt->second)) {
                  const int rrc = m2->resolve(ctx, imu);
                  if (rrc != PB_OK) last_status = rrc;
                }
                held = 2;
                break;
              }
              if (third != map.end() && run_fused3(imu, nxt->second, third->second, rc)) {
                if (rc != PB_OK) {
                  last_status = rc;
                  fprintf(stderr, "MavStateEstimator::addUpdate: fused ins+legodo+correction step failed: %s\n", pb_last_error(ctx));
                }
                fused_triples++;
                device_head = third->second;
                head_utime = third->second->utime;
                current_it = ++third;
                continue;
              }
            }
            // a fused pair counts as two updates towards the checkpoint cadence and writes its posterior straight into
            // the slot of its SECOND half (a replay that starts there continues behind the pair)
            int pslot = -1;
            if (history_slots > 0 && fusible_pair(imu, nxt->second) && (since_checkpoint += 2) >= checkpoint_every &&
                (pslot = reserve_slot(nxt->second)) >= 0)
              pb_set_output_slot(ctx, pslot);
            if (run_fused(imu, nxt->second, rc)) {
              if (rc != PB_OK) {
                last_status = rc;
                fprintf(stderr, "MavStateEstimator::addUpdate: fused ins+legodo step failed: %s\n", pb_last_error(ctx));
              }
              fused_pairs++;
              device_head = nxt->second;
              head_utime = nxt->second->utime;
              if (pslot >= 0) {
                pb_set_output_slot(ctx, -1);
                rc = pb_state_save(ctx, pslot);  // a no-op: the step wrote there
                if (rc != PB_OK) last_status = rc;
                checkpoint_of[nxt->second] = pslot;
                since_checkpoint = 0;
              }
              current_it = ++nxt;
              continue;
            }
          }
        }
      }
      // a checkpointed update writes its posterior straight into the checkpoint slot (no copy afterwards)
      int slot = -1;
      if (history_slots > 0 && ++since_checkpoint >= checkpoint_every && (slot = reserve_slot(u)) >= 0)
        pb_set_output_slot(ctx, slot);
      int rc = u->updateFilter(ctx);
      if (rc != PB_OK) {
        last_status = rc;
        fprintf(stderr, "MavStateEstimator::addUpdate: %s update failed: %s\n",
                RBISUpdateInterface::sensor_enum_string(u->sensor_id), pb_last_error(ctx));
      }
      device_head = u;
      head_utime = u->utime;  // posterior_state.utime = update->utime (:60)
      if (slot >= 0) {
        pb_set_output_slot(ctx, -1);
        rc = pb_state_save(ctx, slot);  // a no-op when the update wrote there; a copy for updates that cannot (reset)
        if (rc != PB_OK) last_status = rc;
        checkpoint_of[u] = slot;
        since_checkpoint = 0;
      }
      ++current_it;
    }
    pb_set_utime(ctx, head_utime);
    holding_ = held;
    clearHistoryBeforeUtime(head_utime - utime_history_span);                 // :72-77
    unprocessed_updates_start = held ? current_it : map.end();
  }

  // the INS step fuse_ins_legodo is holding back, when it is the ONLY pending update (nullptr otherwise): a handler whose
  // measurement depends on the head state (leg kinematic odometry) can ask the device for the state "after that step"
  // instead of flushing it, so that the pair still runs as one fused kernel
  RBISIMUProcessStep *pendingImu()
  {
    if (!fuse_ins_legodo || holding_ != 1 || unprocessed_updates_start == history.updateMap.end()) return nullptr;
    auto it = unprocessed_updates_start;
    auto *imu = dynamic_cast<RBISIMUProcessStep *>(it->second);
    if (imu == nullptr || ++it != history.updateMap.end()) return nullptr;
    return imu;
  }
  // apply an INS step that fuse_ins_legodo is holding back (no-op otherwise)
  void flushPending()
  {
    if (!fuse_ins_legodo || unprocessed_updates_start == history.updateMap.end()) return;
    flushing_ = true;
    auto &map = history.updateMap;
    for (auto it = unprocessed_updates_start; it != map.end(); ++it) {
      int rc = it->second->updateFilter(ctx);
      if (rc != PB_OK) last_status = rc;
      device_head = it->second;
      head_utime = it->second->utime;
    }
    flushing_ = false;
    holding_ = 0;
    pb_set_utime(ctx, head_utime);
    unprocessed_updates_start = map.end();
  }

  // the same for the pending updates IN FRONT of `stop` only; `stop` and what follows it stay unprocessed
  void flushPendingBefore(updateHistory::historyMapIterator stop)
  {
    if (!fuse_ins_legodo || unprocessed_updates_start == history.updateMap.end() || unprocessed_updates_start == stop) return;
    flushing_ = true;
    for (auto it = unprocessed_updates_start; it != stop && it != history.updateMap.end(); ++it) {
      int rc = it->second->updateFilter(ctx);
      if (rc != PB_OK) last_status = rc;
      device_head = it->second;
      head_utime = it->second->utime;
    }
    flushing_ = false;
    holding_ = 0;
    pb_set_utime(ctx, head_utime);
    unprocessed_updates_start = stop;
  }

  void getHeadState(RBIS &head_state, RBIM &head_cov)
  {
    flushPending();
    head_state = RBIS(n, B);
    head_cov = RBIM(n, B);
    last_status = pb_get_head(ctx, 0, B, head_state.vec.data(), head_state.quat.data(), head_cov.m.data(), nullptr, PB_HOST);
    head_state.utime = head_utime;
  }
  std::vector<double> getMeasurementsLogLikelihood()
  {
    flushPending();
    std::vector<double> ll(B);
    last_status = pb_get_head(ctx, 0, B, nullptr, nullptr, nullptr, ll.data(), PB_HOST);
    return ll;
  }

  // (position, quaternion) of the posterior of the update at `it` into device snapshot slot `snap_slot`: what the reference
  // reads as lower_it->second->posterior_state (rbis_fovis_update.cpp:196-206) -- every update keeps its posterior there.
  // Here only every checkpoint_every-th update (and never the INS half of a fused pair) has a saved posterior; for the others
  // it is re-derived: the nearest earlier checkpoint into the context's own array, the updates up to `it` re-applied, the
  // snapshot taken, and the head put back.  Costs at most checkpoint_every re-applied updates per call; FovisHandler calls it
  // once per keyframe change.  false: `it` has not been applied yet, or no slot is free to park the head in.
  int64_t rederived_posteriors = 0;
  bool snapshotPosteriorOf(updateHistory::historyMapIterator it, int snap_slot)
  {
    auto &map = history.updateMap;
    flushPending();
    for (auto u = unprocessed_updates_start; u != map.end(); ++u)
      if (u == it) return false;  // (added without roll_forward: no posterior exists yet)
    auto ck = checkpoint_of.find(it->second);
    if (ck != checkpoint_of.end()) return (last_status = pb_snapshot_from_slot(ctx, snap_slot, ck->second)) == PB_OK;
    if (it->second == device_head) return (last_status = pb_snapshot(ctx, snap_slot)) == PB_OK;
    // where the head goes meanwhile: its own checkpoint if it has one, else a spare slot
    int park = -1;
    bool park_is_spare = false;
    auto hk = device_head ? checkpoint_of.find(device_head) : checkpoint_of.end();
    if (hk != checkpoint_of.end()) park = hk->second;
    else if (!free_slots.empty()) {
      park = free_slots.back();
      free_slots.pop_back();
      park_is_spare = true;
      if ((last_status = pb_state_save(ctx, park)) != PB_OK) return false;
    } else {
      return false;
    }
    auto origin = it;
    while (checkpoint_of.find(origin->second) == checkpoint_of.end()) --origin;  // begin() always has one
    int rc = pb_state_restore(ctx, checkpoint_of[origin->second]);
    for (auto u = std::next(origin); rc == PB_OK; ++u) {
      rc = u->second->updateFilter(ctx);
      if (u == it) break;
    }
    if (rc == PB_OK) rc = pb_snapshot(ctx, snap_slot);
    const int rc2 = pb_state_restore(ctx, park);   // the head again (a copy in the context's own array)
    if (park_is_spare) free_slots.push_back(park);
    pb_set_utime(ctx, head_utime);
    rederived_posteriors++;
    if (rc != PB_OK || rc2 != PB_OK) last_status = rc != PB_OK ? rc : rc2;
    return rc == PB_OK && rc2 == PB_OK;
  }

  // EKFSmoothBackwardsPass (mav_state_est.cpp:98-189): walk the history backwards; at every INS update k apply
  // ekfSmoothingStep with  next_pred = posterior of INS_{k+1},  next = smoothed posterior of step k+1 (for the newest
  // step: its last measurement's posterior),  cur = posterior of the last measurement that followed INS_k (or INS_k's
  // own when none did).  The reference reads these posteriors out of its update objects, which keep them by value; here a
  // posterior exists where an update has a CHECKPOINT slot.  With a checkpoint on every update of the window
  // (history_checkpoint_every = 1) the pass only reads; with sparser checkpoints -- the only way a long window of a big batch
  // fits the device: 64k 21-state filters are 135 MB per posterior -- it re-derives what is missing, stretch by stretch from the
  // newest: the updates between two checkpoints are re-applied from the older one into a window of free slots (checkpoint and
  // recompute: history_slots >= window / every + every + 3 instead of one per update; pb_smooth_log is the same idea for device
  // streams).  The smoothed posteriors are bit for bit those of the all-checkpoints pass.
  // The reference overwrites the updates' posteriors with the smoothed ones for later republishing; here
  // on_smoothed(utime of INS_k, slot) is called newest-first with a slot that holds the smoothed posterior until the next call
  // (pb_get_slot reads it; pb_state_restore(slot) + getHeadState too).  Returns the number of steps, -1 on an error.
  int64_t smoother_reapplied_updates = 0;   // statistics: updates re-applied to re-derive posteriors that had no checkpoint
  int EKFSmoothBackwardsPass(double dt, const std::function<void(int64_t, int)> &on_smoothed)
  {
    auto &map = history.updateMap;
    if (fuse_ins_legodo) {
      fprintf(stderr, "EKFSmoothBackwardsPass: needs the posterior of every INS update; run with state_estimator.fuse_ins_legodo = false\n");
      return -1;
    }
    flushPending();
    // time-ordered list of (update, slot or -1)
    std::vector<std::pair<RBISUpdateInterface *, int>> seq;
    for (auto u = map.begin(); u != unprocessed_updates_start; ++u) {
      auto it = checkpoint_of.find(u->second);
      seq.push_back({ u->second, it == checkpoint_of.end() ? -1 : it->second });
    }
    const int N = (int) seq.size();
    if (N == 0 || seq[0].second < 0) {
      fprintf(stderr, "EKFSmoothBackwardsPass: the oldest update of the history has no checkpoint (state_estimator.history_slots = 0?)\n");
      return -1;
    }
    std::vector<int> ins, cks;
    for (int i = 0; i < N; i++) {
      if (seq[(size_t) i].first->sensor_id == RBISUpdateInterface::ins) ins.push_back(i);
      if (seq[(size_t) i].second >= 0) cks.push_back(i);
    }
    if (ins.size() < 2) return 0;
    // the longest run of updates without a checkpoint decides the window
    int maxgap = N - 1 - cks.back();
    for (size_t m = 0; m + 1 < cks.size(); m++) maxgap = std::max(maxgap, cks[m + 1] - cks[m] - 1);
    const bool head_loose = seq.back().second < 0;   // the newest posterior exists only as the device head
    const int need = maxgap + 2 + ((maxgap > 0 || head_loose) ? 1 : 0);
    if ((int) free_slots.size() < need) {
      fprintf(stderr, "EKFSmoothBackwardsPass: needs %d free checkpoint slots (two for the smoothed posteriors%s), %zu are free: raise "
                      "state_estimator.history_slots or lower history_checkpoint_every\n",
              need, maxgap > 0 ? ", the longest run of updates without a checkpoint and one for the head" : "", free_slots.size());
      return -1;
    }
    const size_t nf = free_slots.size();
    const int spare[2] = { free_slots[nf - 1], free_slots[nf - 2] };
    const int head_keep = (maxgap > 0 || head_loose) ? free_slots[nf - 3] : -1;
    auto W = [&](int i) { return free_slots[nf - 4 - (size_t) i]; };   // window slots
    int rc = PB_OK;
    auto bail = [&](const char *what) {
      last_status = rc;
      fprintf(stderr, "EKFSmoothBackwardsPass: %s: %s\n", what, pb_last_error(ctx));
      return -1;
    };
    if (head_keep >= 0) {
      if (device_head != seq.back().first) {
        fprintf(stderr, "EKFSmoothBackwardsPass: the device does not hold the newest posterior (call it right after addUpdate)\n");
        return -1;
      }
      if ((rc = pb_state_save(ctx, head_keep)) != PB_OK) return bail("saving the head");
    }
    int next = head_loose ? head_keep : seq.back().second, steps = 0, toggle = 0;
    int j = (int) ins.size() - 2;   // the step being smoothed: needs the posteriors of updates ins[j+1] - 1 and ins[j+1]
    // stretches (a, e]: a = a checkpointed update, e = the next checkpointed update (or the newest update)
    for (int m = (int) cks.size() - 1; m >= 0 && j >= 0; m--) {
      const int a = cks[(size_t) m], e = (m + 1 < (int) cks.size()) ? cks[(size_t) m + 1] : N - 1;
      if (ins[(size_t) j + 1] <= a) continue;          // (no INS update in this stretch)
      const int last_missing = (seq[(size_t) e].second >= 0) ? e - 1 : e;
      if (last_missing > a) {                          // re-derive the posteriors of a+1 .. last_missing into the window
        if ((rc = pb_state_restore(ctx, seq[(size_t) a].second)) != PB_OK) return bail("restoring a checkpoint");
        for (int i = a + 1; i <= last_missing; i++) {
          pb_set_output_slot(ctx, W(i - a - 1));
          if ((rc = seq[(size_t) i].first->updateFilter(ctx)) != PB_OK) return bail("re-applying an update");
          smoother_reapplied_updates++;
        }
      }
      auto slot_at = [&](int i) { return seq[(size_t) i].second >= 0 ? seq[(size_t) i].second : W(i - a - 1); };
      while (j >= 0 && ins[(size_t) j + 1] > a) {
        const int ip = ins[(size_t) j + 1];
        const int out = spare[toggle];
        if ((rc = pb_smooth_step(ctx, slot_at(ip), next, slot_at(ip - 1), out, dt)) != PB_OK) return bail("smoother step");
        if (on_smoothed) on_smoothed(seq[(size_t) ins[(size_t) j]].first->utime, out);
        next = out;
        toggle ^= 1;
        steps++;
        j--;
      }
    }
    if (head_keep >= 0) {
      if ((rc = pb_state_restore(ctx, head_keep)) != PB_OK) return bail("putting the head back");   // the newest posterior again
      pb_set_utime(ctx, head_utime);
    }
    device_head = nullptr;  // callers may have restored slots into the head: force a restore on the next replay
    return steps;
  }

private:
  bool flushing_ = false;
  int holding_ = 0;  // updates at the end of the history that have not been applied yet (0, 1 = an INS step, 2 = INS + legodo)
  // an INS step followed by the velocity measurement LegOdoCommon's lin_rate mode produces (what run_fused accepts)
  // the measurement object that carries a deferred leg odometry (pair_kernel / make_measurement): the update itself, or the
  // six-row half of LegOdoCommon's pos_and_lin_rate either-update (its block holds both halves' rows and masks)
  static RBISIndexedMeasurement *deferredMeasurement(RBISUpdateInterface *u)
  {
    if (auto *e = dynamic_cast<RBISEitherUpdate *>(u)) u = e->first;
    return dynamic_cast<RBISIndexedMeasurement *>(u);
  }
  bool fusible_pair(RBISIMUProcessStep *imu, RBISUpdateInterface *next)
  {
    if (auto *d = deferredMeasurement(next))
      if (d->pair_kernel) return true;   // any of LegOdoCommon's modes: the pair kernel forms and applies it
    if (dynamic_cast<RBISIndexedPlusOrientationMeasurement *>(next) != nullptr) return false;
    auto *m = dynamic_cast<RBISIndexedMeasurement *>(next);
    if (m == nullptr || m->index != RBIS::velocityInds()) return false;
    if (imu->imu_block.mem == PB_HOST_BROADCAST && m->measurement.mem == PB_HOST_BROADCAST && m->r_kind == PB_R_DIAG_BROADCAST && m->mask == nullptr)
      return true;
    if (device_lo_block(m)) return true;
    return imu->imu_block.mem == PB_HOST && m->measurement.mem == PB_HOST && m->cov_mem == PB_HOST &&
           (m->r_kind == PB_R_DIAG || m->r_kind == PB_R_DIAG_BROADCAST);
  }
  // ... followed by a position_orient / position_yaw correction with a diagonal R -> one pb_step_legodo_correct
  bool run_fused3(RBISIMUProcessStep *imu, RBISUpdateInterface *second, RBISUpdateInterface *third, int &rc)
  {
    auto *m = dynamic_cast<RBISIndexedMeasurement *>(second);
    auto *o = dynamic_cast<RBISIndexedPlusOrientationMeasurement *>(third);
    if (m == nullptr || o == nullptr) return false;
    if (m->measurement.mem == PB_DEVICE) return false;  // (a device-resident leg-odometry block pairs, it does not triple)
    int kind;
    if (o->index == std::vector<int>{ 9, 10, 11, 6, 7, 8 }) kind = PB_CORR_POS_ORIENT;
    else if (o->index == std::vector<int>{ 9, 10, 11, 8 }) kind = PB_CORR_POS_YAW;
    else return false;
    if (o->r_kind != PB_R_DIAG && o->r_kind != PB_R_DIAG_BROADCAST) return false;
    if (o->measurement.mem != o->orientation.mem) return false;
    if (o->r_kind == PB_R_DIAG && o->cov_mem != o->measurement.mem) return false;
    const double q[4] = { imu->q_gyro, imu->q_accel, imu->q_gyro_bias, imu->q_accel_bias };
    const double *lo;
    const uint8_t *mask1 = nullptr;
    int mem1;
    double lo6[6];
    if (imu->imu_block.mem == PB_HOST_BROADCAST) {
      for (int i = 0; i < 3; i++) { lo6[i] = m->measurement.p[i]; lo6[3 + i] = m->measurement_cov[i]; }
      lo = lo6;
      mem1 = PB_HOST_BROADCAST;
    } else {
      fuse_lo_.resize((size_t) 6 * B);
      memcpy(fuse_lo_.data(), m->measurement.p, sizeof(double) * 3 * B);
      if (m->r_kind == PB_R_DIAG) memcpy(fuse_lo_.data() + (size_t) 3 * B, m->measurement_cov, sizeof(double) * 3 * B);
      else
        for (int i = 0; i < 3; i++) std::fill_n(fuse_lo_.begin() + (size_t) (3 + i) * B, B, m->measurement_cov[i]);
      lo = fuse_lo_.data();
      mask1 = m->mask;
      mem1 = PB_HOST;
    }
    imu->announce(ctx);
    rc = pb_step_legodo_correct(ctx, imu->imu_block.p, lo, mask1, q, mem1, kind, o->measurement.p, o->measurement_cov,
                                o->r_kind, o->orientation.p, o->mask, o->measurement.mem);
    return true;
  }
  // a velocity measurement that lives on the device as ONE [6][B] block (z, diagonal R) + mask: what
  // LegOdoHandler::processMessageFeet / pb_legodo_update(_after_predict) leave there
  bool device_lo_block(const RBISIndexedMeasurement *m) const
  {
    return m->measurement.mem == PB_DEVICE && m->r_kind == PB_R_DIAG && m->measurement_cov == m->measurement.p + (size_t) 3 * B;
  }
  // imu followed by a velocity measurement LegOdoCommon's lin_rate mode produces -> one pb_step_legodo; false = not fusible
  bool run_fused(RBISIMUProcessStep *imu, RBISUpdateInterface *next, int &rc)
  {
    if (auto *d = deferredMeasurement(next))
      if (d->pair_kernel) {  // the measurement is made inside the step kernel (leg odometry, any of LegOdoCommon's modes): one launch for the message pair
        rc = d->pair_kernel(ctx, imu, history_slots > 0);
        d->pair_kernel = nullptr;
        d->make_measurement = nullptr;
        leg_kernel_pairs++;
        return true;
      }
    if (dynamic_cast<RBISIndexedPlusOrientationMeasurement *>(next) != nullptr) return false;
    auto *m = dynamic_cast<RBISIndexedMeasurement *>(next);
    if (m == nullptr || m->index != RBIS::velocityInds()) return false;
    const double q[4] = { imu->q_gyro, imu->q_accel, imu->q_gyro_bias, imu->q_accel_bias };
    imu->announce(ctx);   // (taken by whichever step call follows)
    if (device_lo_block(m)) {  // IMU block from the host (broadcast or per filter), measurement on the device
      rc = pb_step_legodo_split(ctx, imu->imu_block.p, imu->imu_block.mem, m->measurement.p, m->mask, PB_DEVICE, q);
      return true;
    }
    if (imu->imu_block.mem == PB_HOST_BROADCAST && m->measurement.mem == PB_HOST_BROADCAST && m->r_kind == PB_R_DIAG_BROADCAST &&
        m->mask == nullptr) {
      const double lo[6] = { m->measurement.p[0], m->measurement.p[1], m->measurement.p[2],
                             m->measurement_cov[0], m->measurement_cov[1], m->measurement_cov[2] };
      rc = pb_step_legodo(ctx, imu->imu_block.p, lo, nullptr, q, PB_HOST_BROADCAST);
      return true;
    }
    if (imu->imu_block.mem == PB_HOST && m->measurement.mem == PB_HOST && m->cov_mem == PB_HOST &&
        (m->r_kind == PB_R_DIAG || m->r_kind == PB_R_DIAG_BROADCAST)) {
      fuse_lo_.resize((size_t) 6 * B);
      memcpy(fuse_lo_.data(), m->measurement.p, sizeof(double) * 3 * B);
      if (m->r_kind == PB_R_DIAG) memcpy(fuse_lo_.data() + (size_t) 3 * B, m->measurement_cov, sizeof(double) * 3 * B);
      else
        for (int i = 0; i < 3; i++) std::fill_n(fuse_lo_.begin() + (size_t) (3 + i) * B, B, m->measurement_cov[i]);
      rc = pb_step_legodo(ctx, imu->imu_block.p, fuse_lo_.data(), m->mask, q, PB_HOST);
      return true;
    }
    return false;
  }
  std::vector<double> fuse_lo_;
  void drop_checkpoint(RBISUpdateInterface *u)
  {
    auto it = checkpoint_of.find(u);
    if (it != checkpoint_of.end()) {
      free_slots.push_back(it->second);
      checkpoint_of.erase(it);
    }
  }
  // a free checkpoint slot for update u, recycling the oldest part of the window when the pool is exhausted; -1 = none
  int reserve_slot(RBISUpdateInterface *u)
  {
    if (free_slots.empty()) {
      // pool exhausted: the window shrinks to what the pool covers -- drop everything before the second-oldest
      // checkpoint (begin() must keep one: it is the prior of the oldest replayable update)
      auto &map = history.updateMap;
      auto it = map.begin();
      ++it;
      while (it != map.end() && checkpoint_of.find(it->second) == checkpoint_of.end()) ++it;
      if (it == map.end() || it->second == u) return -1;  // nothing to recycle: skip this checkpoint
      erase_before(it);
    }
    const int slot = free_slots.back();
    free_slots.pop_back();
    return slot;
  }
  void save_checkpoint(RBISUpdateInterface *u)
  {
    const int slot = reserve_slot(u);
    if (slot < 0) return;
    int rc = pb_state_save(ctx, slot);
    if (rc != PB_OK) last_status = rc;
    checkpoint_of[u] = slot;
    since_checkpoint = 0;
  }
  void erase_before(updateHistory::historyMapIterator keep)
  {
    auto &map = history.updateMap;
    for (auto it = map.begin(); it != keep; ++it) {
      drop_checkpoint(it->second);
      delete it->second;
    }
    map.erase(map.begin(), keep);
  }
  // update_history.cpp:44-55, with the extra rule that the new first element must hold a checkpoint
  void clearHistoryBeforeUtime(int64_t utime)
  {
    auto &map = history.updateMap;
    if (history_slots == 0) {  // in-order only: keep just the head
      auto last = map.end();
      --last;
      for (int h = 0; h < holding_; h++) --last;  // held-back updates have not been applied: the head is before them
      erase_before(last);
      return;
    }
    auto keep = map.begin();
    for (auto it = map.begin(); it != map.end() && it->first <= utime; ++it)
      if (checkpoint_of.find(it->second) != checkpoint_of.end()) keep = it;
    if (keep != map.begin()) erase_before(keep);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// messages (batched stand-ins for the LCM types at the seam; every array is [k][B])
// ---------------------------------------------------------------------------------------------------------------
namespace msgs {
struct ins_t {               // bot_core::ins_t
  int64_t utime;
  BatchArray gyro, accel;    // [3][B] each, sensor frame
  BatchArray mag;            // [3][B] magnetometer, sensor frame; optional (p == NULL: zeros, like the Atlas path :274)
  const uint8_t *valid = nullptr;  // [B], in the memory space of gyro / accel (independent log segments, SegmentBatcher /
                                   // SegmentStreamer): 0 = this filter has no message -- its segment has ended -- and idles: its
                                   // step is taken with dt = 0
};
struct kvh_raw_imu_t {       // the newest packet of bot_core::kvh_raw_imu_batch_t (atlas_filter == false path)
  int64_t utime;
  BatchArray delta_rotation, linear_acceleration;  // [3][B]
  double raw_dt;             // (raw_imu[0].utime - raw_imu[1].utime) * 1E-6
};
struct legodo_delta_t {      // what leg_estimate hands to LegOdoCommon::createMeasurement
  int64_t utime, prev_utime;
  const double *position;        // [3][B] pelvis position (may be NULL unless mode pos_and_lin_rate), host
  const double *delta_trans;     // [3][B] translation of the pelvis delta over (prev_utime, utime], host
  const double *delta_quat;      // [4][B] rotation of the delta (NULL = identity), host
  const int *position_status;    // [B] or NULL (= valid)
  const float *delta_status;     // [B]: < 0 skip (leg_estimate.hpp:84-93), < 0.5 certain, else uncertain
};
struct foot_state_t {        // what forward kinematics + the foot sensors give per joint-state message (leg_estimate.cpp:430-447)
  int64_t utime;
  BatchArray feet;           // [14][B]: body-to-left-foot (t3, q4 = w,x,y,z), body-to-right-foot (t3, q4)
  BatchArray forces;         // [2][B]: left, right vertical foot force
};
struct joint_state_t {       // bot_core::joint_state_t: the arrays are float on the wire and here
  int64_t utime;
  std::vector<std::string> joint_name;      // num_joints names, the same for every filter
  const float *joint_position = nullptr;    // [num_joints][B] (PB_HOST / PB_DEVICE) or [num_joints] (PB_HOST_BROADCAST)
  const float *joint_velocity = nullptr;    // (only the joint Kalman filter reads it, leg_estimate.cpp:418-426)
  const float *joint_effort = nullptr;      // same shape as joint_position; read when legodo.torque_adjustment is set
  int mem = PB_HOST;                        // PB_DEVICE arrays must stay valid until the next message (see LegOdoHandler::forceTorqueDevice)
  // independent log segments (one recorded robot per filter, SegmentBatcher): every filter's message has its OWN time stamp
  // -- utimes [B], HOST; `utime` above then is the batch-level time that orders the update in the history -- and a filter whose
  // segment has ended has no message at all: valid [B], HOST, 0 = none.  NULL = one time for all / all valid.
  const int64_t *utimes = nullptr;
  const uint8_t *valid = nullptr;
  int times_mem = PB_HOST;                  // where utimes / valid live: PB_HOST, or PB_DEVICE (chunked replays: read in place)
};
struct six_axis_force_torque_array_t {   // bot_core::six_axis_force_torque_array_t: what the leg odometry reads of it
  int64_t utime;
  BatchArray force_z;        // [2][B] (PB_HOST) or [2] (PB_HOST_BROADCAST): sensors[0].force[2], sensors[1].force[2] (left, right foot)
};
struct controller_foot_contact_t {       // pronto::controller_foot_contact_t
  int64_t utime;
  int32_t num_left_foot_contacts, num_right_foot_contacts;
};
struct update_t {            // pronto::update_t (fovis)
  int64_t timestamp, prev_timestamp;
  const uint8_t *estimate_valid;   // [B] or NULL: estimate_status == ESTIMATE_VALID
  BatchArray translation, rotation;  // [3][B], [4][B]
};
struct pose_t {              // bot_core::pose_t
  int64_t utime;
  BatchArray pos, vel, orientation;  // [3][B], [3][B], [4][B]
  const uint8_t *valid = nullptr;    // [B] HOST, PB_HOST messages only: 0 = this filter has no message (independent log segments)
};
struct indexed_measurement_t {  // pronto::indexed_measurement_t
  int64_t utime;
  std::vector<int> z_indices;
  BatchArray z_effective;      // [m][B]
  const double *R_effective;   // [m*m][B] column-major, same memory space as z_effective
};
struct rigid_transform_t {  // bot_core::rigid_transform_t (Vicon)
  int64_t utime;
  BatchArray trans, quat;    // [3][B], [4][B] host (or host-broadcast) arrays
};
struct gps_data_t {
  int64_t utime;
  const uint8_t *has_lock;     // [B]: gps_lock >= 3
  BatchArray xyz_pos;
};
struct kvh_raw_imu_packet_t {  // bot_core::kvh_raw_imu_t
  int64_t utime, packet_count;
  const double *delta_rotation, *linear_acceleration;  // [3][B] host arrays
};
struct kvh_raw_imu_batch_t {   // bot_core::kvh_raw_imu_batch_t: raw_imu[0] is the NEWEST packet, as on the wire
  int64_t utime;
  std::vector<kvh_raw_imu_packet_t> raw_imu;
  int mem = PB_HOST;           // PB_HOST_BROADCAST: the packets' arrays are [3], one robot's IMU for every filter
};
// B DIFFERENT robots' kvh_raw_imu_batch_t messages as one batched message (independent log segments): every robot's message has
// been through ITS OWN IMUStream::convertFromLCMBatch (imu_stream.cpp:62-98) where it was decoded, and what is left of it is
// what InsHandler::processMessageAtlas reads (sensor_handlers.cpp:173-204).  All arrays live in `mem` (PB_HOST or PB_DEVICE).
struct kvh_raw_imu_segments_t {
  int64_t utime;                 // batch-level time (orders the update in the history)
  int max_new = 0;               // rows of new_accel: the most new packets any filter can have in one message
  const int32_t *n_new = nullptr;         // [B] new packets of filter b; 0 = "no new IMU packets": no update for it (:181-187)
  const uint8_t *valid = nullptr;         // [B] n_new[b] > 0 and the filter has a message at all
  const double *new_accel = nullptr;      // [max_new][3][B] linear_acceleration of the new packets, oldest first
  const double *delta_rotation = nullptr; // [3][B] of the newest new packet
  const double *raw_dt = nullptr;         // [B] that packet's utime_delta * 1E-6 (atlas_filter), or (raw_imu[0].utime - raw_imu[1].utime) * 1E-6
  const int64_t *utimes = nullptr;        // [B] every robot's own message utime
  int mem = PB_HOST;
};
}  // namespace msgs

// estimate_tools/src/estimate_tools/imu_stream.hpp:10-37, imu_stream.cpp:62-98: KVH batch messages repeat packets
// (1 kHz packets in ~333 Hz batches); only packets with a packet_count above the last one seen are new.
struct IMUPacket {
  int64_t utime_raw, utime_batch, utime, utime_delta, packet_count;
  const double *delta_rotation, *linear_acceleration;  // [3][B]
};
struct IMUBatch {
  int64_t utime;
  std::vector<IMUPacket> packets;      // new packets, [0] oldest ... [end] newest
  std::vector<IMUPacket> packets_old;  // packets already seen in an earlier message
};
// estimate_tools/src/estimate_tools/iir_notch.cpp:3-61: 2nd-order IIR notch, host version.  The batch runs the cascade
// on the device (pb_imu_notch, one state per filter); when ONE robot's IMU feeds every filter (PB_HOST_BROADCAST) every
// filter would compute the same numbers, so the handler filters once on the host -- the reference's own code path.
class IIRNotch {
public:
  double b[3], a[3], x[2] = { 0, 0 }, y[2] = { 0, 0 };
  IIRNotch(double notch_freq, double fs)
  {
    const double Wo0 = notch_freq / (fs / 2), Ab = fabs(10 * log10(.5));
    const double BW = Wo0 * M_PI, Wo = Wo0 * M_PI;      // "Inputs are normalized by pi" (:21-23); Bw = Wo (:8)
    const double Gb = pow(10, -Ab / 20.);
    const double beta = (sqrt(1.0 - Gb * Gb) / Gb) * tan(BW / 2.0);
    const double gain = 1 / (1 + beta);
    b[0] = gain; b[1] = gain * (-2.0 * cos(Wo)); b[2] = gain;
    a[0] = 1.0; a[1] = -2 * gain * cos(Wo); a[2] = 2 * gain - 1;
  }
  double processSample(double input)
  {
    const double output = (input * b[0] + x[0] * b[1] + x[1] * b[2]) - (y[0] * a[1] + y[1] * a[2]);  // :52
    x[1] = x[0]; x[0] = input;
    y[1] = y[0]; y[0] = output;
    return output;
  }
};

class IMUStream {
public:
  IMUStream() : last_packet_(-1), last_packet_utime_(0), counter_(0) {}
  IMUBatch convertFromLCMBatch(const msgs::kvh_raw_imu_batch_t *msg)
  {
    if (!msg->raw_imu.empty() && msg->raw_imu[0].packet_count < last_packet_) {
      fprintf(stdout, "Detected time skip, resetting IMUStream\n");  // imu_stream.cpp:63-68
      last_packet_ = -1;
      last_packet_utime_ = 0;
      counter_ = 0;
    }
    IMUBatch batch;
    batch.utime = msg->utime;
    for (int i = (int) msg->raw_imu.size() - 1; i >= 0; i--) {  // oldest first (:76)
      const auto &m = msg->raw_imu[i];
      IMUPacket raw;
      raw.utime_raw = m.utime;
      raw.utime_batch = msg->utime;
      raw.utime = m.utime;
      raw.packet_count = m.packet_count;
      raw.delta_rotation = m.delta_rotation;
      raw.linear_acceleration = m.linear_acceleration;
      if (m.packet_count > last_packet_) {
        raw.utime_delta = m.utime - last_packet_utime_;
        batch.packets.push_back(raw);
        last_packet_ = m.packet_count;
        last_packet_utime_ = m.utime;
      } else {
        raw.utime_delta = m.utime - (-m.utime);  // "deliberately obfuscate the delta field" (:87-88)
        batch.packets_old.push_back(raw);
      }
    }
    counter_++;
    return batch;
  }
  int getNumberBatchSinceReset() const { return counter_; }

private:
  int64_t last_packet_, last_packet_utime_;
  int counter_;
};

// ---------------------------------------------------------------------------------------------------------------
// handlers
// ---------------------------------------------------------------------------------------------------------------
// RBISInitializer's bookkeeping helpers (rbis_initializer.cpp:96-113)
struct RBISInitializer {
  static bool allInitializedExcept(const std::map<std::string, bool> &sensors_initialized, const std::string &sensor_prefix)
  {
    for (const auto &kv : sensors_initialized) {
      if (kv.first == sensor_prefix) continue;
      if (!kv.second) return false;
    }
    return true;
  }
  static bool initializingWith(const std::map<std::string, bool> &sensors_initialized, const std::string &sensor_prefix)
  {
    return sensors_initialized.count(sensor_prefix) > 0;
  }
};

// Eigen's Quaternion::setFromTwoVectors(a, b) (w,x,y,z): the smallest rotation with q * a parallel to b; opposite
// vectors take any axis orthogonal to a (Eigen takes one from an SVD)
inline void quat_from_two_vectors(const double a[3], const double b[3], double q[4])
{
  const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  const double v0[3] = { a[0] / na, a[1] / na, a[2] / na }, v1[3] = { b[0] / nb, b[1] / nb, b[2] / nb };
  double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
  if (c < -1.0 + 1e-12) {
    const int k = (fabs(v0[0]) < fabs(v0[1])) ? (fabs(v0[0]) < fabs(v0[2]) ? 0 : 2) : (fabs(v0[1]) < fabs(v0[2]) ? 1 : 2);
    double e[3] = { 0, 0, 0 };
    e[k] = 1.0;
    const double ax[3] = { v0[1] * e[2] - v0[2] * e[1], v0[2] * e[0] - v0[0] * e[2], v0[0] * e[1] - v0[1] * e[0] };
    const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (c < -1.0) c = -1.0;
    const double w2 = (1.0 + c) * 0.5;
    q[0] = sqrt(w2);
    for (int i = 0; i < 3; i++) q[1 + i] = ax[i] / n * sqrt(1.0 - w2);
    return;
  }
  const double axis[3] = { v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0] };
  const double sq = sqrt((1.0 + c) * 2.0), invs = 1.0 / sq;
  q[0] = sq * 0.5;
  for (int i = 0; i < 3; i++) q[1 + i] = axis[i] * invs;
}

class InsHandler {
public:
  std::string channel;
  BotTrans ins_to_body;
  double cov_accel, cov_gyro, cov_accel_bias, cov_gyro_bias, dt;
  bool atlas_filter, accel_bias_update_online, gyro_bias_update_online;
  int64_t prev_utime_atlas = 0;

  InsHandler(BotParam *_param, const BotTrans *ins_to_body_ = nullptr)
  {
    param_ = _param;
    channel = bot_param_get_str_or_fail(_param, "state_estimator.ins.channel");
    cov_gyro = bot_sq(bot_to_radians(bot_param_get_double_or_fail(_param, "state_estimator.ins.q_gyro")));          // :18-19
    cov_accel = bot_sq(bot_param_get_double_or_fail(_param, "state_estimator.ins.q_accel"));                        // :20-21
    cov_gyro_bias = bot_sq(bot_to_radians(bot_param_get_double_or_fail(_param, "state_estimator.ins.q_gyro_bias")));  // :22-23
    cov_accel_bias = bot_sq(bot_param_get_double_or_fail(_param, "state_estimator.ins.q_accel_bias"));              // :24-25
    dt = bot_param_get_double_or_fail(_param, "state_estimator.ins.timestep_dt");                                  // :27
    atlas_filter = bot_param_get_boolean_or_fail(_param, "state_estimator.ins.atlas_filter");
    accel_bias_update_online = bot_param_get_boolean_or_fail(_param, "state_estimator.ins.accel_bias_update_online");
    if (!accel_bias_update_online) cov_accel_bias = 0.0;                                                          // :70-72
    gyro_bias_update_online = bot_param_get_boolean_or_fail(_param, "state_estimator.ins.gyro_bias_update_online");
    if (!gyro_bias_update_online) cov_gyro_bias = 0.0;                                                            // :89-91
    if (ins_to_body_) ins_to_body = *ins_to_body_;
    notch_freq = atlas_filter ? bot_param_get_double_or_fail(_param, "state_estimator.ins.atlas_filter_freq") : 0.0;  // :31
  }

  // Atlas KVH path WITH the front end (sensor_handlers.cpp:173-197): de-duplicate the batch message, run every NEW
  // packet through the 3-stage notch cascade (on the device: pb_imu_notch), use the newest filtered packet.
  RBISUpdateInterface *processMessageAtlas(const msgs::kvh_raw_imu_batch_t *msg, MavStateEstimator *est)
  {
    const int B = est->B;
    if (msg->mem != PB_HOST && msg->mem != PB_HOST_BROADCAST) return nullptr;
    if (!atlas_filter) {
      // :199-204: newest packet unfiltered, raw_dt from the two newest packets
      if (msg->raw_imu.size() < 2) return nullptr;
      msgs::kvh_raw_imu_t m{ msg->utime, BatchArray(msg->raw_imu[0].delta_rotation, msg->mem),
                             BatchArray(msg->raw_imu[0].linear_acceleration, msg->mem),
                             (msg->raw_imu[0].utime - msg->raw_imu[1].utime) * 1E-6 };
      return processMessageAtlasPacket(&m, est);
    }
    if (msg->mem == PB_HOST_BROADCAST) {
      // one robot for every filter: de-duplicate and filter ONCE on the host (IIRNotch above), hand the newest filtered
      // packet over as a broadcast block.  (Do not mix with per-filter messages in one run: that state lives on the device.)
      if (host_notch_.empty())
        for (int ax = 0; ax < 3; ax++)
          for (int i = 0; i < 3; i++) host_notch_.emplace_back(notch_freq * (double) (1 << i), 1000.0);  // :33-40: f, 2f, 4f
      IMUBatch batch = imu_data_.convertFromLCMBatch(msg);
      if (batch.packets.empty()) return nullptr;
      double filt[3] = { 0, 0, 0 };
      for (const IMUPacket &pk : batch.packets)
        for (int ax = 0; ax < 3; ax++) {
          double v = pk.linear_acceleration[ax];
          for (int i = 0; i < 3; i++) v = host_notch_[(size_t) ax * 3 + i].processSample(v);
          filt[ax] = v;
        }
      const IMUPacket &p = batch.packets.back();
      msgs::kvh_raw_imu_t m{ msg->utime, BatchArray(p.delta_rotation, PB_HOST_BROADCAST), BatchArray(filt, PB_HOST_BROADCAST),
                             p.utime_delta * 1E-6 };
      return processMessageAtlasPacket(&m, est);
    }
    if (!notch_initialised) {
      if (pb_imu_notch_init(est->ctx, notch_freq, 1000) != PB_OK) {  // fs = 1000 (:32)
        fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
        exit(1);
      }
      notch_initialised = true;
    }
    IMUBatch batch = imu_data_.convertFromLCMBatch(msg);
    if (batch.packets.empty()) return nullptr;  // :181-187 "happens all the time at 1kHz"
    const size_t np = batch.packets.size();
    std::vector<double> acc((size_t) np * 3 * B), filt((size_t) 3 * B);
    for (size_t p = 0; p < np; p++) memcpy(&acc[p * 3 * B], batch.packets[p].linear_acceleration, sizeof(double) * 3 * B);
    if (pb_imu_notch(est->ctx, (int) np, acc.data(), filt.data(), PB_HOST) != PB_OK) {
      fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
      return nullptr;
    }
    const IMUPacket &p = batch.packets.back();  // the most recent filtered packet (:191-196)
    msgs::kvh_raw_imu_t m{ msg->utime, BatchArray(p.delta_rotation, PB_HOST), BatchArray(filt.data(), PB_HOST),
                           p.utime_delta * 1E-6 };
    return processMessageAtlasPacket(&m, est);
  }

  double notch_freq = 0.0;
  bool notch_initialised = false;
  IMUStream imu_data_;
  std::vector<IIRNotch> host_notch_;  // [axis][stage], broadcast messages only

  // Microstrain path (sensor_handlers.cpp:96-131): rotate accel and gyro into the body frame, dt = param
  RBISUpdateInterface *processMessage(const msgs::ins_t *msg, MavStateEstimator *est)
  {
    if (msg->gyro.mem == PB_DEVICE && msg->accel.mem == PB_DEVICE)   // per-filter samples already in HBM: the frame rotation runs there
      return buildOnDevice(est, msg->gyro.p, msg->accel.p, nullptr, nullptr, msg->utime, msg->valid, false, PB_DEVICE);
    return build(msg->gyro, msg->accel, 1.0, false, dt, msg->utime, est->B, msg->valid);
  }
  // Atlas KVH path for B DIFFERENT robots (independent log segments): the de-duplication ran per robot where the messages were
  // decoded (msgs::kvh_raw_imu_segments_t); here the notch cascade of every filter takes ITS new packets (pb_imu_notch_counts), and
  // the newest filtered packet becomes the process step with the filter's own raw_dt and its own message-time dt
  // (sensor_handlers.cpp:173-252).  A filter without a new packet -- the reference returns NULL for its message -- idles (dt = 0).
  // Returns NULL when the message carries nothing at all (max_new == 0).
  RBISUpdateInterface *processMessageAtlasSegments(const msgs::kvh_raw_imu_segments_t *msg, MavStateEstimator *est)
  {
    if ((msg->mem != PB_HOST && msg->mem != PB_DEVICE) || msg->max_new < 1 || !msg->new_accel || !msg->delta_rotation || !msg->raw_dt) return nullptr;
    const size_t B = (size_t) est->B;
    const double *accel = msg->new_accel;   // atlas_filter == false: row 0 is the newest packet, unfiltered (:199-204)
    if (atlas_filter) {
      if (!notch_initialised) {
        if (pb_imu_notch_init(est->ctx, notch_freq, 1000) != PB_OK) {  // fs = 1000 (:32)
          fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
          exit(1);
        }
        notch_initialised = true;
      }
      if (msg->n_new == nullptr) return nullptr;
      if (msg->mem == PB_DEVICE) {
        if (notch_out_dev_ == nullptr) {
          void *d = nullptr;
          if (pb_malloc(est->ctx, sizeof(double) * 3 * B, &d) != PB_OK) {
            fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
            exit(1);
          }
          notch_out_dev_ = std::make_shared<DeviceBlock>(std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(double) * 3 * B), d);
        }
        accel = (const double *) notch_out_dev_->p;
        if (pb_imu_notch_counts(est->ctx, msg->max_new, msg->n_new, msg->new_accel, (double *) notch_out_dev_->p, PB_DEVICE) != PB_OK) {
          fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
          return nullptr;
        }
      } else {
        notch_out_host_.resize(3 * B);
        if (pb_imu_notch_counts(est->ctx, msg->max_new, msg->n_new, msg->new_accel, notch_out_host_.data(), PB_HOST) != PB_OK) {
          fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
          return nullptr;
        }
        accel = notch_out_host_.data();
      }
    }
    return buildOnDevice(est, msg->delta_rotation, accel, msg->raw_dt, msg->utimes, msg->utime, msg->valid, true, msg->mem);
  }
  // Atlas KVH path without the notch (sensor_handlers.cpp:199-252): gyro = delta_rotation/raw_dt, accel through the
  // full ins_to_body transform (rotation + translation, :227), dt from message timestamps (:239-249)
  RBISUpdateInterface *processMessageAtlasPacket(const msgs::kvh_raw_imu_t *msg, MavStateEstimator *est)
  {
    double integration_dt = (prev_utime_atlas == 0) ? dt : (msg->utime - prev_utime_atlas) * 1E-6;
    if (integration_dt > 0.1) fprintf(stdout, "dt was : %f - there is an issue with timestamps\n", integration_dt);
    prev_utime_atlas = msg->utime;
    return build(msg->delta_rotation, msg->linear_acceleration, 1.0 / msg->raw_dt, true, integration_dt, msg->utime, est->B);
  }

  // ---- initialisation from gravity (processMessageInit / processMessageInitCommon, sensor_handlers.cpp:254-364) ----
  // Averages num_to_init body-frame IMU samples per filter: roll and pitch from the mean specific force
  // (init quat = init quat * setFromTwoVectors(mean(-accel), -z)), the gyro bias from the mean rate (0 if any axis exceeds
  // max_initial_gyro_bias), the matching covariance blocks from default_cov.  As in the reference the INS goes last
  // (allInitializedExcept "ins") and the bias that ends up in init_state is *_bias_initial -- the estimate only if
  // *_bias_recalc_at_start.  When the filter also initialises with "gps", the yaw comes from the mean magnetometer vector
  // (msgs::ins_t::mag, moved to the body frame with bot_trans_apply_vec like :148): its horizontal part is turned onto +y
  // (ENU, :338-351).  Keys are read (or_fail) on first use.
  int num_to_init = 0, init_counter = 0;
  double max_initial_gyro_bias = 0;
  bool accel_bias_recalc_at_start = false, gyro_bias_recalc_at_start = false;
  std::vector<double> g_vec_sum, gyro_bias_sum, mag_vec_sum;  // [3][B]
  std::vector<double> gyro_bias_initial, accel_bias_initial;  // [3] (same for every filter) or [3][B] after a recalc
  bool processMessageInit(const msgs::ins_t *msg, const std::map<std::string, bool> &sensors_initialized,
                          const RBIS & /*default_state*/, const RBIM &default_cov, RBIS &init_state, RBIM &init_cov)
  {
    const int B = init_state.B, n = init_state.n;
    if (!init_params_read_) {
      num_to_init = (int) bot_param_get_int_or_fail(param_, "state_estimator.ins.num_to_init");
      max_initial_gyro_bias = bot_param_get_double_or_fail(param_, "state_estimator.ins.max_initial_gyro_bias");
      gyro_bias_initial.resize(3);
      accel_bias_initial.resize(3);
      bot_param_get_double_array_or_fail(param_, "state_estimator.ins.accel_bias_initial", accel_bias_initial.data(), 3);
      bot_param_get_double_array_or_fail(param_, "state_estimator.ins.gyro_bias_initial", gyro_bias_initial.data(), 3);
      accel_bias_recalc_at_start = bot_param_get_boolean_or_fail(param_, "state_estimator.ins.accel_bias_recalc_at_start");
      gyro_bias_recalc_at_start = bot_param_get_boolean_or_fail(param_, "state_estimator.ins.gyro_bias_recalc_at_start");
      g_vec_sum.assign((size_t) 3 * B, 0.0);
      gyro_bias_sum.assign((size_t) 3 * B, 0.0);
      mag_vec_sum.assign((size_t) 3 * B, 0.0);
      init_params_read_ = true;
    }
    init_state.utime = msg->utime;
    auto *update = static_cast<RBISIMUProcessStep *>(build(msg->gyro, msg->accel, 1.0, false, dt, msg->utime, B));
    if (update == nullptr) return false;
    if (!RBISInitializer::allInitializedExcept(sensors_initialized, "ins")) {  // force the INS to go last (:266-267)
      delete update;
      return false;
    }
    init_counter++;
    const bool bcast = update->imu_block.mem == PB_HOST_BROADCAST;
    const size_t per = bcast ? 1 : (size_t) B;
    for (int b = 0; b < B; b++) {
      const size_t sb = bcast ? 0 : (size_t) b;
      for (int i = 0; i < 3; i++) {
        gyro_bias_sum[(size_t) i * B + b] += update->owned[(size_t) i * per + sb];            // :291
        g_vec_sum[(size_t) i * B + b] += -update->owned[(size_t) (3 + i) * per + sb];         // :289
      }
    }
    delete update;
    if (msg->mag.p != nullptr && msg->mag.mem != PB_DEVICE) {  // :147-149, :290
      const size_t mper = (msg->mag.mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
      for (int b = 0; b < B; b++) {
        const size_t sb = (mper == 1) ? 0 : (size_t) b;
        const double ms[3] = { msg->mag.p[sb], msg->mag.p[mper + sb], msg->mag.p[2 * mper + sb] };
        double mb[3];
        bot_trans_apply_vec(&ins_to_body, ms, mb);
        for (int i = 0; i < 3; i++) mag_vec_sum[(size_t) i * B + b] += mb[i];
      }
    }
    if (init_counter < num_to_init) return false;
    bool warned = false;
    std::vector<double> gb_est((size_t) 3 * B);
    for (int b = 0; b < B; b++) {
      if (!warned && (init_cov(RBIS::chi_ind, RBIS::chi_ind, b) > 0 || init_cov(RBIS::chi_ind + 1, RBIS::chi_ind + 1, b) > 0)) {
        fprintf(stderr, "Warning: overriding initial roll, pitch with IMU values\n");
        warned = true;
      }
      double g_est[3], gb[3];
      bool too_big = false;
      for (int i = 0; i < 3; i++) {
        g_est[i] = g_vec_sum[(size_t) i * B + b] / (double) init_counter;
        gb[i] = gyro_bias_sum[(size_t) i * B + b] / (double) init_counter;
        too_big = too_big || fabs(gb[i]) > max_initial_gyro_bias;
      }
      if (too_big) gb[0] = gb[1] = gb[2] = 0.0;                                                 // :303-311
      const double minus_z[3] = { 0.0, 0.0, -1.0 };
      double qg[4];
      quat_from_two_vectors(g_est, minus_z, qg);                                               // :314-315
      const double q0[4] = { init_state.q(0, b), init_state.q(1, b), init_state.q(2, b), init_state.q(3, b) };
      const double qn[4] = { q0[0] * qg[0] - q0[1] * qg[1] - q0[2] * qg[2] - q0[3] * qg[3],
                             q0[0] * qg[1] + q0[1] * qg[0] + q0[2] * qg[3] - q0[3] * qg[2],
                             q0[0] * qg[2] - q0[1] * qg[3] + q0[2] * qg[0] + q0[3] * qg[1],
                             q0[0] * qg[3] + q0[1] * qg[2] - q0[2] * qg[1] + q0[3] * qg[0] };
      for (int i = 0; i < 4; i++) init_state.q(i, b) = qn[i];                                   // :320
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++)
          init_cov(RBIS::chi_ind + r, RBIS::chi_ind + c, b) = default_cov(RBIS::chi_ind + r, RBIS::chi_ind + c, b);  // :321-322
      if (n == RBIS::rbis_num_states)
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++)
            init_cov(RBIS::gyro_bias_ind + r, RBIS::gyro_bias_ind + c, b) = default_cov(RBIS::gyro_bias_ind + r, RBIS::gyro_bias_ind + c, b);
      for (int i = 0; i < 3; i++) gb_est[(size_t) i * B + b] = gb[i];
      if (RBISInitializer::initializingWith(sensors_initialized, "gps")) {                      // :338-351
        double m_est[3] = { mag_vec_sum[b] / (double) init_counter, mag_vec_sum[(size_t) B + b] / (double) init_counter, 0.0 };
        const double unit_y[3] = { 0.0, 1.0, 0.0 };  // "in ENU, the magnetic vector should be aligned with Y axis"
        double qm[4];
        quat_from_two_vectors(m_est, unit_y, qm);
        const double q1[4] = { init_state.q(0, b), init_state.q(1, b), init_state.q(2, b), init_state.q(3, b) };
        const double qy[4] = { qm[0] * q1[0] - qm[1] * q1[1] - qm[2] * q1[2] - qm[3] * q1[3],
                               qm[0] * q1[1] + qm[1] * q1[0] + qm[2] * q1[3] - qm[3] * q1[2],
                               qm[0] * q1[2] - qm[1] * q1[3] + qm[2] * q1[0] + qm[3] * q1[1],
                               qm[0] * q1[3] + qm[1] * q1[2] - qm[2] * q1[1] + qm[3] * q1[0] };
        for (int i = 0; i < 4; i++) init_state.q(i, b) = qy[i];                                 // quat_mag * orientation
        init_cov(RBIS::chi_ind + 2, RBIS::chi_ind + 2, b) = default_cov(RBIS::chi_ind + 2, RBIS::chi_ind + 2, b);
      }
    }
    if (accel_bias_recalc_at_start && n == RBIS::rbis_num_states) {  // :353-356: whatever init_state holds now
      accel_bias_initial.resize((size_t) 3 * B);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) accel_bias_initial[(size_t) i * B + b] = init_state(RBIS::accel_bias_ind + i, b);
    }
    if (gyro_bias_recalc_at_start) gyro_bias_initial = gb_est;     // :358-361
    if (n == RBIS::rbis_num_states) {
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 3; i++) {                               // :363-364
          init_state(RBIS::gyro_bias_ind + i, b) = gyro_bias_initial[gyro_bias_initial.size() == 3 ? (size_t) i : (size_t) i * B + b];
          init_state(RBIS::accel_bias_ind + i, b) = accel_bias_initial[accel_bias_initial.size() == 3 ? (size_t) i : (size_t) i * B + b];
        }
    }
    return true;
  }

private:
  BotParam *param_ = nullptr;
  bool init_params_read_ = false;
  std::shared_ptr<DevicePool> ins_pool_;         // [7][B] process-step blocks made on the device
  std::shared_ptr<DeviceBlock> notch_out_dev_;   // [3][B] newest filtered packet (stream order: read by the step made from it)
  std::vector<double> notch_out_host_;
  // The per-sample arithmetic of processMessage / processMessageAtlas for B robots on the device (pb_ins_body_block): the update
  // owns the [7][B] block it is given.  atlas: gyro holds delta_rotation (divided by raw_dt), the acceleration is translated too,
  // dt comes from every filter's own message times.
  RBISUpdateInterface *buildOnDevice(MavStateEstimator *est, const double *gyro, const double *accel, const double *raw_dt, const int64_t *utimes,
                                     int64_t utime, const uint8_t *valid, bool atlas, int mem)
  {
    if (!ins_pool_) ins_pool_ = std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(double) * 7 * (size_t) est->B + (size_t) est->B);
    bool fresh = false;
    void *blk = ins_pool_->get(fresh);
    // (the block is [7][B] doubles + a [B] mask behind it: the update OWNS its copy of the mask -- a host mask is gone when the handler
    // returns, and a device mask inside a replayer's chunk buffer may be overwritten before a held-back step has run)
    uint8_t *mask_out = (blk != nullptr && valid != nullptr) ? (uint8_t *) ((double *) blk + (size_t) 7 * est->B) : nullptr;
    if (blk == nullptr ||
        pb_ins_body_block(est->ctx, gyro, accel, raw_dt, utimes, utime, valid, ins_to_body.rot_quat, atlas ? ins_to_body.trans_vec : nullptr, dt,
                          atlas ? 1 : 0, mem, (double *) blk, mask_out) != PB_OK) {
      fprintf(stderr, "InsHandler: %s\n", pb_last_error(est->ctx));
      if (blk) ins_pool_->free_.push_back(blk);
      return nullptr;
    }
    auto *u = new RBISIMUProcessStep(BatchArray((const double *) blk, PB_DEVICE), cov_gyro, cov_accel, cov_gyro_bias, cov_accel_bias, utime);
    u->owned_dev = std::make_shared<DeviceBlock>(ins_pool_, blk);
    u->valid_dev = mask_out;
    return u;
  }
  // valid [B] (per-filter host messages only): a filter without a message takes its step with dt = 0 -- with the sample it was
  // handed (its own last one) that leaves its state and covariance where they are
  RBISUpdateInterface *build(BatchArray gyro, BatchArray accel, double gyro_scale, bool accel_translate, double dt_,
                             int64_t utime, int B, const uint8_t *valid = nullptr)
  {
    if (gyro.mem == PB_DEVICE || accel.mem == PB_DEVICE || gyro.mem != accel.mem) {
      fprintf(stderr, "InsHandler: sensor-frame inputs must be host arrays (frame rotation is a host pass)\n");
      return nullptr;
    }
    // PB_HOST_BROADCAST: one IMU for every filter -> rotate once, hand a [7] block over (expanded on the device)
    const int mem = gyro.mem;
    if (mem == PB_HOST_BROADCAST) B = 1;
    std::vector<double> blk((size_t) 7 * B);
    const bool ident = ins_to_body.isIdentityRotation();
    PB_SHIM_PARALLEL_FOR
    for (int b = 0; b < B; b++) {
      double g[3] = { gyro.p[b] * gyro_scale, gyro.p[(size_t) B + b] * gyro_scale, gyro.p[(size_t) 2 * B + b] * gyro_scale };
      double a[3] = { accel.p[b], accel.p[(size_t) B + b], accel.p[(size_t) 2 * B + b] };
      double gb[3], ab[3];
      if (ident) {
        memcpy(gb, g, sizeof g);
        memcpy(ab, a, sizeof a);
      } else {
        bot_quat_rotate_to(ins_to_body.rot_quat, g, gb);
        bot_quat_rotate_to(ins_to_body.rot_quat, a, ab);
      }
      if (accel_translate)
        for (int i = 0; i < 3; i++) ab[i] += ins_to_body.trans_vec[i];
      for (int i = 0; i < 3; i++) {
        blk[(size_t) i * B + b] = gb[i];
        blk[(size_t) (3 + i) * B + b] = ab[i];
      }
      blk[(size_t) 6 * B + b] = (valid != nullptr && mem == PB_HOST && !valid[b]) ? 0.0 : dt_;
    }
    return new RBISIMUProcessStep(std::move(blk), cov_gyro, cov_accel, cov_gyro_bias, cov_accel_bias, utime, mem);
  }
};

// pronto_math.cpp:25-61 and pronto_conversions_lcm.hpp:38-87
inline void quat_to_euler(const double q[4], double &roll, double &pitch, double &yaw)
{
  const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  roll = atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
  pitch = asin(2 * (q0 * q2 - q3 * q1));
  yaw = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
}

class LegOdoCommon {
public:
  typedef enum { MODE_LIN_RATE, MODE_ROT_RATE, MODE_LIN_AND_ROT_RATE, MODE_POSITION_AND_LIN_RATE } LegOdoCommonMode;
  LegOdoCommonMode mode_;
  double R_legodo_xyz_, R_legodo_vxyz_, R_legodo_vang_, R_legodo_vxyz_uncertain_, R_legodo_vang_uncertain_;

  explicit LegOdoCommon(BotParam *param)
  {
    std::string mode_str = bot_param_get_str_or_fail(param, "state_estimator.legodo.mode");
    if (mode_str == "lin_rot_rate") mode_ = MODE_LIN_AND_ROT_RATE;
    else if (mode_str == "lin_rate") mode_ = MODE_LIN_RATE;
    else if (mode_str == "pos_and_lin_rate") mode_ = MODE_POSITION_AND_LIN_RATE;
    else {
      fprintf(stdout, "Legodo not understood! [LegOdoCommon].\n");  // rbis_legodo_common.cpp:19-21 (mode_ left unset there)
      mode_ = MODE_LIN_RATE;
    }
    R_legodo_xyz_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.r_xyz");
    R_legodo_vxyz_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.r_vxyz");
    R_legodo_vang_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.r_vang");
    R_legodo_vxyz_uncertain_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.r_vxyz_uncertain");
    R_legodo_vang_uncertain_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.r_vang_uncertain");
  }

  // getCovariance (rbis_legodo_common.cpp:34-88) for one filter: fills Rdiag[m], z_indices; returns m
  // getCovariance (rbis_legodo_common.cpp:34-91): diagonal of R and, if asked for, the index list
  int getCovariance(LegOdoCommonMode mode_current, bool delta_certain, double *Rdiag, std::vector<int> *z_indices) const
  {
    const double rv = delta_certain ? R_legodo_vxyz_ : R_legodo_vxyz_uncertain_;
    const double ra = delta_certain ? R_legodo_vang_ : R_legodo_vang_uncertain_;
    if (mode_current == MODE_LIN_AND_ROT_RATE) {
      for (int i = 0; i < 3; i++) { Rdiag[i] = bot_sq(rv); Rdiag[3 + i] = bot_sq(ra); }
      if (z_indices) *z_indices = { 3, 4, 5, 0, 1, 2 };
      return 6;
    } else if (mode_current == MODE_POSITION_AND_LIN_RATE) {
      for (int i = 0; i < 3; i++) { Rdiag[i] = bot_sq(R_legodo_xyz_); Rdiag[3 + i] = bot_sq(rv); }
      if (z_indices) *z_indices = { 9, 10, 11, 3, 4, 5 };
      return 6;
    }
    for (int i = 0; i < 3; i++) Rdiag[i] = bot_sq(rv);
    if (z_indices) *z_indices = { 3, 4, 5 };
    return 3;
  }

  // createMeasurement (rbis_legodo_common.cpp:110-169) for B filters.  Filters whose delta status is < 0 (the
  // handler's "return NULL", rbis_legodo_update.cpp:242-255) get mask 0.  In mode pos_and_lin_rate the reference
  // falls back to lin_rate per message when the position is invalid (:118-122): those filters take an m = 3 velocity
  // update, the others the m = 6 one (RBISEitherUpdate), so every filter gets exactly the reference's update and
  // log-likelihood increment.
  RBISUpdateInterface *createMeasurement(const msgs::legodo_delta_t *msg, int B) const
  {
    if (mode_ == MODE_POSITION_AND_LIN_RATE && msg->position_status != nullptr) {
      bool any_bad = false, any_good = false;
      for (int b = 0; b < B; b++) (msg->position_status[b] ? any_good : any_bad) = true;
      if (any_bad) {
        LegOdoCommon lin(*this);
        lin.mode_ = MODE_LIN_RATE;
        msgs::legodo_delta_t m3 = *msg;
        m3.position_status = nullptr;
        auto *fallback = static_cast<RBISIndexedMeasurement *>(lin.createMeasurement(&m3, B));
        for (int b = 0; b < B; b++) fallback->owned_mask[(size_t) b] &= (uint8_t) !msg->position_status[b];
        if (!any_good) return fallback;
        msgs::legodo_delta_t m6 = *msg;
        m6.position_status = nullptr;
        auto *full = static_cast<RBISIndexedMeasurement *>(createMeasurement(&m6, B));
        for (int b = 0; b < B; b++) full->owned_mask[(size_t) b] &= (uint8_t) (msg->position_status[b] != 0);
        return new RBISEitherUpdate(full, fallback);
      }
    }
    const double elapsed = (double) (msg->utime - msg->prev_utime) * 1E-6;  // pronto_conversions_lcm.hpp:45
    const int m = (mode_ == MODE_LIN_RATE) ? 3 : 6;
    std::vector<double> z((size_t) m * B), R((size_t) m * B);
    std::vector<uint8_t> mask(B);
    std::vector<int> idx;
    {
      double unused[6];
      getCovariance(mode_, true, unused, &idx);
    }
    PB_SHIM_PARALLEL_FOR
    for (int b = 0; b < B; b++) {
      const float st = msg->delta_status ? msg->delta_status[b] : 0.f;
      mask[b] = st >= 0;
      const bool certain = st < 0.5;                                        // :124-129
      double Rd[6];
      getCovariance(mode_, certain, Rd, nullptr);
      double vel[3];
      for (int i = 0; i < 3; i++) vel[i] = msg->delta_trans[(size_t) i * B + b] / elapsed;   // getDeltaAsVelocity :73
      if (mode_ == MODE_LIN_RATE) {
        for (int i = 0; i < 3; i++) { z[(size_t) i * B + b] = vel[i]; R[(size_t) i * B + b] = Rd[i]; }
      } else if (mode_ == MODE_LIN_AND_ROT_RATE) {
        double q[4] = { 1, 0, 0, 0 }, rpy[3];
        if (msg->delta_quat) for (int i = 0; i < 4; i++) q[i] = msg->delta_quat[(size_t) i * B + b];
        quat_to_euler(q, rpy[0], rpy[1], rpy[2]);                           // :141-147
        for (int i = 0; i < 3; i++) {
          z[(size_t) i * B + b] = vel[i];
          z[(size_t) (3 + i) * B + b] = rpy[i] / (((double) msg->utime - msg->prev_utime) / 1000000);
          R[(size_t) i * B + b] = Rd[i];
          R[(size_t) (3 + i) * B + b] = Rd[3 + i];
        }
      } else {
        for (int i = 0; i < 3; i++) {
          z[(size_t) i * B + b] = msg->position ? msg->position[(size_t) i * B + b] : 0.0;
          z[(size_t) (3 + i) * B + b] = vel[i];
          R[(size_t) i * B + b] = Rd[i];
          R[(size_t) (3 + i) * B + b] = Rd[3 + i];
        }
      }
    }
    return new RBISIndexedMeasurement(idx, std::move(z), std::move(R), PB_R_DIAG, std::move(mask),
                                      RBISUpdateInterface::legodo, msg->utime);
  }
};

// ModelClient (the reference hands its URDF string to kdl_parser, leg_estimate.cpp:68-73): here it carries what the
// forward kinematics of the two standing links needs of the robot model -- per leg the joints from the root link down to
// the standing link with their <origin>, <axis> and type -- filled in by hand or by fromURDFString(), a reader for exactly
// those URDF elements (<joint name type> with <parent link>, <child link>, <origin xyz rpy>, <axis xyz>).
class ModelClient {
public:
  struct Joint {
    std::string name;
    int type = 1;  // 0 fixed, 1 revolute / continuous, 2 prismatic (pb_legodo_set_chain)
    double xyz[3] = { 0, 0, 0 }, rpy[3] = { 0, 0, 0 }, axis[3] = { 1, 0, 0 };  // (URDF defaults)
  };
  std::vector<Joint> left_chain, right_chain;
  std::string urdf;
  std::string getURDFString() const { return urdf; }

  // the chain of joints from the root link to `link`; false if the link is not the child of any joint, a joint type is
  // not one the chain table knows (floating, planar), or an <origin> / <axis> attribute does not parse
  static bool chainTo(const std::string &urdf_with_comments, const std::string &link, std::vector<Joint> &chain)
  {
    struct J { Joint j; std::string parent, child; };
    std::vector<J> all;
    std::string urdf_xml;  // without <!-- comments -->: a commented-out joint must not be read
    for (size_t p = 0; p < urdf_with_comments.size();) {
      const size_t c0 = urdf_with_comments.find("<!--", p);
      urdf_xml.append(urdf_with_comments, p, (c0 == std::string::npos ? urdf_with_comments.size() : c0) - p);
      if (c0 == std::string::npos) break;
      const size_t c1 = urdf_with_comments.find("-->", c0 + 4);
      if (c1 == std::string::npos) break;
      p = c1 + 3;
    }
    // name = "value" (XML allows white space around '='; either quote)
    auto attr = [](const std::string &tag, const char *name, std::string &out) {
      const std::string key(name);
      auto space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r'; };
      size_t p = 0;
      while ((p = tag.find(key, p)) != std::string::npos) {
        const bool starts_word = p == 0 || !(isalnum((unsigned char) tag[p - 1]) || tag[p - 1] == '_' || tag[p - 1] == ':');
        p += key.size();
        if (!starts_word) continue;
        size_t v = p;
        while (v < tag.size() && space(tag[v])) v++;
        if (v >= tag.size() || tag[v] != '=') continue;
        v++;
        while (v < tag.size() && space(tag[v])) v++;
        if (v >= tag.size() || (tag[v] != '"' && tag[v] != '\'')) continue;
        const char qc = tag[v];
        const size_t e = tag.find(qc, v + 1);
        if (e == std::string::npos) return false;
        out = tag.substr(v + 1, e - v - 1);
        return true;
      }
      return false;
    };
    // exactly three numbers (urdfdom's Vector3::init throws on anything else; the reference exits when kdl_parser fails)
    auto three = [](const std::string &v, double *o) {
      char extra;
      return sscanf(v.c_str(), " %lf %lf %lf %c", o, o + 1, o + 2, &extra) == 3;
    };
    size_t pos = 0;
    while ((pos = urdf_xml.find("<joint", pos)) != std::string::npos) {
      const size_t head_end = urdf_xml.find('>', pos);
      if (head_end == std::string::npos) break;
      const std::string head = urdf_xml.substr(pos, head_end - pos + 1);
      const char nx = urdf_xml[pos + 6];
      if (!(nx == ' ' || nx == '\t' || nx == '\n' || nx == '\r')) { pos = head_end; continue; }  // e.g. <joint_properties>
      std::string type;
      J j;
      if (!attr(head, "name", j.j.name) || !attr(head, "type", type) || head[head.size() - 2] == '/') { pos = head_end; continue; }  // (a <transmission>'s <joint name=.../>)
      const size_t end = urdf_xml.find("</joint>", head_end);
      if (end == std::string::npos) break;
      const std::string body = urdf_xml.substr(head_end + 1, end - head_end - 1);
      pos = end;
      j.j.type = (type == "fixed") ? 0 : (type == "revolute" || type == "continuous") ? 1 : (type == "prismatic") ? 2 : -1;
      auto child_tag = [&](const char *name, std::string &tag) {
        const size_t p = body.find(std::string("<") + name);
        if (p == std::string::npos) return false;
        const size_t e = body.find('>', p);
        if (e == std::string::npos) return false;
        tag = body.substr(p, e - p + 1);
        return true;
      };
      std::string tag, v;
      if (!child_tag("parent", tag) || !attr(tag, "link", j.parent)) continue;
      if (!child_tag("child", tag) || !attr(tag, "link", j.child)) continue;
      // an <origin> / <axis> attribute that is present but does not hold three numbers is an error, not a default
      if (child_tag("origin", tag)) {
        if (attr(tag, "xyz", v) && !three(v, j.j.xyz)) return false;
        if (attr(tag, "rpy", v) && !three(v, j.j.rpy)) return false;
      }
      if (child_tag("axis", tag) && attr(tag, "xyz", v) && !three(v, j.j.axis)) return false;
      all.push_back(j);
    }
    chain.clear();
    std::string cur = link;
    for (size_t guard = 0; guard <= all.size(); guard++) {
      const J *up = nullptr;
      for (const J &j : all)
        if (j.child == cur) up = &j;
      if (up == nullptr) break;  // cur is the root link
      if (up->j.type < 0) return false;
      chain.insert(chain.begin(), up->j);
      cur = up->parent;
    }
    return !chain.empty();
  }
  bool fromURDFString(const std::string &urdf_xml, const std::string &left_standing_link, const std::string &right_standing_link)
  {
    urdf = urdf_xml;
    return chainTo(urdf_xml, left_standing_link, left_chain) && chainTo(urdf_xml, right_standing_link, right_chain);
  }
};

// LegOdoHandler (rbis_legodo_update.cpp:10-280).  processMessage(joint_state_t) is the reference's handler: torque
// adjustment, forward kinematics, leg_estimate::updateOdometry and LegOdoCommon::createMeasurement -- all of it per filter
// on the device, because its world_to_body_ is the filter's own head pose (:214-229).  processMessageFeet takes the two
// body-to-foot transforms instead of the joint angles (a caller with its own kinematics), processMessageDelta the finished
// increment.
class LegOdoHandler {
public:
  LegOdoCommon *leg_odo_common_;
  int zero_initial_velocity;      // rbis_legodo_update.cpp:58,265-269
  bool force_torque_init_ = true; // rbis_legodo_update.cpp:98,197,208-211: nothing is integrated before the first
                                  // force/torque message; the joint-state path starts with false like the reference (its foot
                                  // forces come from forceTorqueHandler); the increment / foot-state paths, whose callers
                                  // already hold the foot sensing, start with true
  explicit LegOdoHandler(BotParam *param) : leg_odo_common_(new LegOdoCommon(param)), zero_initial_velocity(0), param_(param)
  {
    auto it = param->kv.find("state_estimator.legodo.zero_initial_velocity");
    if (it != param->kv.end()) zero_initial_velocity = atoi(it->second.c_str());
  }
  // the reference's constructor (lcm_recv, lcm_pub, param, model, frames) without the LCM objects and BotFrames
  LegOdoHandler(BotParam *param, const ModelClient *model) : LegOdoHandler(param)
  {
    model_ = model;
    force_torque_init_ = false;
    // leg_estimate's constructor (leg_estimate.cpp:29-142)
    const std::string init_mode = bot_param_get_str_or_fail(param, "state_estimator.legodo.initialization_mode");
    if (init_mode != "zero") fprintf(stdout, "Leg Odometry Initialize Mode: %s (only \"zero\" moves the pose, leg_estimate.cpp:172-190)\n", init_mode.c_str());
    // leg_estimate.cpp:43-61: "lowpass" | "kalman" | anything else = no filtering
    const std::string filt = bot_param_get_str_or_fail(param, "state_estimator.legodo.filter_joint_positions");
    if (filt == "lowpass") {
      filter_joint_positions_ = 1;
    } else if (filt == "kalman") {
      filter_joint_positions_ = 2;
      joint_process_noise_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.joint_process_noise");
      joint_observation_noise_ = bot_param_get_double_or_fail(param, "state_estimator.legodo.joint_observation_noise");
    }
    use_torque_adjustment_ = bot_param_get_boolean_or_fail(param, "state_estimator.legodo.torque_adjustment");
    if (use_torque_adjustment_) {  // rbis_legodo_update.cpp:29-53
      const std::string &names = bot_param_get_raw_or_fail(param, "state_estimator.legodo.adjustment_joints");
      size_t p = 0;
      while (p < names.size()) {
        while (p < names.size() && (names[p] == ' ' || names[p] == ',' || names[p] == '[' || names[p] == ']' || names[p] == '"')) p++;
        size_t e = p;
        while (e < names.size() && names[e] != ' ' && names[e] != ',' && names[e] != ']' && names[e] != '"') e++;
        if (e > p) adjustment_joints_.push_back(names.substr(p, e - p));
        p = e;
      }
      std::vector<double> g(adjustment_joints_.size());
      bot_param_get_double_array_or_fail(param, "state_estimator.legodo.adjustment_gain", g.data(), (int) g.size());
      for (double v : g) adjustment_gain_.push_back((float) v);
    }
  }
  ~LegOdoHandler() { delete leg_odo_common_; }
  void forceTorqueHandler() { force_torque_init_ = true; }
  // rbis_legodo_update.cpp:195-204: keep the last force/torque message; processMessage reads |force z| of the two feet
  void forceTorqueHandler(const msgs::six_axis_force_torque_array_t *msg, int B)
  {
    const size_t n = (msg->force_z.mem == PB_HOST_BROADCAST) ? 2 : (size_t) 2 * B;
    foot_force_.resize(n);
    for (size_t i = 0; i < n; i++) foot_force_[i] = (float) fabs(msg->force_z.p[i]);  // FootSensing(fabs(...)) (:234-235), float members
    foot_force_mem_ = (msg->force_z.mem == PB_HOST_BROADCAST) ? PB_HOST_BROADCAST : PB_HOST;
    foot_force_dev_ = nullptr;
    force_torque_init_ = true;
  }
  // device-resident replays: |force z| [2][B] floats in HBM.  Like the device arrays of a PB_DEVICE joint_state_t it must stay
  // valid until the update made from it has been APPLIED: with fuse_ins_legodo the odometry of a message runs inside the pair
  // kernel of the addUpdate call that follows processMessage (same dispatch, FrontEnd::addSensor), so "until the next message"
  // is enough -- the estimator makes a deferred measurement at once whenever it holds the pair back or does not roll forward
  void forceTorqueDevice(const float *abs_force_z)
  {
    foot_force_dev_ = abs_force_z;
    force_torque_init_ = true;
  }
  // rbis_legodo_update.cpp:190-193
  void controllerInputHandler(const msgs::controller_foot_contact_t *msg)
  {
    n_control_contacts_[0] = msg->num_left_foot_contacts;
    n_control_contacts_[1] = msg->num_right_foot_contacts;
    control_contacts_dirty_ = true;
  }

  BotParam *param_ = nullptr;
  const ModelClient *model_ = nullptr;
  int filter_joint_positions_ = 0;   // 0 none, 1 lowpass, 2 kalman (leg_estimate.cpp:43-61)
  double joint_process_noise_ = 0, joint_observation_noise_ = 0;
  std::shared_ptr<DevicePool> jf_pool_;   // filtered joint blocks [rows][B] floats (per-filter joint states)
  std::shared_ptr<DevicePool> ff_pool_;   // foot forces [2][B] floats uploaded for device-resident joint blocks
  bool use_torque_adjustment_ = false;
  std::vector<std::string> adjustment_joints_;
  std::vector<float> adjustment_gain_;
  std::vector<float> foot_force_;
  int foot_force_mem_ = PB_HOST;
  const float *foot_force_dev_ = nullptr;
  int32_t n_control_contacts_[2] = { -1, -1 };
  bool control_contacts_dirty_ = false;
  std::shared_ptr<DevicePool> pool_;
  bool legodo_ready_ = false, chain_ready_ = false;
  std::vector<std::string> chain_names_;  // the joint_name list the chain rows were resolved against

  // leg_estimate's constructor parameters (leg_estimate.cpp:63,93-121) -> the device side, once
  void initLegEstimate(MavStateEstimator *est)
  {
    const int B = est->B;
    const double lt = bot_param_get_double_or_fail(param_, "state_estimator.legodo.schmitt_low_threshold");
    const double ht = bot_param_get_double_or_fail(param_, "state_estimator.legodo.schmitt_high_threshold");
    const int64_t ld = (int64_t) bot_param_get_double_or_fail(param_, "state_estimator.legodo.schmitt_low_delay");
    const int64_t hd = (int64_t) bot_param_get_double_or_fail(param_, "state_estimator.legodo.schmitt_high_delay");
    const bool fce = bot_param_get_boolean_or_fail(param_, "state_estimator.legodo.filter_contact_events");
    bool bad = pb_legodo_init(est->ctx, lt, ht, ld, hd, fce) != PB_OK;
    // the contact mode keys are read by leg_estimate's constructor, i.e. only when the handler was given a robot model
    if (!bad && model_ != nullptr) {
      const bool standing = bot_param_get_str_or_fail(param_, "state_estimator.legodo.init_contact_mode") == "standing";
      const double total_force = bot_param_get_double_or_fail(param_, "state_estimator.legodo.total_force");
      const double level = bot_param_get_double_or_fail(param_, "state_estimator.legodo.standing_schmitt_level");
      const bool ctrl = bot_param_get_boolean_or_fail(param_, "state_estimator.legodo.use_controller_input");
      bad = pb_legodo_set_contact_mode(est->ctx, standing, total_force, level, ctrl) != PB_OK;
    }
    // "ignore the calculated velocity at launch" is counted per filter on the device (valid ticks only, like :243-268)
    if (!bad) bad = pb_legodo_set_zero_initial_velocity(est->ctx, zero_initial_velocity) != PB_OK;
    if (bad) {
      fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
      exit(1);
    }
    // per update: the measurement block -- z and the diagonal of R, [6][B] (lin_rate) or [12][B] doubles -- and its mask(s) [2][B]
    pool_ = std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(double) * 12 * (size_t) B + 2 * (size_t) B);
    legodo_ready_ = true;
  }
  // fksolver_->JntToCart looks joints up by name (leg_estimate.cpp:432-436): resolve the chain's names against the message's
  void initChain(const msgs::joint_state_t *msg, MavStateEstimator *est)
  {
    std::vector<int> type, row;
    std::vector<double> org, axis;
    std::vector<float> gain;
    for (const std::vector<ModelClient::Joint> *chain : { &model_->left_chain, &model_->right_chain })
      for (const ModelClient::Joint &j : *chain) {
        int r = 0;
        if (j.type != 0) {
          const auto it = std::find(msg->joint_name.begin(), msg->joint_name.end(), j.name);
          if (it == msg->joint_name.end()) {  // KDL treats a missing joint as an error; the reference then exit(-1)s (:438-441)
            fprintf(stderr, "Error: could not calculate forward kinematics! (joint %s is not in the joint state)\n", j.name.c_str());
            exit(-1);
          }
          r = (int) (it - msg->joint_name.begin());
        }
        type.push_back(j.type);
        row.push_back(r);
        for (int i = 0; i < 3; i++) org.push_back(j.xyz[i]);
        for (int i = 0; i < 3; i++) org.push_back(j.rpy[i]);
        for (int i = 0; i < 3; i++) axis.push_back(j.axis[i]);
        float g = 0.0f;
        if (use_torque_adjustment_) {
          const auto it = std::find(adjustment_joints_.begin(), adjustment_joints_.end(), j.name);
          if (it != adjustment_joints_.end()) g = adjustment_gain_[(size_t) (it - adjustment_joints_.begin())];
        }
        gain.push_back(g);
      }
    if (use_torque_adjustment_)
      for (const std::string &n : adjustment_joints_)
        if (std::find(msg->joint_name.begin(), msg->joint_name.end(), n) == msg->joint_name.end()) {
          fprintf(stdout, "TorqueAdjustment: %s joint not found\n", n.c_str());  // torque_adjustment.cpp:44-46
          exit(-1);
        }
    if (pb_legodo_set_chain(est->ctx, (int) model_->left_chain.size(), (int) model_->right_chain.size(), type.data(), row.data(), org.data(),
                            axis.data(), use_torque_adjustment_ ? gain.data() : nullptr) != PB_OK) {
      fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
      exit(1);
    }
    if (filter_joint_positions_ != 0) {
      // SimpleKalmanFilter(joint_process_noise, joint_observation_noise) (leg_estimate.cpp:56): the two values land in
      // process_noise_pos_ and process_noise_vel_, observation_noise_ keeps its default 5E-4 (simple_kalman_filter.hpp:15).
      // A new joint order (a new chain) starts the filters over.
      if (pb_joint_filter_init(est->ctx, filter_joint_positions_, joint_process_noise_, joint_observation_noise_, 5E-4) != PB_OK) {
        fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
        exit(1);
      }
      // per-filter joint states: the filtered block [rows][B] and, for host messages, the two foot forces [2][B] behind it
      jf_pool_ = std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(float) * (msg->joint_name.size() + 2) * (size_t) est->B);
    }
    chain_names_ = msg->joint_name;
    chain_ready_ = true;
  }

  // one message's leg inputs, kept alive by the update made from them: broadcast values are copied (they are a few dozen
  // numbers), device arrays are referenced (the caller keeps them, as everywhere), host blocks are used before the handler
  // returns
  struct LegMsg {
    int kind = 0, mem = PB_HOST, rows = 0;       // kind 0: foot poses, 1: joint state
    int64_t utime = 0;
    const float *jp = nullptr, *je = nullptr, *ff = nullptr;
    const double *feet = nullptr, *forces = nullptr;
    std::vector<float> own_f;
    std::vector<double> own_d;
    std::shared_ptr<DeviceBlock> filtered;   // the joint filters' output block, when they run on the device
    std::shared_ptr<DeviceBlock> forces_dev; // the foot forces of a host force/torque message uploaded next to a device joint block
    std::vector<int64_t> own_ut;             // per-filter message times / validity (independent log segments), copied: the odometry
    std::vector<uint8_t> own_valid;          // may run after the handler has returned
    const int64_t *dev_ut = nullptr;         // ... or device arrays, referenced like every PB_DEVICE input (valid until the update
    const uint8_t *dev_valid = nullptr;      // made from this message has been applied)
    double r = 0, ru = 0;
    int times(pb_ctx *ctx) const
    {
      if (dev_ut != nullptr || dev_valid != nullptr) return pb_legodo_set_message_times(ctx, dev_ut, dev_valid, PB_DEVICE);
      if (own_ut.empty() && own_valid.empty()) return PB_OK;
      return pb_legodo_set_message_times(ctx, own_ut.empty() ? nullptr : own_ut.data(), own_valid.empty() ? nullptr : own_valid.data(), PB_HOST);
    }
    // odometry alone (imu == NULL) or slaved to the state after `imu`; outputs as pb_legodo_update_joints
    int odometry(pb_ctx *ctx, const BatchArray *imu, double *o_delta, double *o_status, double *d_lo, uint8_t *d_mask, double *o_pos,
                 uint8_t *o_pos_ok) const
    {
      const int trc = times(ctx);
      if (trc != PB_OK) return trc;
      if (kind == 1)
        return pb_legodo_update_joints(ctx, imu ? imu->p : nullptr, imu ? imu->mem : PB_DEVICE, utime, rows, jp, je, ff, mem, 0, r, ru, o_delta,
                                       o_status, d_lo, d_mask, o_pos, o_pos_ok);
      return imu ? pb_legodo_update_after_predict(ctx, imu->p, imu->mem, utime, feet, forces, mem, 0, r, ru, o_delta, o_status, d_lo, d_mask)
                 : pb_legodo_update(ctx, utime, feet, forces, mem, 0, r, ru, o_delta, o_status, d_lo, d_mask);
    }
    // INS step + odometry + update in one call (one kernel where the library has it)
    int pair(pb_ctx *ctx, const RBISIMUProcessStep *imu, double *d_lo, uint8_t *d_mask) const
    {
      const double q[4] = { imu->q_gyro, imu->q_accel, imu->q_gyro_bias, imu->q_accel_bias };
      const int trc = times(ctx);
      if (trc != PB_OK) return trc;
      imu->announce(ctx);
      if (kind == 1)
        return pb_step_legodo_joints(ctx, imu->imu_block.p, imu->imu_block.mem, q, utime, rows, jp, je, ff, mem, r, ru, d_lo, d_mask);
      return pb_step_legodo_feet(ctx, imu->imu_block.p, imu->imu_block.mem, q, utime, feet, forces, mem, r, ru, d_lo, d_mask);
    }
  };

  // what the device-side entry points share: the pending-IMU decision, the odometry and the measurement LegOdoCommon forms
  // from its result
  RBISUpdateInterface *odometryUpdate(MavStateEstimator *est, std::shared_ptr<LegMsg> lm)
  {
    const int B = est->B;
    const int64_t utime = lm->utime;
    const LegOdoCommon *lc = leg_odo_common_;
    lm->r = lc->R_legodo_vxyz_;
    lm->ru = lc->R_legodo_vxyz_uncertain_;
    // The odometry reads the head pose.  An INS step that fuse_ins_legodo is holding back is either applied first, or -- when
    // the measurement made here will pair with it -- left pending: the pair then runs as ONE kernel that does the INS step, the
    // odometry slaved to the pose after it and the update in LegOdoCommon's mode (pb_step_legodo_joints / _feet).
    RBISIMUProcessStep *ahead = est->pendingImu();
    if (ahead != nullptr && ahead->imu_block.mem == PB_HOST && lm->mem == PB_HOST) ahead = nullptr;  // one host staging area
    if (ahead == nullptr) est->flushPending();
    if (!legodo_ready_) initLegEstimate(est);
    if (control_contacts_dirty_) {
      if (pb_legodo_set_control_contacts(est->ctx, n_control_contacts_, PB_HOST_BROADCAST) != PB_OK) fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
      control_contacts_dirty_ = false;
    }
    bool fresh = false;
    void *blk = pool_->get(fresh);
    if (blk == nullptr) {
      fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
      return nullptr;
    }
    auto block = std::make_shared<DeviceBlock>(pool_, blk);
    double *d_lo = (double *) blk;
    uint8_t *d_mask = (uint8_t *) (d_lo + (size_t) 6 * B);   // lin_rate: z [3][B] | R diagonal [3][B] | mask [B]
    // LegOdoCommon::createMeasurement runs on the DEVICE for every mode (pb_legodo_set_measurement_mode): the increment, the status
    // and the position never cross PCIe.  A foot-state message has no pelvis position: mode pos_and_lin_rate then is the
    // reference's own fall-back to lin_rate (rbis_legodo_common.cpp:118-122) for every filter.
    const int dmode = (lc->mode_ == LegOdoCommon::MODE_LIN_AND_ROT_RATE) ? 1 : ((lc->mode_ == LegOdoCommon::MODE_POSITION_AND_LIN_RATE && lm->kind == 1) ? 2 : 0);
    if (dmode != device_meas_mode_) {
      if (pb_legodo_set_measurement_mode(est->ctx, dmode, lc->R_legodo_xyz_, lc->R_legodo_vang_, lc->R_legodo_vang_uncertain_) != PB_OK) {
        fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
        return nullptr;
      }
      device_meas_mode_ = dmode;
    }
    const bool lin = dmode == 0;
    if (!lin) d_mask = (uint8_t *) (d_lo + (size_t) 12 * B);   // z [6][B] | R diagonal [6][B] | masks [2][B]
    // the measurement can be made later, inside the step kernel, when its inputs outlive this call
    const bool defer = ahead != nullptr && one_kernel_pairs && lm->mem != PB_HOST;
    if (!defer) {
      const int lrc = lm->odometry(est->ctx, ahead ? &ahead->imu_block : nullptr, nullptr, nullptr, d_lo, d_mask, nullptr, nullptr);
      if (lrc != PB_OK) {
        fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
        return nullptr;
      }
    }
    prev_legodo_utime_ = utime;
    if (lin) {
      auto *u = new RBISIndexedMeasurement(RBIS::velocityInds(), BatchArray(d_lo, PB_DEVICE), d_lo + (size_t) 3 * B, PB_R_DIAG, d_mask,
                                           RBISUpdateInterface::legodo, utime);
      u->owned_dev = block;
      if (defer) {
        u->pair_kernel = [lm, d_lo, d_mask](pb_ctx *ctx, const RBISIMUProcessStep *imu, bool keep) {
          return lm->pair(ctx, imu, keep ? d_lo : nullptr, keep ? d_mask : nullptr);
        };
        u->make_measurement = [lm, d_lo, d_mask](pb_ctx *ctx, const RBISIMUProcessStep *ahead_of) {
          return lm->odometry(ctx, ahead_of ? &ahead_of->imu_block : nullptr, nullptr, nullptr, d_lo, d_mask, nullptr, nullptr);
        };
      }
      return u;
    }
    std::vector<int> idx;
    {
      double unused[6];
      lc->getCovariance(lc->mode_, true, unused, &idx);
    }
    auto *full = new RBISIndexedMeasurement(idx, BatchArray(d_lo, PB_DEVICE), d_lo + (size_t) 6 * B, PB_R_DIAG, d_mask, RBISUpdateInterface::legodo, utime);
    full->owned_dev = block;
    if (defer) {  // (the block holds the rows and masks of BOTH halves of mode 2's either-update)
      full->pair_kernel = [lm, d_lo, d_mask](pb_ctx *ctx, const RBISIMUProcessStep *imu, bool keep) {
        return lm->pair(ctx, imu, keep ? d_lo : nullptr, keep ? d_mask : nullptr);
      };
      full->make_measurement = [lm, d_lo, d_mask](pb_ctx *ctx, const RBISIMUProcessStep *ahead_of) {
        return lm->odometry(ctx, ahead_of ? &ahead_of->imu_block : nullptr, nullptr, nullptr, d_lo, d_mask, nullptr, nullptr);
      };
    }
    if (dmode == 1) return full;
    // pos_and_lin_rate: the filters whose position is not valid take the lin_rate update on the velocity rows of the same block
    auto *fallback = new RBISIndexedMeasurement(RBIS::velocityInds(), BatchArray(d_lo + (size_t) 3 * B, PB_DEVICE), d_lo + (size_t) 9 * B, PB_R_DIAG,
                                                d_mask + (size_t) B, RBISUpdateInterface::legodo, utime);
    fallback->owned_dev = block;
    return new RBISEitherUpdate(full, fallback);
  }
  int device_meas_mode_ = 0;
  int64_t prev_legodo_utime_ = 0;
  // false: always make the measurement in the handler (k_legodo), then the fused step reads it -- round 2's two launches
  bool one_kernel_pairs = true;

  // the reference's handler (rbis_legodo_update.cpp:206-280)
  RBISUpdateInterface *processMessage(const msgs::joint_state_t *msg, MavStateEstimator *est)
  {
    if (!force_torque_init_) {
      fprintf(stdout, "Force/Torque message not received yet, not integrating leg odometry =========================\n");
      return nullptr;
    }
    if (model_ == nullptr) {
      fprintf(stderr, "LegOdoHandler: a joint_state_t needs the robot model (construct the handler with a ModelClient)\n");
      exit(1);
    }
    if (!legodo_ready_) initLegEstimate(est);
    if (!chain_ready_ || msg->joint_name != chain_names_) initChain(msg, est);
    const float *forces = foot_force_dev_ ? foot_force_dev_ : foot_force_.data();
    int fmem = foot_force_dev_ ? PB_DEVICE : foot_force_mem_;
    auto lm = std::make_shared<LegMsg>();
    if (fmem == PB_HOST && msg->mem == PB_DEVICE) {
      // a device-resident joint block with the force/torque message still on the host (SegmentBatcher uploads the big blocks
      // itself): the 2 x B floats follow the joints into HBM, in a block this message owns
      if (!ff_pool_) ff_pool_ = std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(float) * 2 * (size_t) est->B);
      bool fresh = false;
      void *blk = ff_pool_->get(fresh);
      if (blk == nullptr || pb_memcpy_h2d(est->ctx, blk, forces, sizeof(float) * 2 * (size_t) est->B) != PB_OK) {
        fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
        return nullptr;
      }
      lm->forces_dev = std::make_shared<DeviceBlock>(ff_pool_, blk);
      forces = (const float *) blk;
      fmem = PB_DEVICE;
    }
    if (fmem != msg->mem) {
      fprintf(stderr, "LegOdoHandler: the joint state and the force/torque message must live in the same memory space\n");
      return nullptr;
    }
    lm->kind = 1;
    lm->mem = msg->mem;
    lm->rows = (int) msg->joint_name.size();
    lm->utime = msg->utime;
    if (msg->times_mem == PB_DEVICE) {
      lm->dev_ut = msg->utimes;
      lm->dev_valid = msg->valid;
    } else {
      if (msg->utimes != nullptr) lm->own_ut.assign(msg->utimes, msg->utimes + est->B);
      if (msg->valid != nullptr) lm->own_valid.assign(msg->valid, msg->valid + est->B);
    }
    if (msg->utimes != nullptr && filter_joint_positions_ == 2) {
      fprintf(stderr, "LegOdoHandler: per-filter message times with filter_joint_positions = kalman are not supported (the joint Kalman filters keep one clock per batch)\n");
      return nullptr;
    }
    const float *eff = use_torque_adjustment_ ? msg->joint_effort : nullptr;
    if (filter_joint_positions_ != 0) {
      // leg_estimate.cpp:411-428, after the torque adjustment (rbis_legodo_update.cpp:231-241): pb_joint_filter does both, the
      // kinematics then read the filtered block without an effort
      if (filter_joint_positions_ == 2 && msg->joint_velocity == nullptr) {
        fprintf(stderr, "LegOdoHandler: filter_joint_positions = kalman reads joint_velocity\n");
        return nullptr;
      }
      float *out = nullptr;
      if (msg->mem == PB_HOST_BROADCAST) {
        lm->own_f.assign((size_t) lm->rows + 2, 0.0f);
        out = lm->own_f.data();
      } else {
        bool fresh = false;
        void *blk = jf_pool_->get(fresh);
        if (blk == nullptr) {
          fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
          return nullptr;
        }
        lm->filtered = std::make_shared<DeviceBlock>(jf_pool_, blk);
        out = (float *) blk;
      }
      if (pb_joint_filter(est->ctx, msg->utime, lm->rows, msg->joint_position, msg->joint_velocity, eff, msg->mem, out) != PB_OK) {
        fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
        return nullptr;
      }
      lm->jp = out;
      lm->je = nullptr;
      if (msg->mem == PB_HOST_BROADCAST) {
        lm->own_f[(size_t) lm->rows] = forces[0];
        lm->own_f[(size_t) lm->rows + 1] = forces[1];
        lm->ff = lm->own_f.data() + lm->rows;
      } else {
        lm->ff = forces;
        if (msg->mem == PB_HOST) {  // the filtered joints are in HBM: the foot forces of a host message follow them there
          float *dff = out + (size_t) lm->rows * est->B;
          if (pb_memcpy_h2d(est->ctx, dff, forces, sizeof(float) * 2 * (size_t) est->B) != PB_OK) {
            fprintf(stderr, "LegOdoHandler: %s\n", pb_last_error(est->ctx));
            return nullptr;
          }
          lm->ff = dff;
          lm->mem = PB_DEVICE;
        }
      }
    } else if (msg->mem == PB_HOST_BROADCAST) {  // one robot's message: own the few numbers
      lm->own_f.assign(msg->joint_position, msg->joint_position + lm->rows);
      if (eff) lm->own_f.insert(lm->own_f.end(), eff, eff + lm->rows);
      lm->own_f.insert(lm->own_f.end(), forces, forces + 2);
      lm->jp = lm->own_f.data();
      lm->je = eff ? lm->own_f.data() + lm->rows : nullptr;
      lm->ff = lm->own_f.data() + (eff ? 2 : 1) * (size_t) lm->rows;
    } else {
      lm->jp = msg->joint_position;
      lm->je = eff;
      lm->ff = forces;
    }
    return odometryUpdate(est, lm);
  }

  // the same from what forward kinematics produces (msgs::foot_state_t); no pelvis position: mode pos_and_lin_rate falls back
  // to lin_rate like the reference does while its world constraint is not initialised (rbis_legodo_common.cpp:118-122)
  RBISUpdateInterface *processMessageFeet(const msgs::foot_state_t *msg, MavStateEstimator *est)
  {
    if (!force_torque_init_) {
      fprintf(stdout, "Force/Torque message not received yet, not integrating leg odometry =========================\n");
      return nullptr;
    }
    auto lm = std::make_shared<LegMsg>();
    lm->kind = 0;
    lm->mem = msg->feet.mem;
    lm->utime = msg->utime;
    if (msg->feet.mem == PB_HOST_BROADCAST) {
      lm->own_d.assign(msg->feet.p, msg->feet.p + 14);
      lm->own_d.insert(lm->own_d.end(), msg->forces.p, msg->forces.p + 2);
      lm->feet = lm->own_d.data();
      lm->forces = lm->own_d.data() + 14;
    } else {
      lm->feet = msg->feet.p;
      lm->forces = msg->forces.p;
    }
    return odometryUpdate(est, lm);
  }

  // from a finished increment (a caller that runs its own leg_estimate): createMeasurement and the handler's gates
  RBISUpdateInterface *processMessageDelta(const msgs::legodo_delta_t *msg, MavStateEstimator *est)
  {
    if (!force_torque_init_) {
      fprintf(stdout, "Force/Torque message not received yet, not integrating leg odometry =========================\n");
      return nullptr;
    }
    // (:243-255) "Leg Odometry is not valid not integrating": no update and no decrement when no filter has a valid increment
    if (msg->delta_status != nullptr) {
      bool any = false;
      for (int b = 0; b < est->B && !any; b++) any = msg->delta_status[b] >= 0;
      if (!any) return nullptr;
    }
    // "Ignore the calculated velocity at launch" (:264-269): decrement FIRST, then compare -- N zeroes N-1 ticks
    zero_initial_velocity--;
    if (zero_initial_velocity > 0) {
      std::vector<double> zero((size_t) 3 * est->B, 0.0);
      msgs::legodo_delta_t m2 = *msg;
      m2.delta_trans = zero.data();   // odo_delta.setIdentity()
      m2.delta_quat = nullptr;
      if (msg->position != nullptr) m2.position = zero.data();  // odo_position.setIdentity(), status passed on as is
      return leg_odo_common_->createMeasurement(&m2, est->B);
    }
    return leg_odo_common_->createMeasurement(msg, est->B);
  }
};

// The reference's update objects own their measurement (Eigen copies in their constructors); so do ours for host data,
// which the history may re-apply long after the message buffer is gone.  Device blocks are referenced, not copied: the
// caller keeps them alive for utime_history_span.  cov_diag must outlive the update (handlers are app-lifetime singletons).
inline RBISIndexedMeasurement *makeIndexedMeasurement(const std::vector<int> &idx, BatchArray z, int B,
                                                      const std::vector<double> &cov_diag, const uint8_t *mask,
                                                      RBISUpdateInterface::sensor_enum sensor, int64_t utime)
{
  if (z.mem == PB_DEVICE) return new RBISIndexedMeasurement(idx, z, cov_diag.data(), PB_R_DIAG_BROADCAST, mask, sensor, utime);
  const size_t per = (z.mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
  std::vector<double> zc(z.p, z.p + idx.size() * per);
  std::vector<uint8_t> mc;
  if (mask != nullptr && z.mem == PB_HOST) mc.assign(mask, mask + B);
  auto *u = new RBISIndexedMeasurement(idx, std::move(zc), std::vector<double>(cov_diag), PB_R_DIAG_BROADCAST, std::move(mc), sensor,
                                       utime);
  u->measurement.mem = z.mem;
  return u;
}

class ScanMatcherHandler {
public:
  typedef enum { MODE_POSITION, MODE_POSITION_YAW, MODE_VELOCITY, MODE_VELOCITY_YAW, MODE_YAW } ScanMatchingMode;
  ScanMatchingMode mode;
  std::vector<int> z_indices;
  std::vector<double> cov_scan_match;  // diagonal

  explicit ScanMatcherHandler(BotParam *param)
  {
    std::string mode_str = bot_param_get_str_or_fail(param, "state_estimator.scan_matcher.mode");
    if (mode_str == "position") mode = MODE_POSITION;
    else if (mode_str == "position_yaw") mode = MODE_POSITION_YAW;
    else if (mode_str == "velocity") mode = MODE_VELOCITY;
    else if (mode_str == "velocity_yaw") mode = MODE_VELOCITY_YAW;
    else if (mode_str == "yaw") mode = MODE_YAW;
    else mode = MODE_VELOCITY;  // sensor_handlers.cpp:636-639
    if (mode == MODE_POSITION || mode == MODE_POSITION_YAW) {
      double rxy = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_pxy");
      double rz = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_pz");
      cov_scan_match = { bot_sq(rxy), bot_sq(rxy), bot_sq(rz) };
      z_indices = RBIS::positionInds();
    } else if (mode == MODE_YAW) {
      double ry = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_yaw");
      cov_scan_match = { bot_sq(bot_to_radians(ry)) };
      z_indices = { RBIS::chi_ind + 2 };
    } else {
      double rxy = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_vxy");
      double rz = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_vz");
      cov_scan_match = { bot_sq(rxy), bot_sq(rxy), bot_sq(rz) };
      z_indices = RBIS::velocityInds();
    }
    if (mode == MODE_POSITION_YAW || mode == MODE_VELOCITY_YAW) {
      double ry = bot_param_get_double_or_fail(param, "state_estimator.scan_matcher.r_yaw");
      cov_scan_match.push_back(bot_sq(bot_to_radians(ry)));
      z_indices.push_back(RBIS::chi_ind + 2);
    }
  }

  // sensor_handlers.cpp:689-724.  For the *_yaw modes z is [4][B] whose 4th row is ignored (chi index), so the
  // message's pos/vel block is used in place when it already has 4 rows of storage; otherwise it is padded on the host.
  RBISUpdateInterface *processMessage(const msgs::pose_t *msg, MavStateEstimator *est)
  {
    const int B = est->B;
    if (mode == MODE_POSITION)
      return makeIndexedMeasurement(RBIS::positionInds(), msg->pos, B, cov_scan_match, msg->valid, RBISUpdateInterface::scan_matcher,
                                    msg->utime);
    if (mode == MODE_VELOCITY)
      return makeIndexedMeasurement(RBIS::velocityInds(), msg->vel, B, cov_scan_match, msg->valid, RBISUpdateInterface::scan_matcher,
                                    msg->utime);
    const BatchArray src = (mode == MODE_POSITION_YAW) ? msg->pos : msg->vel;
    const int m = (int) z_indices.size();
    const int omem = msg->orientation.mem;
    if (omem == PB_DEVICE || (mode != MODE_YAW && src.mem != omem)) {
      fprintf(stderr, "ScanMatcherHandler: *_yaw modes take host (or host-broadcast) pos/vel and orientation arrays\n");
      return nullptr;
    }
    const size_t per = (omem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
    std::vector<double> z((size_t) m * per, 0.0);
    if (mode != MODE_YAW) memcpy(z.data(), src.p, sizeof(double) * 3 * per);
    std::vector<double> q(msg->orientation.p, msg->orientation.p + 4 * per);
    std::vector<double> R(cov_scan_match);
    std::vector<uint8_t> vmask;
    if (msg->valid != nullptr && omem == PB_HOST) vmask.assign(msg->valid, msg->valid + B);
    auto *u = new RBISIndexedPlusOrientationMeasurement(z_indices, std::move(z), std::move(R), PB_R_DIAG_BROADCAST, std::move(q),
                                                        std::move(vmask), RBISUpdateInterface::scan_matcher, msg->utime);
    u->measurement.mem = u->orientation.mem = omem;
    return u;
  }
};

class IndexedMeasurementHandler {
public:
  explicit IndexedMeasurementHandler(RBISUpdateInterface::sensor_enum this_sensor) : indexed_sensor(this_sensor) {}
  RBISUpdateInterface *processMessage(const msgs::indexed_measurement_t *msg, MavStateEstimator *est)
  {
    // sensor_handlers.cpp:576-582.  Like the reference's update object, ours owns a copy of host data (it may be
    // re-applied later from the history); device blocks stay where they are.
    const int m = (int) msg->z_indices.size();
    if (msg->z_effective.mem == PB_DEVICE)
      return new RBISIndexedMeasurement(msg->z_indices, msg->z_effective, msg->R_effective, PB_R_FULL, nullptr, indexed_sensor,
                                        msg->utime);
    const size_t per = (msg->z_effective.mem == PB_HOST_BROADCAST) ? 1 : (size_t) est->B;
    std::vector<double> z(msg->z_effective.p, msg->z_effective.p + (size_t) m * per);
    std::vector<double> R(msg->R_effective, msg->R_effective + (size_t) m * m * per);
    auto *u = new RBISIndexedMeasurement(msg->z_indices, std::move(z), std::move(R), PB_R_FULL, std::vector<uint8_t>(),
                                         indexed_sensor, msg->utime);
    u->measurement.mem = u->cov_mem = msg->z_effective.mem;
    return u;
  }
  // processMessageInit (sensor_handlers.cpp:584-610): the measured entries and their covariance block initialise the
  // state; chi entries are folded into the quaternion (init_state.chiToQuat()).  Host messages only.
  double chi_tol = 1e-6;  // eigen_utils' chiToQuat tolerance (the same constant as pb_set_constants' chi_tol)
  bool processMessageInit(const msgs::indexed_measurement_t *msg, const std::map<std::string, bool> & /*sensors_initialized*/,
                          const RBIS & /*default_state*/, const RBIM & /*default_cov*/, RBIS &init_state, RBIM &init_cov)
  {
    if (msg->z_effective.mem == PB_DEVICE) return false;
    const int B = init_state.B, m = (int) msg->z_indices.size();
    const size_t per = (msg->z_effective.mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
    for (int b = 0; b < B; b++) {
      const size_t sb = (per == 1) ? 0 : (size_t) b;
      for (int ii = 0; ii < m; ii++) {
        if (msg->z_indices[(size_t) ii] >= init_state.n) continue;  // a 15-state batch has no bias entries to initialise
        for (int jj = 0; jj < m; jj++)
          if (msg->z_indices[(size_t) jj] < init_state.n)
            init_cov(msg->z_indices[(size_t) ii], msg->z_indices[(size_t) jj], b) = msg->R_effective[(size_t) (jj * m + ii) * per + sb];
        init_state(msg->z_indices[(size_t) ii], b) = msg->z_effective.p[(size_t) ii * per + sb];
      }
      // chiToQuat: quat <- quat * Exp(chi), chi <- 0, when |chi| exceeds the tolerance
      const double c[3] = { init_state(6, b), init_state(7, b), init_state(8, b) };
      const double ang = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      if (ang > chi_tol) {
        const double sh = sin(0.5 * ang) / ang, dq[4] = { cos(0.5 * ang), c[0] * sh, c[1] * sh, c[2] * sh };
        const double q0[4] = { init_state.q(0, b), init_state.q(1, b), init_state.q(2, b), init_state.q(3, b) };
        init_state.q(0, b) = q0[0] * dq[0] - q0[1] * dq[1] - q0[2] * dq[2] - q0[3] * dq[3];
        init_state.q(1, b) = q0[0] * dq[1] + q0[1] * dq[0] + q0[2] * dq[3] - q0[3] * dq[2];
        init_state.q(2, b) = q0[0] * dq[2] - q0[1] * dq[3] + q0[2] * dq[0] + q0[3] * dq[1];
        init_state.q(3, b) = q0[0] * dq[3] + q0[1] * dq[2] - q0[2] * dq[1] + q0[3] * dq[0];
        init_state(6, b) = init_state(7, b) = init_state(8, b) = 0.0;
      }
    }
    return true;
  }
private:
  RBISUpdateInterface::sensor_enum indexed_sensor;
};

class GpsHandler {
public:
  std::vector<double> cov_xyz;  // diagonal
  explicit GpsHandler(BotParam *_param)
  {
    double rxy = bot_param_get_double_or_fail(_param, "state_estimator.gps.r_xy");
    double rz = bot_param_get_double_or_fail(_param, "state_estimator.gps.r_z");
    cov_xyz = { rxy * rxy, rxy * rxy, rz * rz };
  }
  RBISUpdateInterface *processMessage(const msgs::gps_data_t *msg, MavStateEstimator *est)
  {
    return makeIndexedMeasurement(RBIS::positionInds(), msg->xyz_pos, est->B, cov_xyz, msg->has_lock, RBISUpdateInterface::gps,
                                  msg->utime);  // sensor_handlers.cpp:374-381
  }
  // processMessageInit (sensor_handlers.cpp:383-403): position and its covariance from the fix (filters without lock keep theirs)
  bool processMessageInit(const msgs::gps_data_t *msg, const std::map<std::string, bool> & /*sensors_initialized*/,
                          const RBIS & /*default_state*/, const RBIM & /*default_cov*/, RBIS &init_state, RBIM &init_cov)
  {
    if (msg->xyz_pos.mem == PB_DEVICE) return false;
    const int B = init_state.B;
    const size_t per = (msg->xyz_pos.mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
    init_state.utime = msg->utime;
    bool any = false;
    for (int b = 0; b < B; b++) {
      if (msg->has_lock != nullptr && !msg->has_lock[b]) continue;
      any = true;
      for (int i = 0; i < 3; i++) {
        init_state(RBIS::position_ind + i, b) = msg->xyz_pos.p[(size_t) i * per + (per == 1 ? 0 : (size_t) b)];
        for (int j = 0; j < 3; j++) init_cov(RBIS::position_ind + i, RBIS::position_ind + j, b) = (i == j) ? cov_xyz[(size_t) i] : 0.0;
      }
    }
    return any;
  }
};

// ViconHandler (sensor_handlers.cpp:406-574): motion-capture pose as position / orientation / yaw / position+orientation
// measurement.  local_to_body = local_to_vicon o body_to_vicon when apply_frame (libbot's bot_trans_apply_trans_to
// [NOT IN TREE]: rotation q1 * q2, translation R(q1) t2 + t1).  Host (or host-broadcast) messages.
class ViconHandler {
public:
  typedef enum { MODE_POSITION, MODE_POSITION_ORIENT, MODE_ORIENTATION, MODE_YAW } ViconMode;
  ViconMode mode;
  bool apply_frame;
  BotTrans body_to_vicon;
  std::vector<int> z_indices;
  std::vector<double> cov_vicon;  // diagonal, of the measured entries
  double r_xyz2, r_chi2;
  ViconHandler(BotParam *param, const BotTrans *body_to_vicon_ = nullptr)
  {
    const std::string mode_str = bot_param_get_str_or_fail(param, "state_estimator.vicon.mode");
    if (mode_str == "position") mode = MODE_POSITION;
    else if (mode_str == "position_orient") mode = MODE_POSITION_ORIENT;
    else if (mode_str == "orientation") mode = MODE_ORIENTATION;
    else if (mode_str == "yaw") mode = MODE_YAW;
    else {
      mode = MODE_POSITION;
      fprintf(stdout, "Unrecognized Vicon mode. Using position mode by default.\n");
    }
    apply_frame = bot_param_get_boolean_or_fail(param, "state_estimator.vicon.apply_frame");
    if (apply_frame && body_to_vicon_) body_to_vicon = *body_to_vicon_;
    r_xyz2 = bot_sq(bot_param_get_double_or_fail(param, "state_estimator.vicon.r_xyz"));
    r_chi2 = bot_sq(bot_to_radians(bot_param_get_double_or_fail(param, "state_estimator.vicon.r_chi")));
    if (mode == MODE_POSITION) { z_indices = RBIS::positionInds(); cov_vicon = { r_xyz2, r_xyz2, r_xyz2 }; }
    else if (mode == MODE_YAW) { z_indices = { RBIS::chi_ind + 2 }; cov_vicon = { r_chi2 }; }
    else if (mode == MODE_ORIENTATION) { z_indices = RBIS::chiInds(); cov_vicon = { r_chi2, r_chi2, r_chi2 }; }
    else { z_indices = { 9, 10, 11, 6, 7, 8 }; cov_vicon = { r_xyz2, r_xyz2, r_xyz2, r_chi2, r_chi2, r_chi2 }; }
  }
  RBISUpdateInterface *processMessage(const msgs::rigid_transform_t *msg, MavStateEstimator *est)
  {
    std::vector<double> t, q;
    std::vector<uint8_t> mask;
    int mem;
    if (!to_body(msg, est->B, t, q, mask, mem)) return nullptr;
    const int m = (int) z_indices.size();
    const size_t per = t.size() / 3;
    if (mode == MODE_POSITION) {
      auto *u = new RBISIndexedMeasurement(z_indices, std::move(t), std::vector<double>(cov_vicon), PB_R_DIAG_BROADCAST, std::move(mask),
                                           RBISUpdateInterface::vicon, msg->utime);
      u->measurement.mem = mem;
      return u;
    }
    std::vector<double> z((size_t) m * per, 0.0);  // entries at chi indices are ignored (rbis.cpp:203-205)
    if (mode == MODE_POSITION_ORIENT) memcpy(z.data(), t.data(), sizeof(double) * 3 * per);
    auto *u = new RBISIndexedPlusOrientationMeasurement(z_indices, std::move(z), std::vector<double>(cov_vicon), PB_R_DIAG_BROADCAST,
                                                        std::move(q), std::move(mask), RBISUpdateInterface::vicon, msg->utime);
    u->measurement.mem = u->orientation.mem = mem;
    return u;
  }
  bool processMessageInit(const msgs::rigid_transform_t *msg, const std::map<std::string, bool> & /*sensors_initialized*/,
                          const RBIS & /*default_state*/, const RBIM & /*default_cov*/, RBIS &init_state, RBIM &init_cov)
  {
    std::vector<double> t, q;
    std::vector<uint8_t> mask;
    int mem;
    const int B = init_state.B;
    if (!to_body(msg, B, t, q, mask, mem, /*for_init=*/true)) return false;
    const size_t per = t.size() / 3;
    init_state.utime = msg->utime;
    for (int b = 0; b < B; b++) {
      const size_t sb = per == 1 ? 0 : (size_t) b;
      for (int i = 0; i < 3; i++) {
        init_state(RBIS::position_ind + i, b) = t[(size_t) i * per + sb];
        for (int j = 0; j < 3; j++) {
          init_cov(RBIS::position_ind + i, RBIS::position_ind + j, b) = (i == j) ? r_xyz2 : 0.0;
          init_cov(RBIS::chi_ind + i, RBIS::chi_ind + j, b) = (i == j) ? r_chi2 : 0.0;
        }
      }
      for (int i = 0; i < 4; i++) init_state.q(i, b) = q[(size_t) i * per + sb];
    }
    return true;
  }
private:
  // msg -> body pose per filter; mask 0 where |translation| < 1e-5 on all axes (":493-494 return NULL")
  bool to_body(const msgs::rigid_transform_t *msg, int B, std::vector<double> &t, std::vector<double> &q, std::vector<uint8_t> &mask,
               int &mem, bool for_init = false) const
  {
    mem = msg->trans.mem;
    if (mem == PB_DEVICE || msg->quat.mem != mem) {
      fprintf(stderr, "ViconHandler: host (or host-broadcast) trans and quat arrays expected\n");
      return false;
    }
    const size_t per = (mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
    t.resize(3 * per);
    q.resize(4 * per);
    bool any = false;
    if (per > 1) mask.assign(per, 1);
    for (size_t b = 0; b < per; b++) {
      const double tv[3] = { msg->trans.p[b], msg->trans.p[per + b], msg->trans.p[2 * per + b] };
      const double qv[4] = { msg->quat.p[b], msg->quat.p[per + b], msg->quat.p[2 * per + b], msg->quat.p[3 * per + b] };
      double tb[3] = { tv[0], tv[1], tv[2] }, qb[4] = { qv[0], qv[1], qv[2], qv[3] };
      if (apply_frame) {
        bot_quat_rotate_to(qv, body_to_vicon.trans_vec, tb);
        for (int i = 0; i < 3; i++) tb[i] += tv[i];
        const double *a = qv, *c = body_to_vicon.rot_quat;
        qb[0] = a[0] * c[0] - a[1] * c[1] - a[2] * c[2] - a[3] * c[3];
        qb[1] = a[0] * c[1] + a[1] * c[0] + a[2] * c[3] - a[3] * c[2];
        qb[2] = a[0] * c[2] - a[1] * c[3] + a[2] * c[0] + a[3] * c[1];
        qb[3] = a[0] * c[3] + a[1] * c[2] - a[2] * c[1] + a[3] * c[0];
      }
      const bool dropped = !for_init && fabs(tv[0]) < 1e-5 && fabs(tv[1]) < 1e-5 && fabs(tv[2]) < 1e-5;
      if (per > 1) mask[b] = !dropped;
      any = any || !dropped;
      for (int i = 0; i < 3; i++) t[(size_t) i * per + b] = tb[i];
      for (int i = 0; i < 4; i++) q[(size_t) i * per + b] = qb[i];
    }
    return any;
  }
};

// PoseMeasHandler (motion_estimate/src/pose_meas/pose_meas.cpp:7-131): a bot_core::pose_t as a position or position +
// orientation measurement, `no_corrections` messages long, then silent.  Same update kernels as the VO / Vicon lists.
class PoseMeasHandler {
public:
  typedef enum { MODE_POSITION, MODE_POSITION_ORIENT } PoseMeasMode;
  PoseMeasMode mode;
  int no_corrections;  // no of corrections to make before going silent
  std::vector<int> z_indices;
  std::vector<double> cov_pose_meas;  // diagonal, of the measured entries
  double r_xyz2, r_chi2;
  explicit PoseMeasHandler(BotParam *param)
  {
    const std::string mode_str = bot_param_get_str_or_fail(param, "state_estimator.pose_meas.mode");
    if (mode_str == "position") mode = MODE_POSITION;
    else if (mode_str == "position_orient") mode = MODE_POSITION_ORIENT;
    else {
      mode = MODE_POSITION;
      fprintf(stdout, "Unrecognized PoseMeas mode. Using position mode by default.\n");
    }
    no_corrections = (int) bot_param_get_double_or_fail(param, "state_estimator.pose_meas.no_corrections");
    r_xyz2 = bot_sq(bot_param_get_double_or_fail(param, "state_estimator.pose_meas.r_xyz"));
    r_chi2 = bot_sq(bot_to_radians(bot_param_get_double_or_fail(param, "state_estimator.pose_meas.r_chi")));
    if (mode == MODE_POSITION) { z_indices = RBIS::positionInds(); cov_pose_meas = { r_xyz2, r_xyz2, r_xyz2 }; }
    else { z_indices = { 9, 10, 11, 6, 7, 8 }; cov_pose_meas = { r_xyz2, r_xyz2, r_xyz2, r_chi2, r_chi2, r_chi2 }; }
  }
  RBISUpdateInterface *processMessage(const msgs::pose_t *msg, MavStateEstimator *est)
  {
    // "If we have created no_corrections, go silent afterwards" (:56-64): decrement first, then compare
    no_corrections--;
    if (no_corrections == 1) fprintf(stdout, "Finished making PoseMeas corrections\n");
    if (no_corrections <= 0) return nullptr;
    const int mem = msg->pos.mem;
    if (mem == PB_DEVICE || msg->orientation.mem != mem) {
      fprintf(stderr, "PoseMeasHandler: host (or host-broadcast) pos and orientation arrays expected\n");
      return nullptr;
    }
    const size_t per = (mem == PB_HOST_BROADCAST) ? 1 : (size_t) est->B;
    // a pose at the origin is "no pose" (:74-75), per filter: mask
    std::vector<uint8_t> mask;
    if (per > 1) mask.assign(per, 1);
    bool any = false;
    for (size_t b = 0; b < per; b++) {
      const bool dropped = fabs(msg->pos.p[b]) < 1e-5 && fabs(msg->pos.p[per + b]) < 1e-5 && fabs(msg->pos.p[2 * per + b]) < 1e-5;
      if (per > 1) mask[b] = !dropped;
      any = any || !dropped;
    }
    if (!any) return nullptr;
    std::vector<double> t(msg->pos.p, msg->pos.p + 3 * per);
    if (mode == MODE_POSITION) {
      auto *u = new RBISIndexedMeasurement(z_indices, std::move(t), std::vector<double>(cov_pose_meas), PB_R_DIAG_BROADCAST, std::move(mask),
                                           RBISUpdateInterface::pose_meas, msg->utime);
      u->measurement.mem = mem;
      return u;
    }
    std::vector<double> z(6 * per, 0.0);  // entries at chi indices are ignored (rbis.cpp:203-205)
    memcpy(z.data(), t.data(), sizeof(double) * 3 * per);
    std::vector<double> q(msg->orientation.p, msg->orientation.p + 4 * per);
    auto *u = new RBISIndexedPlusOrientationMeasurement(z_indices, std::move(z), std::vector<double>(cov_pose_meas), PB_R_DIAG_BROADCAST,
                                                        std::move(q), std::move(mask), RBISUpdateInterface::pose_meas, msg->utime);
    u->measurement.mem = u->orientation.mem = mem;
    return u;
  }
  // :98-129
  bool processMessageInit(const msgs::pose_t *msg, const std::map<std::string, bool> & /*sensors_initialized*/,
                          const RBIS & /*default_state*/, const RBIM & /*default_cov*/, RBIS &init_state, RBIM &init_cov)
  {
    const int mem = msg->pos.mem, B = init_state.B;
    if (mem == PB_DEVICE || msg->orientation.mem != mem) return false;
    const size_t per = (mem == PB_HOST_BROADCAST) ? 1 : (size_t) B;
    init_state.utime = msg->utime;
    for (int b = 0; b < B; b++) {
      const size_t sb = per == 1 ? 0 : (size_t) b;
      for (int i = 0; i < 3; i++) {
        init_state(RBIS::position_ind + i, b) = msg->pos.p[(size_t) i * per + sb];
        for (int j = 0; j < 3; j++) {
          init_cov(RBIS::position_ind + i, RBIS::position_ind + j, b) = (i == j) ? r_xyz2 : 0.0;
          init_cov(RBIS::chi_ind + i, RBIS::chi_ind + j, b) = (i == j) ? r_chi2 : 0.0;
        }
      }
      for (int i = 0; i < 4; i++) init_state.q(i, b) = msg->orientation.p[(size_t) i * per + sb];
    }
    return true;
  }
};

class FovisHandler {
public:
  typedef enum { MODE_VELOCITY, MODE_VELOCITY_ROTATION_RATE, MODE_POSITION, MODE_POSITION_ORIENT } FovisMode;
  FovisMode mode;
  std::vector<int> z_indices;
  std::vector<double> cov_fovis;  // diagonal
  int64_t prev_t0_body_utime_ = 0;
  int slot;                       // device snapshot slot holding the filter posterior at prev_timestamp
  // one device block per update, owned by that update and recycled through this pool when the history drops it:
  // z [6][B] (rows 3-5 stay zero) | composed orientation [4][B] | copy of msg->estimate_valid [B]
  std::shared_ptr<DevicePool> pool;

  explicit FovisHandler(BotParam *param, int snapshot_slot = 0) : slot(snapshot_slot)
  {
    std::string mode_str = bot_param_get_str_or_fail(param, "state_estimator.fovis.mode");
    if (mode_str == "velocity_rotation_rate") mode = MODE_VELOCITY_ROTATION_RATE;
    else if (mode_str == "velocity") mode = MODE_VELOCITY;
    else if (mode_str == "position") mode = MODE_POSITION;
    else if (mode_str == "position_orient") mode = MODE_POSITION_ORIENT;
    else {
      fprintf(stdout, "FOVIS mode not understood %s, exiting\n", mode_str.c_str());
      exit(-1);  // rbis_fovis_update.cpp:36-38
    }
    if (mode == MODE_VELOCITY || mode == MODE_VELOCITY_ROTATION_RATE) {
      double r = bot_param_get_double_or_fail(param, "state_estimator.fovis.r_vxyz");
      cov_fovis = { bot_sq(r), bot_sq(r), bot_sq(r) };
      z_indices = RBIS::velocityInds();
      if (mode == MODE_VELOCITY_ROTATION_RATE) {
        double ra = bot_param_get_double_or_fail(param, "state_estimator.fovis.r_vang");
        for (int i = 0; i < 3; i++) cov_fovis.push_back(bot_sq(ra));
        for (int i : RBIS::angularVelocityInds()) z_indices.push_back(i);
      }
    } else {
      double r = bot_param_get_double_or_fail(param, "state_estimator.fovis.r_pxyz");
      cov_fovis = { bot_sq(r), bot_sq(r), bot_sq(r) };
      z_indices = RBIS::positionInds();
      if (mode == MODE_POSITION_ORIENT) {
        double rc = bot_param_get_double_or_fail(param, "state_estimator.fovis.r_chi");  // NOT deg->rad (:99-102)
        for (int i = 0; i < 3; i++) cov_fovis.push_back(bot_sq(rc));
        for (int i : RBIS::chiInds()) z_indices.push_back(i);
      }
    }
  }
  // The estimator calls this when the posterior at a future `prev_timestamp` is the head: the device-side
  // replacement of history.updateMap.lower_bound(prev_timestamp) (rbis_fovis_update.cpp:184-207).
  void markKeyframe(MavStateEstimator *est)
  {
    est->flushPending();
    pb_snapshot(est->ctx, slot);
    prev_t0_body_utime_ = est->head_utime;
  }

  RBISUpdateInterface *processMessage(const msgs::update_t *msg, MavStateEstimator *est)
  {
    const int B = est->B;
    // estimate_status != ESTIMATE_VALID -> NULL (:160-164); per filter that is a mask, and NULL when no filter is valid
    if (msg->estimate_valid != nullptr && std::find_if(msg->estimate_valid, msg->estimate_valid + B,
                                                       [](uint8_t v) { return v != 0; }) == msg->estimate_valid + B) {
      fprintf(stdout, "FovisHandler: FOVIS failure, not integrating this measurement\n");
      return nullptr;
    }
    if (mode == MODE_VELOCITY_ROTATION_RATE) {
      fprintf(stdout, "FovisHandler Mode not supported, exiting\n");  // :276-285
      return nullptr;
    }
    if (mode == MODE_VELOCITY) {
      if (msg->translation.mem == PB_DEVICE) return nullptr;
      const double elapsed = (double) (msg->timestamp - msg->prev_timestamp) * 1E-6;
      const bool bcast = msg->translation.mem == PB_HOST_BROADCAST;
      std::vector<double> z((size_t) 3 * (bcast ? 1 : B));
      for (size_t i = 0; i < z.size(); i++) z[i] = msg->translation.p[i] / elapsed;
      std::vector<uint8_t> mask;
      if (msg->estimate_valid && !bcast) mask.assign(msg->estimate_valid, msg->estimate_valid + B);
      auto *u = new RBISIndexedMeasurement(RBIS::velocityInds(), std::move(z), std::vector<double>(cov_fovis), PB_R_DIAG_BROADCAST,
                                           std::move(mask), RBISUpdateInterface::fovis, msg->timestamp);
      if (bcast) u->measurement.mem = PB_HOST_BROADCAST;
      return u;
    }
    // position / position_orient: T1 = T0(posterior at prev_timestamp) * (t, q).  T0 comes from a checkpoint (history) or
    // from the snapshot markKeyframe() took, never from the head: updates the estimator is holding back for fusion
    // (fuse_ins_legodo / fuse_corrections) stay held, so that this measurement can ride in the same kernel.
    if (est->history_slots > 0 && msg->prev_timestamp != prev_t0_body_utime_) {
      // The reference's own look-up (:177-213): the posterior of the first update at or after prev_timestamp, at most
      // 25 ms later, becomes the cached T0 -- here: copied from that update's checkpoint into the snapshot slot.
      auto &map = est->history.updateMap;
      auto lower_it = map.lower_bound(msg->prev_timestamp);
      // an update that applied to NO filter is a message for which the reference's handler returned NULL: it is not in the
      // reference's history, the look-up lands on the update after it
      while (lower_it != map.end() && lower_it->second->appliesToNoFilter(est->ctx, B)) ++lower_it;
      if (lower_it == map.end()) {
        fprintf(stdout, "%lld at the end\n", (long long) msg->prev_timestamp);  // :192-195
        return nullptr;
      }
      const double diff_utime = (double) (lower_it->first - msg->prev_timestamp) * 1E-6;
      if (diff_utime > 0.025) {
        fprintf(stdout, "FOIVS: time difference for VO delta root pose is too great (%fsec). Will not use\n", diff_utime);
        return nullptr;
      }
      // lower_it->second->posterior_state (:196-206): from that update's checkpoint, or re-derived from the nearest earlier
      // one when the checkpoints are sparser than the updates (history_checkpoint_every > 1, the default for a configuration
      // that only sets utime_history_span) or the update is the INS half of a fused pair
      if (!est->snapshotPosteriorOf(lower_it, slot)) {
        fprintf(stdout, "FovisHandler: the posterior of the update at %lld cannot be had (not applied yet, or no free checkpoint "
                        "slot): %s. Will not use\n", (long long) lower_it->first, pb_last_error(est->ctx));
        return nullptr;
      }
      prev_t0_body_utime_ = msg->prev_timestamp;  // :213
    }
    const double diff = (double) (prev_t0_body_utime_ - msg->prev_timestamp) * 1E-6;
    if (diff > 0.025 || diff < -0.025) {  // without a history the keyframe is whatever markKeyframe() last captured
      fprintf(stdout, "FOIVS: time difference for VO delta root pose is too great (%fsec). Will not use\n", diff);  // :186-189
      return nullptr;
    }
    // The measurement (T1's translation and rotation) is formed on the device and belongs to THIS update: a later message
    // must not overwrite it, because the history re-applies the update when an older measurement arrives late.
    if (!pool) pool = std::make_shared<DevicePool>(est->ctx, est->ctx_alive, sizeof(double) * 10 * (size_t) B + (size_t) B);
    bool fresh = false;
    void *blk = pool->get(fresh);
    if (blk == nullptr) {
      fprintf(stderr, "FovisHandler: %s\n", pb_last_error(est->ctx));
      return nullptr;
    }
    auto block = std::make_shared<DeviceBlock>(pool, blk);
    double *d_z6 = (double *) blk, *d_q = d_z6 + (size_t) 6 * B;
    uint8_t *d_valid = (uint8_t *) (d_q + (size_t) 4 * B);
    if (fresh) {  // rows 3..5 of z are never written again: z at chi indices is ignored (rbis.cpp:203-205)
      std::vector<double> zero((size_t) 6 * B, 0.0);
      pb_memcpy_h2d(est->ctx, d_z6, zero.data(), sizeof(double) * 6 * B);
    }
    if (pb_compose_delta(est->ctx, slot, msg->translation.p, msg->rotation.p, d_z6, d_q, msg->translation.mem) != PB_OK) {
      fprintf(stderr, "FovisHandler: %s\n", pb_last_error(est->ctx));
      return nullptr;
    }
    const uint8_t *valid = nullptr;  // the composed z / q live on the device, so does the mask that goes with them
    if (msg->estimate_valid != nullptr && msg->translation.mem != PB_HOST_BROADCAST) {  // broadcast: all valid here
      pb_memcpy_h2d(est->ctx, d_valid, msg->estimate_valid, (size_t) B);
      valid = d_valid;
    }
    RBISIndexedMeasurement *u;
    if (mode == MODE_POSITION)
      u = new RBISIndexedMeasurement(RBIS::positionInds(), BatchArray(d_z6, PB_DEVICE), cov_fovis.data(), PB_R_DIAG_BROADCAST,
                                     valid, RBISUpdateInterface::fovis, msg->timestamp);
    else
      u = new RBISIndexedPlusOrientationMeasurement(z_indices, BatchArray(d_z6, PB_DEVICE), cov_fovis.data(), PB_R_DIAG_BROADCAST,
                                                    BatchArray(d_q, PB_DEVICE), valid, RBISUpdateInterface::fovis,
                                                    msg->timestamp);
    u->owned_dev = block;
    u->owned_R = cov_fovis;  // the update owns its R too (the reference copies cov_fovis into the update object)
    u->measurement_cov = u->owned_R.data();
    return u;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// dispatch: LCMFrontEnd::addSensor / SensorHandler::lcm_message_handler without LCM (lcm_front_end.hpp:82-94,139-203)
// ---------------------------------------------------------------------------------------------------------------
class FrontEnd {
public:
  BotParam *param;
  MavStateEstimator *state_estimator = nullptr;
  explicit FrontEnd(BotParam *p) : param(p) {}
  void setStateEstimator(MavStateEstimator *e) { state_estimator = e; }

  template <class MsgT, class HandlerT>
  std::function<void(const MsgT *)> addSensor(const std::string &sensor_prefix,
                                              RBISUpdateInterface *(HandlerT::*handler_function)(const MsgT *, MavStateEstimator *),
                                              HandlerT *handler)
  {
    const std::string pre = "state_estimator." + sensor_prefix;
    const int downsample_factor = (int) bot_param_get_int_or_fail(param, (pre + ".downsample_factor").c_str());
    const bool roll_forward = bot_param_get_boolean_or_fail(param, (pre + ".roll_forward_on_receive").c_str());
    const int64_t utime_delay = bot_param_get_int_or_fail(param, (pre + ".utime_offset").c_str());
    auto counter = std::make_shared<int64_t>(0);
    return [=](const MsgT *msg) {
      if (state_estimator == nullptr) return;
      if ((*counter)++ % downsample_factor != 0) return;                     // :147
      RBISUpdateInterface *update = (handler->*handler_function)(msg, state_estimator);
      if (update != nullptr) {
        update->utime -= utime_delay;                                        // :157
        state_estimator->addUpdate(update, roll_forward);                    // :158
      }
    };
  }
};

// ---------------------------------------------------------------------------------------------------------------
// wire side (pronto_wire.hpp): publish the head of one filter, replay a recorded LCM log into the batch
// ---------------------------------------------------------------------------------------------------------------

// rbisCreateFilterStateMessageCPP (rbis.cpp:287-304) for filter b: always rbis_num_states = 21 entries and the full
// 21 x 21 column-major covariance; a 15-state batch publishes zero bias entries like the reference run with q_bias = 0.
inline pronto_wire::filter_state_t rbisCreateFilterStateMessageCPP(const RBIS &state, const RBIM &cov, int b)
{
  pronto_wire::filter_state_t msg;
  msg.utime = state.utime;
  msg.num_states = RBIS::rbis_num_states;
  msg.num_cov_elements = msg.num_states * msg.num_states;
  for (int i = 0; i < 4; i++) msg.quat[i] = state.q(i, b);
  msg.state.assign((size_t) msg.num_states, 0.0);
  msg.cov.assign((size_t) msg.num_cov_elements, 0.0);
  for (int i = 0; i < state.n; i++) msg.state[(size_t) i] = state(i, b);
  for (int c = 0; c < cov.n; c++)
    for (int r = 0; r < cov.n; r++) msg.cov[(size_t) c * msg.num_states + r] = cov(r, c, b);
  return msg;
}

// InitMessageHandler (rbis_initializer.cpp:162-184): a pronto::filter_state_t resets the estimator on the fly -- the
// reference's checkpoint / resume mechanism (what FilterStatePublisher writes can be read back here).
// RBIS(msg) (rbis.hpp:69-78): vec <- state[0..n), quat <- msg.quat.  The RigidBodyState(vec) constructor folds a chi beyond
// the tolerance into its own quaternion before msg.quat overwrites that quaternion, i.e. such a chi is dropped; published
// heads carry chi = 0 anyway.
class InitMessageHandler {
public:
  double chi_tol = 1e-6;  // eigen_utils' chiToQuat tolerance: the same constant as pb_set_constants' chi_tol
  // one message for every filter of the batch
  RBISUpdateInterface *processMessage(const pronto_wire::filter_state_t *msg, MavStateEstimator *est)
  {
    return processMessages(std::vector<pronto_wire::filter_state_t>((size_t) est->B, *msg), est);
  }
  // one message per filter (msgs.size() == batch): resume a whole batch from what was published
  RBISUpdateInterface *processMessages(const std::vector<pronto_wire::filter_state_t> &msgs, MavStateEstimator *est)
  {
    const int n = est->n, B = est->B;
    if ((int) msgs.size() != B) return nullptr;
    RBIS x(n, B);
    RBIM P(n, B);
    for (int b = 0; b < B; b++) {
      const pronto_wire::filter_state_t &m = msgs[(size_t) b];
      if (m.num_states != RBIS::rbis_num_states || m.num_cov_elements != m.num_states * m.num_states) {
        fprintf(stderr, "error, constructed RBIS from rbis_filter_state_t of wrong size\n");  // rbis.hpp:72-74
        return nullptr;
      }
      for (int i = 0; i < n; i++) x(i, b) = m.state[(size_t) i];
      const double chi2 = x(6, b) * x(6, b) + x(7, b) * x(7, b) + x(8, b) * x(8, b);
      if (chi2 > chi_tol * chi_tol) x(6, b) = x(7, b) = x(8, b) = 0.0;
      for (int i = 0; i < 4; i++) x.q(i, b) = m.quat[i];
      for (int c = 0; c < n; c++)
        for (int r = 0; r < n; r++) P(r, c, b) = m.cov[(size_t) c * m.num_states + r];
    }
    x.utime = msgs[0].utime;
    return new RBISResetUpdate(x, P, RBISUpdateInterface::init_message, msgs[0].utime);
  }
};

// LCMFrontEnd::publishState (lcm_front_end.cpp:144-157), filter-state half: appends the head of the chosen filters
// to an LCM log on `state_estimator.filter_state_channel` when `state_estimator.publish_filter_state` is set.
class FilterStatePublisher {
public:
  std::string filter_state_channel;
  bool publish_filter_state;
  FilterStatePublisher(BotParam *param, pronto_wire::LogWriter *log, std::vector<int> filters)
      : filter_state_channel(bot_param_get_str_or_fail(param, "state_estimator.filter_state_channel")),
        publish_filter_state(bot_param_get_boolean_or_fail(param, "state_estimator.publish_filter_state")),
        log_(log), filters_(std::move(filters)) {}
  // one channel per filter beyond the first so that a reader can tell them apart: CHANNEL, CHANNEL_1, CHANNEL_2 ...
  void publishHead(MavStateEstimator *est)
  {
    if (!publish_filter_state || log_ == nullptr) return;
    RBIS s;
    RBIM c;
    est->getHeadState(s, c);
    std::vector<uint8_t> buf;
    for (size_t k = 0; k < filters_.size(); k++) {
      rbisCreateFilterStateMessageCPP(s, c, filters_[k]).encode(buf);
      log_->write(s.utime, k == 0 ? filter_state_channel : filter_state_channel + "_" + std::to_string(k), buf);
    }
  }
private:
  pronto_wire::LogWriter *log_;
  std::vector<int> filters_;
};

// Log replay (lcm_front_end.cpp:223-229 handle loop, reading a recorded segment instead of the network): events are
// dispatched in file order to the subscribed channels.  One recorded robot feeds EVERY filter of the batch -- the
// batch differs in parameters / initial state, not in data (param_sweep.py:39-52) -- so a decoded message is handed to
// the callback LCMFrontEnd::addSensor returns as PB_HOST_BROADCAST blocks ([rows] values, expanded on the device:
// nothing of batch size is built on the host or crosses PCIe).
class LogPlayer {
public:
  explicit LogPlayer(int batch) : B_(batch) {}
  void subscribeRaw(const std::string &channel, std::function<void(const pronto_wire::LogEvent &)> cb)
  {
    subs_[channel] = std::move(cb);
  }
  void subscribeIndexedMeasurement(const std::string &channel, std::function<void(const msgs::indexed_measurement_t *)> cb)
  {
    subs_[channel] = [this, cb](const pronto_wire::LogEvent &ev) {
      pronto_wire::indexed_measurement_t w;
      if (w.decode(ev.data.data(), ev.data.size()) < 0 || w.measured_cov_dim != w.measured_dim * w.measured_dim) {
        n_bad_++;
        return;
      }
      // one recorded message for every filter: hand the [m] / [m*m] values over as PB_HOST_BROADCAST blocks
      msgs::indexed_measurement_t msg;
      msg.utime = w.utime;
      msg.z_indices.assign(w.z_indices.begin(), w.z_indices.end());
      msg.z_effective = BatchArray(w.z_effective.data(), PB_HOST_BROADCAST);
      msg.R_effective = w.R_effective.data();
      cb(&msg);
    };
  }
  // Messages whose .lcm definition the caller supplies at run time (lcm_schema.hpp; libbot's bot_core types are not in
  // the reference tree): decoded into a value tree and handed over with the raw event.  `schema` must outlive the player.
  void subscribeSchema(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                       std::function<void(const pronto_wire::Value &, const pronto_wire::LogEvent &)> cb)
  {
    subs_[channel] = [this, schema, type, cb](const pronto_wire::LogEvent &ev) {
      pronto_wire::Value v;
      if (!schema->decode(type, ev.data.data(), ev.data.size(), v)) {
        n_bad_++;
        return;
      }
      cb(v, ev);
    };
  }
  // bot_core::ins_t by field name (utime, gyro[3], accel[3]: what InsHandler::processMessage reads, sensor_handlers.cpp:96-131)
  void subscribeIns(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                    std::function<void(const msgs::ins_t *)> cb)
  {
    subscribeSchema(channel, schema, type, [this, cb](const pronto_wire::Value &v, const pronto_wire::LogEvent &) {
      double g[3], a[3];
      int64_t utime;
      if (!v.integer("utime", utime) || !v.numbers("gyro", g, 3) || !v.numbers("accel", a, 3)) {
        n_bad_++;
        return;
      }
      msgs::ins_t m{ utime, BatchArray(g, PB_HOST_BROADCAST), BatchArray(a, PB_HOST_BROADCAST) };
      cb(&m);
    });
  }
  // bot_core::pose_t (utime, pos[3], vel[3], orientation[4]: ScanMatcherHandler::processMessage, sensor_handlers.cpp:689-724)
  void subscribePose(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                     std::function<void(const msgs::pose_t *)> cb)
  {
    subscribeSchema(channel, schema, type, [this, cb](const pronto_wire::Value &v, const pronto_wire::LogEvent &) {
      double p[3], vel[3], q[4];
      int64_t utime;
      if (!v.integer("utime", utime) || !v.numbers("pos", p, 3) || !v.numbers("vel", vel, 3) || !v.numbers("orientation", q, 4)) {
        n_bad_++;
        return;
      }
      msgs::pose_t m{ utime, BatchArray(p, PB_HOST_BROADCAST), BatchArray(vel, PB_HOST_BROADCAST), BatchArray(q, PB_HOST_BROADCAST) };
      cb(&m);
    });
  }
  // bot_core::rigid_transform_t (utime, trans[3], quat[4]: ViconHandler::processMessage, sensor_handlers.cpp:476-536)
  void subscribeRigidTransform(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                               std::function<void(const msgs::rigid_transform_t *)> cb)
  {
    subscribeSchema(channel, schema, type, [this, cb](const pronto_wire::Value &v, const pronto_wire::LogEvent &) {
      double t[3], q[4];
      int64_t utime;
      if (!v.integer("utime", utime) || !v.numbers("trans", t, 3) || !v.numbers("quat", q, 4)) {
        n_bad_++;
        return;
      }
      msgs::rigid_transform_t m{ utime, BatchArray(t, PB_HOST_BROADCAST), BatchArray(q, PB_HOST_BROADCAST) };
      cb(&m);
    });
  }
  // bot_core::kvh_raw_imu_batch_t (utime, raw_imu[]{utime, packet_count, delta_rotation[3], linear_acceleration[3]}:
  // IMUStream::convertFromLCMBatch, imu_stream.cpp:62-98; newest packet first, as on the wire)
  void subscribeKvhBatch(const std::string &channel, const pronto_wire::Schema *schema, const std::string &type,
                         std::function<void(const msgs::kvh_raw_imu_batch_t *)> cb)
  {
    subscribeSchema(channel, schema, type, [this, cb](const pronto_wire::Value &v, const pronto_wire::LogEvent &) {
      msgs::kvh_raw_imu_batch_t m;
      m.mem = PB_HOST_BROADCAST;
      const pronto_wire::Value *raw = v.get("raw_imu");
      if (!v.integer("utime", m.utime) || raw == nullptr || raw->kind != pronto_wire::Value::ARRAY) {
        n_bad_++;
        return;
      }
      std::vector<double> store(raw->items.size() * 6);
      for (size_t k = 0; k < raw->items.size(); k++) {
        const pronto_wire::Value &pk = raw->items[k];
        msgs::kvh_raw_imu_packet_t p;
        if (!pk.integer("utime", p.utime) || !pk.integer("packet_count", p.packet_count) ||
            !pk.numbers("delta_rotation", &store[k * 6], 3) || !pk.numbers("linear_acceleration", &store[k * 6 + 3], 3)) {
          n_bad_++;
          return;
        }
        p.delta_rotation = &store[k * 6];
        p.linear_acceleration = &store[k * 6 + 3];
        m.raw_imu.push_back(p);
      }
      cb(&m);
    });
  }
  void subscribeUpdate(const std::string &channel, std::function<void(const msgs::update_t *)> cb)
  {
    subs_[channel] = [this, cb](const pronto_wire::LogEvent &ev) {
      pronto_wire::update_t w;
      if (w.decode(ev.data.data(), ev.data.size()) < 0) {
        n_bad_++;
        return;
      }
      // estimate_status != ESTIMATE_VALID for the one robot = for every filter: an all-zero per-filter mask
      if (w.estimate_status != pronto_wire::update_t::ESTIMATE_VALID) valid_.assign((size_t) B_, (uint8_t) 0);
      msgs::update_t msg;
      msg.timestamp = w.timestamp;
      msg.prev_timestamp = w.prev_timestamp;
      msg.estimate_valid = (w.estimate_status == pronto_wire::update_t::ESTIMATE_VALID) ? nullptr : valid_.data();
      msg.translation = BatchArray(w.translation, PB_HOST_BROADCAST);
      msg.rotation = BatchArray(w.rotation, PB_HOST_BROADCAST);
      cb(&msg);
    };
  }
  // returns the number of events dispatched, or -1 if the file cannot be opened
  int64_t run(const std::string &path)
  {
    pronto_wire::LogReader rd(path);
    if (!rd.good()) return -1;
    pronto_wire::LogEvent ev;
    int64_t n = 0;
    while (rd.next(ev)) {
      auto it = subs_.find(ev.channel);
      if (it == subs_.end()) continue;
      it->second(ev);
      n++;
    }
    return n;
  }
  int64_t undecodable() const { return n_bad_; }
private:
  int B_;
  int64_t n_bad_ = 0;
  std::map<std::string, std::function<void(const pronto_wire::LogEvent &)>> subs_;
  std::vector<uint8_t> valid_;
};

}  // namespace MavStateEst
