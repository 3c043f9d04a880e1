// rbis_frontend.hpp -- the IMU front end per filter on the device: what InsHandler does to ONE robot's message before it
// becomes an RBISIMUProcessStep, for B robots at once (independent log segments: every filter has its own IMU stream).
//   state-estimator/src/mav_state_est/sensor_handlers.cpp:96-131    processMessage (Microstrain): rotate into the body frame, dt = param
//   state-estimator/src/mav_state_est/sensor_handlers.cpp:165-252   processMessageAtlas (KVH): newest NEW packet after the notch
//                                                                   cascade, gyro = delta_rotation / raw_dt, accel through the
//                                                                   whole ins_to_body transform, dt from the message times
//   estimate_tools/src/estimate_tools/iir_notch.cpp:34-61           IIRNotch::processSample
// The de-duplication (IMUStream::convertFromLCMBatch, imu_stream.cpp:62-98) stays where the messages are decoded -- it is integer
// book-keeping on packet counters, one state per recorded stream; what comes here is its result: per filter the number of NEW
// packets, their accelerations oldest first, and the newest new packet's delta_rotation / utime_delta.
// Plumbing kernels: one small launch per message, a few dozen bytes per filter.  Included by pronto_batch.hip only.
#pragma once

#include <stdint.h>

#include "rbis_kernels.hpp"

namespace pb {

// The notch cascade with a per-filter packet count: filter b runs its first counts[b] packets (at most n_packets) through its
// three stages per axis; a filter with no new packet keeps its state and gets no output.  counts == NULL: n_packets for every
// filter (pb_imu_notch).  One lane per (filter, axis) like k_notch.
static __global__ void k_notch_counts(double *__restrict__ nst, long stride, int B, int n_packets, const int32_t *__restrict__ counts,
                                      const double *__restrict__ acc_in, double *__restrict__ acc_out, NotchCoef k)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ax = blockIdx.y;
  if (b >= B) return;
  int np = n_packets;
  if (counts) {
    const int cb = counts[b];
    np = cb < 0 ? 0 : (cb < n_packets ? cb : n_packets);
  }
  if (np == 0) return;
  double s[3][4];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) s[i][t] = nst[(long) ((ax * 3 + i) * 4 + t) * stride + b];
  double v = 0.0;
  for (int p0 = 0; p0 < np; p0 += 4) {
    double pk[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = (p0 + j < np) ? p0 + j : np - 1;
      pk[j] = acc_in[((long) p * 3 + ax) * B + b];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (p0 + j >= np) break;
      v = pk[j];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double in = v;
        const double xb = in * k.b[i][0] + s[i][0] * k.b[i][1] + s[i][1] * k.b[i][2];   // iir_notch.cpp:52
        const double ya = s[i][2] * k.a[i][1] + s[i][3] * k.a[i][2];
        const double out = xb - ya;
        s[i][1] = s[i][0]; s[i][0] = in;
        s[i][3] = s[i][2]; s[i][2] = out;
        v = out;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) nst[(long) ((ax * 3 + i) * 4 + t) * stride + b] = s[i][t];
  acc_out[(long) ax * B + b] = v;
}

struct InsFrame {
  double rot[4];     // ins_to_body.rot_quat (w, x, y, z)
  double trans[3];   // ins_to_body.trans_vec, added to the rotated acceleration when `translate` (bot_trans_apply_vec, :227)
  int translate;
  int dt_from_utimes;   // 1: dt = (utime - previous utime of this filter) * 1E-6, the configured dt on its first message (:239-249)
  double dt_default;
};

// libbot's bot_quat_rotate_to, the expression the shim's host pass evaluates (mav_state_est_batch.hpp)
PB_HD void ins_rotate(const double rot[4], const double v[3], double r[3])
{
  const double ab = rot[0] * rot[1], ac = rot[0] * rot[2], ad = rot[0] * rot[3];
  const double nbb = -rot[1] * rot[1], bc = rot[1] * rot[2], bd = rot[1] * rot[3];
  const double ncc = -rot[2] * rot[2], cd = rot[2] * rot[3], ndd = -rot[3] * rot[3];
  r[0] = 2 * ((ncc + ndd) * v[0] + (bc - ad) * v[1] + (ac + bd) * v[2]) + v[0];
  r[1] = 2 * ((ad + bc) * v[0] + (nbb + ndd) * v[1] + (cd - ab) * v[2]) + v[1];
  r[2] = 2 * ((bd - ac) * v[0] + (ab + cd) * v[1] + (nbb + ncc) * v[2]) + v[2];
}

// One robot's IMU message -> the [7][B] block of RBISIMUProcessStep (gyro xyz | accel xyz | dt, body frame), one lane per filter.
//   gyro [3][B]: angular rate, or delta_rotation when raw_dt != NULL (then divided by raw_dt[b], :207-210);  accel [3][B];
//   utimes [B] or NULL (= utime for every filter);  valid [B] or NULL (= all): a filter WITHOUT a message -- its segment has ended,
//   or its KVH message carried no new packet (:181-187: the reference returns NULL) -- gets dt = 0 on its own last sample; valid_out
//   [B] (optional) receives the mask for pb_set_imu_valid, which makes the step that consumes the block a true no-op for it.
// State per filter: the last body-frame sample `last` [6][stride] and the previous message time `prev_ut` [stride].
static __global__ void k_ins_body(int B, long stride, const double *__restrict__ gyro, const double *__restrict__ accel,
                                  const double *__restrict__ raw_dt, const int64_t *__restrict__ utimes, int64_t utime,
                                  const uint8_t *__restrict__ valid, InsFrame f, double *__restrict__ last, int64_t *__restrict__ prev_ut,
                                  double *__restrict__ out, uint8_t *__restrict__ valid_out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const bool on = valid ? valid[b] != 0 : true;
  if (valid_out) valid_out[b] = on ? 1 : 0;
  double blk[7];
  if (on) {
    double g[3] = { gyro[b], gyro[(long) B + b], gyro[2L * B + b] };
    const double a[3] = { accel[b], accel[(long) B + b], accel[2L * B + b] };
    if (raw_dt) {
      const double rd = raw_dt[b];
      for (int i = 0; i < 3; i++) g[i] = g[i] / rd;
    }
    ins_rotate(f.rot, g, blk);
    ins_rotate(f.rot, a, blk + 3);
    if (f.translate)
      for (int i = 0; i < 3; i++) blk[3 + i] += f.trans[i];
    double dt = f.dt_default;
    if (f.dt_from_utimes) {
      const int64_t ut = utimes ? utimes[b] : utime;
      const int64_t prev = prev_ut[b];
      if (prev != 0) dt = (double) (ut - prev) * 1E-6;
      prev_ut[b] = ut;
    }
    blk[6] = dt;
    for (int i = 0; i < 6; i++) last[(long) i * stride + b] = blk[i];
  } else {
    for (int i = 0; i < 6; i++) blk[i] = last[(long) i * stride + b];
    blk[6] = 0.0;   // dt = 0 on its own last sample: pose, velocity, biases and covariance stay (the step call that consumes the block
                    // keeps the angular-velocity / acceleration entries too when it is given the mask: pb_set_imu_valid)
  }
  for (int i = 0; i < 7; i++) out[(long) i * B + b] = blk[i];
}

// A filter WITHOUT an IMU message in a batched message: the reference does nothing at all for it (its handler returned NULL).  Its
// block entry says dt = 0 -- pose, velocity, biases and covariance stay whatever the sample -- but the step also writes the
// angular-velocity and acceleration entries, omega = gyro - gyro_bias, a = accel - accel_bias, with the biases of NOW.  Right in
// front of the step kernel, for the filters whose mask is 0, the sample becomes (omega + gyro_bias, a + accel_bias) of the PRIOR the
// step is about to read: the step then re-derives what is there (exactly for 15 states, to the last bit of the sum for 21) and the
// update that shares the kernel sees the same state the reference's would.  One small launch and a 56-byte copy per filter in
// front of the hot kernel instead of selects inside it -- measured: the selects cost k_step_quad 6-7 % (39.7 vs 37.0-37.9 us at
// 64k filters, same box); this costs the hot kernels nothing, and nothing at all when no mask is given.
template <int NS>
static __global__ void k_imu_idle_prepare(const double *__restrict__ st, const uint8_t *__restrict__ valid, const double *__restrict__ imu,
                                          double *__restrict__ imu_out, int B)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double v[7];
#pragma unroll
  for (int i = 0; i < 7; i++) v[i] = imu[(long) i * B + b];
  if (valid[b] == 0) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      v[i] = st[S::eidx(L::OFF_VEC + i, b)] + (NS == 21 ? st[S::eidx(L::OFF_VEC + (NS == 21 ? 15 : 0) + i, b)] : 0.0);
      v[3 + i] = st[S::eidx(L::OFF_VEC + 12 + i, b)] + (NS == 21 ? st[S::eidx(L::OFF_VEC + (NS == 21 ? 18 : 0) + i, b)] : 0.0);
    }
    v[6] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 7; i++) imu_out[(long) i * B + b] = v[i];
}

}  // namespace pb

