// rbis_jointfilt.hpp -- the joint-position filters in front of the leg kinematics: step 0 of leg_estimate::updateOdometry
// (motion_estimate/src/leg_estimate/leg_estimate.cpp:411-428, state_estimator.legodo.filter_joint_positions = lowpass |
// kalman), one filter per joint and robot:
//   estimate_tools/src/filter_tools/Filter.cpp:4-65                          LowPassFilter: 14-tap FIR, renormalised taps, the first
//                                                                            sample fills the whole window
//   estimate_tools/src/kalman_filter_tools/simple_kalman_filter.cpp:11-50    SimpleKalmanFilter: [position, velocity] with float
//                                                                            noise members, float residual and innovation
//                                                                            variance, the gain taken from P (not from Pprior)
// The reference filters joint_position[i] for i < NUM_FILT_JOINTS = 28 (leg_estimate.hpp:59) in place in a std::vector<float>
// (leg_estimate.hpp:91-93): a filtered position is ROUNDED TO FLOAT before the kinematics read it, and so it is here.  Only
// the rows the two kinematic chains read are filtered (nothing else reads a joint position); the others are copied.
// The handler's torque adjustment runs BEFORE the filters (rbis_legodo_update.cpp:231-241): k_joint_filter applies it to the
// chain rows itself, the caller then passes joint_effort = NULL to the kinematics.
// The arithmetic is written out in the reference's operation order with contraction into fused multiply-adds switched off,
// so that the host and the device version (and the CPU oracle) agree to the bit.
// All robots of a batch receive their messages in lock-step (one call = one message each), so the FIR window's write position,
// the "first sample" flag and the previous time stamp are per-context host values and kernel arguments, not per-robot state.
#pragma once

#include <stdint.h>

#include "rbis_legodo.hpp"

namespace pb {

constexpr int JF_NONE = 0, JF_LOWPASS = 1, JF_KALMAN = 2;
constexpr int JF_TAPS = 14;              // Filter.cpp:16
constexpr int JF_NUM_FILT_JOINTS = 28;   // leg_estimate.hpp:59
constexpr int JF_MAXROWS = 2 * LEG_MAXJ; // the rows two chains can name
constexpr int JF_KSTATE = 6;             // x_est (2), P (4, row-major: it is not kept symmetric)

// Filter.cpp:18-37: the taps as printed, divided by their (sequential) sum
inline void jf_lowpass_coeffs(double c[JF_TAPS])
{
  const double taps[JF_TAPS] = { 0.005271208909706, 0.05204636786996, 0.05315761628452, 0.07562063364867, 0.09406855250555,
                                 0.108343855546,    0.1160610649931,  0.1160610649931,  0.108343855546,   0.09406855250555,
                                 0.07562063364867,  0.05315761628452, 0.05204636786996, 0.005271208909706 };
  double sum = 0;
  for (int i = 0; i < JF_TAPS; i++) sum += taps[i];
  for (int i = 0; i < JF_TAPS; i++) c[i] = taps[i] / sum;
}

struct JfPar {
  int mode;                 // JF_*
  int nf;                   // filtered rows (chain rows < 28, each once)
  int nadj;                 // rows with a torque-adjustment gain
  int row[JF_MAXROWS];      // [nf]
  int adj_row[JF_MAXROWS];  // [nadj]
  float adj_gain[JF_MAXROWS];
  double coef[JF_TAPS];     // lowpass
  float pn_pos, pn_vel, r;  // kalman: process_noise_pos_, process_noise_vel_, R (float members, simple_kalman_filter.hpp:27-41)
};

// LowPassFilter::processSample (Filter.cpp:44-65).  win(i) = the i-th oldest sample of the window AFTER the new sample has
// been pushed (samples_buf.at(i)); the samples are float values (the message's, or the float torque adjustment's), kept as floats.
template <class WIN>
PB_HD float jf_lowpass(const double (&coef)[JF_TAPS], WIN &&win)
{
#pragma clang fp contract(off)
  double acc = 0.;
#pragma unroll
  for (int i = 0; i < JF_TAPS; i++) acc += coef[JF_TAPS - i - 1] * (double) win(i);
  return (float) acc;
}

// SimpleKalmanFilter::processSample (simple_kalman_filter.cpp:25-50) after the first sample; s = { x_est, P row-major }.
// The Eigen expressions written out: 2 x 2 products are a(i,0) b(0,j) + a(i,1) b(1,j); a product with an exact 0 or 1 factor
// is dropped where it cannot change the value of a finite operand.
PB_HD float jf_kalman(double (&s)[JF_KSTATE], double dt, float x_in, float pn_pos, float pn_vel, float r)
{
#pragma clang fp contract(off)
  const double x0 = s[0], x1 = s[1], P00 = s[2], P01 = s[3], P10 = s[4], P11 = s[5];
  const double q0 = (double) pn_pos * dt, q1 = (double) pn_vel / dt;  // Q (:37)
  const double j0 = x0 + dt * x1, j1 = x1;                             // jprior = F x_est (:38)
  const double fp00 = P00 + dt * P10, fp01 = P01 + dt * P11;           // F P
  const double pp00 = (fp00 + fp01 * dt) + q0, pp01 = fp01;            // Pprior = F P F^T + Q (:39)
  const double pp10 = P10 + P11 * dt, pp11 = P11 + q1;
  const float resid = (float) ((double) x_in - j0);                    // meas_resid is a float (:40, hpp:32)
  const float S = (float) (pp00 + (double) r);                         // S is a float (:41, hpp:34)
  const double k0 = P00 / (double) S, k1 = P10 / (double) S;           // K = (P Hk) / S: P, not Pprior (:42)
  s[0] = j0 + k0 * (double) resid;                                     // :43
  s[1] = j1 + k1 * (double) resid;
  const double m00 = 1.0 - k0, m10 = 0.0 - k1;                         // (I - K Hk^T) Pprior (:44)
  s[2] = m00 * pp00;
  s[3] = m00 * pp01;
  s[4] = m10 * pp00 + pp10;
  s[5] = m10 * pp01 + pp11;
  return (float) s[0];                                                 // x_filtered -> joint_position[i] (a float vector)
}

#if defined(__HIPCC__)
// V consecutive robots per lane (V = 4: every access is 16 bytes -- the scalar version's 4-byte accesses moved 256 B per wave
// instruction and reached 0.54 of the HBM roofline at 64k robots; V = 1 for batches that are not a multiple of four) and row of the
// message; blockIdx.y = row.  A row no filter owns is copied (torque-adjusted where it has a gain: chain rows >= 28).
// in/out: [rows][B] floats; ring: [JF_TAPS][nf][B] floats; kst: [JF_KSTATE][nf][B] doubles.
// head = window slot of the OLDEST sample (the one this call overwrites); first = this is the first message since
// pb_joint_filter_init.
template <int V>
struct JfVec {
  typedef float f_t __attribute__((ext_vector_type(V)));
  typedef double d_t __attribute__((ext_vector_type(V)));
};
template <>
struct JfVec<1> {
  typedef float f_t;
  typedef double d_t;
};
template <int V>
__device__ __forceinline__ void jf_ldf(const float *p, float (&o)[V])
{
  const typename JfVec<V>::f_t v = *reinterpret_cast<const typename JfVec<V>::f_t *>(p);
  if constexpr (V == 1) o[0] = v;
  else {
#pragma unroll
    for (int i = 0; i < V; i++) o[i] = v[i];
  }
}
template <int V>
__device__ __forceinline__ void jf_stf(float *p, const float (&o)[V])
{
  typename JfVec<V>::f_t v;
  if constexpr (V == 1) v = o[0];
  else {
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = o[i];
  }
  *reinterpret_cast<typename JfVec<V>::f_t *>(p) = v;
}
template <int V>
__device__ __forceinline__ void jf_ldd(const double *p, double (&o)[V])
{
  const typename JfVec<V>::d_t v = *reinterpret_cast<const typename JfVec<V>::d_t *>(p);
  if constexpr (V == 1) o[0] = v;
  else {
#pragma unroll
    for (int i = 0; i < V; i++) o[i] = v[i];
  }
}
template <int V>
__device__ __forceinline__ void jf_std(double *p, const double (&o)[V])
{
  typename JfVec<V>::d_t v;
  if constexpr (V == 1) v = o[0];
  else {
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = o[i];
  }
  *reinterpret_cast<typename JfVec<V>::d_t *>(p) = v;
}
template <int V>
static __global__ __launch_bounds__(256) void k_joint_filter(JfPar par, int B, const float *__restrict__ pos, const float *__restrict__ vel,
                                                             const float *__restrict__ eff, float *__restrict__ out,
                                                             float *__restrict__ ring, double *__restrict__ kst, int head, int first,
                                                             double dt)
{
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) * V;   // (B is a multiple of V: the launcher's choice of V)
  const int row = blockIdx.y;
  if (b >= B) return;
  int f = -1;  // (uniform)
  for (int i = 0; i < par.nf; i++) f = (par.row[i] == row) ? i : f;
  float x[V];
  jf_ldf<V>(pos + (long) row * B + b, x);
  if (eff) {
    float g = 0.0f;
    for (int a = 0; a < par.nadj; a++) g = (par.adj_row[a] == row) ? par.adj_gain[a] : g;  // (uniform)
    float e[V];
    jf_ldf<V>(eff + (long) row * B + b, e);
#pragma unroll
    for (int v = 0; v < V; v++) x[v] = torque_adjust(x[v], e[v], g);
  }
  float y[V];
#pragma unroll
  for (int v = 0; v < V; v++) y[v] = x[v];
  const long nfB = (long) par.nf * B, fb = (long) f * B + b;
  if (f >= 0 && par.mode == JF_LOWPASS) {
    float w[JF_TAPS][V];
#pragma unroll
    for (int i = 0; i < JF_TAPS - 1; i++) {  // the 13 samples that stay, oldest first: slots head + 1 ... head + 13 (mod 14)
      int s = head + 1 + i;
      s = s >= JF_TAPS ? s - JF_TAPS : s;
      if (first) {
#pragma unroll
        for (int v = 0; v < V; v++) w[i][v] = x[v];
      } else {
        jf_ldf<V>(ring + (long) s * nfB + fb, w[i]);
      }
    }
#pragma unroll
    for (int v = 0; v < V; v++) {
      w[JF_TAPS - 1][v] = x[v];
      y[v] = jf_lowpass(par.coef, [&](int i) { return w[i][v]; });
    }
    if (first) {
#pragma unroll
      for (int s = 0; s < JF_TAPS; s++) jf_stf<V>(ring + (long) s * nfB + fb, x);
    } else {
      jf_stf<V>(ring + (long) head * nfB + fb, x);
    }
  } else if (f >= 0) {
    double s[JF_KSTATE][V];
    if (first) {  // simple_kalman_filter.cpp:27-34: x_est = (x, x_dot), P stays the identity of the constructor, output = input
      float xd[V];
      jf_ldf<V>(vel + (long) row * B + b, xd);
#pragma unroll
      for (int v = 0; v < V; v++) {
        s[0][v] = (double) x[v];
        s[1][v] = (double) xd[v];
        s[2][v] = 1.0; s[3][v] = 0.0; s[4][v] = 0.0; s[5][v] = 1.0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < JF_KSTATE; i++) jf_ldd<V>(kst + (long) i * nfB + fb, s[i]);
#pragma unroll
      for (int v = 0; v < V; v++) {
        double sv[JF_KSTATE];
#pragma unroll
        for (int i = 0; i < JF_KSTATE; i++) sv[i] = s[i][v];
        y[v] = jf_kalman(sv, dt, x[v], par.pn_pos, par.pn_vel, par.r);
#pragma unroll
        for (int i = 0; i < JF_KSTATE; i++) s[i][v] = sv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < JF_KSTATE; i++) jf_std<V>(kst + (long) i * nfB + fb, s[i]);
  }
  jf_stf<V>(out + (long) row * B + b, y);
}
#endif

}  // namespace pb
