// rbis_device.hpp -- per-filter RBIS EKF arithmetic, one filter per lane, everything in registers.
//
// Structured restatement of the reference maths (paths relative to the reference tree):
//   state-estimator/src/mav_state_est/rbis.cpp:12-35    process linearisation   -> make_process_blocks
//   state-estimator/src/mav_state_est/rbis.cpp:37-75    insUpdateState          -> ins_update_state
//   state-estimator/src/mav_state_est/rbis.cpp:77-122   insUpdateCovariance     -> ins_update_covariance
//   state-estimator/src/mav_state_est/rbis.cpp:124-227  K / dP / apply-delta    -> measurement_update
//
// Design (gfx950): the covariance lives symmetric-packed in VGPRs (n=15: 120 doubles = 240 VGPRs, one wave
// per SIMD).  Ad = I + Ac*dt has 5 (n=15) or 8 (n=21) non-zero 3x3 blocks and factors EXACTLY into three
// elementary block-row operations Ad = E2*E1*E3 (E3: row Delta, E1: row v, E2: row chi; the cross products of
// their off-identity parts vanish), so P <- Ad P Ad^T is three in-place symmetric congruences (`rowop`) that
// touch only the affected block row: ~0.9k FMA instead of the reference's two dense 21^3 GEMMs (18.5k FMA,
// rbis.cpp:118).  The measurement update is the rank-m downdate P -= W D^-1 W^T with W = P[:,idx] L^-T from an
// unpivoted LDL^T of S (the reference forms K*C*P densely, rbis.cpp:140).
//
// All loops have compile-time bounds and are fully unrolled so that every array index is a constant and the
// arrays stay in registers (cdna_hip_programming.md rule 20: runtime-indexed arrays go to scratch).
#pragma once

#include <cmath>
#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PB_HD __host__ __device__ __forceinline__
#else
#define PB_HD inline
#endif

namespace pb {

template <int NS>
struct Lay {
  static constexpr int NB = NS / 3;             // 3x3 block rows
  static constexpr int NP = NS * (NS + 1) / 2;  // packed covariance entries
  // component order of the device state array [NC][stride]
  static constexpr int OFF_VEC = 0;
  static constexpr int OFF_QUAT = NS;
  static constexpr int OFF_LL = NS + 4;
  static constexpr int OFF_P = NS + 5;
  static constexpr int NC = NS + 5 + NP;
};

// packed index of P(i,j): lower triangle, row-major
PB_HD constexpr int pk(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// compiler-level memory clobber: loads behind it are re-issued instead of being shared with loads in front of it
PB_HD void reload_fence() { asm volatile("" ::: "memory"); }

// row / column of a packed index
PB_HD constexpr int pk_row(int p)
{
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= p) i++;
  return i;
}
PB_HD constexpr int pk_col(int p) { return p - pk_row(p) * (pk_row(p) + 1) / 2; }

// compile-time loop: fn(std::integral_constant<int, 0>) ... fn(std::integral_constant<int, N-1>)
template <class Fn, int... I>
PB_HD void static_for_impl(Fn &&fn, std::integer_sequence<int, I...>)
{
  (fn(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
PB_HD void static_for(Fn &&fn)
{
  static_for_impl(fn, std::make_integer_sequence<int, N>{});
}

// The process model cuts the state into a dynamic core c = {v, chi, Delta} [+ b = {gyro bias, accel bias}] and a passive
// part p = {omega, accel} (identity rows/cols of Ad, never a source; rbis.cpp:12-35, see rbis_coop.hpp).
//   core sub index 0..NSC-1 -> full state index;  passive index 0..5 -> full state index (omega 0..2, accel 12..14)
PB_HD constexpr int core_full(int s) { return s < 9 ? 3 + s : 15 + (s - 9); }
PB_HD constexpr int passive_full(int p) { return p < 3 ? p : 12 + (p - 3); }

// ------------------------------------------------------------------------------------------------------------
// Storage layout in HBM.  The state array is cut into TILES of 64 filters; one tile holds the whole state
// of its 64 filters as NROW rows of 64 x 16 bytes: row r = the component PAIR (slot 2r, slot 2r+1) of each filter, so a
// lane moves 16 bytes per access (buffer_load/store_dwordx4; 1 KiB per wave instruction) and a wave's whole round trip
// stays inside one contiguous 70 KiB (n=15) / 129 KiB (n=21) block.  "Slot" order is a permutation of the canonical
// component order of Lay<NS>, chosen so that the waves that share a tile (the two roles of rbis_coop.hpp for n = 15, the
// four of rbis_quad.hpp for n = 21) own disjoint row ranges and store their rows in increasing order:
//   n=15 role C: x[v chi Delta] | quat | P_cc packed by core sub index
//        role P: P_c,omega panel | P_c,accel panel | P_pp | loglik | x[omega accel]
//   n=21 wave 0: P_cc | loglik              wave 1: P_bc, P_bb by bias row | x[bg ba] | x[omega]
//        wave 2: P_(cb),omega panel | P_omega,omega | x[v chi Delta] | quat
//        wave 3: P_(cb),accel panel | P_accel,(omega accel) | x[accel] | pad
//   (the two-role kernels that remain for n = 21 -- stand-alone updates, replay -- write the two rows whose slots belong
//    to different roles, Tab::split2, with 8-byte stores)
// Element (component c, filter b) lives at double index  (b/64)*NSLOT*64 + (slot(c)/2)*128 + (b%64)*2 + slot(c)%2.
// ------------------------------------------------------------------------------------------------------------
template <int NS>
struct Slots {
  using L = Lay<NS>;
  static constexpr bool HB = (NS == 21);
  static constexpr int NSC = HB ? 15 : 9;   // core sub-state
  static constexpr int NSB = HB ? 5 : 3;    // its 3x3 block rows
  static constexpr int NROW = (L::NC + 1) / 2, NSLOT = 2 * NROW;
  static constexpr int TILE = 64;
  static constexpr long TILE_DOUBLES = (long) NSLOT * TILE;
  static constexpr unsigned TILE_BYTES = (unsigned) (NSLOT * TILE * 8);
  struct Tab {
    short slot_of[L::NC];
    short comp_of[NSLOT];
    signed char role2[NSLOT];  // two-role mapping (rbis_coop.hpp): 0 = role C writes this slot, 1 = role P
    bool split2[NROW];         // the row's two slots belong to different roles of the two-role mapping (8-byte stores)
    int ncore;                 // n = 15: slots of role C's rows
    int nq[4];                 // n = 21: end slot of each of the four waves' rows (rbis_quad.hpp)
  };
  static constexpr Tab make()
  {
    Tab t{};
    int s = 0, role = 0;
    auto put = [&](int c) { t.slot_of[c] = (short) s; t.role2[s] = (signed char) role; t.comp_of[s++] = (short) c; };
    if (!HB) {
      // ---- n = 15: role C's rows, then role P's ----
      for (int i = 0; i < 9; i++) put(L::OFF_VEC + 3 + i);
      for (int i = 0; i < 4; i++) put(L::OFF_QUAT + i);
      for (int i = 0; i < 9; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(core_full(i), core_full(j)));
      t.ncore = s;
      t.nq[0] = t.nq[1] = s;
      role = 1;
      for (int J = 0; J < 2; J++)
        for (int i = 0; i < NSC; i++)
          for (int cc = 0; cc < 3; cc++) put(L::OFF_P + pk(core_full(i), passive_full(3 * J + cc)));
      for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(passive_full(i), passive_full(j)));
      put(L::OFF_LL);
      for (int i = 0; i < 6; i++) put(L::OFF_VEC + passive_full(i));
      t.nq[2] = s;
    } else {
      // ---- n = 21: the four waves of rbis_quad.hpp, balanced by ARITHMETIC (the state vector and the quaternion, whose
      // update is the expensive part, go with the lightest panel).  `role` = the writer in the two-role mapping. ----
      // wave 0 (CC): P_cc, loglik
      for (int i = 0; i < 9; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(core_full(i), core_full(j)));
      put(L::OFF_LL);
      t.nq[0] = s;
      // wave 1 (CB): P_bc and P_bb by bias row, x[bg ba], x[omega]
      for (int i = 9; i < NSC; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(core_full(i), core_full(j)));
      for (int i = 0; i < 6; i++) put(L::OFF_VEC + 15 + i);
      role = 1;
      for (int i = 0; i < 3; i++) put(L::OFF_VEC + passive_full(i));
      t.nq[1] = s;
      t.ncore = s;
      // wave 2 (PW): the omega block column, P_omega,omega, x[v chi Delta], quat
      for (int i = 0; i < NSC; i++)
        for (int cc = 0; cc < 3; cc++) put(L::OFF_P + pk(core_full(i), passive_full(cc)));
      for (int i = 0; i < 3; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(passive_full(i), passive_full(j)));
      role = 0;
      for (int i = 0; i < 9; i++) put(L::OFF_VEC + 3 + i);
      for (int i = 0; i < 4; i++) put(L::OFF_QUAT + i);
      t.nq[2] = s;
      // wave 3 (PA): the accel block column, P_accel,(omega accel), x[accel], pad
      role = 1;
      for (int i = 0; i < NSC; i++)
        for (int cc = 0; cc < 3; cc++) put(L::OFF_P + pk(core_full(i), passive_full(3 + cc)));
      for (int i = 3; i < 6; i++)
        for (int j = 0; j <= i; j++) put(L::OFF_P + pk(passive_full(i), passive_full(j)));
      for (int i = 3; i < 6; i++) put(L::OFF_VEC + passive_full(i));
    }
    while (s < NSLOT) { t.role2[s] = (signed char) role; t.comp_of[s++] = -1; }  // padding slot (n=21: 257 components in 129 rows)
    t.nq[3] = s;
    for (int r = 0; r < NROW; r++) t.split2[r] = t.role2[2 * r] != t.role2[2 * r + 1];
    return t;
  }
  static constexpr Tab T = make();
  static_assert(T.nq[0] % 2 == 0 && T.nq[1] % 2 == 0 && T.nq[2] % 2 == 0, "every wave must own whole rows");
  // prefetch hint of the two-role kernels: rows [0, ROW_SPLIT) are (mostly) role C's, [ROW_SPLIT, NROW) role P's
  static constexpr int ROW_SPLIT = T.ncore / 2;
  // n = 21, four-wave mapping: rows [QROW[w], QROW[w+1]) belong to wave w
  static constexpr int QROW[5] = { 0, T.nq[0] / 2, T.nq[1] / 2, T.nq[2] / 2, T.nq[3] / 2 };
  PB_HD static constexpr int slot(int comp) { return T.slot_of[comp]; }
  // double index of (component, filter) in a state array
  PB_HD static long eidx(int comp, long b)
  {
    const int sl = T.slot_of[comp];
    return (b >> 6) * TILE_DOUBLES + (long) (sl >> 1) * (2 * TILE) + (b & 63) * 2 + (sl & 1);
  }
  PB_HD static long eidx_slot(int sl, long b) { return (b >> 6) * TILE_DOUBLES + (long) (sl >> 1) * (2 * TILE) + (b & 63) * 2 + (sl & 1); }
};

// block rows of the RBIS vector (rbis.hpp:22-24 + eigen_utils::RigidBodyState)
enum { BW = 0, BV = 1, BCHI = 2, BPOS = 3, BACC = 4, BBG = 5, BBA = 6 };

struct Consts {
  double g;        // |g_vec| (eigen_utils g_val)
  double chi_tol;  // chiToQuat fold tolerance
  // optional per-filter process noise [4][B] = q_gyro, q_accel, q_gyro_bias, q_accel_bias (device memory); used by
  // parameter sweeps / noise identification (noise_id.cpp:9-42) where every filter of the batch carries its own q
  const double *qblk = nullptr;
  // k_step_coop: 1 = give each of the 8 XCDs one contiguous filter range (workgroups are dealt round-robin to XCDs)
  int xcd_remap = 0;
  // k_step_coop: 1 = TWO workgroups per 64-filter tile, each with its 64 lanes on one half of it (lane l and lane l + 32 carry the
  // same filter: the same addresses, the same values) -- twice the workgroups in flight for batches that do not fill the chip
  int half_tiles = 0;
};

// ------------------------------------------------------------------------------------------------------------
// quaternion helpers (Eigen conventions: Hamilton product, w first)
// ------------------------------------------------------------------------------------------------------------
PB_HD void quat_mul(const double (&a)[4], const double (&b)[4], double (&o)[4])
{
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

// QuaternionBase::toRotationMatrix, row-major
PB_HD void quat_to_rot(const double (&q)[4], double (&R)[9])
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// sin and cos of an angle with |x| < 2^20 (joint angles, half rotation angles): Cody-Waite reduction by pi/2 in two pieces, then the classic minimax
// polynomials on [-pi/4, pi/4] (coefficients: fdlibm's __kernel_sin / __kernel_cos, < 1 ulp).  No large-argument path, so
// ~35 instructions for the pair where the library call costs ~3x that plus the registers of its Payne-Hanek branch.
PB_HD void sincos_joint(double x, double &s, double &c)
{
  const double kf = rint(x * 6.36619772367581382433e-01);
  double r = fma(-kf, 1.57079632673412561417e+00, x);
  r = fma(-kf, 6.07710050650619224932e-11, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double sn = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int k = (int) kf & 3;
  const double a = (k & 1) ? cs : sn, b = (k & 1) ? sn : cs;
  s = (k & 2) ? -a : a;
  c = ((k + 1) & 2) ? -b : b;
}

// eigen_utils chiToQuat: if |chi| > tol { q <- q * AngleAxis(|chi|, chi/|chi|); chi <- 0 }
// (sin / cos through sincos_joint: the library call carries a large-argument branch and ~2x the instructions, four times per step
// on the critical wave of every step kernel)
PB_HD void fold_chi(double (&chi)[3], double (&q)[4], double tol)
{
  const double n = sqrt(chi[0] * chi[0] + chi[1] * chi[1] + chi[2] * chi[2]);
  if (n > tol) {
    double s, c;
    sincos_joint(0.5 * n, s, c);
    const double f = s / n;
    const double dq[4] = { c, f * chi[0], f * chi[1], f * chi[2] };
    double o[4];
    quat_mul(q, dq, o);
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = o[3];
    chi[0] = chi[1] = chi[2] = 0.0;
  }
}

// RigidBodyState::addState(d) where d = RBIS(dvec) (ctor folds d's chi first):
//   vec += d.vec; chiToQuat(); quat *= d.quat
template <int NS>
PB_HD void add_delta(double (&x)[NS], double (&q)[4], double (&d)[NS], double tol)
{
  double dchi[3] = { d[6], d[7], d[8] };
  double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
  fold_chi(dchi, dq, tol);
  d[6] = dchi[0]; d[7] = dchi[1]; d[8] = dchi[2];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] += d[i];
  double chi[3] = { x[6], x[7], x[8] };
  fold_chi(chi, q, tol);
  x[6] = chi[0]; x[7] = chi[1]; x[8] = chi[2];
  double o[4];
  quat_mul(q, dq, o);
  q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = o[3];
}

// eigen_utils subtractQuats(q1, q2): rotation vector of q2^-1 * q1 with the angle in [-pi, pi]
PB_HD void subtract_quats(const double (&q1)[4], const double (&q2)[4], double (&out)[3])
{
  const double n2 = q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3];
  const double in2 = 1.0 / n2;
  const double q2i[4] = { q2[0] * in2, -q2[1] * in2, -q2[2] * in2, -q2[3] * in2 };
  double r[4];
  quat_mul(q2i, q1, r);
  const double n = sqrt(r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  if (n != 0.0) {
    // angle = 2 atan2(n, |w|) in [0, pi]; axis = vec / (sign(w) n); bot_mod2pi maps an angle of exactly pi (w == 0 to
    // rounding) to -pi
    double angle = 2.0 * atan2(n, fabs(r[0]));
    if (angle >= 3.141592653589793) angle -= 6.283185307179586;
    const double f = (r[0] < 0 ? -angle : angle) / n;
    out[0] = r[1] * f; out[1] = r[2] * f; out[2] = r[3] * f;
  } else {
    out[0] = out[1] = out[2] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------------------------

// rbis.cpp:37-75.  x is updated in place; uses the PRE-update v and quat on every right-hand side.
template <int NS>
PB_HD void ins_update_state(double (&x)[NS], double (&q)[4], const double (&gyro)[3], const double (&accel)[3],
                            double dt, const Consts &k)
{
  double R[9];
  quat_to_rot(q, R);
  double w[3], a[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    w[i] = gyro[i] - (NS == 21 ? x[15 + i] : 0.0);   // :50
    a[i] = accel[i] - (NS == 21 ? x[18 + i] : 0.0);  // :51
  }
  const double v[3] = { x[3], x[4], x[5] };
  // -w x v + R^T g + a   (:55-56); R^T g = -g * (third row of R)
  const double wxv[3] = { w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0] };
  double d[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) d[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    d[3 + i] = ((-wxv[i]) + ((-k.g) * R[6 + i] + a[i])) * dt;
    d[6 + i] = w[i] * dt;                                                        // :58
    d[9 + i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2]) * dt;  // :59
    x[i] = w[i];
    x[12 + i] = a[i];
  }
  add_delta<NS>(x, q, d, k.chi_tol);  // :63,:69
}

// The quaternion part of ins_update_state alone -- bit-identical to what it leaves in q -- from the three things it depends
// on: the prior chi (x[6..8]), the gyro bias (x[15..17], 21 states) and the prior quaternion.  For callers that only need the
// orientation after an IMU step (the leg odometry slaved to it): no velocity / position arithmetic, 10 components instead of
// the whole state vector.
template <int NS>
PB_HD void ins_update_quat(const double (&chi_prior)[3], const double (&gyro_bias)[3], double (&q)[4], const double (&gyro)[3], double dt,
                           const Consts &k)
{
  double dchi[3], chi[3];
#pragma unroll
  for (int i = 0; i < 3; i++) dchi[i] = (gyro[i] - (NS == 21 ? gyro_bias[i] : 0.0)) * dt;  // rbis.cpp:50,58
  // add_delta: the increment's own chi is folded first (RigidBodyState(vec)), the vector added, the state's chi folded, then
  // the two rotations composed
  double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
  fold_chi(dchi, dq, k.chi_tol);
#pragma unroll
  for (int i = 0; i < 3; i++) chi[i] = chi_prior[i] + dchi[i];
  fold_chi(chi, q, k.chi_tol);
  double o[4];
  quat_mul(q, dq, o);
  q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = o[3];
}

// One elementary block-row congruence P <- E P E^T, E = I + (block row I <- sum_s A[s] * block row SRC[s]).
// KIND[s]: 0 dense 3x3 (A[s][0..8] row-major), 1 hat(a) = [[0,-a2,a1],[a2,0,-a0],[-a1,a0,0]] (A[s][0..2] = a),
//          2 scalar*identity (A[s][0] holds the scalar).
// Only the T_j of the "core" columns j in {I} u SRC are live together; every other column is updated as soon
// as its T_j exists, which keeps the temporaries at <= 4 blocks instead of NB.
template <int NS, int I, int NSRC>
struct RowOp {
  static constexpr int NB = NS / 3;
  static constexpr int NP = NS * (NS + 1) / 2;

  PB_HD static bool nz(int kind, int r, int c) { return kind == 0 || (kind == 1 && r != c) || (kind == 2 && r == c); }

  // element (r, c) of source s's 3x3 block
  PB_HD static double el(const double (&A)[9], int kind, int r, int c)
  {
    if (kind == 0) return A[3 * r + c];
    if (kind == 2) return A[0];
    // hat(a)
    if (r == 0 && c == 1) return -A[2];
    if (r == 0 && c == 2) return A[1];
    if (r == 1 && c == 0) return A[2];
    if (r == 1 && c == 2) return -A[0];
    if (r == 2 && c == 0) return -A[1];
    return A[0];  // (2,1)
  }

  PB_HD static bool is_core(const int (&src)[NSRC], int j)
  {
    bool c = (j == I);
#pragma unroll
    for (int s = 0; s < NSRC; s++) c = c || (src[s] == j);
    return c;
  }

  // T_j = sum_s A_s P_{src_s, j}
  PB_HD static void make_T(const double (&P)[NP], const int (&src)[NSRC], const int (&kind)[NSRC],
                           const double (&A)[NSRC][9], int j, double (&T)[9])
  {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double acc = 0.0;
        bool first = true;
#pragma unroll
        for (int s = 0; s < NSRC; s++) {
#pragma unroll
          for (int qq = 0; qq < 3; qq++) {
            if (nz(kind[s], r, qq)) {
              const double a = el(A[s], kind[s], r, qq);
              const double p = P[pk(3 * src[s] + qq, 3 * j + c)];
              acc = first ? a * p : fma(a, p, acc);
              first = false;
            }
          }
        }
        T[3 * r + c] = acc;
      }
    }
  }

  PB_HD static void apply(double (&P)[NP], const int (&src)[NSRC], const int (&kind)[NSRC],
                          const double (&A)[NSRC][9])
  {
    double T[NB][9];
    // 1. core columns, all from the ORIGINAL P
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (is_core(src, j)) make_T(P, src, kind, A, j, T[j]);
    // 2. diagonal block: P_II += T_I + T_I^T + sum_s T_{src_s} A_s^T   (lower triangle only)
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c <= r; c++) {
        double acc = T[I][3 * r + c] + T[I][3 * c + r];
#pragma unroll
        for (int s = 0; s < NSRC; s++) {
#pragma unroll
          for (int qq = 0; qq < 3; qq++) {
            if (nz(kind[s], c, qq)) acc = fma(T[src[s]][3 * r + qq], el(A[s], kind[s], c, qq), acc);
          }
        }
        P[pk(3 * I + r, 3 * I + c)] += acc;
      }
    }
    // 3. off-diagonal core blocks, 4. every other column as soon as its T exists
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (j == I) continue;
      if (is_core(src, j)) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) P[pk(3 * I + r, 3 * j + c)] += T[j][3 * r + c];
      } else {
        double Tj[9];
        make_T(P, src, kind, A, j, Tj);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) P[pk(3 * I + r, 3 * j + c)] += Tj[3 * r + c];
      }
    }
  }
};

// rbis.cpp:77-122 about the PRIOR state (rbis_update_interface.cpp:39): xp = prior vec (its omega is the
// previous step's), qp = prior quat.
template <int NS>
PB_HD void ins_update_covariance(double (&P)[NS * (NS + 1) / 2], const double (&xp)[NS], const double (&qp)[4],
                                 double dt, double q_gyro, double q_accel, double q_gyro_bias,
                                 double q_accel_bias, const Consts &k)
{
  double R[9];
  quat_to_rot(qp, R);
  const double w[3] = { xp[0], xp[1], xp[2] };
  const double v[3] = { xp[3], xp[4], xp[5] };
  const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };  // R^T g_vec
  // blocks of Ac*dt (rbis.cpp:20-33)
  const double wd[3] = { w[0] * dt, w[1] * dt, w[2] * dt };
  const double gd[3] = { gb[0] * dt, gb[1] * dt, gb[2] * dt };
  const double vd[3] = { v[0] * dt, v[1] * dt, v[2] * dt };
  // hat() operands of the skew blocks: -skew(w) dt = hat(-w dt), skew(R^T g) dt = hat(g dt), -skew(v) dt = hat(-v dt)
  const double a_mw[3] = { -wd[0], -wd[1], -wd[2] };
  const double a_g[3] = { gd[0], gd[1], gd[2] };
  const double a_mv[3] = { -vd[0], -vd[1], -vd[2] };
  double A_R[9], A_RV[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    A_R[3 * i + 0] = R[3 * i + 0] * dt;
    A_R[3 * i + 1] = R[3 * i + 1] * dt;
    A_R[3 * i + 2] = R[3 * i + 2] * dt;
    // (-R * skew(v)) dt : row i = -(R_i x-cross) -> column c of skew(v)
    A_RV[3 * i + 0] = -(R[3 * i + 1] * vd[2] - R[3 * i + 2] * vd[1]);
    A_RV[3 * i + 1] = -(R[3 * i + 2] * vd[0] - R[3 * i + 0] * vd[2]);
    A_RV[3 * i + 2] = -(R[3 * i + 0] * vd[1] - R[3 * i + 1] * vd[0]);
  }

  // E3: row Delta <- + R dt * row v + (-R vhat dt) * row chi
  {
    const int src[2] = { BV, BCHI };
    const int kind[2] = { 0, 0 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = A_R[i]; A[1][i] = A_RV[i]; }
    RowOp<NS, BPOS, 2>::apply(P, src, kind, A);
  }
  // E1: row v <- + (-what dt) * row v + (ghat dt) * row chi [+ (-vhat dt) * row bg + (-dt I) * row ba]
  if constexpr (NS == 21) {
    const int src[4] = { BV, BCHI, BBG, BBA };
    const int kind[4] = { 1, 1, 1, 2 };
    double A[4][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = a_mw[i % 3]; A[1][i] = a_g[i % 3]; A[2][i] = a_mv[i % 3]; A[3][i] = -dt; }
    RowOp<NS, BV, 4>::apply(P, src, kind, A);
  } else {
    const int src[2] = { BV, BCHI };
    const int kind[2] = { 1, 1 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = a_mw[i % 3]; A[1][i] = a_g[i % 3]; }
    RowOp<NS, BV, 2>::apply(P, src, kind, A);
  }
  // E2: row chi <- + (-what dt) * row chi [+ (-dt I) * row bg]
  if constexpr (NS == 21) {
    const int src[2] = { BCHI, BBG };
    const int kind[2] = { 1, 2 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = a_mw[i % 3]; A[1][i] = -dt; }
    RowOp<NS, BCHI, 2>::apply(P, src, kind, A);
  } else {
    const int src[1] = { BCHI };
    const int kind[1] = { 1 };
    double A[1][9];
#pragma unroll
    for (int i = 0; i < 9; i++) A[0][i] = a_mw[i % 3];
    RowOp<NS, BCHI, 1>::apply(P, src, kind, A);
  }

  // Qd = Wc Qc Wc^T dt in closed form (rbis.cpp:91-116):
  //   [v,v] += (qg vhat vhat^T + qa I) dt, [chi,v] += qg vhat^T dt, [chi,chi] += qg dt I, bias diag += q_b dt
  const double qgd = q_gyro * dt, qad = q_accel * dt;
  const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  // vhat vhat^T = |v|^2 I - v v^T
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++)
      P[pk(3 + r, 3 + c)] += qgd * ((r == c ? vv : 0.0) - v[r] * v[c]) + (r == c ? qad : 0.0);
  // [chi,v] block (row chi, col v) += qg * vhat^T dt ; vhat^T = -vhat
  {
    const double m[9] = { 0, v[2], -v[1], -v[2], 0, v[0], v[1], -v[0], 0 };
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (r != c) P[pk(6 + r, 3 + c)] += qgd * m[3 * r + c];
  }
#pragma unroll
  for (int r = 0; r < 3; r++) P[pk(6 + r, 6 + r)] += qgd;
  if constexpr (NS == 21) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      P[pk(15 + r, 15 + r)] += q_gyro_bias * dt;
      P[pk(18 + r, 18 + r)] += q_accel_bias * dt;
    }
  }
  // rbis.cpp:120-121: overwrite the accel and gyro diagonal blocks
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) {
      P[pk(12 + r, 12 + c)] = (r == c) ? q_accel : 0.0;
      P[pk(r, c)] = (r == c) ? q_gyro : 0.0;
    }
}

// RBISIMUProcessStep::updateFilter (rbis_update_interface.cpp:30-52)
template <int NS>
PB_HD void imu_process_step(double (&x)[NS], double (&q)[4], double (&P)[NS * (NS + 1) / 2],
                            const double (&gyro)[3], const double (&accel)[3], double dt, double q_gyro,
                            double q_accel, double q_gyro_bias, double q_accel_bias, const Consts &k)
{
  // covariance first: it needs the prior state, which ins_update_state overwrites
  ins_update_covariance<NS>(P, x, q, dt, q_gyro, q_accel, q_gyro_bias, q_accel_bias, k);
  ins_update_state<NS>(x, q, gyro, accel, dt, k);
}

// ------------------------------------------------------------------------------------------------------------
// measurement update with compile-time indices (state in registers)
// ------------------------------------------------------------------------------------------------------------

// unpivoted LDL^T of the m x m SPD innovation covariance S (lower, packed row-major); on return the strict
// lower part holds L and d[] the pivots.
template <int M>
PB_HD void ldlt(double (&S)[M * (M + 1) / 2], double (&d)[M])
{
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double dk = S[pk(kk, kk)];
#pragma unroll
    for (int j = 0; j < kk; j++) dk -= S[pk(kk, j)] * S[pk(kk, j)] * d[j];
    d[kk] = dk;
    const double idk = 1.0 / dk;
#pragma unroll
    for (int i = kk + 1; i < M; i++) {
      double s = S[pk(i, kk)];
#pragma unroll
      for (int j = 0; j < kk; j++) s -= S[pk(i, j)] * S[pk(kk, j)] * d[j];
      S[pk(i, kk)] = s * idk;
    }
  }
}

template <int M>
struct Idx {
  int v[M];
};

// matrixMeasurementGetKandCovDelta + rbisApplyDelta (rbis.cpp:124-143,219-227) for a selector C.
//   resid [M]  (already formed), S = R + P[idx,idx] passed in packed-lower form with R added by the caller.
// IDX is a constexpr index list so that P[pk(i, idx_k)] is a register, not a scratch access.
struct NoSink {
  PB_HD void operator()(int, double) const {}
};

// `sink(packed_index, value)` is called for every entry of the posterior covariance as soon as it is final, so a
// kernel can store it straight away and free its register.
// measurement_update_cols: the measured columns W = P[:, idx] arrive already gathered (compile-time indices: below;
// wave-uniform run-time indices: k_update_lane_rt); they are turned into P[:, idx] L^-T in place.
template <int NS, int M, typename SINK = NoSink>
PB_HD void measurement_update_cols(double (&x)[NS], double (&q)[4], double (&P)[NS * (NS + 1) / 2], double &ll,
                                   const double (&resid)[M], double (&S)[M * (M + 1) / 2], double (&W)[NS][M], const Consts &k,
                                   SINK sink = SINK(), bool active = true)
{
  // `active == false` = "the handler returned NULL for this filter": the lane runs the same instruction stream
  // with D^-1 forced to 0, so P, x and ll come out bit-identical to their inputs, and -- the point -- every lane
  // of the wave issues the SAME store instructions (whole 512-byte rows; a divergent skip path splits each row
  // store into two partial-line writes, measured as +28 % HBM write traffic).
  double d[M];
  ldlt<M>(S, d);
  // y = L^-1 r ; ll += -log det S - r^T S^-1 r  (rbis.cpp:142)
  double y[M], id[M];
  double quad = 0.0, det = 1.0;
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s = resid[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s -= S[pk(kk, j)] * y[j];
    y[kk] = active ? s : 0.0;
    id[kk] = active ? 1.0 / d[kk] : 0.0;
    det *= d[kk];
    quad += s * s * id[kk];
  }
  // the reference takes ONE log of S.determinant() (rbis.cpp:142): log of the product of the pivots, not a sum of logs
  // (the two differ when S is indefinite with an even number of negative pivots: finite there, NaN here)
  if (active) ll += -log(det) - quad;
  // W = P[:, idx] L^-T  (row i: W_ik = P(i,idx_k) - sum_{j<k} W_ij L_kj)
#pragma unroll
  for (int i = 0; i < NS; i++) {
#pragma unroll
    for (int kk = 0; kk < M; kk++) {
      double s = W[i][kk];
#pragma unroll
      for (int j = 0; j < kk; j++) s -= W[i][j] * S[pk(kk, j)];
      W[i][kk] = s;
    }
  }
  // dx = K r = W D^-1 y ;  P -= W D^-1 W^T
  double yd[M];
#pragma unroll
  for (int kk = 0; kk < M; kk++) yd[kk] = y[kk] * id[kk];
  double dx[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    double s = 0.0;
    double wd[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) {
      s = (kk == 0) ? W[i][0] * yd[0] : fma(W[i][kk], yd[kk], s);
      wd[kk] = W[i][kk] * id[kk];
    }
    dx[i] = s;
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double acc = P[pk(i, j)];
#pragma unroll
      for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], W[j][kk], acc);
      P[pk(i, j)] = acc;
      sink(pk(i, j), acc);
    }
  }
  if (active) add_delta<NS>(x, q, dx, k.chi_tol);
}

// IDX is a constexpr index list so that P[pk(i, idx_k)] is a register, not a scratch access.
template <int NS, int M, typename IDXT, typename SINK = NoSink>
PB_HD void measurement_update(double (&x)[NS], double (&q)[4], double (&P)[NS * (NS + 1) / 2], double &ll,
                              const double (&resid)[M], double (&S)[M * (M + 1) / 2], IDXT, const Consts &k,
                              SINK sink = SINK(), bool active = true)
{
  constexpr Idx<M> idx = IDXT::value;
  double W[NS][M];
#pragma unroll
  for (int i = 0; i < NS; i++)
#pragma unroll
    for (int kk = 0; kk < M; kk++) W[i][kk] = P[pk(i, idx.v[kk])];
  measurement_update_cols<NS, M>(x, q, P, ll, resid, S, W, k, sink, active);
}

}  // namespace pb
