// rbis_coop.hpp -- the predict(+update) step split over TWO cooperating waves per 64 filters.
//
// Why: a 21-state filter has 231 packed covariance entries = 462 registers per lane before any temporary; one lane
// per filter cannot hold it (k_step<21,*> spills to scratch).  The process model gives a natural cut
// (rbis.cpp:12-35): in the order p = {omega, accel} (passive: identity rows/cols of Ad, never a source),
// c = {v, chi, Delta} (dynamic core), b = {gyro bias, accel bias},
//
//        | I   0     0   |          P'_cc = G P_(cb)(cb) G^T,  P'_cb = F_cc P_cb + F_cb P_bb,  P'_bb = P_bb + Q_b dt
//   Ad = | 0  F_cc  F_cb | ,        P'_cp = F_cc P_cp + F_cb P_bp,   P'_bp = P_bp,   P'_pp = P_pp (+ overwrites)
//        | 0   0     I   |          (G = [F_cc F_cb])
//
// so the (c,b) x (c,b) sub-matrix evolves by itself and the p-panels only consume F.  Role C (wave 0 of the
// workgroup) owns the (c,b) sub-matrix, the state and the quaternion; role P (wave 1) owns P_cp, P_bp, P_pp and the
// omega/accel entries of x.  A measurement on core states (legodo idx 3..5) needs ONE hand-off: role C publishes the
// LDL^T factors and its rows of W = P[:,idx] L^-T through LDS, both waves meet at one barrier, then each downdates
// and stores its own entries.  No lane divergence, no cross-lane traffic, whole-row (512 B) global accesses.
//
// The role bodies are PB_HD templates over load/store/exchange functors so that tests/host_harness.cpp can run the
// two roles back to back on the CPU against the oracle.
#pragma once

#include "rbis_device.hpp"

namespace pb {

template <int NS>
struct Coop {
  static constexpr bool HB = (NS == 21);            // has bias states
  static constexpr int NSC = HB ? 15 : 9;           // role C sub-state: v chi Delta [bg ba]
  static constexpr int NPC = NSC * (NSC + 1) / 2;
  static constexpr int NB_ = HB ? 6 : 0;            // bias states
  // The log-likelihood lives in role C's rows for n = 21 and in role P's rows for n = 15 (Slots<NS>: both roles must
  // own an even number of components); for n = 15 role C hands its increment over with the factors.
  static constexpr bool LL_IN_P = !HB;
  static constexpr int XCH_LLI = 9 + 3 * NSC;       // hand-off slot of the log-likelihood increment (LL_IN_P only)
  static constexpr int NXCH = 9 + 3 * NSC + (LL_IN_P ? 1 : 0);  // LDS hand-off doubles per filter: L(3) id(3) yd(3) W_c,b [lli]
  // sub index -> full state index
  PB_HD static constexpr int fullc(int s) { return core_full(s); }
  // passive index 0..5 -> full state index (omega 0..2, accel 12..14)
  PB_HD static constexpr int fullp(int p) { return passive_full(p); }
};

// blocks of Ac*dt about the prior state (rbis.cpp:12-35), shared by both roles
struct ProcBlocks {
  double a_mw[3];  // -skew(w) dt      = hat(a_mw)
  double a_g[3];   // skew(R^T g) dt   = hat(a_g)
  double a_mv[3];  // -skew(v) dt      = hat(a_mv)
  double A_R[9];   // R dt
  double A_RV[9];  // -R skew(v) dt
  double v[3];
};

template <int NS>
PB_HD void make_proc_blocks(const double (&xp)[NS], const double (&qp)[4], double dt, const Consts &k, ProcBlocks &pb_)
{
  double R[9];
  quat_to_rot(qp, R);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pb_.v[i] = xp[3 + i];
    pb_.a_mw[i] = -(xp[i] * dt);
    pb_.a_g[i] = (-k.g * R[6 + i]) * dt;
    pb_.a_mv[i] = -(xp[3 + i] * dt);
  }
  const double vd[3] = { xp[3] * dt, xp[4] * dt, xp[5] * dt };
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pb_.A_R[3 * i + 0] = R[3 * i + 0] * dt;
    pb_.A_R[3 * i + 1] = R[3 * i + 1] * dt;
    pb_.A_R[3 * i + 2] = R[3 * i + 2] * dt;
    pb_.A_RV[3 * i + 0] = -(R[3 * i + 1] * vd[2] - R[3 * i + 2] * vd[1]);
    pb_.A_RV[3 * i + 1] = -(R[3 * i + 2] * vd[0] - R[3 * i + 0] * vd[2]);
    pb_.A_RV[3 * i + 2] = -(R[3 * i + 0] * vd[1] - R[3 * i + 1] * vd[0]);
  }
}

// hat(a) * X for a 3x3 block X (row-major), accumulated into Y
PB_HD void hat_mul_acc(const double (&a)[3], const double (&X)[9], double (&Y)[9])
{
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Y[0 + c] += a[1] * X[6 + c] - a[2] * X[3 + c];
    Y[3 + c] += a[2] * X[0 + c] - a[0] * X[6 + c];
    Y[6 + c] += a[0] * X[3 + c] - a[1] * X[0 + c];
  }
}
PB_HD void mat_mul_acc(const double (&A)[9], const double (&X)[9], double (&Y)[9])
{
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++)
      Y[3 * r + c] += A[3 * r] * X[c] + A[3 * r + 1] * X[3 + c] + A[3 * r + 2] * X[6 + c];
}

// One robot's IMU (+ leg-odometry) message for EVERY filter of the batch (a parameter sweep replaying one log,
// PB_HOST_BROADCAST): the 13 values travel as a kernel argument -- no device block, no fill launch, no input traffic.
struct StepBcast {
  double imu[7] = { 0, 0, 0, 0, 0, 0, 0 };
  double lo[6] = { 0, 0, 0, 1, 1, 1 };
  int on = 0;  // bit 0: the IMU block is broadcast, bit 1: the leg-odometry block is (the other one is read from memory)
};

struct StepInputs {
  double gyro[3], accel[3], dt;
  double z[3], rd[3];
  bool upd;
  double qg, qa, qbg, qba;
};

// A SECOND measurement applied in the same state round trip, behind the leg-odometry update: what the reference does as
// a third updateFilter call when a visual-odometry or scan-match message follows the IMU / leg-odometry pair
// (rbis_fovis_update.cpp:299-305, sensor_handlers.cpp:689-724).  Its indices are a compile-time list of CORE sub indices
// (v 0-2, chi 3-5, Delta 6-8, gyro bias 9-11, accel bias 12-14); chi entries take the orientation residual
// (indexedPlusOrientationMeasurement, rbis.cpp:189-217); R is diagonal.
template <bool ORIENT_, int... SUB>
struct Corr {
  static constexpr int M = (int) sizeof...(SUB);
  static constexpr bool ORIENT = ORIENT_;
  static constexpr int MM = M > 0 ? M : 1;
  static constexpr int sub[MM] = { SUB... };
};
using NoCorr = Corr<false>;
using CorrPosOrient = Corr<true, 6, 7, 8, 3, 4, 5>;  // idx 9,10,11,6,7,8: FovisHandler position_orient, ViconHandler
using CorrPosYaw = Corr<true, 6, 7, 8, 5>;           // idx 9,10,11,8:     ScanMatcherHandler position_yaw
using CorrVelYaw = Corr<true, 0, 1, 2, 5>;           // idx 3,4,5,8:       ScanMatcherHandler velocity_yaw
using CorrYaw = Corr<true, 5>;                       // idx 8:             ScanMatcherHandler yaw
using CorrVel = Corr<false, 0, 1, 2>;                // idx 3,4,5:         LegOdoCommon lin_rate, Fovis / ScanMatcher velocity
using CorrPos = Corr<false, 6, 7, 8>;                // idx 9,10,11:       GpsHandler, Fovis / ScanMatcher position
using CorrPosVel = Corr<false, 6, 7, 8, 0, 1, 2>;    // idx 9,10,11,3,4,5: LegOdoCommon pos_and_lin_rate
// the laser / RGB-D GPF's substates (rgbd_gpf_lib.cpp:71-96), plain indexed measurements (chi entries are vector states here)
using CorrGpfYawPos = Corr<false, 5, 6, 7, 8>;       // idx 8,9,10,11:     pos_yaw
using CorrGpfChiPos = Corr<false, 3, 4, 5, 6, 7, 8>; // idx 6,7,8,9,10,11: pos_chi
using CorrGpfZ = Corr<false, 8>;                     // idx 11:            z_only
struct CorrInputs {
  double z[6], rd[6], qm[4];
  double ro[15] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };  // strictly-lower part of a FULL R, packed by rows (i > j: i(i-1)/2 + j)
  bool upd;
};
// LDS hand-off of the second update, behind the first one's: L2 (strict lower, packed by rows), id2, yd2, W2 rows, [lli2]
template <int NS, class CORR>
struct CoopX {
  using C = Coop<NS>;
  static constexpr int M = CORR::M;
  static constexpr int X2_L = C::NXCH, X2_ID = X2_L + M * (M - 1) / 2, X2_YD = X2_ID + M, X2_W = X2_YD + M,
                       X2_LLI = X2_W + C::NSC * M;
  static constexpr int NXCH = (M == 0) ? C::NXCH : X2_LLI + (C::LL_IN_P ? 1 : 0);
  // k_step_leg: the measurement from the odometry wave -- z[3], R, valid, and a six-row mode's second block z[3], R, on -- and the
  // two foot poses (2 x 7) from the other wave.  All of them are consumed before role C writes its SECOND hand-off (behind the
  // first one's barrier), so they share its slots: four workgroups of the pair kernel must fit a CU's 160 KB.
  static constexpr int XCH_LEG = C::NXCH, XCH_FOOT = XCH_LEG + 10, NXCH_LEG = (NXCH > XCH_FOOT + 14) ? NXCH : XCH_FOOT + 14;
  // SIX == 1 (see coop_role_core): role P's omega stage -- 1/d, y/d, its log-likelihood term and the propagated P(c, omega) rows --
  // lies in the FIRST hand-off's slots: role C reads it behind barrier L and only then writes its own factors there.
  static constexpr int X6_ID = 0, X6_YD = 3, X6_LLI = 6, X6_A = 9;
};

// The second 3-row block of LegOdoCommon's six-row measurements (rbis_legodo_common.cpp:46-79), when it is NOT a set of role C's
// states: lin_rot_rate's angular-velocity rows (idx 0..2).  R is diagonal, so the six-row update equals the two blocks applied one
// after the other with ONE addState of the summed correction (the residual of the later block taken at x + dx of the earlier one;
// log-likelihood = sum of the two conditional terms) -- to rounding, not bit for bit: the oracle factors the 6 x 6 S at once.
// "this value exists HERE": an empty asm that reads and writes v.  Arithmetic has no position of its own in the compiler's schedule
// (it sinks to its first use, across barriers too); a side-effecting user in front of a barrier keeps it there.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void pb_pin(double &v) { asm volatile("" : "+v"(v)); }
#else
inline void pb_pin(double &) {}
#endif
struct SixIn {
  double z[3] = { 0.0, 0.0, 0.0 }, r = 1.0;
  bool on = false;
};

// ------------------------------------------------------------------------------------------------------------
// role C: (c,b) sub-matrix, state, quaternion, log-likelihood
//   LD(comp) -> double, ST(comp, v), XW(slot, v) writes the hand-off, SYNC() is the workgroup barrier
// ------------------------------------------------------------------------------------------------------------
// What one launch does is three compile-time switches: PREDICT (the IMU process step), UPDATE (the leg-odometry velocity
// update behind it), CORR (one more measurement with compile-time core indices).  PREDICT + UPDATE is the BASELINE hot step;
// CORR alone is a stand-alone indexed / indexed+orientation update on the same two-role mapping.
// LEG: the leg-odometry measurement (z, R, valid) is not an input but made by the OTHER wave of the tile while this one
// propagates (k_step_leg, rbis_legstep.hpp): it arrives in the hand-off slots XCH_LEG.. behind one more barrier.
// SIX: LegOdoCommon's six-row measurements in the SAME state round trip, as two 3-row blocks with one summed correction (SixIn):
//   1  lin_rot_rate: the angular-velocity block FIRST, by role P -- behind a predict P(omega, omega) = q_gyro I (rbis.cpp:121), so
//      S = (q_gyro + r) I needs no factorisation; role P publishes 1/d, y/d and its P(c, omega) rows BEFORE barrier L, role C
//      downdates its sub-matrix with them and runs the velocity block on the result: no extra barrier.
//   2  pos_and_lin_rate: the velocity block, then CORR = CorrPos on its posterior with the correction summed (the CORR stage
//      otherwise is a second, separate update: two addState calls, like the reference's two updateFilter calls).
template <int NS, bool UPDATE, class CORR = NoCorr, bool PREDICT = true, bool LEG = false, int SIX = 0, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void coop_role_core(LD ld, ST st, XW xw, XR xr, SYNC sync, const StepInputs &in, const Consts &k,
                          const CorrInputs &cin_ = CorrInputs())
{
  static_assert(!LEG || (UPDATE && PREDICT), "the odometry wave feeds a predict + update step");
  static_assert(SIX == 0 || (UPDATE && PREDICT), "the six-row leg-odometry modes ride on a predict + velocity update");
  static_assert(SIX != 2 || (CORR::M == 3 && !CORR::ORIENT), "SIX == 2: the second block is a 3-row vector block of role C's states");
  static_assert(SIX != 1 || CORR::M == 0, "SIX == 1 has no CORR stage");
  using L = Lay<NS>;
  using C = Coop<NS>;
  constexpr int NSC = C::NSC;
  static_assert(PREDICT || !UPDATE, "the leg-odometry update rides behind a predict");
  static_assert(PREDICT || CORR::M > 0, "nothing to do");
  double x[NS], q[4], ll = 0.0;
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
  if constexpr (!C::LL_IN_P) ll = ld(L::OFF_LL);
  double Pc[C::NPC];
#pragma unroll
  for (int i = 0; i < NSC; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Pc[pk(i, j)] = ld(L::OFF_P + pk(C::fullc(i), C::fullc(j)));

  if constexpr (PREDICT) {
  // ---- covariance propagate on the sub-matrix (blocks: v=0 chi=1 Delta=2 bg=3 ba=4) ----
  ProcBlocks f;
  make_proc_blocks<NS>(x, q, in.dt, k, f);
  {
    const int src[2] = { 0, 1 };
    const int kind[2] = { 0, 0 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = f.A_R[i]; A[1][i] = f.A_RV[i]; }
    RowOp<NSC, 2, 2>::apply(Pc, src, kind, A);  // E3: row Delta
  }
  if constexpr (C::HB) {
    const int src[4] = { 0, 1, 3, 4 };
    const int kind[4] = { 1, 1, 1, 2 };
    double A[4][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = f.a_mw[i % 3]; A[1][i] = f.a_g[i % 3]; A[2][i] = f.a_mv[i % 3]; A[3][i] = -in.dt; }
    RowOp<NSC, 0, 4>::apply(Pc, src, kind, A);  // E1: row v
    const int src2[2] = { 1, 3 };
    const int kind2[2] = { 1, 2 };
    double A2[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A2[0][i] = f.a_mw[i % 3]; A2[1][i] = -in.dt; }
    RowOp<NSC, 1, 2>::apply(Pc, src2, kind2, A2);  // E2: row chi
  } else {
    const int src[2] = { 0, 1 };
    const int kind[2] = { 1, 1 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = f.a_mw[i % 3]; A[1][i] = f.a_g[i % 3]; }
    RowOp<NSC, 0, 2>::apply(Pc, src, kind, A);
    const int src2[1] = { 1 };
    const int kind2[1] = { 1 };
    double A2[1][9];
#pragma unroll
    for (int i = 0; i < 9; i++) A2[0][i] = f.a_mw[i % 3];
    RowOp<NSC, 1, 1>::apply(Pc, src2, kind2, A2);
  }
  // Qd (closed form of rbis.cpp:91-116) on sub indices: v 0..2, chi 3..5, bg 9..11, ba 12..14
  {
    const double qgd = in.qg * in.dt, qad = in.qa * in.dt;
    const double vv = f.v[0] * f.v[0] + f.v[1] * f.v[1] + f.v[2] * f.v[2];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) Pc[pk(r, c)] += qgd * ((r == c ? vv : 0.0) - f.v[r] * f.v[c]) + (r == c ? qad : 0.0);
    const double m[9] = { 0, f.v[2], -f.v[1], -f.v[2], 0, f.v[0], f.v[1], -f.v[0], 0 };
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (r != c) Pc[pk(3 + r, c)] += qgd * m[3 * r + c];
#pragma unroll
    for (int r = 0; r < 3; r++) Pc[pk(3 + r, 3 + r)] += qgd;
    if constexpr (C::HB) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        Pc[pk(9 + r, 9 + r)] += in.qbg * in.dt;
        Pc[pk(12 + r, 12 + r)] += in.qba * in.dt;
      }
    }
  }
  // ---- state propagate (rbis.cpp:37-75); omega/accel entries are role P's to store ----
  ins_update_state<NS>(x, q, in.gyro, in.accel, in.dt, k);
  }  // PREDICT

  double dfull[NS];   // the velocity block's correction (SIX == 2: applied together with the second block's)
#pragma unroll
  for (int i = 0; i < NS; i++) dfull[i] = 0.0;
  bool upd1 = false;
  double leg2[5] = { 0.0, 0.0, 0.0, 1.0, 0.0 };
  if constexpr (UPDATE) {
    // S = R + P[v,v]; unpivoted LDL^T; y = L^-1 r  (rbis.cpp:124-143)
    double resid[3], S[6], d[3], y[3], id[3], yd[3];
    double mz[3], mr[3];
    double dx1[SIX == 1 ? NSC : 1], idw = 0.0;   // SIX == 1: the omega block's share of the correction; 1/d (0: no such block)
    dx1[0] = 0.0;
    bool mupd = in.upd;
    if constexpr (LEG || SIX == 1) {
      // The predict ENDS here, in front of barrier L -- while the other wave is still busy with the odometry.  Without the pins the
      // compiler sinks the covariance propagation behind the barrier (arithmetic has no position of its own): this wave then waits
      // for the odometry first and propagates afterwards, and with the omega stage behind the barrier as well the two overlap into
      // 280 live registers (456-528 bytes of scratch, 33-35 us at 64k filters instead of 26).
#pragma unroll
      for (int i = 0; i < C::NPC; i++) pb_pin(Pc[i]);
#pragma unroll
      for (int i = 0; i < 9; i++) pb_pin(x[C::fullc(i)]);
#pragma unroll
      for (int i = 0; i < 4; i++) pb_pin(q[i]);
      sync();  // barrier L
    }
    if constexpr (SIX == 1) {
      // the angular-velocity block, applied by role P to its panels; here: P_cc -= A A^T / d, dx_c = A (y / d) with A = P'(c, omega)
      using CX = CoopX<NS, CORR>;
      idw = xr(CX::X6_ID);
      const double ydw[3] = { xr(CX::X6_YD), xr(CX::X6_YD + 1), xr(CX::X6_YD + 2) };
      if constexpr (!C::LL_IN_P) ll += xr(CX::X6_LLI);
      // one column of A at a time: NSC values live next to the sub-matrix instead of 3 NSC
#pragma unroll
      for (int i = 0; i < NSC; i++) dx1[i] = 0.0;
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        double a[NSC];
#pragma unroll
        for (int i = 0; i < NSC; i++) a[i] = xr(CX::X6_A + 3 * i + kk);
#pragma unroll
        for (int i = 0; i < NSC; i++) {
          const double ad = a[i] * idw;
          dx1[i] = fma(a[i], ydw[kk], dx1[i]);
#pragma unroll
          for (int j = 0; j <= i; j++) Pc[pk(i, j)] = fma(-ad, a[j], Pc[pk(i, j)]);
        }
        reload_fence();
      }
    }
    if constexpr (LEG) {
      const double r = xr(CoopX<NS, CORR>::XCH_LEG + 3);
      mupd = in.upd && xr(CoopX<NS, CORR>::XCH_LEG + 4) != 0.0;
      if constexpr (SIX == 2) {
#pragma unroll
        for (int i = 0; i < 5; i++) leg2[i] = xr(CoopX<NS, CORR>::XCH_LEG + 5 + i);
      }
#pragma unroll
      for (int i = 0; i < 3; i++) { mz[i] = xr(CoopX<NS, CORR>::XCH_LEG + i); mr[i] = r; }
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) { mz[i] = in.z[i]; mr[i] = in.rd[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) resid[i] = mupd ? mz[i] - (SIX == 1 ? x[3 + i] + dx1[i] : x[3 + i]) : 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) S[pk(i, j)] = Pc[pk(i, j)] + (i == j ? (mupd ? mr[i] : 1.0) : 0.0);
    ldlt<3>(S, d);
    double quad = 0.0, det = 1.0;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      double s = resid[kk];
#pragma unroll
      for (int j = 0; j < kk; j++) s -= S[pk(kk, j)] * y[j];
      y[kk] = mupd ? s : 0.0;
      id[kk] = mupd ? 1.0 / d[kk] : 0.0;
      yd[kk] = y[kk] * id[kk];
      det *= d[kk];
      quad += s * s * id[kk];
    }
    const double lli = -log(det) - quad;  // -log(S.determinant()) - r^T S^-1 r (rbis.cpp:142): ONE log of the product
    if constexpr (C::LL_IN_P) xw(C::XCH_LLI, lli);
    else if (mupd) ll += lli;
    double W[NSC][3];
#pragma unroll
    for (int i = 0; i < NSC; i++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        double s = Pc[pk(i, kk)];
#pragma unroll
        for (int j = 0; j < kk; j++) s -= W[i][j] * S[pk(kk, j)];
        W[i][kk] = s;
      }
    // hand-off to role P
    xw(0, S[pk(1, 0)]); xw(1, S[pk(2, 0)]); xw(2, S[pk(2, 1)]);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { xw(3 + kk, id[kk]); xw(6 + kk, yd[kk]); }
#pragma unroll
    for (int i = 0; i < NSC; i++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) xw(9 + 3 * i + kk, W[i][kk]);
    sync();
    // downdate + store own entries, dx for own states
#pragma unroll
    for (int i = 0; i < NSC; i++) {
      double wd[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) wd[kk] = W[i][kk] * id[kk];
      dfull[C::fullc(i)] = fma(W[i][2], yd[2], fma(W[i][1], yd[1], W[i][0] * yd[0])) + (SIX == 1 ? dx1[i] : 0.0);
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double acc = Pc[pk(i, j)];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], W[j][kk], acc);
        if constexpr (CORR::M == 0) st(L::OFF_P + pk(C::fullc(i), C::fullc(j)), acc);
        else Pc[pk(i, j)] = acc;  // row i of W is not needed for any later row j' > i's column i: W[i] stays as is
      }
    }
    upd1 = mupd || idw != 0.0;
    if constexpr (SIX != 2) {
      if (upd1) add_delta<NS>(x, q, dfull, k.chi_tol);
    }
  }
  if constexpr (CORR::M > 0) {
    // ---- the second update on the posterior of the first (indexedPlusOrientationMeasurement, rbis.cpp:189-217) ----
    constexpr int M = CORR::M;
    using CX = CoopX<NS, CORR>;
    CorrInputs cin = cin_;
    if constexpr (LEG && SIX == 2) {  // the position block of the odometry wave's measurement (read behind barrier L, before
      // this role's first hand-off: the slots are the second hand-off's)
#pragma unroll
      for (int i = 0; i < 3; i++) { cin.z[i] = leg2[i]; cin.rd[i] = leg2[3]; }
      cin.upd = in.upd && leg2[4] != 0.0;
    }
    double r2[M], S2[M * (M + 1) / 2], d2[M], y2[M], id2[M], yd2[M];
    double dq3[3] = { 0.0, 0.0, 0.0 };
    if constexpr (CORR::ORIENT) subtract_quats(cin.qm, q, dq3);
#pragma unroll
    for (int kk = 0; kk < M; kk++) {
      const int ii = C::fullc(CORR::sub[kk]);
      const double r = (CORR::ORIENT && ii >= 6 && ii <= 8) ? dq3[ii - 6] : cin.z[kk] - (SIX == 2 ? x[ii] + dfull[ii] : x[ii]);
      r2[kk] = cin.upd ? r : 0.0;
    }
#pragma unroll
    for (int i = 0; i < M; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        S2[pk(i, j)] = Pc[pk(CORR::sub[i], CORR::sub[j])] + (i == j ? (cin.upd ? cin.rd[i] : 1.0) : (cin.upd ? cin.ro[i * (i - 1) / 2 + j] : 0.0));
    ldlt<M>(S2, d2);
    double quad2 = 0.0, det2 = 1.0;
#pragma unroll
    for (int kk = 0; kk < M; kk++) {
      double s2 = r2[kk];
#pragma unroll
      for (int j = 0; j < kk; j++) s2 -= S2[pk(kk, j)] * y2[j];
      y2[kk] = cin.upd ? s2 : 0.0;
      id2[kk] = cin.upd ? 1.0 / d2[kk] : 0.0;
      yd2[kk] = y2[kk] * id2[kk];
      det2 *= d2[kk];
      quad2 += s2 * s2 * id2[kk];
    }
    const double lli2 = -log(det2) - quad2;
    if constexpr (C::LL_IN_P) xw(CX::X2_LLI, lli2);
    else if (cin.upd) ll += lli2;
    // W2 = P[:, idx2] L2^-T row by row, published as it is formed.  n = 21: the 15 x 15 sub-matrix (240 registers) and W2
    // (15 x M) do not fit together, so W2 is NOT kept: the downdate below reads its rows back from the hand-off area
    // (this wave's own LDS writes are in order); n = 15 keeps it in registers.
    constexpr bool W2_LDS = C::HB;
    double W2[W2_LDS ? 1 : NSC][M];
    double dfull2[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) dfull2[i] = 0.0;
#pragma unroll
    for (int i = 1; i < M; i++)
#pragma unroll
      for (int j = 0; j < i; j++) xw(CX::X2_L + i * (i - 1) / 2 + j, S2[pk(i, j)]);
#pragma unroll
    for (int kk = 0; kk < M; kk++) { xw(CX::X2_ID + kk, id2[kk]); xw(CX::X2_YD + kk, yd2[kk]); }
#pragma unroll
    for (int i = 0; i < NSC; i++) {
      double w2[M], dxs = 0.0;
#pragma unroll
      for (int kk = 0; kk < M; kk++) {
        double s2 = Pc[pk(i, CORR::sub[kk])];
#pragma unroll
        for (int j = 0; j < kk; j++) s2 -= w2[j] * S2[pk(kk, j)];
        w2[kk] = s2;
        xw(CX::X2_W + M * i + kk, s2);
        dxs = (kk == 0) ? s2 * yd2[0] : fma(s2, yd2[kk], dxs);
        if constexpr (!W2_LDS) W2[i][kk] = s2;
      }
      dfull2[C::fullc(i)] = dxs;
    }
    sync();
#pragma unroll
    for (int i = 0; i < NSC; i++) {
      // (without the clobber the compiler shares every LDS read of W2 between the rows: all of W2 back in registers)
      if constexpr (W2_LDS) reload_fence();
      double wd2[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) wd2[kk] = (W2_LDS ? xr(CX::X2_W + M * i + kk) : W2[i][kk]) * id2[kk];
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double acc = Pc[pk(i, j)];
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd2[kk], W2_LDS ? xr(CX::X2_W + M * j + kk) : W2[j][kk], acc);
        st(L::OFF_P + pk(C::fullc(i), C::fullc(j)), acc);
      }
    }
    if constexpr (SIX == 2) {  // ONE addState of the summed correction (both terms are zero where their block is masked)
#pragma unroll
      for (int i = 0; i < NS; i++) dfull2[i] += dfull[i];
      if (upd1 || cin.upd) add_delta<NS>(x, q, dfull2, k.chi_tol);
    } else {
      if (cin.upd) add_delta<NS>(x, q, dfull2, k.chi_tol);
    }
  }
  if constexpr (!UPDATE && CORR::M == 0) {
#pragma unroll
    for (int i = 0; i < NSC; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) st(L::OFF_P + pk(C::fullc(i), C::fullc(j)), Pc[pk(i, j)]);
    // role P reads the PRIOR x and quat: it must have consumed them before they are overwritten below (in the
    // UPDATE flavour the hand-off barrier above already orders this)
    sync();
  }
#pragma unroll
  for (int i = 0; i < 9; i++) st(L::OFF_VEC + C::fullc(i), x[C::fullc(i)]);
#pragma unroll
  for (int i = 0; i < 4; i++) st(L::OFF_QUAT + i, q[i]);
  if constexpr (!C::LL_IN_P) st(L::OFF_LL, ll);
#pragma unroll
  for (int i = 9; i < NSC; i++) st(L::OFF_VEC + C::fullc(i), x[C::fullc(i)]);
}

// ------------------------------------------------------------------------------------------------------------
// role P: passive panels P_cp, P_bp, P_pp and the omega / accel entries of x.   XR(slot) reads the hand-off.
// ------------------------------------------------------------------------------------------------------------
// SIX == 1: this role applies the angular-velocity block of lin_rot_rate (see coop_role_core) -- XW publishes its factors.
template <int NS, bool UPDATE, class CORR = NoCorr, bool PREDICT = true, int SIX = 0, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void coop_role_passive_x(LD ld, ST st, XW xw, XR xr, SYNC sync, const StepInputs &in, const Consts &k,
                               const CorrInputs &cin = CorrInputs(), const SixIn &six = SixIn())
{
  static_assert(SIX != 1 || (UPDATE && PREDICT && CORR::M == 0), "SIX == 1 rides on a predict + velocity update");
  using L = Lay<NS>;
  using C = Coop<NS>;
  double x[NS], q[4] = { 1.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = 0.0;
  if constexpr (PREDICT) {  // the process blocks need the whole prior state
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = ld(L::OFF_VEC + i);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
  } else {                  // a stand-alone update only moves this role's own omega / accel entries
#pragma unroll
    for (int i = 0; i < 6; i++) x[C::fullp(i)] = ld(L::OFF_VEC + C::fullp(i));
  }
  // X[sb][J] = 3x3 block P(state block sb, passive block J), row-major; sb: v chi Delta [bg ba]; J: omega, accel
  constexpr int NSB = C::HB ? 5 : 3;
  double X[NSB][2][9];
#pragma unroll
  for (int sb = 0; sb < NSB; sb++)
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) X[sb][J][3 * r + c] = ld(L::OFF_P + pk(C::fullc(3 * sb + r), C::fullp(3 * J + c)));
  double Ppp[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Ppp[pk(i, j)] = ld(L::OFF_P + pk(C::fullp(i), C::fullp(j)));
  double ll = 0.0;
  if constexpr (C::LL_IN_P) ll = ld(L::OFF_LL);

  double xp[6];
  if constexpr (!PREDICT) {
#pragma unroll
    for (int i = 0; i < 6; i++) xp[i] = x[C::fullp(i)];
  }
  if constexpr (PREDICT) {
  ProcBlocks f;
  make_proc_blocks<NS>(x, q, in.dt, k, f);
  // panel propagate: all right-hand sides use the ORIGINAL blocks (Ad = E2 E1 E3)
#pragma unroll
  for (int J = 0; J < 2; J++) {
    double nD[9], nV[9], nC[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { nD[i] = X[2][J][i]; nV[i] = X[0][J][i]; nC[i] = X[1][J][i]; }
    mat_mul_acc(f.A_R, X[0][J], nD);
    mat_mul_acc(f.A_RV, X[1][J], nD);
    hat_mul_acc(f.a_mw, X[0][J], nV);
    hat_mul_acc(f.a_g, X[1][J], nV);
    hat_mul_acc(f.a_mw, X[1][J], nC);
    if constexpr (C::HB) {
      hat_mul_acc(f.a_mv, X[3][J], nV);
#pragma unroll
      for (int i = 0; i < 9; i++) {
        nV[i] -= in.dt * X[4][J][i];
        nC[i] -= in.dt * X[3][J][i];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) { X[2][J][i] = nD[i]; X[0][J][i] = nV[i]; X[1][J][i] = nC[i]; }
  }
  // rbis.cpp:120-121: overwrite the omega and accel diagonal blocks (the [accel,omega] block is untouched)
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) {
      Ppp[pk(r, c)] = (r == c) ? in.qg : 0.0;
      Ppp[pk(3 + r, 3 + c)] = (r == c) ? in.qa : 0.0;
    }
  // rbis.cpp:50-51
#pragma unroll
  for (int i = 0; i < 3; i++) {
    xp[i] = in.gyro[i] - (C::HB ? x[15 + i] : 0.0);
    xp[3 + i] = in.accel[i] - (C::HB ? x[18 + i] : 0.0);
  }
  }  // PREDICT

  if constexpr (SIX == 1) {
    // ---- the angular-velocity block: S = (q_gyro + r) I, W = P'[:, omega], no factorisation ----
    using CX = CoopX<NS, CORR>;
    const double idw = six.on ? 1.0 / (in.qg + six.r) : 0.0;
    double ydw[3], quad = 0.0;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const double res = six.on ? six.z[kk] - xp[kk] : 0.0;
      ydw[kk] = res * idw;
      quad = fma(res, ydw[kk], quad);
    }
    const double dw = in.qg + six.r;
    const double lliw = six.on ? -log(dw * dw * dw) - quad : 0.0;   // -log det S - r^T S^-1 r of this block
    xw(CX::X6_ID, idw);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) xw(CX::X6_YD + kk, ydw[kk]);
    if constexpr (!C::LL_IN_P) xw(CX::X6_LLI, lliw);
    else ll += lliw;
#pragma unroll
    for (int sb = 0; sb < NSB; sb++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) xw(CX::X6_A + 3 * (3 * sb + r) + kk, X[sb][0][3 * r + kk]);
    sync();  // barrier L
    // own entries: the accel panel with P'(a, omega) (untouched by the predict), the omega panel and P_pp in closed form
    const double keep = fma(-in.qg, idw, 1.0);   // 1 - q_gyro / d
#pragma unroll
    for (int sb = 0; sb < NSB; sb++)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const double ad[3] = { X[sb][0][3 * r] * idw, X[sb][0][3 * r + 1] * idw, X[sb][0][3 * r + 2] * idw };
#pragma unroll
        for (int c = 0; c < 3; c++) {
          X[sb][1][3 * r + c] = fma(-ad[2], Ppp[pk(3 + c, 2)], fma(-ad[1], Ppp[pk(3 + c, 1)], fma(-ad[0], Ppp[pk(3 + c, 0)], X[sb][1][3 * r + c])));
          X[sb][0][3 * r + c] *= keep;
        }
      }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double ad[3] = { Ppp[pk(3 + r, 0)] * idw, Ppp[pk(3 + r, 1)] * idw, Ppp[pk(3 + r, 2)] * idw };
      xp[3 + r] += fma(Ppp[pk(3 + r, 2)], ydw[2], fma(Ppp[pk(3 + r, 1)], ydw[1], Ppp[pk(3 + r, 0)] * ydw[0]));
#pragma unroll
      for (int c = 0; c <= r; c++)
        Ppp[pk(3 + r, 3 + c)] = fma(-ad[2], Ppp[pk(3 + c, 2)], fma(-ad[1], Ppp[pk(3 + c, 1)], fma(-ad[0], Ppp[pk(3 + c, 0)], Ppp[pk(3 + r, 3 + c)])));
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) Ppp[pk(3 + r, c)] *= keep;
      Ppp[pk(r, r)] = in.qg * keep;
      xp[r] = fma(in.qg, ydw[r], xp[r]);
    }
  }
  if constexpr (!UPDATE && CORR::M == 0) sync();  // pairs with role C's barrier before it overwrites x / quat
  if constexpr (UPDATE) {
    sync();
    const double L10 = xr(0), L20 = xr(1), L21 = xr(2);
    double id[3], yd[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { id[kk] = xr(3 + kk); yd[kk] = xr(6 + kk); }
    // W_p = P'[p, v] L^-T : P'(p_i, v_k) = X[v][J][k][c]
    double Wp[6][3];
#pragma unroll
    for (int pi = 0; pi < 6; pi++) {
      const double c0 = X[0][pi / 3][0 * 3 + pi % 3], c1 = X[0][pi / 3][1 * 3 + pi % 3], c2 = X[0][pi / 3][2 * 3 + pi % 3];
      Wp[pi][0] = c0;
      Wp[pi][1] = c1 - Wp[pi][0] * L10;
      Wp[pi][2] = c2 - Wp[pi][0] * L20 - Wp[pi][1] * L21;
      xp[pi] += fma(Wp[pi][2], yd[2], fma(Wp[pi][1], yd[1], Wp[pi][0] * yd[0]));
    }
    if constexpr (C::LL_IN_P) {
      const double lli = xr(C::XCH_LLI);
      if (in.upd) ll += lli;
    }
    // downdate (c,p) and (b,p) panels with role C's rows of W; every entry is stored the moment it is final, in the
    // row order of the storage layout (Slots<NS>: omega panel, accel panel, P_pp)
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int sb = 0; sb < NSB; sb++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double wd[3];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) wd[kk] = xr(9 + 3 * (3 * sb + r) + kk) * id[kk];
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double acc = X[sb][J][3 * r + c];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[3 * J + c][kk], acc);
            if constexpr (CORR::M == 0) st(L::OFF_P + pk(C::fullc(3 * sb + r), C::fullp(3 * J + c)), acc);
            else X[sb][J][3 * r + c] = acc;
          }
        }
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double wd[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) wd[kk] = Wp[i][kk] * id[kk];
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double acc = Ppp[pk(i, j)];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[j][kk], acc);
        if constexpr (CORR::M == 0) st(L::OFF_P + pk(C::fullp(i), C::fullp(j)), acc);
        else Ppp[pk(i, j)] = acc;
      }
    }
  }
  if constexpr (CORR::M > 0) {
    // ---- the second update: role C's factors and rows of W2 arrive through the second hand-off ----
    constexpr int M = CORR::M;
    using CX = CoopX<NS, CORR>;
    sync();
    double id2[M], yd2[M], L2[M * (M - 1) / 2 + 1];
#pragma unroll
    for (int i = 0; i < M * (M - 1) / 2; i++) L2[i] = xr(CX::X2_L + i);
#pragma unroll
    for (int kk = 0; kk < M; kk++) { id2[kk] = xr(CX::X2_ID + kk); yd2[kk] = xr(CX::X2_YD + kk); }
    if constexpr (C::LL_IN_P) {
      const double lli2 = xr(CX::X2_LLI);
      if (cin.upd) ll += lli2;
    }
    // W2_p = P''[p, idx2] L2^-T with P''(p_i, core s) = X[s / 3][i / 3][(s % 3) * 3 + i % 3]
    double W2p[6][M];
#pragma unroll
    for (int pi = 0; pi < 6; pi++) {
      double dxs = 0.0;
#pragma unroll
      for (int kk = 0; kk < M; kk++) {
        double s2 = X[CORR::sub[kk] / 3][pi / 3][(CORR::sub[kk] % 3) * 3 + pi % 3];
#pragma unroll
        for (int j = 0; j < kk; j++) s2 -= W2p[pi][j] * L2[kk * (kk - 1) / 2 + j];
        W2p[pi][kk] = s2;
        dxs = (kk == 0) ? s2 * yd2[0] : fma(s2, yd2[kk], dxs);
      }
      xp[pi] += dxs;
    }
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int sb = 0; sb < NSB; sb++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double wd2[M];
#pragma unroll
          for (int kk = 0; kk < M; kk++) wd2[kk] = xr(CX::X2_W + M * (3 * sb + r) + kk) * id2[kk];
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double acc = X[sb][J][3 * r + c];
#pragma unroll
            for (int kk = 0; kk < M; kk++) acc = fma(-wd2[kk], W2p[3 * J + c][kk], acc);
            st(L::OFF_P + pk(C::fullc(3 * sb + r), C::fullp(3 * J + c)), acc);
          }
        }
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double wd2[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) wd2[kk] = W2p[i][kk] * id2[kk];
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double acc = Ppp[pk(i, j)];
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd2[kk], W2p[j][kk], acc);
        st(L::OFF_P + pk(C::fullp(i), C::fullp(j)), acc);
      }
    }
  }
  if constexpr (!UPDATE && CORR::M == 0) {
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int sb = 0; sb < NSB; sb++)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) st(L::OFF_P + pk(C::fullc(3 * sb + r), C::fullp(3 * J + c)), X[sb][J][3 * r + c]);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) st(L::OFF_P + pk(C::fullp(i), C::fullp(j)), Ppp[pk(i, j)]);
  }
  if constexpr (C::LL_IN_P) st(L::OFF_LL, ll);
#pragma unroll
  for (int i = 0; i < 6; i++) st(L::OFF_VEC + C::fullp(i), xp[i]);
}
template <int NS, bool UPDATE, class CORR = NoCorr, bool PREDICT = true, class LD, class ST, class XR, class SYNC>
PB_HD void coop_role_passive(LD ld, ST st, XR xr, SYNC sync, const StepInputs &in, const Consts &k,
                             const CorrInputs &cin = CorrInputs())
{
  coop_role_passive_x<NS, UPDATE, CORR, PREDICT, 0>(ld, st, [](int, double) {}, xr, sync, in, k, cin);
}

}  // namespace pb
