// rbis_quad.hpp -- the 21-state predict(+update) step split over FOUR cooperating waves per 64 filters.
//
// Why: the two-role mapping (rbis_coop.hpp) leaves role C with the 15 x 15 (c,b) sub-matrix = 240 registers of state
// per lane: one wave per SIMD, so loads, arithmetic and stores of a tile never overlap with another tile's.  The
// process model (rbis.cpp:12-35) splits once more.  With c = {v, chi, Delta}, b = {gyro bias, accel bias},
// p = {omega, accel} and Ad = I + N (N's only non-zero block rows: v <- v chi bg ba, chi <- chi bg, Delta <- v chi):
//
//   P'_cc = F_cc P_cc F_cc^T + H,     H = P'_cb F_cb^T + F_cb T1^T,   T1 = F_cc P_cb,   P'_cb = T1 + F_cb P_bb
//   P'_bb = P_bb + Q_b dt,            P'_(cb)p = [F_cc F_cb; 0 I] P_(cb)p  column by column of p
//
//   wave 0 (role CC):  P_cc (45), loglik                        -- the LDL^T of S and the rows of W for the c-states
//   wave 1 (role CB):  P_cb (54), P_bb (21), x[bg ba], x[omega]  -- publishes H (39 non-zero entries) and P'_bv
//   wave 2 (role PW):  P_(cb),omega (45), P_omega,omega, x[v chi Delta], quat -- the state / quaternion propagate and update
//   wave 3 (role PA):  P_(cb),accel (45), P_accel,accel, P_accel,omega, x[accel]
//
// 46 / 84 / 64 / 63 components, balanced by ARITHMETIC rather than by bytes: the quaternion work (two exponential maps
// per step) rides with the lightest panel instead of with the wave every other wave waits for.  Every role fits 256
// registers -> TWO waves per SIMD, two workgroups per CU, and one tile's loads overlap another tile's arithmetic and stores.  Two barriers: A (H is published; nobody has overwritten
// the prior x / quat that all four roles linearise about) and B (wave 0 has published the LDL^T factors and its rows of
// W = P[:,idx] L^-T).  W_b and W_omega are recomputed by their consumers from the raw P'_bv / P'_v,omega columns that
// their owners publish before B, so that no third barrier is needed.
//
// The role bodies are PB_HD templates over load/store/exchange functors (tests/host_harness.cpp runs the four roles as
// four threads with a real barrier on the CPU against the oracle).
#pragma once

#include "rbis_coop.hpp"

namespace pb {

struct Quad {
  // LDS hand-off, doubles per filter.  [0, 45): H packed by (i, j <= i) over the 9 c-states before barrier A; wave 0 is
  // its only reader and re-uses the area for L(3) id(3) yd(3) W_c(27) dx_c(9) before barrier B.
  static constexpr int X_H = 0, X_L = 0, X_ID = 3, X_YD = 6, X_WC = 9, X_DX = 36;
  static constexpr int X_BV = 45;  // raw P'(v_k, b_j) at X_BV + 3 j + k   (18)
  static constexpr int X_VW = 63;  // raw P'(v_k, omega_c) at X_VW + 3 c + k (9)
  static constexpr int X_H2 = 72;  // the F_cb T1^T part of H: chi-v block (9, row-major), chi-chi block (6, packed)
  static constexpr int X_XV = 87;  // the propagated velocity x'[3..5] (role PW -> role CC, for the residual)
  static constexpr int NXCH = 90;
  // k_step_quad_leg: z[3], R, valid -- and a six-row mode's second block z[3], R, on -- from the odometry (role PW) before barrier
  // A; SIX == 2: the propagated position x'[9..11] (role PW -> role CC); the two foot poses (2 x 7) from two other roles
  static constexpr int X_LEG = 90, X_XD = 100, X_FOOT = 103;
  // SIX == 1 (rbis_coop.hpp, coop_role_core): role PW's omega stage -- 1/d, y/d (3), its log-likelihood term, P'(c b, omega) (45) --
  // written before barrier A over the foot poses (role PW has consumed them by then)
  static constexpr int X6_ID = 103, X6_YD = 104, X6_LLI = 107, X6_A = 108;
  PB_HD static constexpr int nxch_leg(int six) { return six == 1 ? X6_A + 45 : X_FOOT + 14; }   // 153 | 117: two workgroups per CU fit 160 KB
  static constexpr int NXCH_SIX = X6_A + 45;
};

// (X hat(m)^T)[r][c] = (m x X_r)[c] for a row-major 3x3 block X
PB_HD double x_hat_t(const double (&X)[9], const double (&m)[3], int r, int c)
{
  if (c == 0) return X[3 * r + 2] * m[1] - X[3 * r + 1] * m[2];
  if (c == 1) return X[3 * r + 0] * m[2] - X[3 * r + 2] * m[0];
  return X[3 * r + 1] * m[0] - X[3 * r + 0] * m[1];
}

// the c-rows of one block column of a panel: (V, C, D) <- F_cc (V, C, D)   (all right-hand sides ORIGINAL blocks)
PB_HD void fcc_apply(const ProcBlocks &f, double (&V)[9], double (&Cc)[9], double (&D)[9])
{
  double nD[9], nV[9], nC[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { nD[i] = D[i]; nV[i] = V[i]; nC[i] = Cc[i]; }
  mat_mul_acc(f.A_R, V, nD);
  mat_mul_acc(f.A_RV, Cc, nD);
  hat_mul_acc(f.a_mw, V, nV);
  hat_mul_acc(f.a_g, Cc, nV);
  hat_mul_acc(f.a_mw, Cc, nC);
#pragma unroll
  for (int i = 0; i < 9; i++) { D[i] = nD[i]; V[i] = nV[i]; Cc[i] = nC[i]; }
}

// ------------------------------------------------------------------------------------------------------------
// wave 0, role CC: P_cc, loglik
// ------------------------------------------------------------------------------------------------------------
// LEG: the leg-odometry measurement is made by role PW of the same tile before barrier A (k_step_quad_leg, rbis_legstep.hpp)
// SIX: LegOdoCommon's six-row measurements as two 3-row blocks with ONE summed correction (rbis_coop.hpp, coop_role_core):
//   1  the angular-velocity block first, by role PW before barrier A (S = (q_gyro + r) I): every role downdates its entries with
//      P'(:, omega) behind barrier A, then the velocity block runs on the result -- no extra barrier
//   2  the velocity block, then the position block on its posterior: barrier B2 (every role has consumed the first hand-off, whose
//      slots the second one re-uses) and barrier C (role CC has published the second factors)
template <bool UPDATE, bool LEG = false, int SIX = 0, bool PIN = true, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_role_cc(LD ld, ST st, XW xw, XR xr, SYNC sync, const StepInputs &in, const Consts &k, const SixIn &six = SixIn())
{
  static_assert(SIX == 0 || UPDATE, "the six-row leg-odometry modes ride on the velocity update");
  constexpr int NS = 21;
  using L = Lay<NS>;
  double x[NS], q[4];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
  double ll = ld(L::OFF_LL);
  double Pc[45];
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Pc[pk(i, j)] = ld(L::OFF_P + pk(core_full(i), core_full(j)));
  ProcBlocks f;
  make_proc_blocks<NS>(x, q, in.dt, k, f);
  {  // F_cc P_cc F_cc^T as three elementary block-row congruences (blocks v=0 chi=1 Delta=2)
    const int src[2] = { 0, 1 };
    const int kind[2] = { 0, 0 };
    double A[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[0][i] = f.A_R[i]; A[1][i] = f.A_RV[i]; }
    RowOp<9, 2, 2>::apply(Pc, src, kind, A);
    const int kind1[2] = { 1, 1 };
    double A1[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A1[0][i] = f.a_mw[i % 3]; A1[1][i] = f.a_g[i % 3]; }
    RowOp<9, 0, 2>::apply(Pc, src, kind1, A1);
    const int src2[1] = { 1 };
    const int kind2[1] = { 1 };
    double A2[1][9];
#pragma unroll
    for (int i = 0; i < 9; i++) A2[0][i] = f.a_mw[i % 3];
    RowOp<9, 1, 1>::apply(Pc, src2, kind2, A2);
  }
  {  // Qd (closed form of rbis.cpp:91-116), c-part
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
  }
  // F_cc P_cc F_cc^T + Q is done HERE, in front of the barrier, while role CB is still busy with H (pb_pin, rbis_coop.hpp): the
  // compiler otherwise sinks it behind the barrier, into the stretch every other wave waits for (fused step at 64k filters
  // 40.6 -> 38.6 us on one box, both libraries in one run; pair kernels 1-4 us)
  // (PIN = false: the time-fused replay kernels, whose state lives in registers incl. AGPRs from step to step -- the pin wants
  // VGPRs: write-through replay 34.4 -> 39.0 us per step with it)
#ifndef PB_NO_PIN_CC   // (A/B builds)
  if constexpr (PIN) {
#pragma unroll
    for (int i = 0; i < 45; i++) pb_pin(Pc[i]);
  }
#endif
  sync();  // A: H and the propagated velocity are there; every role has consumed the prior x / quat
  double leg_z[3] = { 0.0, 0.0, 0.0 }, leg_r = 1.0, leg_valid = 0.0;
  if constexpr (LEG) {  // (read first: this role sits exactly at 256 registers and the allocation is fragile)
#pragma unroll
    for (int i = 0; i < 3; i++) leg_z[i] = xr(Quad::X_LEG + i);
    leg_r = xr(Quad::X_LEG + 3);
    leg_valid = xr(Quad::X_LEG + 4);
  }
  double z2[3] = { six.z[0], six.z[1], six.z[2] }, r2 = six.r, xd[3] = { 0.0, 0.0, 0.0 };
  bool upd2 = six.on;
  if constexpr (SIX == 2) {
    if constexpr (LEG) {
#pragma unroll
      for (int i = 0; i < 3; i++) z2[i] = xr(Quad::X_LEG + 5 + i);
      r2 = xr(Quad::X_LEG + 8);
      upd2 = in.upd && xr(Quad::X_LEG + 9) != 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) xd[i] = xr(Quad::X_XD + i);
  }
  double xv[3] = { 0.0, 0.0, 0.0 };
  if constexpr (UPDATE) {
#pragma unroll
    for (int i = 0; i < 3; i++) xv[i] = xr(Quad::X_XV + i);   // (SIX == 1: role PW has added the omega block's correction)
  }
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j <= i; j++)
      if (i < 6 || j < 6) Pc[pk(i, j)] += xr(Quad::X_H + pk(i, j));
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Pc[pk(3 + r, c)] += xr(Quad::X_H2 + 3 * r + c);
      if (c <= r) Pc[pk(3 + r, 3 + c)] += xr(Quad::X_H2 + 9 + pk(r, c));
    }
  if constexpr (SIX == 1) {  // the angular-velocity block: P_cc -= A A^T / d with A = P'(c, omega) from role PW
    const double idw = xr(Quad::X6_ID);
    ll += xr(Quad::X6_LLI);
    double A[9][3];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) A[i][kk] = xr(Quad::X6_A + 3 * i + kk);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const double ad[3] = { A[i][0] * idw, A[i][1] * idw, A[i][2] * idw };
#pragma unroll
      for (int j = 0; j <= i; j++) Pc[pk(i, j)] = fma(-ad[2], A[j][2], fma(-ad[1], A[j][1], fma(-ad[0], A[j][0], Pc[pk(i, j)])));
    }
  }

  if constexpr (UPDATE) {
    // S = R + P[v,v]; unpivoted LDL^T; y = L^-1 r  (rbis.cpp:124-143)
    double resid[3], S[6], d[3], y[3], id[3], yd[3];
    double mz[3], mr[3];
    bool mupd = in.upd;
    if constexpr (LEG) {
      mupd = in.upd && leg_valid != 0.0;
#pragma unroll
      for (int i = 0; i < 3; i++) { mz[i] = leg_z[i]; mr[i] = leg_r; }
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) { mz[i] = in.z[i]; mr[i] = in.rd[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) resid[i] = mupd ? mz[i] - xv[i] : 0.0;
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
    double W[9][3];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        double s = Pc[pk(i, kk)];
#pragma unroll
        for (int j = 0; j < kk; j++) s -= W[i][j] * S[pk(kk, j)];
        W[i][kk] = s;
      }
    xw(Quad::X_L + 0, S[pk(1, 0)]); xw(Quad::X_L + 1, S[pk(2, 0)]); xw(Quad::X_L + 2, S[pk(2, 1)]);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { xw(Quad::X_ID + kk, id[kk]); xw(Quad::X_YD + kk, yd[kk]); }
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
      for (int kk = 0; kk < 3; kk++) xw(Quad::X_WC + 3 * i + kk, W[i][kk]);
      xw(Quad::X_DX + i, fma(W[i][2], yd[2], fma(W[i][1], yd[1], W[i][0] * yd[0])));  // role PW applies it to x / quat
    }
    sync();  // B
#pragma unroll
    for (int i = 0; i < 9; i++) {
      double wd[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) wd[kk] = W[i][kk] * id[kk];
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double acc = Pc[pk(i, j)];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], W[j][kk], acc);
        if constexpr (SIX == 2) Pc[pk(i, j)] = acc;
        else st(L::OFF_P + pk(core_full(i), core_full(j)), acc);
      }
    }
    // -log(S.determinant()) - r^T S^-1 r (rbis.cpp:142): ONE log of the product, behind everything the other waves or
    // the memory system wait for
    if (mupd) ll += -log(det) - quad;
    if constexpr (SIX == 2) {
      // ---- the position block on the posterior of the velocity block; residual at x + dx of the first ----
      double rs2[3], S2[6], d2[3], y2[3], id2[3], yd2[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double dxd = fma(W[6 + i][2], yd[2], fma(W[6 + i][1], yd[1], W[6 + i][0] * yd[0]));
        rs2[i] = upd2 ? z2[i] - (xd[i] + dxd) : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) S2[pk(i, j)] = Pc[pk(6 + i, 6 + j)] + (i == j ? (upd2 ? r2 : 1.0) : 0.0);
      ldlt<3>(S2, d2);
      double quad2 = 0.0, det2 = 1.0;
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        double s2 = rs2[kk];
#pragma unroll
        for (int j = 0; j < kk; j++) s2 -= S2[pk(kk, j)] * y2[j];
        y2[kk] = upd2 ? s2 : 0.0;
        id2[kk] = upd2 ? 1.0 / d2[kk] : 0.0;
        yd2[kk] = y2[kk] * id2[kk];
        det2 *= d2[kk];
        quad2 += s2 * s2 * id2[kk];
      }
#pragma unroll
      for (int i = 0; i < 9; i++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
          double s2 = Pc[pk(i, 6 + kk)];
#pragma unroll
          for (int j = 0; j < kk; j++) s2 -= W[i][j] * S2[pk(kk, j)];
          W[i][kk] = s2;
        }
      sync();  // B2: every role has consumed the first hand-off
      xw(Quad::X_L + 0, S2[pk(1, 0)]); xw(Quad::X_L + 1, S2[pk(2, 0)]); xw(Quad::X_L + 2, S2[pk(2, 1)]);
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { xw(Quad::X_ID + kk, id2[kk]); xw(Quad::X_YD + kk, yd2[kk]); }
#pragma unroll
      for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int kk = 0; kk < 3; kk++) xw(Quad::X_WC + 3 * i + kk, W[i][kk]);
        xw(Quad::X_DX + i, fma(W[i][2], yd2[2], fma(W[i][1], yd2[1], W[i][0] * yd2[0])));
      }
      sync();  // C
#pragma unroll
      for (int i = 0; i < 9; i++) {
        double wd[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) wd[kk] = W[i][kk] * id2[kk];
#pragma unroll
        for (int j = 0; j <= i; j++) {
          double acc = Pc[pk(i, j)];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], W[j][kk], acc);
          st(L::OFF_P + pk(core_full(i), core_full(j)), acc);
        }
      }
      if (upd2) ll += -log(det2) - quad2;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) st(L::OFF_P + pk(core_full(i), core_full(j)), Pc[pk(i, j)]);
  }
  st(L::OFF_LL, ll);
}

// ------------------------------------------------------------------------------------------------------------
// wave 1, role CB: P_cb, P_bb, x[bg ba], x[omega]
// ------------------------------------------------------------------------------------------------------------
template <bool UPDATE, int SIX = 0, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_role_cb(LD ld, ST st, XW xw, XR xr, SYNC sync, const StepInputs &in, const Consts &k)
{
  constexpr int NS = 21;
  using L = Lay<NS>;
  double x[NS], q[4];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
  // Y[sb][J] = 3x3 block P(c block sb, bias block J), row-major; sb: v chi Delta; J: gyro bias, accel bias
  double Y[3][2][9];
#pragma unroll
  for (int sb = 0; sb < 3; sb++)
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Y[sb][J][3 * r + c] = ld(L::OFF_P + pk(core_full(3 * sb + r), core_full(9 + 3 * J + c)));
  double Pbb[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Pbb[pk(i, j)] = ld(L::OFF_P + pk(core_full(9 + i), core_full(9 + j)));
  double xb[6], xw_[3];
#pragma unroll
  for (int i = 0; i < 6; i++) xb[i] = x[15 + i];
#pragma unroll
  for (int i = 0; i < 3; i++) xw_[i] = in.gyro[i] - x[15 + i];  // rbis.cpp:50

  ProcBlocks f;
  make_proc_blocks<NS>(x, q, in.dt, k, f);
  // T1 = F_cc P_cb, block column by block column; P'_cb = T1 + F_cb P_bb; H = P'_cb F_cb^T + F_cb T1^T.  The F_cb T1^T part
  // (non-zero in the vv, chi-v and chi-chi blocks only) is published the moment a block column of T1 exists, so that T1
  // need not be kept next to P'_cb (27 doubles per lane).
  double hvv[6];
#pragma unroll
  for (int J = 0; J < 2; J++) {
    fcc_apply(f, Y[0][J], Y[1][J], Y[2][J]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (J == 0) {
          if (c <= r) hvv[pk(r, c)] = x_hat_t(Y[0][0], f.a_mv, c, r);
          xw(Quad::X_H2 + 3 * r + c, -in.dt * Y[0][0][3 * c + r]);
          if (c <= r) xw(Quad::X_H2 + 9 + pk(r, c), -in.dt * Y[1][0][3 * c + r]);
        } else if (c <= r) {
          hvv[pk(r, c)] -= in.dt * Y[0][1][3 * c + r];
        }
      }
    // blocks (bg, J) and (ba, J) of the symmetric P_bb
    double Bg[9], Ba[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Bg[3 * r + c] = Pbb[pk(r, 3 * J + c)];
        Ba[3 * r + c] = Pbb[pk(3 + r, 3 * J + c)];
      }
    hat_mul_acc(f.a_mv, Bg, Y[0][J]);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      Y[0][J][i] -= in.dt * Ba[i];
      Y[1][J][i] -= in.dt * Bg[i];
    }
  }
  // the P'_cb F_cb^T part, lower block triangle of the 9 x 9 (the Delta,Delta block is zero)
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (c <= r) xw(Quad::X_H + pk(r, c), x_hat_t(Y[0][0], f.a_mv, r, c) - in.dt * Y[0][1][3 * r + c] + hvv[pk(r, c)]);
      xw(Quad::X_H + pk(3 + r, c), x_hat_t(Y[1][0], f.a_mv, r, c) - in.dt * Y[1][1][3 * r + c]);
      if (c <= r) xw(Quad::X_H + pk(3 + r, 3 + c), -in.dt * Y[1][0][3 * r + c]);
      xw(Quad::X_H + pk(6 + r, c), x_hat_t(Y[2][0], f.a_mv, r, c) - in.dt * Y[2][1][3 * r + c]);
      xw(Quad::X_H + pk(6 + r, 3 + c), -in.dt * Y[2][0][3 * r + c]);
    }
  if constexpr (UPDATE && SIX != 1) {
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) xw(Quad::X_BV + 3 * j + kk, Y[0][j / 3][3 * kk + j % 3]);
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    Pbb[pk(r, r)] += in.qbg * in.dt;
    Pbb[pk(3 + r, 3 + r)] += in.qba * in.dt;
  }
  sync();  // A
  if constexpr (SIX == 1) {
    // the angular-velocity block (role PW's factors): P_cb -= A_c A_b^T / d, P_bb -= A_b A_b^T / d, x_b += A_b (y / d),
    // x_omega += q_gyro (y / d); only then the raw P'(v, b) columns of the velocity block (needed behind barrier B)
    const double idw = xr(Quad::X6_ID);
    const double ydw[3] = { xr(Quad::X6_YD), xr(Quad::X6_YD + 1), xr(Quad::X6_YD + 2) };
    double Ab[6][3];
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) Ab[j][kk] = xr(Quad::X6_A + 3 * (9 + j) + kk);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const double ad[3] = { xr(Quad::X6_A + 3 * i) * idw, xr(Quad::X6_A + 3 * i + 1) * idw, xr(Quad::X6_A + 3 * i + 2) * idw };
#pragma unroll
      for (int j = 0; j < 6; j++) {
        double &e = Y[i / 3][j / 3][3 * (i % 3) + j % 3];
        e = fma(-ad[2], Ab[j][2], fma(-ad[1], Ab[j][1], fma(-ad[0], Ab[j][0], e)));
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double ad[3] = { Ab[j][0] * idw, Ab[j][1] * idw, Ab[j][2] * idw };
      xb[j] += fma(Ab[j][2], ydw[2], fma(Ab[j][1], ydw[1], Ab[j][0] * ydw[0]));
#pragma unroll
      for (int j2 = 0; j2 <= j; j2++) Pbb[pk(j, j2)] = fma(-ad[2], Ab[j2][2], fma(-ad[1], Ab[j2][1], fma(-ad[0], Ab[j2][0], Pbb[pk(j, j2)])));
    }
#pragma unroll
    for (int c = 0; c < 3; c++) xw_[c] = fma(in.qg, ydw[c], xw_[c]);
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int kk = 0; kk < 3; kk++) xw(Quad::X_BV + 3 * j + kk, Y[0][j / 3][3 * kk + j % 3]);
  }
  if constexpr (UPDATE) {
    sync();  // B
    const double L10 = xr(Quad::X_L + 0), L20 = xr(Quad::X_L + 1), L21 = xr(Quad::X_L + 2);
    double id[3], yd[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { id[kk] = xr(Quad::X_ID + kk); yd[kk] = xr(Quad::X_YD + kk); }
    double Wb[6][3];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double c0 = Y[0][j / 3][0 + j % 3], c1 = Y[0][j / 3][3 + j % 3], c2 = Y[0][j / 3][6 + j % 3];
      Wb[j][0] = c0;
      Wb[j][1] = c1 - Wb[j][0] * L10;
      Wb[j][2] = c2 - Wb[j][0] * L20 - Wb[j][1] * L21;
      xb[j] += fma(Wb[j][2], yd[2], fma(Wb[j][1], yd[1], Wb[j][0] * yd[0]));
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {  // W_omega from role PW's raw column
      const double w0 = xr(Quad::X_VW + 3 * c + 0);
      const double w1 = xr(Quad::X_VW + 3 * c + 1) - w0 * L10;
      const double w2 = xr(Quad::X_VW + 3 * c + 2) - w0 * L20 - w1 * L21;
      xw_[c] += fma(w2, yd[2], fma(w1, yd[1], w0 * yd[0]));
    }
    // rows of the storage layout: bias j: its 9 c-entries, then its P_bb entries
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double wd[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) wd[kk] = Wb[j][kk] * id[kk];
#pragma unroll
      for (int i = 0; i < 9; i++) {
        double acc = Y[i / 3][j / 3][3 * (i % 3) + j % 3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], xr(Quad::X_WC + 3 * i + kk), acc);
        if constexpr (SIX == 2) Y[i / 3][j / 3][3 * (i % 3) + j % 3] = acc;
        else st(L::OFF_P + pk(core_full(9 + j), core_full(i)), acc);
      }
#pragma unroll
      for (int j2 = 0; j2 <= j; j2++) {
        double acc = Pbb[pk(j, j2)];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wb[j2][kk], acc);
        if constexpr (SIX == 2) Pbb[pk(j, j2)] = acc;
        else st(L::OFF_P + pk(core_full(9 + j), core_full(9 + j2)), acc);
      }
    }
    if constexpr (SIX == 2) {
      // ---- the position block: this role's raw P''(Delta, b) columns go where the first block's P'(v, b) were ----
      sync();  // B2
#pragma unroll
      for (int j = 0; j < 6; j++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) xw(Quad::X_BV + 3 * j + kk, Y[2][j / 3][3 * kk + j % 3]);
      sync();  // C
      const double M10 = xr(Quad::X_L + 0), M20 = xr(Quad::X_L + 1), M21 = xr(Quad::X_L + 2);
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { id[kk] = xr(Quad::X_ID + kk); yd[kk] = xr(Quad::X_YD + kk); }
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const double c0 = Y[2][j / 3][0 + j % 3], c1 = Y[2][j / 3][3 + j % 3], c2 = Y[2][j / 3][6 + j % 3];
        Wb[j][0] = c0;
        Wb[j][1] = c1 - Wb[j][0] * M10;
        Wb[j][2] = c2 - Wb[j][0] * M20 - Wb[j][1] * M21;
        xb[j] += fma(Wb[j][2], yd[2], fma(Wb[j][1], yd[1], Wb[j][0] * yd[0]));
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double w0 = xr(Quad::X_VW + 3 * c + 0);
        const double w1 = xr(Quad::X_VW + 3 * c + 1) - w0 * M10;
        const double w2 = xr(Quad::X_VW + 3 * c + 2) - w0 * M20 - w1 * M21;
        xw_[c] += fma(w2, yd[2], fma(w1, yd[1], w0 * yd[0]));
      }
#pragma unroll
      for (int j = 0; j < 6; j++) {
        double wd[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) wd[kk] = Wb[j][kk] * id[kk];
#pragma unroll
        for (int i = 0; i < 9; i++) {
          double acc = Y[i / 3][j / 3][3 * (i % 3) + j % 3];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], xr(Quad::X_WC + 3 * i + kk), acc);
          st(L::OFF_P + pk(core_full(9 + j), core_full(i)), acc);
        }
#pragma unroll
        for (int j2 = 0; j2 <= j; j2++) {
          double acc = Pbb[pk(j, j2)];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wb[j2][kk], acc);
          st(L::OFF_P + pk(core_full(9 + j), core_full(9 + j2)), acc);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 6; j++) {
#pragma unroll
      for (int i = 0; i < 9; i++) st(L::OFF_P + pk(core_full(9 + j), core_full(i)), Y[i / 3][j / 3][3 * (i % 3) + j % 3]);
#pragma unroll
      for (int j2 = 0; j2 <= j; j2++) st(L::OFF_P + pk(core_full(9 + j), core_full(9 + j2)), Pbb[pk(j, j2)]);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) st(L::OFF_VEC + 15 + i, xb[i]);
#pragma unroll
  for (int i = 0; i < 3; i++) st(L::OFF_VEC + i, xw_[i]);
}

// ------------------------------------------------------------------------------------------------------------
// waves 2 and 3, roles PW (J = 0: omega; + x[v chi Delta], quat) and PA (J = 1: accel; + x[accel]): one block column of
// the passive panels each
// ------------------------------------------------------------------------------------------------------------
// PIN: the panel propagation is finished in FRONT of barrier A (pb_pin, rbis_coop.hpp) -- the plain step gains 0.8 us of 40 at 64k
// filters; the pair kernels, whose role PW runs the odometry first, lose 1-2 us with it and leave the order to the compiler.
template <bool UPDATE, int J, int SIX = 0, bool PIN = false, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_role_passive(LD ld, ST st, XW xw, XR xr, SYNC sync, const StepInputs &in, const Consts &k, const SixIn &six = SixIn())
{
  constexpr int NS = 21;
  using L = Lay<NS>;
  double x[NS], q[4];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
  // X[sb] = 3x3 block P(core block sb, passive block J), row-major; sb: v chi Delta bg ba
  double X[5][9];
#pragma unroll
  for (int sb = 0; sb < 5; sb++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) X[sb][3 * r + c] = ld(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)));
  double Pjj[6];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) Pjj[pk(r, c)] = ld(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)));
  double Paw[J == 1 ? 9 : 1];  // P(accel_r, omega_c): untouched by the process step (rbis.cpp:120-121)
  if constexpr (J == 1) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Paw[3 * r + c] = ld(L::OFF_P + pk(passive_full(3 + r), passive_full(c)));
  }
  ProcBlocks f;
  make_proc_blocks<NS>(x, q, in.dt, k, f);
  {
    double G0[9], A0[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { G0[i] = X[3][i]; A0[i] = X[4][i]; }
    fcc_apply(f, X[0], X[1], X[2]);
    hat_mul_acc(f.a_mv, G0, X[0]);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      X[0][i] -= in.dt * A0[i];
      X[1][i] -= in.dt * G0[i];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) Pjj[pk(r, c)] = (r == c) ? (J == 0 ? in.qg : in.qa) : 0.0;
  double xp[3];  // J == 1: rbis.cpp:51
#pragma unroll
  for (int i = 0; i < 3; i++) xp[i] = in.accel[i] - x[18 + i];
  double dx1[9];    // SIX == 1, role PW: the omega block's correction of x[v chi Delta]
  double dsum[9];   // SIX == 2, role PW: the velocity block's correction, applied together with the position block's
#pragma unroll
  for (int i = 0; i < 9; i++) dx1[i] = dsum[i] = 0.0;
  if constexpr (J == 0) {
    // state propagate (rbis.cpp:37-75) on this role's copy: x[v chi Delta] and quat are its to store
    ins_update_state<NS>(x, q, in.gyro, in.accel, in.dt, k);
    if constexpr (SIX == 1) {
      // ---- the angular-velocity block: S = (q_gyro + r) I, W = P'[:, omega] = this role's panel; no factorisation ----
      const double dw = in.qg + six.r;
      const double idw = six.on ? 1.0 / dw : 0.0;
      double ydw[3], quad = 0.0;
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        const double res = six.on ? six.z[kk] - x[kk] : 0.0;   // x[0..2] = gyro - bias (rbis.cpp:50)
        ydw[kk] = res * idw;
        quad = fma(res, ydw[kk], quad);
      }
      xw(Quad::X6_ID, idw);
#pragma unroll
      for (int kk = 0; kk < 3; kk++) xw(Quad::X6_YD + kk, ydw[kk]);
      xw(Quad::X6_LLI, six.on ? -log(dw * dw * dw) - quad : 0.0);
      const double keep = fma(-in.qg, idw, 1.0);   // 1 - q_gyro / d
#pragma unroll
      for (int sb = 0; sb < 5; sb++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
          for (int kk = 0; kk < 3; kk++) xw(Quad::X6_A + 3 * (3 * sb + r) + kk, X[sb][3 * r + kk]);
          if (sb < 3) dx1[3 * sb + r] = fma(X[sb][3 * r + 2], ydw[2], fma(X[sb][3 * r + 1], ydw[1], X[sb][3 * r] * ydw[0]));
#pragma unroll
          for (int kk = 0; kk < 3; kk++) X[sb][3 * r + kk] *= keep;
        }
#pragma unroll
      for (int r = 0; r < 3; r++) Pjj[pk(r, r)] = in.qg * keep;
    }
    if constexpr (UPDATE) {
#pragma unroll
      for (int i = 0; i < 3; i++) xw(Quad::X_XV + i, SIX == 1 ? x[3 + i] + dx1[i] : x[3 + i]);
      if constexpr (SIX == 2) {
#pragma unroll
        for (int i = 0; i < 3; i++) xw(Quad::X_XD + i, x[9 + i]);
      }
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) xw(Quad::X_VW + 3 * c + kk, X[0][3 * kk + c]);
    }
  }
  if constexpr (PIN) {
#pragma unroll
    for (int sb = 0; sb < 5; sb++)
#pragma unroll
      for (int i = 0; i < 9; i++) pb_pin(X[sb][i]);
  }
  sync();  // A
  if constexpr (SIX == 1 && J == 1) {
    // the angular-velocity block on the accel panel: P'(a, omega) is this role's (untouched by the predict)
    const double idw = xr(Quad::X6_ID);
    const double ydw[3] = { xr(Quad::X6_YD), xr(Quad::X6_YD + 1), xr(Quad::X6_YD + 2) };
#pragma unroll
    for (int sb = 0; sb < 5; sb++)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int i = 3 * sb + r;
        const double ad[3] = { xr(Quad::X6_A + 3 * i) * idw, xr(Quad::X6_A + 3 * i + 1) * idw, xr(Quad::X6_A + 3 * i + 2) * idw };
#pragma unroll
        for (int c = 0; c < 3; c++)
          X[sb][3 * r + c] = fma(-ad[2], Paw[3 * c + 2], fma(-ad[1], Paw[3 * c + 1], fma(-ad[0], Paw[3 * c], X[sb][3 * r + c])));
      }
    const double keep = fma(-in.qg, idw, 1.0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double ad[3] = { Paw[3 * r] * idw, Paw[3 * r + 1] * idw, Paw[3 * r + 2] * idw };
      xp[r] += fma(Paw[3 * r + 2], ydw[2], fma(Paw[3 * r + 1], ydw[1], Paw[3 * r] * ydw[0]));
#pragma unroll
      for (int c = 0; c <= r; c++) Pjj[pk(r, c)] = fma(-ad[2], Paw[3 * c + 2], fma(-ad[1], Paw[3 * c + 1], fma(-ad[0], Paw[3 * c], Pjj[pk(r, c)])));
    }
#pragma unroll
    for (int i = 0; i < 9; i++) Paw[i] *= keep;
  }
  if constexpr (UPDATE) {
    sync();  // B
    const double L10 = xr(Quad::X_L + 0), L20 = xr(Quad::X_L + 1), L21 = xr(Quad::X_L + 2);
    double id[3], yd[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { id[kk] = xr(Quad::X_ID + kk); yd[kk] = xr(Quad::X_YD + kk); }
    double Wp[3][3];  // W_p = P'[p, v] L^-T
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Wp[c][0] = X[0][0 + c];
      Wp[c][1] = X[0][3 + c] - Wp[c][0] * L10;
      Wp[c][2] = X[0][6 + c] - Wp[c][0] * L20 - Wp[c][1] * L21;
      if constexpr (J == 1) xp[c] += fma(Wp[c][2], yd[2], fma(Wp[c][1], yd[1], Wp[c][0] * yd[0]));
    }
    if constexpr (J == 0) {  // apply role CC's dx to the state vector and the quaternion (addState, rbis.cpp:219-227)
      double dfull[NS];
#pragma unroll
      for (int i = 0; i < NS; i++) dfull[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 9; i++) dfull[core_full(i)] = (SIX == 1) ? xr(Quad::X_DX + i) + dx1[i] : xr(Quad::X_DX + i);
      if constexpr (SIX == 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) dsum[i] = dfull[core_full(i)];   // applied together with the second block's
      } else {
        if (in.upd || (SIX == 1 && six.on)) add_delta<NS>(x, q, dfull, k.chi_tol);
      }
    }
#pragma unroll
    for (int sb = 0; sb < 5; sb++)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double wd[3];
        if (sb < 3) {  // role CC's rows of W
#pragma unroll
          for (int kk = 0; kk < 3; kk++) wd[kk] = xr(Quad::X_WC + 3 * (3 * sb + r) + kk) * id[kk];
        } else {       // W_b = P'[b, v] L^-T from role CB's raw column
          const int j = 3 * (sb - 3) + r;
          const double w0 = xr(Quad::X_BV + 3 * j + 0);
          const double w1 = xr(Quad::X_BV + 3 * j + 1) - w0 * L10;
          const double w2 = xr(Quad::X_BV + 3 * j + 2) - w0 * L20 - w1 * L21;
          wd[0] = w0 * id[0]; wd[1] = w1 * id[1]; wd[2] = w2 * id[2];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double acc = X[sb][3 * r + c];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
          if constexpr (SIX == 2) X[sb][3 * r + c] = acc;
          else st(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)), acc);
        }
      }
    double Ww[3][3];  // J == 1: W_omega from role PW's raw column
    if constexpr (J == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Ww[c][0] = xr(Quad::X_VW + 3 * c + 0);
        Ww[c][1] = xr(Quad::X_VW + 3 * c + 1) - Ww[c][0] * L10;
        Ww[c][2] = xr(Quad::X_VW + 3 * c + 2) - Ww[c][0] * L20 - Ww[c][1] * L21;
      }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double wd[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) wd[kk] = Wp[r][kk] * id[kk];
      if constexpr (J == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double acc = Paw[3 * r + c];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Ww[c][kk], acc);
          if constexpr (SIX == 2) Paw[3 * r + c] = acc;
          else st(L::OFF_P + pk(passive_full(3 + r), passive_full(c)), acc);
        }
      }
#pragma unroll
      for (int c = 0; c <= r; c++) {
        double acc = Pjj[pk(r, c)];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
        if constexpr (SIX == 2) Pjj[pk(r, c)] = acc;
        else st(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)), acc);
      }
    }
    if constexpr (SIX == 2) {
      // ---- the position block: raw P''(Delta, omega) columns where the first block's P'(v, omega) were ----
      sync();  // B2
      if constexpr (J == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int kk = 0; kk < 3; kk++) xw(Quad::X_VW + 3 * c + kk, X[2][3 * kk + c]);
      }
      sync();  // C
      const double M10 = xr(Quad::X_L + 0), M20 = xr(Quad::X_L + 1), M21 = xr(Quad::X_L + 2);
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { id[kk] = xr(Quad::X_ID + kk); yd[kk] = xr(Quad::X_YD + kk); }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Wp[c][0] = X[2][0 + c];
        Wp[c][1] = X[2][3 + c] - Wp[c][0] * M10;
        Wp[c][2] = X[2][6 + c] - Wp[c][0] * M20 - Wp[c][1] * M21;
        if constexpr (J == 1) xp[c] += fma(Wp[c][2], yd[2], fma(Wp[c][1], yd[1], Wp[c][0] * yd[0]));
      }
      if constexpr (J == 0) {  // ONE addState of the two blocks' summed correction
        double dfull[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) dfull[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; i++) dfull[core_full(i)] = dsum[i] + xr(Quad::X_DX + i);
        if (in.upd || six.on) add_delta<NS>(x, q, dfull, k.chi_tol);
      }
#pragma unroll
      for (int sb = 0; sb < 5; sb++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double wd[3];
          if (sb < 3) {
#pragma unroll
            for (int kk = 0; kk < 3; kk++) wd[kk] = xr(Quad::X_WC + 3 * (3 * sb + r) + kk) * id[kk];
          } else {
            const int j = 3 * (sb - 3) + r;
            const double w0 = xr(Quad::X_BV + 3 * j + 0);
            const double w1 = xr(Quad::X_BV + 3 * j + 1) - w0 * M10;
            const double w2 = xr(Quad::X_BV + 3 * j + 2) - w0 * M20 - w1 * M21;
            wd[0] = w0 * id[0]; wd[1] = w1 * id[1]; wd[2] = w2 * id[2];
          }
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double acc = X[sb][3 * r + c];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
            st(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)), acc);
          }
        }
      if constexpr (J == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          Ww[c][0] = xr(Quad::X_VW + 3 * c + 0);
          Ww[c][1] = xr(Quad::X_VW + 3 * c + 1) - Ww[c][0] * M10;
          Ww[c][2] = xr(Quad::X_VW + 3 * c + 2) - Ww[c][0] * M20 - Ww[c][1] * M21;
        }
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double wd[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) wd[kk] = Wp[r][kk] * id[kk];
        if constexpr (J == 1) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double acc = Paw[3 * r + c];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Ww[c][kk], acc);
            st(L::OFF_P + pk(passive_full(3 + r), passive_full(c)), acc);
          }
        }
#pragma unroll
        for (int c = 0; c <= r; c++) {
          double acc = Pjj[pk(r, c)];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
          st(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)), acc);
        }
      }
    }
  } else {
#pragma unroll
    for (int sb = 0; sb < 5; sb++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) st(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)), X[sb][3 * r + c]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if constexpr (J == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) st(L::OFF_P + pk(passive_full(3 + r), passive_full(c)), Paw[3 * r + c]);
      }
#pragma unroll
      for (int c = 0; c <= r; c++) st(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)), Pjj[pk(r, c)]);
    }
  }
  if constexpr (J == 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) st(L::OFF_VEC + core_full(i), x[core_full(i)]);
#pragma unroll
    for (int i = 0; i < 4; i++) st(L::OFF_QUAT + i, q[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) st(L::OFF_VEC + passive_full(3 + i), xp[i]);
  }
}

// =================================================================================================================
// Stand-alone indexed (+ orientation) update on the four-wave mapping: the measurement indices are a compile-time list of
// c-states (v chi Delta: every index list the reference's handlers produce), R is diagonal.  S = R + P_cc[idx,idx] lives
// entirely in role CC, so ONE barrier is enough: before it role CC publishes the LDL^T factors, its rows of
// W = P[:,idx] L^-T and dx for the c-states, role CB its raw P[idx, b] columns and role PW its raw P[idx, omega] columns;
// behind it every role forms the rows of W it needs and downdates its own entries.
// (RBISIndexedMeasurement / RBISIndexedPlusOrientationMeasurement::updateFilter, rbis_update_interface.cpp:54-107.)
// =================================================================================================================
template <class CORR>
struct QuadU {
  static constexpr int M = CORR::M;
  static constexpr int X_L = 0, X_ID = X_L + M * (M - 1) / 2, X_YD = X_ID + M, X_WC = X_YD + M, X_DX = X_WC + 9 * M,
                       X_BV = X_DX + 9,        // raw P(idx_k, b_j) at X_BV + M j + k
                       X_VW = X_BV + 6 * M,    // raw P(idx_k, omega_c) at X_VW + M c + k
                       NXCH = X_VW + 3 * M;
  PB_HD static constexpr bool all_core()
  {
    for (int i = 0; i < M; i++)
      if (CORR::sub[i] >= 9) return false;
    return true;
  }
};

// rows of W for a raw column set: w[kk] = raw[kk] - sum_{j<kk} w[j] L[kk][j]   (L packed strictly-lower by rows)
template <int M>
PB_HD void quad_fsub(const double (&raw)[M], const double (&Lp)[M * (M - 1) / 2 + 1], double (&w)[M])
{
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s2 = raw[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s2 -= w[j] * Lp[kk * (kk - 1) / 2 + j];
    w[kk] = s2;
  }
}

template <class CORR, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_upd_cc(LD ld, ST st, XW xw, XR xr, SYNC sync, const CorrInputs &cin, const Consts &k)
{
  constexpr int NS = 21, M = CORR::M;
  using L = Lay<NS>;
  using QU = QuadU<CORR>;
  static_assert(QU::all_core(), "measurement indices must be c-states");
  double ll = ld(L::OFF_LL);
  double Pc[45];
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Pc[pk(i, j)] = ld(L::OFF_P + pk(core_full(i), core_full(j)));
  double r2[M], S2[M * (M + 1) / 2], d2[M], y2[M], id2[M], yd2[M];
  double dq3[3] = { 0.0, 0.0, 0.0 };
  if constexpr (CORR::ORIENT) {
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
    subtract_quats(cin.qm, q, dq3);
  }
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    const int ii = core_full(CORR::sub[kk]);
    const double r = (CORR::ORIENT && ii >= 6 && ii <= 8) ? dq3[ii - 6] : cin.z[kk] - ld(L::OFF_VEC + ii);
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
#pragma unroll
  for (int i = 1; i < M; i++)
#pragma unroll
    for (int j = 0; j < i; j++) xw(QU::X_L + i * (i - 1) / 2 + j, S2[pk(i, j)]);
#pragma unroll
  for (int kk = 0; kk < M; kk++) { xw(QU::X_ID + kk, id2[kk]); xw(QU::X_YD + kk, yd2[kk]); }
  // W = P_cc[:, idx] L^-T row by row, published as it is formed.  Six measurement rows: P_cc (90 registers) and W (108) do
  // not fit together at two waves per SIMD, so W is NOT kept: the downdate reads its rows back from the hand-off area
  // (this wave's own LDS writes are in order).
  constexpr bool W_LDS = (M > 4);
  double W[W_LDS ? 1 : 9][M];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    double w2[M], dxs = 0.0;
#pragma unroll
    for (int kk = 0; kk < M; kk++) {
      double s2 = Pc[pk(i, CORR::sub[kk])];
#pragma unroll
      for (int j = 0; j < kk; j++) s2 -= w2[j] * S2[pk(kk, j)];
      w2[kk] = s2;
      if constexpr (!W_LDS) W[i][kk] = s2;
      xw(QU::X_WC + M * i + kk, s2);
      dxs = (kk == 0) ? s2 * yd2[0] : fma(s2, yd2[kk], dxs);
    }
    xw(QU::X_DX + i, dxs);
  }
  sync();
#pragma unroll
  for (int i = 0; i < 9; i++) {
    // (without the clobber the compiler shares every LDS read of W between the rows: all of W back in registers)
    if constexpr (W_LDS) reload_fence();
    double wd[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) wd[kk] = (W_LDS ? xr(QU::X_WC + M * i + kk) : W[i][kk]) * id2[kk];
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double acc = Pc[pk(i, j)];
#pragma unroll
      for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], W_LDS ? xr(QU::X_WC + M * j + kk) : W[j][kk], acc);
      st(L::OFF_P + pk(core_full(i), core_full(j)), acc);
    }
  }
  if (cin.upd) ll += -log(det2) - quad2;
  st(L::OFF_LL, ll);
}

template <class CORR, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_upd_cb(LD ld, ST st, XW xw, XR xr, SYNC sync, const CorrInputs &cin, const Consts &k)
{
  constexpr int NS = 21, M = CORR::M;
  using L = Lay<NS>;
  using QU = QuadU<CORR>;
  double Y[3][2][9];
#pragma unroll
  for (int sb = 0; sb < 3; sb++)
#pragma unroll
    for (int J = 0; J < 2; J++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Y[sb][J][3 * r + c] = ld(L::OFF_P + pk(core_full(3 * sb + r), core_full(9 + 3 * J + c)));
  double Pbb[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Pbb[pk(i, j)] = ld(L::OFF_P + pk(core_full(9 + i), core_full(9 + j)));
  double xb[6], xw_[3];
#pragma unroll
  for (int i = 0; i < 6; i++) xb[i] = ld(L::OFF_VEC + 15 + i);
#pragma unroll
  for (int i = 0; i < 3; i++) xw_[i] = ld(L::OFF_VEC + i);
  auto pcb = [&](int i, int j) { return Y[i / 3][j / 3][3 * (i % 3) + j % 3]; };  // P(c_i, b_j)
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int kk = 0; kk < M; kk++) xw(QU::X_BV + M * j + kk, pcb(CORR::sub[kk], j));
  sync();
  double Lp[M * (M - 1) / 2 + 1], id2[M], yd2[M];
#pragma unroll
  for (int i = 0; i < M * (M - 1) / 2; i++) Lp[i] = xr(QU::X_L + i);
#pragma unroll
  for (int kk = 0; kk < M; kk++) { id2[kk] = xr(QU::X_ID + kk); yd2[kk] = xr(QU::X_YD + kk); }
#pragma unroll
  for (int c = 0; c < 3; c++) {  // omega's share of dx: W_omega from role PW's raw columns
    double raw[M], w[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) raw[kk] = xr(QU::X_VW + M * c + kk);
    quad_fsub<M>(raw, Lp, w);
#pragma unroll
    for (int kk = 0; kk < M; kk++) xw_[c] = fma(w[kk], yd2[kk], xw_[c]);
  }
  // Six measurement rows: P_cb (108 registers), P_bb (42) and all of W_b (72) do not fit together at two waves per SIMD.
  // Two passes then: the P_cb columns with ONE row of W_b at a time, and W_b again (its raw columns are still in this
  // wave's own hand-off slots) for P_bb once the P_cb registers are free.
  constexpr bool TWO_PASS = (M > 4);
  double Wb[6][M];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double raw[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) raw[kk] = pcb(CORR::sub[kk], j);
    quad_fsub<M>(raw, Lp, Wb[j]);
#pragma unroll
    for (int kk = 0; kk < M; kk++) xb[j] = fma(Wb[j][kk], yd2[kk], xb[j]);
    if constexpr (TWO_PASS) {
      double wd[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) wd[kk] = Wb[j][kk] * id2[kk];
#pragma unroll
      for (int i = 0; i < 9; i++) {
        double acc = pcb(i, j);
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], xr(QU::X_WC + M * i + kk), acc);
        st(L::OFF_P + pk(core_full(9 + j), core_full(i)), acc);
      }
      reload_fence();
    }
  }
  if constexpr (TWO_PASS) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double raw[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) raw[kk] = xr(QU::X_BV + M * j + kk);
      quad_fsub<M>(raw, Lp, Wb[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double wd[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) wd[kk] = Wb[j][kk] * id2[kk];
    if constexpr (!TWO_PASS) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        double acc = pcb(i, j);
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], xr(QU::X_WC + M * i + kk), acc);
        st(L::OFF_P + pk(core_full(9 + j), core_full(i)), acc);
      }
    }
#pragma unroll
    for (int j2 = 0; j2 <= j; j2++) {
      double acc = Pbb[pk(j, j2)];
#pragma unroll
      for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], Wb[j2][kk], acc);
      st(L::OFF_P + pk(core_full(9 + j), core_full(9 + j2)), acc);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) st(L::OFF_VEC + 15 + i, xb[i]);
#pragma unroll
  for (int i = 0; i < 3; i++) st(L::OFF_VEC + i, xw_[i]);
}

template <class CORR, int J, class LD, class ST, class XW, class XR, class SYNC>
PB_HD void quad_upd_passive(LD ld, ST st, XW xw, XR xr, SYNC sync, const CorrInputs &cin, const Consts &k)
{
  constexpr int NS = 21, M = CORR::M;
  using L = Lay<NS>;
  using QU = QuadU<CORR>;
  double X[5][9];
#pragma unroll
  for (int sb = 0; sb < 5; sb++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) X[sb][3 * r + c] = ld(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)));
  double Pjj[6];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) Pjj[pk(r, c)] = ld(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)));
  double Paw[J == 1 ? 9 : 1];
  double xa[3] = { 0.0, 0.0, 0.0 };
  if constexpr (J == 1) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Paw[3 * r + c] = ld(L::OFF_P + pk(passive_full(3 + r), passive_full(c)));
#pragma unroll
    for (int i = 0; i < 3; i++) xa[i] = ld(L::OFF_VEC + 12 + i);
  }
  double x[NS], q[4] = { 1.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = 0.0;
  if constexpr (J == 0) {  // role PW owns x[v chi Delta] and the quaternion
#pragma unroll
    for (int i = 0; i < 9; i++) x[core_full(i)] = ld(L::OFF_VEC + core_full(i));
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = ld(L::OFF_QUAT + i);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int kk = 0; kk < M; kk++) xw(QU::X_VW + M * c + kk, X[CORR::sub[kk] / 3][3 * (CORR::sub[kk] % 3) + c]);
  }
  sync();
  double Lp[M * (M - 1) / 2 + 1], id2[M], yd2[M];
#pragma unroll
  for (int i = 0; i < M * (M - 1) / 2; i++) Lp[i] = xr(QU::X_L + i);
#pragma unroll
  for (int kk = 0; kk < M; kk++) { id2[kk] = xr(QU::X_ID + kk); yd2[kk] = xr(QU::X_YD + kk); }
  double Wp[3][M];  // W_p = P[p, idx] L^-T
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double raw[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) raw[kk] = X[CORR::sub[kk] / 3][3 * (CORR::sub[kk] % 3) + c];
    quad_fsub<M>(raw, Lp, Wp[c]);
    if constexpr (J == 1) {
#pragma unroll
      for (int kk = 0; kk < M; kk++) xa[c] = fma(Wp[c][kk], yd2[kk], xa[c]);
    }
  }
#pragma unroll
  for (int sb = 0; sb < 5; sb++)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double wd[M];
      if (sb < 3) {  // role CC's rows of W
#pragma unroll
        for (int kk = 0; kk < M; kk++) wd[kk] = xr(QU::X_WC + M * (3 * sb + r) + kk) * id2[kk];
      } else {       // W_b from role CB's raw column
        double raw[M], w[M];
#pragma unroll
        for (int kk = 0; kk < M; kk++) raw[kk] = xr(QU::X_BV + M * (3 * (sb - 3) + r) + kk);
        quad_fsub<M>(raw, Lp, w);
#pragma unroll
        for (int kk = 0; kk < M; kk++) wd[kk] = w[kk] * id2[kk];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double acc = X[sb][3 * r + c];
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
        st(L::OFF_P + pk(core_full(3 * sb + r), passive_full(3 * J + c)), acc);
      }
    }
  double Ww[J == 1 ? 3 : 1][M];
  if constexpr (J == 1) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double raw[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) raw[kk] = xr(QU::X_VW + M * c + kk);
      quad_fsub<M>(raw, Lp, Ww[c]);
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double wd[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) wd[kk] = Wp[r][kk] * id2[kk];
    if constexpr (J == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double acc = Paw[3 * r + c];
#pragma unroll
        for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], Ww[c][kk], acc);
        st(L::OFF_P + pk(passive_full(3 + r), passive_full(c)), acc);
      }
    }
#pragma unroll
    for (int c = 0; c <= r; c++) {
      double acc = Pjj[pk(r, c)];
#pragma unroll
      for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], Wp[c][kk], acc);
      st(L::OFF_P + pk(passive_full(3 * J + r), passive_full(3 * J + c)), acc);
    }
  }
  if constexpr (J == 0) {
    double dfull[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) dfull[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) dfull[core_full(i)] = xr(QU::X_DX + i);
    if (cin.upd) add_delta<NS>(x, q, dfull, k.chi_tol);
#pragma unroll
    for (int i = 0; i < 9; i++) st(L::OFF_VEC + core_full(i), x[core_full(i)]);
#pragma unroll
    for (int i = 0; i < 4; i++) st(L::OFF_QUAT + i, q[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) st(L::OFF_VEC + 12 + i, xa[i]);
  }
}

}  // namespace pb
