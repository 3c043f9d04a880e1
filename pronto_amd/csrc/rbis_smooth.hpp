// rbis_smooth.hpp -- RTS smoother step (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) on gfx950.
//
//   L      = P_k Ad^T (P^-_{k+1})^-1            (Ad about the filtered state at k; bias blocks of P^- replaced by I
//   P^s_k  = P_k + L (P^s_{k+1} - P^-_{k+1}) L^T    when their variance is < 1e-11, rbis.cpp:244-251)
//   x^s_k  = x_k (+) L (x^s_{k+1} (-) x^-_{k+1})
//
// Unlike the forward step this IS dense n x n work (an SPD solve with n right-hand sides and two dense products,
// ~40 kflop per 21-state filter against ~8 KB of state: ~5 flop/B, at the fp64 balance point), so one lane per filter
// is the wrong shape.  Mapping: a GROUP of G = 16 (n=15) or 32 (n=21) lanes owns one filter, lane r owns matrix row r,
// 4 filters per workgroup.  A group never spans a wave, and a wave's LDS operations execute in order, so the steps
// below are separated by wave-local ordering only -- no workgroup barrier anywhere.
//   1. lane r: row r of P_k stays in registers; row r of P^- (bias-fixed) and of D = P^s - P^- go to LDS
//   2. lane r computes column r of T = Ad P_k from ITS OWN row: T[:,r] = Ad (P_k[r,:])^T  (P_k symmetric) -- no staging
//   3. cooperative LDL^T of P^- in LDS with Eigen's diagonal pivoting (the reference calls .ldlt(); same pivot order
//      keeps parity at 1e-15 instead of cond(P^-) * eps)
//   4. lane r solves for its column: x = P^T L^-T D^-1 L^-1 P T[:,r] = row r of the gain L; published to LDS
//   5. u = L_row D, P^s_row = P_row + u L^T  -> packed lower triangle to HBM          6. state / quaternion update
// fp64 MFMA brings nothing here on MI355X (its f64 matrix rate equals the vector rate), so this is VALU + LDS.
#pragma once

#include <hip/hip_runtime.h>

#include "rbis_device.hpp"

namespace pb {

template <int NS>
struct SmoothCfg {
  static constexpr int G = (NS == 21) ? 32 : 16;  // lanes per filter
  static constexpr int F = 4;                     // filters per workgroup
  static constexpr int THREADS = G * F;
  static constexpr int MAT = NS * NS;             // doubles per full LDS matrix
  static constexpr int NPK = NS * (NS + 1) / 2;
  // LDS per filter: A (P^- -> LDL^T), XB (solution X = L^T), DM (D, full so that a run-time column index is cheap),
  // residual, dx, pivot list
  static constexpr int PER_FILTER = 3 * MAT + 3 * NS + 1 + ((3 * MAT + 3 * NS + 1) & 1);
};

// packed index with a runtime row (lane)
__device__ __forceinline__ int pk_rt(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// Orders this wave's LDS traffic: the hardware executes one wave's DS instructions in issue order, so all that is needed
// is that the compiler keeps them in program order across this point.
__device__ __forceinline__ void group_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int NS>
__global__ __launch_bounds__(SmoothCfg<NS>::THREADS) void k_smooth_step(const double *__restrict__ next_pred,
                                                                           const double *__restrict__ next_sm,
                                                                           const double *__restrict__ cur,
                                                                           double *__restrict__ out, long stride, int B,
                                                                           double dt, Consts k)
{
  using L = Lay<NS>;
  using C = SmoothCfg<NS>;
  constexpr int G = C::G, MAT = C::MAT;
  extern __shared__ double lds[];
  const int g = threadIdx.x / G;          // filter slot inside the workgroup
  const int r = threadIdx.x % G;          // matrix row owned by this lane
  const long b = (long) blockIdx.x * C::F + g;
  const bool live = b < B;                // groups past the batch end shadow the last filter and never store
  const long bb = live ? b : (B - 1);
  const bool row = r < NS;                // lanes beyond the matrix idle through the row work
  const int rr = row ? r : 0;
  double *A = lds + (size_t) g * C::PER_FILTER;
  double *XB = A + MAT, *DM = XB + MAT, *RV = DM + MAT, *DX = RV + NS;
  int *perm = reinterpret_cast<int *>(DX + NS);  // n ints in the last NS doubles

  auto ldP = [&](const double *s, int i, int j) { return s[(long) (L::OFF_P + pk_rt(i, j)) * stride + bb]; };

  // ---- 1. operands: own row of P_k in registers; P^- row and packed D row -> LDS ----
  double prow[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    prow[j] = ldP(cur, rr, j);
    const double pp = ldP(next_pred, rr, j), ps = ldP(next_sm, rr, j);
    if (row) {
      A[rr * NS + j] = pp;
      DM[rr * NS + j] = ps - pp;
    }
  }
  double w[3], v[3], q[4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    w[i] = cur[(long) (L::OFF_VEC + i) * stride + bb];
    v[i] = cur[(long) (L::OFF_VEC + 3 + i) * stride + bb];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = cur[(long) (L::OFF_QUAT + i) * stride + bb];
  // residual x^s_{k+1} (-) x^-_{k+1}: vec difference, chi = Log(q^-^-1 q^s)   (rbis.cpp:259-261)
  {
    double qs[4], qp[4], dchi[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      qs[i] = next_sm[(long) (L::OFF_QUAT + i) * stride + bb];
      qp[i] = next_pred[(long) (L::OFF_QUAT + i) * stride + bb];
    }
    subtract_quats(qs, qp, dchi);
    double res = next_sm[(long) (L::OFF_VEC + rr) * stride + bb] - next_pred[(long) (L::OFF_VEC + rr) * stride + bb];
    if (rr >= 6 && rr <= 8) res = (rr == 6) ? dchi[0] : (rr == 7 ? dchi[1] : dchi[2]);
    if (row) RV[rr] = res;
  }
  group_sync();
  // bias-block fix (rbis.cpp:244-251): P^- bias-bias block <- I when any of its variances is < 1e-11
  if constexpr (NS == 21) {
    bool fix_g = false, fix_a = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      fix_g = fix_g || (A[(15 + i) * NS + 15 + i] < .00000000001);
      fix_a = fix_a || (A[(18 + i) * NS + 18 + i] < .00000000001);
    }
    group_sync();
    if (row && fix_g && rr >= 15 && rr < 18)
      for (int j = 0; j < 3; j++) A[rr * NS + 15 + j] = (rr - 15 == j) ? 1.0 : 0.0;
    if (row && fix_a && rr >= 18 && rr < 21)
      for (int j = 0; j < 3; j++) A[rr * NS + 18 + j] = (rr - 18 == j) ? 1.0 : 0.0;
    group_sync();
  }

  // ---- 2. x = column r of T = Ad P_k = Ad * (row r of P_k)^T, Ad = I + dt Ac (rbis.cpp:12-35), registers only ----
  double x[NS];
  {
    double R[9];
    quat_to_rot(q, R);
    const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };
    const double pv[3] = { prow[3], prow[4], prow[5] }, pc[3] = { prow[6], prow[7], prow[8] };
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = prow[i];
    // v rows: -w x p_v + g_b x p_chi [- v x p_bg - p_ba];  chi rows: -w x p_chi [- p_bg];  Delta rows: R p_v - R (v x p_chi)
    const double wxpv[3] = { w[1] * pv[2] - w[2] * pv[1], w[2] * pv[0] - w[0] * pv[2], w[0] * pv[1] - w[1] * pv[0] };
    const double gxpc[3] = { gb[1] * pc[2] - gb[2] * pc[1], gb[2] * pc[0] - gb[0] * pc[2], gb[0] * pc[1] - gb[1] * pc[0] };
    const double wxpc[3] = { w[1] * pc[2] - w[2] * pc[1], w[2] * pc[0] - w[0] * pc[2], w[0] * pc[1] - w[1] * pc[0] };
    const double vxpc[3] = { v[1] * pc[2] - v[2] * pc[1], v[2] * pc[0] - v[0] * pc[2], v[0] * pc[1] - v[1] * pc[0] };
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double av = -wxpv[i] + gxpc[i], ac = -wxpc[i];
      if constexpr (NS == 21) {
        const double pbg[3] = { prow[15], prow[16], prow[17] };
        const double vxpbg = (i == 0) ? v[1] * pbg[2] - v[2] * pbg[1] : (i == 1 ? v[2] * pbg[0] - v[0] * pbg[2] : v[0] * pbg[1] - v[1] * pbg[0]);
        av += -vxpbg - prow[18 + i];
        ac += -pbg[i];
      }
      const double ad = R[3 * i] * (pv[0] - vxpc[0]) + R[3 * i + 1] * (pv[1] - vxpc[1]) + R[3 * i + 2] * (pv[2] - vxpc[2]);
      x[3 + i] = fma(dt, av, x[3 + i]);
      x[6 + i] = fma(dt, ac, x[6 + i]);
      x[9 + i] = fma(dt, ad, x[9 + i]);
    }
  }

  // ---- 3. LDL^T of A with diagonal pivoting (Eigen LDLT: largest remaining |A_ii|, first on ties) ----
  for (int kk = 0; kk < NS; kk++) {
    int p = kk;
    double big = fabs(A[kk * NS + kk]);
#pragma unroll 8
    for (int i = kk + 1; i < NS; i++) {
      const double d = fabs(A[i * NS + i]);
      if (d > big) { big = d; p = i; }
    }
    if (r == 0) perm[kk] = p;
    group_sync();
    if (row) {  // column swap: each lane in its own row
      const double t0 = A[rr * NS + kk], t1 = A[rr * NS + p];
      A[rr * NS + kk] = t1;
      A[rr * NS + p] = t0;
    }
    group_sync();
    if (row) {  // row swap: lane j handles column j
      const double t0 = A[kk * NS + rr], t1 = A[p * NS + rr];
      A[kk * NS + rr] = t1;
      A[p * NS + rr] = t0;
    }
    group_sync();
    const double d = A[kk * NS + kk];
    if (row && rr > kk) {
      const double l = (fabs(d) > 0.0) ? A[rr * NS + kk] / d : 0.0;
#pragma unroll 8
      for (int j = kk + 1; j < NS; j++) A[rr * NS + j] = fma(-l, A[kk * NS + j], A[rr * NS + j]);
      A[rr * NS + kk] = l;
    }
    group_sync();
  }

  // ---- 4. solve A y = x for this lane's column: y = P^T L^-T D^-1 L^-1 P x  = row r of the gain ----
  {
    if (row) {  // the permutations index the column at run time: go through this lane's own LDS column
#pragma unroll
      for (int i = 0; i < NS; i++) XB[i * NS + rr] = x[i];
      for (int kk = 0; kk < NS; kk++) {
        const int p = perm[kk];
        const double t0 = XB[kk * NS + rr], t1 = XB[p * NS + rr];
        XB[kk * NS + rr] = t1;
        XB[p * NS + rr] = t0;
      }
    }
    // the substitutions run on the lane's own LDS column with run-time loops (registers stay free for occupancy; the
    // in-order LDS pipe makes a lane's own writes visible to its later reads)
    const int col = rr;
#pragma unroll 1
    for (int i = 1; i < NS; i++) {
      double s = XB[i * NS + col];
#pragma unroll 8
      for (int j = 0; j < i; j++) s = fma(-A[i * NS + j], XB[j * NS + col], s);
      if (row) XB[i * NS + col] = s;
    }
#pragma unroll 1
    for (int i = 0; i < NS; i++) {
      const double d = A[i * NS + i];
      const double s = XB[i * NS + col];
      if (row) XB[i * NS + col] = (fabs(d) > 5.562684646268003e-309) ? s / d : 0.0;  // Eigen: tolerance 1/highest
    }
#pragma unroll 1
    for (int i = NS - 2; i >= 0; i--) {
      double s = XB[i * NS + col];
#pragma unroll 8
      for (int j = i + 1; j < NS; j++) s = fma(-A[j * NS + i], XB[j * NS + col], s);
      if (row) XB[i * NS + col] = s;
    }
    if (row) {
      for (int kk = NS - 1; kk >= 0; kk--) {  // P^T
        const int p = perm[kk];
        const double t0 = XB[kk * NS + rr], t1 = XB[p * NS + rr];
        XB[kk * NS + rr] = t1;
        XB[p * NS + rr] = t0;
      }
    }
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = XB[i * NS + rr];  // x = L[r][:]   (X[i][r] = L[r][i])
  }
  group_sync();  // XB now holds X = L^T of every lane:  L[m][a] = XB[a * NS + m]

  // ---- 5. P^s_row = P_row + (L_row D) L^T ----
  {
    // one pass over b (a run-time loop: small code, two live register rows): u_b = L_row . D[:,b], then P_row += u_b L[:,b]^T
#pragma unroll 1
    for (int bcol = 0; bcol < NS; bcol++) {
      double ub = 0.0;
#pragma unroll
      for (int a = 0; a < NS; a++) ub = fma(x[a], DM[a * NS + bcol], ub);
#pragma unroll
      for (int m = 0; m < NS; m++) prow[m] = fma(ub, XB[bcol * NS + m], prow[m]);
    }
#pragma unroll
    for (int m = 0; m < NS; m++)
      if (live && row && m <= rr) out[(long) (L::OFF_P + pk_rt(rr, m)) * stride + b] = prow[m];
  }
  // ---- 6. state: dx = L resid; cur.addState(RBIS(dx))  (rbis.cpp:263-265) ----
  {
    double dx = 0.0;
#pragma unroll
    for (int a = 0; a < NS; a++) dx = fma(x[a], RV[a], dx);
    if (row) DX[rr] = dx;
    group_sync();
    const double xr = cur[(long) (L::OFF_VEC + rr) * stride + bb];
    if (live && row && !(rr >= 6 && rr <= 8)) out[(long) (L::OFF_VEC + rr) * stride + b] = xr + dx;
    if (live && r == 0) {
      double dchi[3] = { DX[6], DX[7], DX[8] };
      double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
      fold_chi(dchi, dq, k.chi_tol);  // RBIS(vec) constructor
      double chi[3];
#pragma unroll
      for (int i = 0; i < 3; i++) chi[i] = cur[(long) (L::OFF_VEC + 6 + i) * stride + bb] + dchi[i];
      double qq[4] = { q[0], q[1], q[2], q[3] };
      fold_chi(chi, qq, k.chi_tol);
      double o[4];
      quat_mul(qq, dq, o);
#pragma unroll
      for (int i = 0; i < 3; i++) out[(long) (L::OFF_VEC + 6 + i) * stride + b] = chi[i];
#pragma unroll
      for (int i = 0; i < 4; i++) out[(long) (L::OFF_QUAT + i) * stride + b] = o[i];
      out[(long) L::OFF_LL * stride + b] = cur[(long) L::OFF_LL * stride + bb];
    }
  }
}

}  // namespace pb
