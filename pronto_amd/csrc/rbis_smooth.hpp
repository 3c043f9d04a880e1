// rbis_smooth.hpp -- RTS smoother step (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) on gfx950.
//
//   L      = P_k Ad^T (P^-_{k+1})^-1            (Ad about the filtered state at k; bias blocks of P^- replaced by I
//   P^s_k  = P_k + L (P^s_{k+1} - P^-_{k+1}) L^T    when their variance is < 1e-11, rbis.cpp:244-251)
//   x^s_k  = x_k (+) L (x^s_{k+1} (-) x^-_{k+1})
//
// Unlike the forward step this IS dense n x n work (an SPD solve with n right-hand sides and two dense products,
// ~40 kflop per 21-state filter against ~7.5 KB of state: ~5 flop/B, at the fp64 balance point), so one lane per
// filter is the wrong shape.  Mapping: a GROUP of G = 16 (n=15) or 32 (n=21) lanes owns one filter, lane r owns matrix
// row r; 4 filters per workgroup; the operand matrices live in LDS (row stride n doubles: conflict-free for both the
// broadcast reads and the row-per-lane reads), results accumulate in registers.  Steps (one workgroup barrier each):
//   1. rows of P_k, P^-_{k+1} (corrected), D = P^s_{k+1} - P^-_{k+1} -> LDS          2. T = Ad P_k (dense n^2 FMA / lane)
//   3. cooperative LDL^T of P^- with Eigen's diagonal pivoting (largest remaining |diagonal|, as the reference's
//      .ldlt())                                                                    4. lane j solves column j: row j of L
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
  static constexpr int MAT = NS * NS;             // doubles per LDS matrix
  // LDS per filter: A (P^- -> LDL^T), PK (P_k), TB (T -> solution X = L^T), DM (D), residual, dx, pivot list
  static constexpr int PER_FILTER = 4 * MAT + 3 * NS + 1;
};

// packed index with a runtime row (lane) and a compile-time or runtime column
__device__ __forceinline__ int pk_rt(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

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
  const bool live = b < B;                // whole group dead past the batch end (still joins every barrier)
  const long bb = live ? b : (B - 1);     // dead groups shadow the last filter and never store
  const bool row = r < NS;                // lanes beyond the matrix idle through the row work
  const int rr = row ? r : 0;
  double *A = lds + (size_t) g * C::PER_FILTER;
  double *PK = A + MAT, *TB = PK + MAT, *DM = TB + MAT, *RV = DM + MAT, *DX = RV + NS, *SC = DX + NS;
  int *perm = reinterpret_cast<int *>(SC);  // n ints in the last NS doubles

  auto ldP = [&](const double *s, int i, int j) { return s[(long) (L::OFF_P + pk_rt(i, j)) * stride + bb]; };

  // ---- 1. operands -> LDS; own row of P_k also stays in registers ----
  double prow[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    prow[j] = ldP(cur, rr, j);
    const double pp = ldP(next_pred, rr, j), ps = ldP(next_sm, rr, j);
    if (row) {
      PK[rr * NS + j] = prow[j];
      A[rr * NS + j] = pp;
      DM[rr * NS + j] = ps - pp;
    }
  }
  // filtered state at k (for Ad) -- every lane needs omega, v, quat
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
  __syncthreads();
  // bias-block fix (rbis.cpp:244-251): P^- bias-bias block <- I when any of its variances is < 1e-11
  if constexpr (NS == 21) {
    bool fix_g = false, fix_a = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      fix_g = fix_g || (A[(15 + i) * NS + 15 + i] < .00000000001);
      fix_a = fix_a || (A[(18 + i) * NS + 18 + i] < .00000000001);
    }
    __syncthreads();
    if (row && fix_g && rr >= 15 && rr < 18)
      for (int j = 0; j < 3; j++) A[rr * NS + 15 + j] = (rr - 15 == j) ? 1.0 : 0.0;
    if (row && fix_a && rr >= 18 && rr < 21)
      for (int j = 0; j < 3; j++) A[rr * NS + 18 + j] = (rr - 18 == j) ? 1.0 : 0.0;
  }

  // ---- 2. T = Ad P_k : row r of Ad = e_r + dt * (row r of Ac, rbis.cpp:12-35) ----
  double arow[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) arow[j] = (j == rr) ? 1.0 : 0.0;
  {
    double R[9];
    quat_to_rot(q, R);
    const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };
    const int blk = rr / 3, i = rr % 3;
    // hat(a)[i][j]
    auto hat = [](const double (&a)[3], int ii, int jj) {
      if (ii == jj) return 0.0;
      if (ii == 0) return jj == 1 ? -a[2] : a[1];
      if (ii == 1) return jj == 0 ? a[2] : -a[0];
      return jj == 0 ? -a[1] : a[0];
    };
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (blk == 1) {  // v rows: [v,v] = -what, [v,chi] = hat(R^T g), [v,bg] = -vhat, [v,ba] = -I
        arow[3 + j] += dt * -hat(w, i, j);
        arow[6 + j] += dt * hat(gb, i, j);
        if constexpr (NS == 21) {
          arow[15 + j] += dt * -hat(v, i, j);
          arow[18 + j] += (i == j) ? -dt : 0.0;
        }
      } else if (blk == 2) {  // chi rows: [chi,chi] = -what, [chi,bg] = -I
        arow[6 + j] += dt * -hat(w, i, j);
        if constexpr (NS == 21) arow[15 + j] += (i == j) ? -dt : 0.0;
      } else if (blk == 3) {  // Delta rows: [Delta,v] = R, [Delta,chi] = -R vhat
        arow[3 + j] += dt * R[3 * i + j];
        double rv = 0.0;
#pragma unroll
        for (int m = 0; m < 3; m++) rv += R[3 * i + m] * hat(v, m, j);
        arow[6 + j] += dt * -rv;
      }
    }
  }
  __syncthreads();
  {
    double trow[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) trow[j] = 0.0;
#pragma unroll
    for (int kk = 0; kk < NS; kk++) {
      const double a = arow[kk];
#pragma unroll
      for (int j = 0; j < NS; j++) trow[j] = fma(a, PK[kk * NS + j], trow[j]);
    }
    if (row)
#pragma unroll
      for (int j = 0; j < NS; j++) TB[rr * NS + j] = trow[j];
  }
  __syncthreads();

  // ---- 3. LDL^T of A with diagonal pivoting (Eigen LDLT: largest remaining |A_ii|, first on ties) ----
  for (int kk = 0; kk < NS; kk++) {
    int p = kk;
    double big = fabs(A[kk * NS + kk]);
    for (int i = kk + 1; i < NS; i++) {
      const double d = fabs(A[i * NS + i]);
      if (d > big) { big = d; p = i; }
    }
    if (r == 0) perm[kk] = p;
    __syncthreads();
    if (row) {  // column swap: each lane in its own row
      const double t0 = A[rr * NS + kk], t1 = A[rr * NS + p];
      A[rr * NS + kk] = t1;
      A[rr * NS + p] = t0;
    }
    __syncthreads();
    if (row) {  // row swap: lane j handles column j
      const double t0 = A[kk * NS + rr], t1 = A[p * NS + rr];
      A[kk * NS + rr] = t1;
      A[p * NS + rr] = t0;
    }
    __syncthreads();
    const double d = A[kk * NS + kk];
    if (row && rr > kk) {
      const double l = (fabs(d) > 0.0) ? A[rr * NS + kk] / d : 0.0;
      for (int j = kk + 1; j < NS; j++) A[rr * NS + j] = fma(-l, A[kk * NS + j], A[rr * NS + j]);
      A[rr * NS + kk] = l;
    }
    __syncthreads();
  }

  // ---- 4. lane j solves A x = T[:, j]  (x = P^T L^-T D^-1 L^-1 P b); x is row j of the gain L ----
  double lg[NS];
  {
    if (row) {  // P b : permute this lane's own column of T in LDS (no other lane touches it)
      for (int kk = 0; kk < NS; kk++) {
        const int p = perm[kk];
        const double t0 = TB[kk * NS + rr], t1 = TB[p * NS + rr];
        TB[kk * NS + rr] = t1;
        TB[p * NS + rr] = t0;
      }
    }
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = TB[i * NS + rr];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
      for (int j = 0; j < i; j++) x[i] = fma(-A[i * NS + j], x[j], x[i]);
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const double d = A[i * NS + i];
      x[i] = (fabs(d) > 5.562684646268003e-309) ? x[i] / d : 0.0;  // Eigen: tolerance 1/highest
    }
#pragma unroll
    for (int i = NS - 1; i >= 0; i--)
#pragma unroll
      for (int j = i + 1; j < NS; j++) x[i] = fma(-A[j * NS + i], x[j], x[i]);
    if (row) {
#pragma unroll
      for (int i = 0; i < NS; i++) TB[i * NS + rr] = x[i];
      for (int kk = NS - 1; kk >= 0; kk--) {  // P^T
        const int p = perm[kk];
        const double t0 = TB[kk * NS + rr], t1 = TB[p * NS + rr];
        TB[kk * NS + rr] = t1;
        TB[p * NS + rr] = t0;
      }
    }
#pragma unroll
    for (int i = 0; i < NS; i++) lg[i] = TB[i * NS + rr];  // L[r][i] = X[i][r]
  }
  __syncthreads();  // TB now holds X = L^T for every lane:  L[m][a] = TB[a * NS + m]

  // ---- 5. P^s_row = P_row + (L_row D) L^T ----
  {
    double u[NS];
#pragma unroll
    for (int bcol = 0; bcol < NS; bcol++) u[bcol] = 0.0;
#pragma unroll
    for (int a = 0; a < NS; a++) {
      const double la = lg[a];
#pragma unroll
      for (int bcol = 0; bcol < NS; bcol++) u[bcol] = fma(la, DM[a * NS + bcol], u[bcol]);
    }
#pragma unroll
    for (int m = 0; m < NS; m++) {
      double acc = prow[m];
#pragma unroll
      for (int bcol = 0; bcol < NS; bcol++) acc = fma(u[bcol], TB[bcol * NS + m], acc);
      if (live && row && m <= rr) out[(long) (L::OFF_P + pk_rt(rr, m)) * stride + b] = acc;
    }
  }
  // ---- 6. state: dx = L resid; cur.addState(RBIS(dx))  (rbis.cpp:263-265) ----
  {
    double dx = 0.0;
#pragma unroll
    for (int a = 0; a < NS; a++) dx = fma(lg[a], RV[a], dx);
    if (row) DX[rr] = dx;
    __syncthreads();
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
