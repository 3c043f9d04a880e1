// rbis_quad_rt.hpp -- the generic indexed (+ orientation) update of a 21-state batch (and, for m = 5, 6, a 15-state one) with a RUN-TIME index list on the
// four-wave mapping (RBISIndexedMeasurement / RBISIndexedPlusOrientationMeasurement::updateFilter,
// rbis_update_interface.cpp:54-107; rbis.cpp:124-227): any indices 0..20 (LegOdoCommon's lin_rot_rate list [3,4,5,0,1,2],
// rbis_legodo_common.cpp:66-67, reaches the angular-velocity states; a free-form pronto_indexed_measurement_t.lcm:3-15 list
// anything), m = 1..6, diagonal / broadcast / full R, skip mask.
//
// It replaces k_update<21,...> (round 1 / 2): that kernel GATHERED the m measured columns with 8-byte run-time-slot loads,
// each touching a whole 1 KiB tile row for 512 useful bytes, and streamed the covariance through one wave in row chunks:
// 54-80 us at 64k filters (0.43 of the roofline for m = 6).  Here the state makes ONE round trip through the four waves that
// own it in the tiled layout (Slots<21>::QROW: wave w owns tile rows [QROW[w], QROW[w+1]) -- the ownership of
// rbis_quad.hpp), 16 bytes per lane and access, and the run-time indices never form a memory address:
//   1. every wave loads its rows; for each measurement k the (wave-uniform) index picks, by a chain of scalar compares over
//      the 21 compile-time candidates, the entries of column P[:, idx_k] the wave holds in ITS registers and puts them into
//      the LDS hand-off area COL[21][m] (plus x[idx_k], and the quaternion for an orientation measurement)     -- barrier 1
//   2. every wave reads S = R + P[idx, idx] out of COL (an LDS address from the index: fine)                 -- barrier 2
//      and factorises it itself (m <= 6: ~100 flops, cheaper than a hand-off);
//   3. the 21 rows of W = P[:, idx] L^-T are formed in place, 6 / 6 / 6 / 3 rows per wave                     -- barrier 3
//   4. every wave downdates and stores its own entries, P_ij -= sum_k (w_i[k] / d_k) w_j[k], reading rows of W from LDS;
//      the wave that owns x[v chi Delta] and the quaternion applies dx like rbisApplyDelta, the others add theirs.
// One generic role body, instantiated per wave with its row range: ownership is a constexpr predicate on the component.
#pragma once

#include "rbis_kernels.hpp"

#if defined(UQ_SKEW) && !defined(PB_EXPERIMENTS)
#error "UQ_SKEW is an experiment knob: it needs -DPB_EXPERIMENTS as well"
#endif

namespace pb {

#if defined(__HIPCC__)
// NS = 21: four waves per tile (rbis_quad.hpp's ownership); NS = 15: the two roles of rbis_coop.hpp.  The 15-state variant is
// used where the one-lane kernel k_update_lane_rt spills (m = 5, 6: the covariance AND 75-90 gathered doubles in one lane).
template <int NS>
struct RtMap {
  static constexpr int NW = (NS == 21) ? 4 : 2;                 // waves per tile
  static constexpr int ROWS_PER_WAVE = (NS + NW - 1) / NW;       // rows of W a wave forms in step 3
  __host__ __device__ static constexpr int row0(int w) { return NS == 21 ? Slots<NS>::QROW[w] : (w == 0 ? 0 : Slots<NS>::ROW_SPLIT); }
  __host__ __device__ static constexpr int row1(int w) { return NS == 21 ? Slots<NS>::QROW[w + 1] : (w == 0 ? Slots<NS>::ROW_SPLIT : Slots<NS>::NROW); }
};
template <int NS, int M>
struct QuadRt {
  static constexpr int X_COL = 0, X_XS = NS * M, X_Q = X_XS + M, NXCH = X_Q + 4;
};

template <int NS, int W>
__host__ __device__ constexpr bool quad_owns(int comp)
{
  const int r = Slots<NS>::T.slot_of[comp] >> 1;
  return r >= RtMap<NS>::row0(W) && r < RtMap<NS>::row1(W);
}

// does wave W hold any entry P(i, j <= i) of row i?
template <int NS, int W>
__host__ __device__ constexpr bool quad_owns_row(int i)
{
  for (int j = 0; j <= i; j++)
    if (quad_owns<NS, W>(Lay<NS>::OFF_P + pk(i, j))) return true;
  return false;
}

// the entries of column I of P (and x[I]) that wave W holds, into COL[.][KK] -- one body per candidate I
template <int NS, int W, int M, int KK, int I, class IO, class XW>
__device__ __forceinline__ void quad_rt_pick(int idxk, IO &io, XW &&xw)
{
  using L = Lay<NS>;
  if constexpr (I < NS) {
    if (idxk == I) {
      // (a distinct marker per branch keeps the compiler from merging the bodies into one that reads through a selected
      // address, which would pin the wave's rows in scratch memory: rbis_kernels.hpp pick_column)
      asm volatile("; wave %0 column %1 -> %2" ::"n"(W), "n"(I), "n"(KK));
      static_for<NS>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (quad_owns<NS, W>(L::OFF_P + pk(I, j))) xw(QuadRt<NS, M>::X_COL + j * M + KK, io.ld(L::OFF_P + pk(I, j)));
      });
      if constexpr (quad_owns<NS, W>(L::OFF_VEC + I)) xw(QuadRt<NS, M>::X_XS + KK, io.ld(L::OFF_VEC + I));
      asm volatile("; wave %0 column %1 -> %2 done" ::"n"(W), "n"(I), "n"(KK));
    } else {
      quad_rt_pick<NS, W, M, KK, I + 1>(idxk, io, xw);
    }
  }
}

template <int NS, int W, int M, int MH>
__device__ __forceinline__ void quad_rt_role(const double *st, double *sto, int B, const IdxArg<M> &idx, const double *__restrict__ z,
                                             const double *__restrict__ R, int rkind, const DiagArg<M> &rb,
                                             const double *__restrict__ qmeas, const uint8_t *__restrict__ mask, const Consts &k,
                                             double (*xch)[64])
{
  using L = Lay<NS>;
  using Q = QuadRt<NS, M>;
  using RM = RtMap<NS>;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  const bool upd = (b < (unsigned) B) && (mask == nullptr || mask[b < (unsigned) B ? b : 0] != 0);  // 0 = handler returned NULL for this filter
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, lane);
  auto xw = [&](int s, double v) { xch[s][lane] = v; };
  auto xr = [&](int s) { return xch[s][lane]; };
  // the measurement first (never cache-resident), then this wave's rows
  const rsrc_t rz = mkbuf(z, (unsigned) M * B8);
  const rsrc_t rR = mkbuf(R, rkind == PB_R_DIAG ? (unsigned) M * B8 : (rkind == PB_R_FULL ? (unsigned) (M * M) * B8 : 0u));
  const bool ORIENT = qmeas != nullptr;  // (wave-uniform; a template parameter would double the 18 instances of this kernel)
  const rsrc_t rq = mkbuf(qmeas, ORIENT ? 4u * B8 : 0u);
  double zz[M], Sm[M * (M + 1) / 2], qm[4] = { 1.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int i = 0; i < M; i++) {
    zz[i] = ldg(rz, i * B8, bo);
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double r;
      if (rkind == PB_R_DIAG_BROADCAST) r = (i == j) ? rb.v[i] : 0.0;
      else if (rkind == PB_R_DIAG) r = (i == j) ? ldg(rR, i * B8, bo) : 0.0;
      else r = ldg(rR, (j * M + i) * B8, bo);
      Sm[pk(i, j)] = upd ? r : (i == j ? 1.0 : 0.0);  // benign R for skipped filters (their R block may hold anything)
    }
  }
  if (ORIENT) {
#pragma unroll
    for (int i = 0; i < 4; i++) qm[i] = ldg(rq, i * B8, bo);
  }
  io.template need<RM::row0(W), RM::row1(W)>();

  // ---- 1. the measured columns, out of this wave's registers ----
  static_for<M>([&](auto KK) { quad_rt_pick<NS, W, M, decltype(KK)::value, 0>(idx.v[decltype(KK)::value], io, xw); });
  if constexpr (quad_owns<NS, W>(L::OFF_QUAT)) {
#pragma unroll
    for (int i = 0; i < 4; i++) xw(Q::X_Q + i, io.ld(L::OFF_QUAT + i));
  }
  __syncthreads();

  // ---- 2. S = R + P[idx, idx], residual; every wave factorises for itself ----
  double resid[M], dq[3] = { 0.0, 0.0, 0.0 };
  if (ORIENT) {
    const double qc[4] = { xr(Q::X_Q), xr(Q::X_Q + 1), xr(Q::X_Q + 2), xr(Q::X_Q + 3) };
    subtract_quats(qm, qc, dq);  // rbis.cpp:199-205
  }
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    const int ii = idx.v[kk];
    double r = zz[kk] - xr(Q::X_XS + kk);  // rbis.cpp:170
    if (ORIENT && ii >= 6 && ii <= 8) r = (ii == 6) ? dq[0] : (ii == 7 ? dq[1] : dq[2]);  // rbis.cpp:206-208
    resid[kk] = upd ? r : 0.0;
#pragma unroll
    for (int j = 0; j <= kk; j++) Sm[pk(kk, j)] += xch[Q::X_COL + ii * M + j][lane];  // P[idx_kk, idx_j]: an LDS address from the index
  }
  __syncthreads();  // every wave has read the raw columns it needs for S: they may be overwritten by W now
  double d[M], y[M], id[M], yd[M], quad = 0.0, det = 1.0;
  ldlt<M>(Sm, d);
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s = resid[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s -= Sm[pk(kk, j)] * y[j];
    y[kk] = s;
    id[kk] = upd ? 1.0 / d[kk] : 0.0;
    yd[kk] = s * id[kk];
    det *= d[kk];
    quad += s * s * id[kk];
  }

  // ---- 3. W = P[:, idx] L^-T in place, rows 6 W .. 6 W + 5 ----
  static_for<RM::ROWS_PER_WAVE>([&](auto RR) {
    constexpr int j = RM::ROWS_PER_WAVE * W + decltype(RR)::value;
    if constexpr (j < NS) {
      double w[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) {
        double s = xr(Q::X_COL + j * M + kk);
#pragma unroll
        for (int jj = 0; jj < kk; jj++) s -= w[jj] * Sm[pk(kk, jj)];
        w[kk] = s;
      }
#pragma unroll
      for (int kk = 0; kk < M; kk++) xw(Q::X_COL + j * M + kk, w[kk]);
    }
  });
  __syncthreads();

  // ---- 4. downdate and store this wave's entries; dx for its states ----
  // (W is read back row by row; the clobber keeps the compiler from sharing every read between the rows, i.e. from pulling
  //  all of W into registers)
  static_for<NS>([&](auto II) {
    constexpr int i = decltype(II)::value;
    if constexpr (quad_owns_row<NS, W>(i)) {
      reload_fence();
      double wd[M];
#pragma unroll
      for (int kk = 0; kk < M; kk++) wd[kk] = xr(Q::X_COL + i * M + kk) * id[kk];
      static_for<i + 1>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (quad_owns<NS, W>(L::OFF_P + pk(i, j))) {
          double acc = io.ld(L::OFF_P + pk(i, j));
#pragma unroll
          for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], xr(Q::X_COL + j * M + kk), acc);
          io.st(L::OFF_P + pk(i, j), acc);
        }
      });
    }
  });
  // dx = K r = W D^-1 y for the states this wave owns
  auto dxi = [&](int i) {
    double s = 0.0;
#pragma unroll
    for (int kk = 0; kk < M; kk++) s = (kk == 0) ? xr(Q::X_COL + i * M) * yd[0] : fma(xr(Q::X_COL + i * M + kk), yd[kk], s);
    return s;
  };
  if constexpr (quad_owns<NS, W>(L::OFF_QUAT)) {
    // the wave with x[v chi Delta] and the quaternion: rbisApplyDelta (RigidBodyState::addState, see add_delta) on its part
    double x[NS], dfull[NS], q[4];
#pragma unroll
    for (int i = 0; i < NS; i++) {
      x[i] = 0.0;
      dfull[i] = 0.0;
    }
    static_for<NS>([&](auto II) {
      constexpr int i = decltype(II)::value;
      if constexpr (quad_owns<NS, W>(L::OFF_VEC + i)) { x[i] = io.ld(L::OFF_VEC + i); dfull[i] = dxi(i); }
    });
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = io.ld(L::OFF_QUAT + i);
    if (upd) add_delta<NS>(x, q, dfull, k.chi_tol);
    static_for<NS>([&](auto II) {
      constexpr int i = decltype(II)::value;
      if constexpr (quad_owns<NS, W>(L::OFF_VEC + i)) io.st(L::OFF_VEC + i, x[i]);
    });
#pragma unroll
    for (int i = 0; i < 4; i++) io.st(L::OFF_QUAT + i, q[i]);
  } else {
    static_for<NS>([&](auto II) {
      constexpr int i = decltype(II)::value;
      if constexpr (quad_owns<NS, W>(L::OFF_VEC + i)) {
        static_assert(i < 6 || i > 8, "chi lives with the quaternion");
        const double xi = io.ld(L::OFF_VEC + i);
        io.st(L::OFF_VEC + i, upd ? xi + dxi(i) : xi);
      }
    });
  }
  if constexpr (quad_owns<NS, W>(L::OFF_LL)) {
    double ll = io.ld(L::OFF_LL);
    if (upd) ll += -log(det) - quad;  // -log(S.determinant()) - r^T S^-1 r (rbis.cpp:142)
    io.st(L::OFF_LL, ll);
  }
}

template <int M, int MH = MH_DEFAULT>
__global__ __launch_bounds__(256, 2) void k_update_quad_rt(const double *st, double *sto, int B, IdxArg<M> idx,
                                                           const double *__restrict__ z, const double *__restrict__ R, int rkind,
                                                           DiagArg<M> rb, const double *__restrict__ qmeas,
                                                           const uint8_t *__restrict__ mask, Consts k)
{
  __shared__ double xch[QuadRt<21, M>::NXCH][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  if (role == 0) quad_rt_role<21, 0, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else if (role == 1) quad_rt_role<21, 1, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else if (role == 2) quad_rt_role<21, 2, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else quad_rt_role<21, 3, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
}

// The same kernel with the index list fixed at compile time (LegOdoCommon's lin_rot_rate list [3,4,5,0,1,2],
// rbis_legodo_common.cpp:66-67: the only handler list that reaches the angular-velocity states): the column pick and every
// LDS address formed from an index fold to constants.
template <int MH, int... I>
__global__ __launch_bounds__(256, 2) void k_update_quad_list(const double *st, double *sto, int B, const double *__restrict__ z,
                                                             const double *__restrict__ R, int rkind, DiagArg<sizeof...(I)> rb,
                                                             const double *__restrict__ qmeas, const uint8_t *__restrict__ mask, Consts k)
{
  constexpr int M = (int) sizeof...(I);
  __shared__ double xch[QuadRt<21, M>::NXCH][64];
#ifdef UQ_SKEW  // experiment (needs -DPB_EXPERIMENTS): the second workgroup of every CU in the first dispatch round starts UQ_SKEW x 0.85 us late
  if (blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < UQ_SKEW; i++) __builtin_amdgcn_s_sleep(32);
#endif
  const IdxArg<M> idx = { { I... } };
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  if (role == 0) quad_rt_role<21, 0, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else if (role == 1) quad_rt_role<21, 1, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else if (role == 2) quad_rt_role<21, 2, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else quad_rt_role<21, 3, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
}

// the same for 15 states on the two roles of rbis_coop.hpp (role C: x[v chi Delta], quat, P_cc; role P: the panels, P_pp,
// loglik, x[omega accel]): run-time lists of five and six indices, where the one-lane kernel spills
template <int M, int MH = MH_DEFAULT>
__global__ __launch_bounds__(128, 2) void k_update_coop_rt(const double *st, double *sto, int B, IdxArg<M> idx,
                                                           const double *__restrict__ z, const double *__restrict__ R, int rkind,
                                                           DiagArg<M> rb, const double *__restrict__ qmeas,
                                                           const uint8_t *__restrict__ mask, Consts k)
{
  __shared__ double xch[QuadRt<15, M>::NXCH][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  if (role == 0) quad_rt_role<15, 0, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
  else quad_rt_role<15, 1, M, MH>(st, sto, B, idx, z, R, rkind, rb, qmeas, mask, k, xch);
}
#endif

}  // namespace pb
