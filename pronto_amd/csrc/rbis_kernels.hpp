// rbis_kernels.hpp -- gfx950 kernels of the batched RBIS EKF (included by pronto_batch.hip only).
//
// Data layout in HBM: st[NC][stride] doubles per context, component-major with the filter index fastest
// (stride = batch rounded up to 64); components = vec[n] | quat[4] | loglik | P packed lower [n(n+1)/2].
// One lane owns one filter, so every global access of a wave is one fully coalesced 512-byte row segment.
//
// Addressing: one 128-bit buffer descriptor per array, the component offset in an SGPR (soffset) and ONE per-lane
// 32-bit byte offset shared by every access: `buffer_load/store_dwordx2 v, v_off, s[rsrc], s_off offen`.  No per-access
// 64-bit VGPR address is materialised (flat addressing cost ~700 instructions, ~60 VGPRs and 164 spilled registers in the
// first version of k_step).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pronto_batch.h"
#include "rbis_coop.hpp"
#include "rbis_device.hpp"

// workgroup size of the hot kernel (one wave per SIMD either way; 64 = one wave per workgroup)
#ifndef PB_STEP_BLOCK
#define PB_STEP_BLOCK 64
#endif

namespace pb {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// 128-bit buffer descriptor over [p, p+bytes): out-of-range lanes read 0 / drop their stores (hardware check)
__device__ __forceinline__ rsrc_t mkbuf(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
// buffer_load_dwordx2 v, v_off, s[rsrc], s_off offen : voff = per-lane byte offset, soff = uniform component offset
// AUX = cache-policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ double ldg(rsrc_t r, unsigned soff, unsigned voff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void stg(rsrc_t r, unsigned soff, unsigned voff, double v)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, AUX);
}

// Memory hint of the state round trip in the step kernels (template parameter MH), picked by the host from the state
// size.  Measured with an in-place streaming copy of the same layout: while the state fits the 256 MB
// memory-side cache, stores with sc1 (write through the XCD's L2) are 3-5 % faster; far beyond it, non-temporal loads
// AND stores are ~5 % faster; in between neither helps.
enum { MH_DEFAULT = 0, MH_STORE_SC1 = 1, MH_STREAM_NT = 2 };
template <int MH> struct MemHint {
  static constexpr int LA = (MH == MH_STREAM_NT) ? 2 : 0;
  static constexpr int SA = (MH == MH_STORE_SC1) ? 16 : (MH == MH_STREAM_NT) ? 2 : 0;
};

// Workgroups are dealt round-robin to the 8 XCDs.  With k.xcd_remap each XCD walks one contiguous filter range
// (bijective for any grid size) instead of every 8th 512-byte segment of each component row (host picks).
__device__ __forceinline__ unsigned xcd_workgroup(const Consts &k)
{
  unsigned wg = blockIdx.x;
  if (k.xcd_remap) {
    const unsigned nq = gridDim.x >> 3, nr = gridDim.x & 7u, xcd = blockIdx.x & 7u, rank = blockIdx.x >> 3;
    wg = (xcd < nr ? xcd * (nq + 1u) : nr * (nq + 1u) + (xcd - nr) * nq) + rank;
  }
  return wg;
}

struct IdxVel {
  static constexpr Idx<3> value = { { 3, 4, 5 } };
};

// RBISIMUProcessStep::updateFilter [+ RBISIndexedMeasurement::updateFilter with idx = {3,4,5}, diagonal R]
// (rbis_update_interface.cpp:30-52, :54-95).  The BASELINE hot step: 2*(n+4+1+n(n+1)/2)*8 + 56 + 48 bytes/filter.
template <int NS, bool UPDATE, int MH = MH_DEFAULT>
__global__ __launch_bounds__(PB_STEP_BLOCK, 1) void k_step(const double *st, double *sto, long stride, int B,
                                                const double *__restrict__ imu, const double *__restrict__ lo,
                                                const uint8_t *__restrict__ mask, double qg, double qa, double qbg,
                                                double qba, Consts k)
{
  using L = Lay<NS>;
  constexpr int LA = MemHint<MH>::LA, SA = MemHint<MH>::SA;
  const unsigned b = xcd_workgroup(k) * blockDim.x + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u;
  const unsigned s8 = (unsigned) stride * 8u, B8 = (unsigned) B * 8u;  // host guarantees NC*stride*8 < 2^32
  const rsrc_t rs = mkbuf(st, (unsigned) L::NC * s8);
  const rsrc_t ro = mkbuf(sto, (unsigned) L::NC * s8);  // posterior: the same array (in place) or a checkpoint slot
  const rsrc_t ri = mkbuf(imu, 7u * B8);
  const rsrc_t rl = mkbuf(lo, UPDATE ? 6u * B8 : 0u);
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ldg<LA>(rs, (L::OFF_VEC + i) * s8, bo);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ldg<LA>(rs, (L::OFF_QUAT + i) * s8, bo);
  ll = ldg<LA>(rs, L::OFF_LL * s8, bo);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = ldg<LA>(rs, (L::OFF_P + i) * s8, bo);
  double gyro[3], accel[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    gyro[i] = ldg(ri, i * B8, bo);
    accel[i] = ldg(ri, (3 + i) * B8, bo);
  }
  const double dt = ldg(ri, 6u * B8, bo);
  double z[3], rd[3];
  bool upd = false;
  if constexpr (UPDATE) {
    upd = (mask == nullptr) || (mask[b] != 0);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      z[i] = ldg(rl, i * B8, bo);
      rd[i] = ldg(rl, (3 + i) * B8, bo);
    }
  }
  if (k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    qg = ldg(rq, 0u, bo); qa = ldg(rq, B8, bo); qbg = ldg(rq, 2u * B8, bo); qba = ldg(rq, 3u * B8, bo);
  }
  imu_process_step<NS>(x, q, P, gyro, accel, dt, qg, qa, qbg, qba, k);
  if constexpr (UPDATE) {
    // Predicated, not branched: lanes whose handler returned NULL (mask 0) run the same stream with D^-1 = 0 and a
    // benign R, so the wave stores whole rows (see measurement_update).
    double resid[3], S[6];
#pragma unroll
    for (int i = 0; i < 3; i++) resid[i] = upd ? z[i] - x[3 + i] : 0.0;  // rbis.cpp:170
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        S[pk(i, j)] = P[pk(3 + i, 3 + j)] + (i == j ? (upd ? rd[i] : 1.0) : 0.0);  // rbis.cpp:134-135
    measurement_update<NS, 3>(x, q, P, ll, resid, S, IdxVel{}, k,
                              [ro, s8, bo](int pi, double v) { stg<SA>(ro, (L::OFF_P + pi) * s8, bo, v); }, upd);
  } else {
#pragma unroll
    for (int i = 0; i < L::NP; i++) stg<SA>(ro, (L::OFF_P + i) * s8, bo, P[i]);
  }
#pragma unroll
  for (int i = 0; i < NS; i++) stg<SA>(ro, (L::OFF_VEC + i) * s8, bo, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) stg<SA>(ro, (L::OFF_QUAT + i) * s8, bo, q[i]);
  stg<SA>(ro, L::OFF_LL * s8, bo, ll);
}

// Time-fused replay: T consecutive predict+update steps per launch with the state and P resident in registers; only the
// 104 B/filter of inputs stream from HBM per step, the posterior is materialised once per launch.  This is NOT the plugin
// path (MavStateEstimator::addUpdate publishes a posterior per message) but what a parameter sweep or a likelihood
// evaluation over a log segment wants (param_sweep.py:39-52).  Accounting: 104 + 2240/T bytes per filter-step, so the
// bound moves from HBM to fp64 VALU issue; bench.py reports it separately (never as the headline value).
// Inputs of step t+1 are prefetched while step t computes (one wave per SIMD: nothing else hides their latency).
template <int NS>
__global__ __launch_bounds__(64, 1) void k_replay_fused(double *__restrict__ st, long stride, int B, int T,
                                                        const double *__restrict__ imu, const double *__restrict__ lo,
                                                        const uint8_t *__restrict__ mask, double qg, double qa, double qbg,
                                                        double qba, Consts k)
{
  using L = Lay<NS>;
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u;
  const unsigned s8 = (unsigned) stride * 8u, B8 = (unsigned) B * 8u;
  const rsrc_t rs = mkbuf(st, (unsigned) L::NC * s8);
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ldg(rs, (L::OFF_VEC + i) * s8, bo);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ldg(rs, (L::OFF_QUAT + i) * s8, bo);
  ll = ldg(rs, L::OFF_LL * s8, bo);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = ldg(rs, (L::OFF_P + i) * s8, bo);
  if (k.qblk != nullptr) {
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    qg = ldg(rq, 0u, bo); qa = ldg(rq, B8, bo); qbg = ldg(rq, 2u * B8, bo); qba = ldg(rq, 3u * B8, bo);
  }
  // step-t input blocks are [7][B] / [6][B] / [B] slabs of the streams; 64-bit slab base, 32-bit offsets inside
  double in[13], nx[13] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  bool upd, nupd = false;
  auto fetch = [&](int t, double (&dst)[13], bool &u) {
    const rsrc_t ri = mkbuf(imu + (size_t) t * 7 * B, 7u * B8);
    const rsrc_t rl = mkbuf(lo + (size_t) t * 6 * B, 6u * B8);
#pragma unroll
    for (int i = 0; i < 7; i++) dst[i] = ldg(ri, i * B8, bo);
#pragma unroll
    for (int i = 0; i < 6; i++) dst[7 + i] = ldg(rl, i * B8, bo);
    u = (mask == nullptr) || (mask[(size_t) t * B + b] != 0);
  };
  fetch(0, in, upd);
  for (int t = 0; t < T; t++) {
#ifdef PB_REPLAY_PREFETCH
    if (t + 1 < T) fetch(t + 1, nx, nupd);
#else
    if (t > 0) fetch(t, in, upd);
#endif
    const double gyro[3] = { in[0], in[1], in[2] }, accel[3] = { in[3], in[4], in[5] };
    imu_process_step<NS>(x, q, P, gyro, accel, in[6], qg, qa, qbg, qba, k);
    double resid[3], S[6];
#pragma unroll
    for (int i = 0; i < 3; i++) resid[i] = upd ? in[7 + i] - x[3 + i] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) S[pk(i, j)] = P[pk(3 + i, 3 + j)] + (i == j ? (upd ? in[10 + i] : 1.0) : 0.0);
    measurement_update<NS, 3>(x, q, P, ll, resid, S, IdxVel{}, k, NoSink(), upd);
#ifdef PB_REPLAY_PREFETCH
#pragma unroll
    for (int i = 0; i < 13; i++) in[i] = nx[i];
    upd = nupd;
#endif
  }
#pragma unroll
  for (int i = 0; i < L::NP; i++) stg(rs, (L::OFF_P + i) * s8, bo, P[i]);
#pragma unroll
  for (int i = 0; i < NS; i++) stg(rs, (L::OFF_VEC + i) * s8, bo, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) stg(rs, (L::OFF_QUAT + i) * s8, bo, q[i]);
  stg(rs, L::OFF_LL * s8, bo, ll);
}

template <int M>
struct IdxArg {
  int v[M];
};
template <int M>
struct DiagArg {
  double v[M];
};

// rows [R0, R1) of the packed covariance: issue ALL their loads, then downdate and store them.  (Interleaving a load
// and a store per entry serialises on the load latency: the compiler may not hoist a load above a possibly aliasing
// store; that version of k_update ran at 0.33-0.45 of the HBM roofline.)
template <int NS, int M, int R0, int R1, int MH>
__device__ __forceinline__ void downdate_rows(rsrc_t rs, rsrc_t ro, unsigned s8, unsigned bo, const double (&W)[NS][M],
                                              const double (&id)[M])
{
  using L = Lay<NS>;
  constexpr int CNT = (R1 * (R1 + 1) - R0 * (R0 + 1)) / 2;
  constexpr int P0 = R0 * (R0 + 1) / 2;
  double buf[CNT];
#pragma unroll
  for (int e = 0; e < CNT; e++) buf[e] = ldg<MemHint<MH>::LA>(rs, (L::OFF_P + P0 + e) * s8, bo);
#pragma unroll
  for (int i = R0; i < R1; i++) {
    double wd[M];
#pragma unroll
    for (int kk = 0; kk < M; kk++) wd[kk] = W[i][kk] * id[kk];
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double acc = buf[pk(i, j) - P0];
#pragma unroll
      for (int kk = 0; kk < M; kk++) acc = fma(-wd[kk], W[j][kk], acc);
      stg<MemHint<MH>::SA>(ro, (L::OFF_P + pk(i, j)) * s8, bo, acc);
    }
  }
}

// Generic RBISIndexedMeasurement / RBISIndexedPlusOrientationMeasurement::updateFilter with a RUNTIME index list
// (rbis_update_interface.cpp:54-107).  The m gathered columns P[:, idx] (wave-uniform component addresses) and x
// live in registers; P itself is streamed through once in row chunks (load a chunk, rank-m downdate, store it).
// The skip mask is predicated like in k_step: every lane stores whole rows with D^-1 = 0 for skipped filters.
// MH: the gathered columns are read with the default policy (they are read again by the row stream), the row stream's
// loads and every store carry the hint.
template <int NS, int M, bool ORIENT, int MH = MH_DEFAULT>
__global__ __launch_bounds__(64, (NS == 15 && M <= 3) ? 2 : 1) void k_update(const double *st, double *sto, long stride, int B, IdxArg<M> idx,
                                                  const double *__restrict__ z, const double *__restrict__ R,
                                                  int rkind, DiagArg<M> rb, const double *__restrict__ qmeas,
                                                  const uint8_t *__restrict__ mask, Consts k)
{
  using L = Lay<NS>;
  const unsigned b = xcd_workgroup(k) * blockDim.x + threadIdx.x;
  if (b >= (unsigned) B) return;
  const bool upd = (mask == nullptr) || (mask[b] != 0);  // 0 = handler returned NULL for this filter
  const unsigned bo = b * 8u;
  const unsigned s8 = (unsigned) stride * 8u, B8 = (unsigned) B * 8u;
  const rsrc_t rs = mkbuf(st, (unsigned) L::NC * s8);
  const rsrc_t ro = mkbuf(sto, (unsigned) L::NC * s8);
  const rsrc_t rz = mkbuf(z, (unsigned) M * B8);
  const rsrc_t rR = mkbuf(R, rkind == PB_R_DIAG ? (unsigned) M * B8 : (rkind == PB_R_FULL ? (unsigned) (M * M) * B8 : 0u));
  const rsrc_t rq = mkbuf(qmeas, ORIENT ? 4u * B8 : 0u);
  double x[NS], q[4];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = ldg(rs, (L::OFF_VEC + i) * s8, bo);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ldg(rs, (L::OFF_QUAT + i) * s8, bo);
  double ll = ldg(rs, L::OFF_LL * s8, bo);
  // gather the measured columns first: all these loads are in flight together
  double W[NS][M];
#pragma unroll
  for (int i = 0; i < NS; i++)
#pragma unroll
    for (int kk = 0; kk < M; kk++) W[i][kk] = ldg(rs, (L::OFF_P + pk(i, idx.v[kk])) * s8, bo);

  // residual (rbis.cpp:169-172 / :199-208)
  double resid[M];
  double dq[3] = { 0, 0, 0 };
  if constexpr (ORIENT) {
    const double qm[4] = { ldg(rq, 0u, bo), ldg(rq, B8, bo), ldg(rq, 2u * B8, bo), ldg(rq, 3u * B8, bo) };
    subtract_quats(qm, q, dq);
  }
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    const int ii = idx.v[kk];
    const double xi = ldg(rs, (L::OFF_VEC + ii) * s8, bo);  // runtime index: re-read instead of x[ii]
    double r = ldg(rz, kk * B8, bo) - xi;
    if constexpr (ORIENT) {
      if (ii >= 6 && ii <= 8) r = (ii == 6) ? dq[0] : (ii == 7 ? dq[1] : dq[2]);
    }
    resid[kk] = upd ? r : 0.0;
  }
  // S = R + P[idx, idx]
  double S[M * (M + 1) / 2], d[M];
#pragma unroll
  for (int i = 0; i < M; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double r;
      if (rkind == PB_R_DIAG_BROADCAST) r = (i == j) ? rb.v[i] : 0.0;
      else if (rkind == PB_R_DIAG) r = (i == j) ? ldg(rR, i * B8, bo) : 0.0;
      else r = ldg(rR, (j * M + i) * B8, bo);
      if (!upd) r = (i == j) ? 1.0 : 0.0;  // benign R for skipped filters (their R block may hold anything)
      S[pk(i, j)] = r + ldg(rs, (L::OFF_P + pk(idx.v[i], idx.v[j])) * s8, bo);
    }
  ldlt<M>(S, d);
  double y[M], id[M], yd[M], lli = 0.0;
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s = resid[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s -= S[pk(kk, j)] * y[j];
    y[kk] = s;
    id[kk] = upd ? 1.0 / d[kk] : 0.0;
    yd[kk] = s * id[kk];
    lli -= log(d[kk]) + s * s * id[kk];
  }
  if (upd) ll += lli;
  // W = P[:, idx] L^-T  (in place on the gathered columns)
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
  double dx[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    double s = 0.0;
#pragma unroll
    for (int kk = 0; kk < M; kk++) s = (kk == 0) ? W[i][0] * yd[0] : fma(W[i][kk], yd[kk], s);
    dx[i] = s;
  }
  // chunk sizes keep (chunk + W) inside the register file: W is NS x M doubles
  if constexpr (NS == 15 && M <= 3) {
    // small W: three chunks fit 256 registers -> two waves per SIMD (m = 3: 25.7 -> 23.2 us at 64k filters; for m = 4 the
    // same split measured 7 % slower than one chunk at one wave per SIMD)
    downdate_rows<NS, M, 0, 8, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 8, 12, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 12, 15, MH>(rs, ro, s8, bo, W, id);
  } else if constexpr (NS == 15 && M == 4) {
    downdate_rows<NS, M, 0, 15, MH>(rs, ro, s8, bo, W, id);
  } else if constexpr (NS == 15) {
    downdate_rows<NS, M, 0, 11, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 11, 15, MH>(rs, ro, s8, bo, W, id);
  } else if constexpr (M <= 4) {
    downdate_rows<NS, M, 0, 12, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 12, 17, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 17, 21, MH>(rs, ro, s8, bo, W, id);
  } else {
    downdate_rows<NS, M, 0, 9, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 9, 13, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 13, 16, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 16, 19, MH>(rs, ro, s8, bo, W, id);
    downdate_rows<NS, M, 19, 21, MH>(rs, ro, s8, bo, W, id);
  }
  if (upd) add_delta<NS>(x, q, dx, k.chi_tol);
  constexpr int SA = MemHint<MH>::SA;
#pragma unroll
  for (int i = 0; i < NS; i++) stg<SA>(ro, (L::OFF_VEC + i) * s8, bo, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) stg<SA>(ro, (L::OFF_QUAT + i) * s8, bo, q[i]);
  stg<SA>(ro, L::OFF_LL * s8, bo, ll);
}

// RBISResetUpdate::updateFilter, per-filter inputs: vec [n][B], quat [4][B], cov [n*n][B] column-major
template <int NS>
__global__ void k_reset(double *__restrict__ st, long stride, int B, const double *__restrict__ vec,
                        const double *__restrict__ quat, const double *__restrict__ cov)
{
  using L = Lay<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < NS; i++) st[(long) (L::OFF_VEC + i) * stride + b] = vec[(long) i * B + b];
  for (int i = 0; i < 4; i++) st[(long) (L::OFF_QUAT + i) * stride + b] = quat[(long) i * B + b];
  st[(long) L::OFF_LL * stride + b] = 0.0;
  for (int i = 0; i < NS; i++)
    for (int j = 0; j <= i; j++) st[(long) (L::OFF_P + pk(i, j)) * stride + b] = cov[(long) (j * NS + i) * B + b];
}

// broadcast reset: comp [NC] already packed on the host
__global__ void k_reset_bcast(double *__restrict__ st, long stride, int B, int NC, const double *__restrict__ comp)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int c = 0; c < NC; c++) st[(long) c * stride + b] = comp[c];
}

template <int NS>
__global__ void k_get_head(const double *__restrict__ st, long stride, int first, int count, double *vec_out,
                           double *quat_out, double *cov_out, double *ll_out)
{
  using L = Lay<NS>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int b = first + t;
  if (vec_out)
    for (int i = 0; i < NS; i++) vec_out[(long) i * count + t] = st[(long) (L::OFF_VEC + i) * stride + b];
  if (quat_out)
    for (int i = 0; i < 4; i++) quat_out[(long) i * count + t] = st[(long) (L::OFF_QUAT + i) * stride + b];
  if (ll_out) ll_out[t] = st[(long) L::OFF_LL * stride + b];
  if (cov_out)
    for (int c = 0; c < NS; c++)
      for (int r = 0; r < NS; r++)
        cov_out[(long) (c * NS + r) * count + t] = st[(long) (L::OFF_P + pk(r, c)) * stride + b];
}

// (position, quat) of the head posterior -> snapshot slot [7][stride]
template <int NS>
__global__ void k_snapshot(const double *__restrict__ st, long stride, int B, double *__restrict__ snap)
{
  using L = Lay<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < 3; i++) snap[(long) i * stride + b] = st[(long) (L::OFF_VEC + 9 + i) * stride + b];
  for (int i = 0; i < 4; i++) snap[(long) (3 + i) * stride + b] = st[(long) (L::OFF_QUAT + i) * stride + b];
}

// T1 = T0 * (t, q)   (rbis_fovis_update.cpp:219-223)
__global__ void k_compose(const double *__restrict__ snap, long stride, int B, const double *__restrict__ t,
                          const double *__restrict__ q, double *__restrict__ z_out, double *__restrict__ q_out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double p0[3] = { snap[b], snap[stride + b], snap[2 * stride + b] };
  const double q0[4] = { snap[3 * stride + b], snap[4 * stride + b], snap[5 * stride + b], snap[6 * stride + b] };
  const double tt[3] = { t[b], t[(long) B + b], t[2L * B + b] };
  const double qq[4] = { q[b], q[(long) B + b], q[2L * B + b], q[3L * B + b] };
  double R[9], o[4];
  quat_to_rot(q0, R);
  for (int i = 0; i < 3; i++)
    z_out[(long) i * B + b] = p0[i] + (R[3 * i] * tt[0] + R[3 * i + 1] * tt[1] + R[3 * i + 2] * tt[2]);
  quat_mul(q0, qq, o);
  for (int i = 0; i < 4; i++) q_out[(long) i * B + b] = o[i];
}

template <int NS>
__global__ void k_summary(const double *__restrict__ st, long stride, int B, double *__restrict__ out)
{
  using L = Lay<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double s_ll = 0, s_abs = 0, qdev = 0, nonfin = 0;
  if (b < B) {
    s_ll = st[(long) L::OFF_LL * stride + b];
    double qn = 0;
    for (int i = 0; i < NS + 4; i++) {
      const double v = st[(long) i * stride + b];
      s_abs += fabs(v);
      if (!isfinite(v)) nonfin += 1;
      if (i >= NS) qn += v * v;
    }
    for (int i = 0; i < L::NP; i++)
      if (!isfinite(st[(long) (L::OFF_P + i) * stride + b])) nonfin += 1;
    if (!isfinite(s_ll)) nonfin += 1;
    qdev = fabs(qn - 1.0);
  }
  for (int off = 32; off > 0; off >>= 1) {
    s_ll += __shfl_down(s_ll, off);
    s_abs += __shfl_down(s_abs, off);
    nonfin += __shfl_down(nonfin, off);
    qdev = fmax(qdev, __shfl_down(qdev, off));
  }
  // one partial per wave, reduced on the host in wave order: bit-reproducible (float atomics are not)
  if ((threadIdx.x & 63) == 0) {
    double *o = out + 4L * blockIdx.x;
    o[0] = s_ll; o[1] = s_abs; o[2] = qdev; o[3] = nonfin;
  }
}

// Two-wave cooperative step (rbis_coop.hpp): 128-thread workgroups, wave 0 = role C (dynamic core sub-matrix, state,
// quaternion), wave 1 = role P (passive omega/accel panels) for the SAME 64 filters; one LDS hand-off + one barrier.
// This is the 21-state hot kernel (231 packed entries do not fit one lane) and an alternative mapping for n = 15.
// No lane returns before the barrier: lanes past the batch end work on the zero-initialised padding columns of the
// state array (stride is the batch rounded up to 64) and on bounds-checked (zero) inputs.
template <int NS, bool UPDATE, int MH = MH_DEFAULT>
__global__ __launch_bounds__(128, 1) void k_step_coop(const double *st, double *sto, long stride, int B,
                                                      const double *__restrict__ imu, const double *__restrict__ lo,
                                                      const uint8_t *__restrict__ mask, double qg, double qa,
                                                      double qbg, double qba, Consts k)
{
  using L = Lay<NS>;
  using C = Coop<NS>;
  __shared__ double xch[UPDATE ? C::NXCH : 1][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wg = xcd_workgroup(k);
  const unsigned b = wg * 64u + lane;
  const unsigned bo = b * 8u;
  const unsigned s8 = (unsigned) stride * 8u, B8 = (unsigned) B * 8u;
  const rsrc_t rs = mkbuf(st, (unsigned) L::NC * s8);
  const rsrc_t ro = mkbuf(sto, (unsigned) L::NC * s8);
  const rsrc_t ri = mkbuf(imu, 7u * B8);
  const rsrc_t rl = mkbuf(lo, UPDATE ? 6u * B8 : 0u);
  StepInputs in;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    in.gyro[i] = ldg(ri, i * B8, bo);
    in.accel[i] = ldg(ri, (3 + i) * B8, bo);
    in.z[i] = UPDATE ? ldg(rl, i * B8, bo) : 0.0;
    in.rd[i] = UPDATE ? ldg(rl, (3 + i) * B8, bo) : 1.0;
  }
  in.dt = ldg(ri, 6u * B8, bo);
  in.upd = UPDATE && (b < (unsigned) B) && (mask == nullptr || mask[b] != 0);
  in.qg = qg; in.qa = qa; in.qbg = qbg; in.qba = qba;
  if (k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    in.qg = ldg(rq, 0u, bo); in.qa = ldg(rq, B8, bo); in.qbg = ldg(rq, 2u * B8, bo); in.qba = ldg(rq, 3u * B8, bo);
  }
  auto ld = [rs, s8, bo](int comp) { return ldg<MemHint<MH>::LA>(rs, (unsigned) comp * s8, bo); };
  auto stf = [ro, s8, bo](int comp, double v) { stg<MemHint<MH>::SA>(ro, (unsigned) comp * s8, bo, v); };
  auto sync = []() { __syncthreads(); };
  if (role == 0) {
    coop_role_core<NS, UPDATE>(ld, stf, [lane](int s, double v) { xch[s][lane] = v; }, sync, in, k);
  } else {
    coop_role_passive<NS, UPDATE>(ld, stf, [lane](int s) { return xch[s][lane]; }, sync, in, k);
  }
}

// PB_HOST_BROADCAST inputs: dst [rows][B] <- one value per row (pronto_batch.hip stage_in)
struct RowVals {
  static constexpr int MAX = 36;  // the largest block of one call: a full 6 x 6 measurement covariance
  double v[MAX];
};
__global__ void k_fill_rows(double *__restrict__ dst, int rows, int B, RowVals vals)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int r = 0; r < rows; r++) dst[(long) r * B + b] = vals.v[r];
}

// Noise identification (state-estimator/src/noise_id/noise_id.cpp:37-38,44-65): window error e = head (-) truth with
// chi = Log(truth.quat^-1 * quat), then over the m active indices log det P_aa and e_a^T P_aa^-1 e_a (the two pieces of
// eigen_utils' loglike_normalized).  out [3][B] = logdet, mahalanobis^2, -0.5*(m log 2pi + logdet + maha).
// err_out [NS][B] (optional) receives the full error vector.  Runtime index list, gathered like k_update.
template <int NS, int M>
__global__ void k_window_nll(const double *__restrict__ st, long stride, int B, IdxArg<M> idx,
                             const double *__restrict__ tvec, const double *__restrict__ tquat, double *__restrict__ out,
                             double *__restrict__ err_out)
{
  using L = Lay<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double q[4], tq[4], dchi[3];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    q[i] = st[(long) (L::OFF_QUAT + i) * stride + b];
    tq[i] = tquat[(long) i * B + b];
  }
  subtract_quats(q, tq, dchi);
  if (err_out != nullptr) {
    for (int i = 0; i < NS; i++) {
      double e = st[(long) (L::OFF_VEC + i) * stride + b] - tvec[(long) i * B + b];
      if (i >= 6 && i <= 8) e = dchi[i - 6];
      err_out[(long) i * B + b] = e;
    }
  }
  double e[M], S[M * (M + 1) / 2], d[M];
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    const int ii = idx.v[kk];
    double v = st[(long) (L::OFF_VEC + ii) * stride + b] - tvec[(long) ii * B + b];
    if (ii >= 6 && ii <= 8) v = (ii == 6) ? dchi[0] : (ii == 7 ? dchi[1] : dchi[2]);
    e[kk] = v;
#pragma unroll
    for (int j = 0; j <= kk; j++) S[pk(kk, j)] = st[(long) (L::OFF_P + pk(ii, idx.v[j])) * stride + b];
  }
  ldlt<M>(S, d);
  double logdet = 0.0, maha = 0.0, y[M];
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s = e[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s -= S[pk(kk, j)] * y[j];
    y[kk] = s;
    logdet += log(d[kk]);
    maha += s * s / d[kk];
  }
  out[b] = logdet;
  out[(long) B + b] = maha;
  out[2L * B + b] = -0.5 * (M * 1.8378770664093453 + logdet + maha);  // log(2 pi)
}

// IMU front end (InsHandler::doFilter, sensor_handlers.cpp:154-162 + iir_notch.cpp:34-61): cascade of three 2nd-order
// IIR notches per accelerometer axis, one filter per lane, n_packets consecutive packets per call (a KVH batch message
// carries ~3 new 1 kHz packets; all are filtered, the newest filtered one feeds the predict).  State per filter:
// [axis][stage]{x0,x1,y0,y1} = 36 doubles in nst[36][stride].  Bytes: 24 B in per packet + 576 B state per call + 24 B out.
struct NotchCoef {
  double b[3][3], a[3][3];  // [stage][tap]
};
// One lane per (filter, axis): blockIdx.y is the axis -- three times the waves of a lane-per-filter mapping, which this
// short, latency-bound kernel needs (rocprof: 14.6 us -> see DESIGN.md 6 for the lane-per-filter version it replaced).
__global__ void k_notch(double *__restrict__ nst, long stride, int B, int n_packets, const double *__restrict__ acc_in,
                        double *__restrict__ acc_out, NotchCoef k)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ax = blockIdx.y;
  if (b >= B) return;
  double s[3][4];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) s[i][t] = nst[(long) ((ax * 3 + i) * 4 + t) * stride + b];
  double v = 0.0;
  for (int p = 0; p < n_packets; p++) {
    v = acc_in[((long) p * 3 + ax) * B + b];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double in = v;
      const double xb = in * k.b[i][0] + s[i][0] * k.b[i][1] + s[i][1] * k.b[i][2];
      const double ya = s[i][2] * k.a[i][1] + s[i][3] * k.a[i][2];
      const double out = xb - ya;
      s[i][1] = s[i][0]; s[i][0] = in;
      s[i][3] = s[i][2]; s[i][2] = out;
      v = out;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) nst[(long) ((ax * 3 + i) * 4 + t) * stride + b] = s[i][t];
  if (n_packets > 0) acc_out[(long) ax * B + b] = v;
}

// Counter calibration: a plain copy with EXACTLY the access pattern of k_step (buffer_load/store_dwordx2, 8 bytes
// per lane, component-major rows of `stride` doubles), so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled
// on a known byte count (MI355X_MICROARCH.md section HBM: widths other than 16 B/lane are uncalibrated).
__global__ __launch_bounds__(64, 1) void k_calib_copy(const double *__restrict__ src, double *__restrict__ dst,
                                                      long stride, int B, int ncomp)
{
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u, s8 = (unsigned) stride * 8u;
  const rsrc_t ri = mkbuf(src, (unsigned) ncomp * s8), ro = mkbuf(dst, (unsigned) ncomp * s8);
  for (int c0 = 0; c0 < ncomp; c0 += 20) {
    double v[20];
#pragma unroll
    for (int i = 0; i < 20; i++) v[i] = (c0 + i < ncomp) ? ldg(ri, (unsigned) (c0 + i) * s8, bo) : 0.0;
#pragma unroll
    for (int i = 0; i < 20; i++)
      if (c0 + i < ncomp) stg(ro, (unsigned) (c0 + i) * s8, bo, v[i]);
  }
}

}  // namespace pb
