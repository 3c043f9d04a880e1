// rbis_kernels.hpp -- gfx950 kernels of the batched RBIS EKF (included by pronto_batch.hip only).
//
// Data layout in HBM (Slots<NS> in rbis_device.hpp): the state array is cut into tiles of 64 filters; a tile is NROW rows
// of 64 x 16 bytes, row r holding the component pair (slot 2r, slot 2r+1) of each of its filters.  One lane owns one
// filter and moves 16 bytes per access, so every global access of a wave is one fully coalesced 1 KiB row and the whole
// round trip of a wave stays inside one contiguous tile (70 KiB for n=15, 129 KiB for n=21).
//
// Addressing: ONE 128-bit buffer descriptor per tile (64-bit tile base in SGPRs, so a context is not limited to 4 GiB),
// the row offset as an immediate / SGPR and ONE per-lane byte offset (lane * 16) shared by every access:
// `buffer_load/store_dwordx4 v, v_off, s[rsrc], s_off offen`.  No per-access 64-bit VGPR address is materialised.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pronto_batch.h"
#include "rbis_coop.hpp"
#include "rbis_quad.hpp"
#include "rbis_device.hpp"

// workgroup size of the one-lane hot kernel = one tile
#define PB_STEP_BLOCK 64

namespace pb {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// 128-bit buffer descriptor over [p, p+bytes): out-of-range lanes read 0 / drop their stores (hardware check)
__device__ __forceinline__ rsrc_t mkbuf(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
// buffer_load_dwordx2 v, v_off, s[rsrc], s_off offen : voff = per-lane byte offset, soff = uniform offset
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
// 16 bytes per lane
template <int AUX = 0>
__device__ __forceinline__ d2_t ldg2(rsrc_t r, unsigned soff, unsigned voff)
{
  return __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
// HAZARD (found on MI355X, ROCm 7.2): a buffer store of more than 64 bits reads its data VGPRs over several cycles; a VALU
// instruction that overwrites one of them in the very next slot can reach the register file first, and the LAST four lanes
// of every 16 then store the new value (observed: `buffer_store_dwordx4 v[130:133] ... s48 offen` directly followed by
// `v_mul_f64 v[130:131], ...` corrupted lanes 12-15, 28-31, 44-47, 60-63 of that row, once per ~1000 launches).  LLVM's
// hazard recognizer inserts the wait state only when soffset is NOT an SGPR (GCNHazardRecognizer::createsVALUHazard);
// here soffset is always an SGPR, so the wait states are placed by hand: the volatile asm keeps its place behind the store
// (side-effect order) and takes the data registers as INPUTS, so they stay untouched until the nop has issued.
template <int AUX = 0>
__device__ __forceinline__ void stg2(rsrc_t r, unsigned soff, unsigned voff, d2_t v)
{
  const v4u d = __builtin_bit_cast(v4u, v);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, AUX);
  asm volatile("s_nop 1" ::"v"(d));
}

// Memory hint of the state round trip in the step kernels (template parameter MH), picked by the host from the state
// size: while the state fits the 256 MB memory-side cache, stores with sc1 (write through the XCD's L2) are
// a few % faster; far beyond it, non-temporal loads AND stores are; in between neither helps.
enum { MH_DEFAULT = 0, MH_STORE_SC1 = 1, MH_STREAM_NT = 2 };
template <int MH> struct MemHint {
  static constexpr int LA = (MH == MH_STREAM_NT) ? 2 : 0;
  static constexpr int SA = (MH == MH_STORE_SC1) ? 16 : (MH == MH_STREAM_NT) ? 2 : 0;
};

// Workgroups are dealt round-robin to the 8 XCDs.  With k.xcd_remap each XCD walks one contiguous range of tiles
// (bijective for any grid size) instead of every 8th tile (host picks).
__device__ __forceinline__ unsigned xcd_workgroup(const Consts &k)
{
  unsigned wg = blockIdx.x;
  if (k.xcd_remap) {
    const unsigned nq = gridDim.x >> 3, nr = gridDim.x & 7u, xcd = blockIdx.x & 7u, rank = blockIdx.x >> 3;
    wg = (xcd < nr ? xcd * (nq + 1u) : nr * (nq + 1u) + (xcd - nr) * nq) + rank;
  }
  return wg;
}

// One lane's window on its tile: component-addressed reads and writes on top of 16-byte row accesses.
//   ld(comp)      value of a canonical component (Lay<NS> numbering); the row is loaded on first use and cached
//   need<R0,R1>() issue the loads of rows [R0, R1) now (before any store that could alias them)
//   st(comp, v)   posterior value of a component; a row is stored the moment both its halves are known
// Every index is a compile-time constant once the callers' loops are unrolled, so `loaded[]` / `have[]` fold away and
// the caches are plain registers (cdna_hip_programming.md rule 20); the kernels' resource usage shows 0 bytes of scratch.
template <int NS, int LA, int SA, bool TWO_ROLE = false>
struct TileIO {
  using S = Slots<NS>;
  rsrc_t rs, ro;
  unsigned vo;  // lane * 16
  d2_t cache[S::NROW];
  bool loaded[S::NROW];
  double outv[S::NSLOT];
  bool have[S::NSLOT];
  __device__ __forceinline__ TileIO(const double *st, double *sto, unsigned tile, unsigned lane)
  {
    rs = mkbuf(reinterpret_cast<const char *>(st) + (size_t) tile * S::TILE_BYTES, S::TILE_BYTES);
    ro = mkbuf(reinterpret_cast<char *>(sto) + (size_t) tile * S::TILE_BYTES, S::TILE_BYTES);
    vo = lane * 16u;
#pragma unroll
    for (int r = 0; r < S::NROW; r++) loaded[r] = false;
#pragma unroll
    for (int s = 0; s < S::NSLOT; s++) {
      have[s] = S::T.comp_of[s] < 0;  // a padding slot is always "known" (0)
      outv[s] = 0.0;
    }
  }
  __device__ __forceinline__ void fetch(int r)
  {
    if (!loaded[r]) {
      cache[r] = ldg2<LA>(rs, (unsigned) r * 1024u, vo);
      loaded[r] = true;
    }
  }
  template <int R0, int R1>
  __device__ __forceinline__ void need()
  {
#pragma unroll
    for (int r = R0; r < R1; r++) fetch(r);
  }
  __device__ __forceinline__ double ld(int comp)
  {
    const int s = S::T.slot_of[comp];
    fetch(s >> 1);
    return (s & 1) ? cache[s >> 1].y : cache[s >> 1].x;
  }
  __device__ __forceinline__ void st(int comp, double v)
  {
    const int s = S::T.slot_of[comp];
    if (TWO_ROLE && S::T.split2[s >> 1]) {  // the other half of this row is the other role's: store 8 bytes
      stg<SA>(ro, (unsigned) (s >> 1) * 1024u + (unsigned) (s & 1) * 8u, vo, v);
      return;
    }
    outv[s] = v;
    have[s] = true;
    if (have[s ^ 1]) {
      const d2_t o = { outv[s & ~1], outv[s | 1] };
      stg2<SA>(ro, (unsigned) (s >> 1) * 1024u, vo, o);
    }
  }
  // one component at a RUN-TIME slot (wave-uniform): an 8-byte access into the row
  __device__ __forceinline__ double ld_slot_rt(int slot) const
  {
    return ldg(rs, (unsigned) (slot >> 1) * 1024u + (unsigned) (slot & 1) * 8u, vo);
  }
};

// slot of a component known only at run time (wave-uniform: a scalar table read)
template <int NS>
__device__ __forceinline__ int slot_rt(int comp)
{
  return Slots<NS>::T.slot_of[comp];
}
__device__ __forceinline__ int pk_rt(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

struct IdxVel {
  static constexpr Idx<3> value = { { 3, 4, 5 } };
};

// RBISIMUProcessStep::updateFilter [+ RBISIndexedMeasurement::updateFilter with idx = {3,4,5}, diagonal R]
// (rbis_update_interface.cpp:30-52, :54-95).  The BASELINE hot step: 2*(n+4+1+n(n+1)/2)*8 + 56 + 48 bytes/filter.
// One lane per filter, one tile per 64-thread workgroup.
template <int NS, bool UPDATE, int MH = MH_DEFAULT>
__global__ __launch_bounds__(PB_STEP_BLOCK, 1) void k_step(const double *st, double *sto, int B,
                                                const double *__restrict__ imu, const double *__restrict__ lo,
                                                const uint8_t *__restrict__ mask, double qg, double qa, double qbg,
                                                double qba, Consts k, StepBcast bc)
{
  using L = Lay<NS>;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, threadIdx.x);  // posterior: in place or a checkpoint slot
  const rsrc_t ri = mkbuf(imu, 7u * B8);
  const rsrc_t rl = mkbuf(lo, UPDATE ? 6u * B8 : 0u);
  // the sensor blocks are requested FIRST: they are the only loads of the step that are never cache-resident (a new block
  // every message) and returns are in order per wave (k_step_quad: 43.5 -> 39.8 us at 64k x 21 states with long streams)
  double gyro[3], accel[3], dt, z[3], rd[3];
  bool upd = false;
  if constexpr (UPDATE) upd = (mask == nullptr) || (mask[b] != 0);
  if (bc.on & 1) {  // one IMU message for every filter: kernel arguments (wave-uniform branch)
#pragma unroll
    for (int i = 0; i < 3; i++) { gyro[i] = bc.imu[i]; accel[i] = bc.imu[3 + i]; }
    dt = bc.imu[6];
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      gyro[i] = ldg(ri, i * B8, bo);
      accel[i] = ldg(ri, (3 + i) * B8, bo);
    }
    dt = ldg(ri, 6u * B8, bo);
  }
  if (bc.on & 2) {
#pragma unroll
    for (int i = 0; i < 3; i++) { z[i] = bc.lo[i]; rd[i] = bc.lo[3 + i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      z[i] = UPDATE ? ldg(rl, i * B8, bo) : 0.0;
      rd[i] = UPDATE ? ldg(rl, (3 + i) * B8, bo) : 1.0;
    }
  }
  if (k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    qg = ldg(rq, 0u, bo); qa = ldg(rq, B8, bo); qbg = ldg(rq, 2u * B8, bo); qba = ldg(rq, 3u * B8, bo);
  }
  io.template need<0, Slots<NS>::NROW>();
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = io.ld(L::OFF_QUAT + i);
  ll = io.ld(L::OFF_LL);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = io.ld(L::OFF_P + i);
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
    measurement_update<NS, 3>(x, q, P, ll, resid, S, IdxVel{}, k, [&io](int pi, double v) { io.st(L::OFF_P + pi, v); }, upd);
  } else {
#pragma unroll
    for (int i = 0; i < L::NP; i++) io.st(L::OFF_P + i, P[i]);
  }
#pragma unroll
  for (int i = 0; i < NS; i++) io.st(L::OFF_VEC + i, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) io.st(L::OFF_QUAT + i, q[i]);
  io.st(L::OFF_LL, ll);
}

// Time-fused replay: T consecutive predict+update steps per launch with the state and P resident in registers; only the
// 104 B/filter of inputs stream from HBM per step, the posterior is materialised once per launch.  This is NOT the plugin
// path (MavStateEstimator::addUpdate publishes a posterior per message) but what a parameter sweep or a likelihood
// evaluation over a log segment wants (param_sweep.py:39-52).  Accounting: 104 + 2240/T bytes per filter-step, so the
// bound moves from HBM to fp64 VALU issue; bench.py reports it separately (never as the headline value).
// Inputs of step t+1 are prefetched while step t computes (one wave per SIMD: nothing else hides their latency).
template <int NS>
__global__ __launch_bounds__(64, 1) void k_replay_fused(double *__restrict__ st, int B, int T,
                                                        const double *__restrict__ imu, const double *__restrict__ lo,
                                                        const uint8_t *__restrict__ mask, double qg, double qa, double qbg,
                                                        double qba, Consts k)
{
  using L = Lay<NS>;
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, 0, 0> io(st, st, blockIdx.x, threadIdx.x);
  io.template need<0, Slots<NS>::NROW>();
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = io.ld(L::OFF_QUAT + i);
  ll = io.ld(L::OFF_LL);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = io.ld(L::OFF_P + i);
  if (k.qblk != nullptr) {
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    qg = ldg(rq, 0u, bo); qa = ldg(rq, B8, bo); qbg = ldg(rq, 2u * B8, bo); qba = ldg(rq, 3u * B8, bo);
  }
  // step-t input blocks are [7][B] / [6][B] / [B] slabs of the streams; 64-bit slab base, 32-bit offsets inside
  double in[13], nx[13] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  bool upd, nupd = false;
  (void) nx; (void) nupd;  // only used with PB_REPLAY_PREFETCH
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
  for (int i = 0; i < L::NP; i++) io.st(L::OFF_P + i, P[i]);
#pragma unroll
  for (int i = 0; i < NS; i++) io.st(L::OFF_VEC + i, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) io.st(L::OFF_QUAT + i, q[i]);
  io.st(L::OFF_LL, ll);
}

template <int M>
struct IdxArg {
  int v[M];
};
template <int M>
struct DiagArg {
  double v[M];
};

// (The generic run-time-index update of a 21-state batch is k_update_quad_rt, rbis_quad_rt.hpp; the 15-state one
//  k_update_lane_rt below (m <= 4) and k_update_coop_rt (m = 5, 6; rbis_quad_rt.hpp).  Round 1 / 2 gathered the measured columns with 8-byte run-time-slot loads and streamed the
//  covariance through one wave, k_update<NS, M, ORIENT>: 54-80 us for 21 states at 64k filters, retired in round 3.)

// RBISResetUpdate::updateFilter, per-filter inputs: vec [n][B], quat [4][B], cov [n*n][B] column-major
template <int NS>
__global__ void k_reset(double *__restrict__ st, int B, const double *__restrict__ vec,
                        const double *__restrict__ quat, const double *__restrict__ cov, const double *__restrict__ ll = nullptr)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < NS; i++) st[S::eidx(L::OFF_VEC + i, b)] = vec[(long) i * B + b];
  for (int i = 0; i < 4; i++) st[S::eidx(L::OFF_QUAT + i, b)] = quat[(long) i * B + b];
  st[S::eidx(L::OFF_LL, b)] = ll ? ll[b] : 0.0;
  for (int i = 0; i < NS; i++)
    for (int j = 0; j <= i; j++) st[S::eidx(L::OFF_P + pk(i, j), b)] = cov[(long) (j * NS + i) * B + b];
}

// broadcast reset: comp [NC] in canonical component order, packed on the host
template <int NS>
__global__ void k_reset_bcast(double *__restrict__ st, int B, const double *__restrict__ comp)
{
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int c = 0; c < Lay<NS>::NC; c++) st[S::eidx(c, b)] = comp[c];
}

template <int NS>
__global__ void k_get_head(const double *__restrict__ st, int first, int count, double *vec_out,
                           double *quat_out, double *cov_out, double *ll_out)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int b = first + t;
  if (vec_out)
    for (int i = 0; i < NS; i++) vec_out[(long) i * count + t] = st[S::eidx(L::OFF_VEC + i, b)];
  if (quat_out)
    for (int i = 0; i < 4; i++) quat_out[(long) i * count + t] = st[S::eidx(L::OFF_QUAT + i, b)];
  if (ll_out) ll_out[t] = st[S::eidx(L::OFF_LL, b)];
  if (cov_out)
    for (int c = 0; c < NS; c++)
      for (int r = 0; r < NS; r++)
        cov_out[(long) (c * NS + r) * count + t] = st[S::eidx(L::OFF_P + pk(r, c), b)];
}

// (position, quat) of the head posterior -> snapshot slot [7][stride]
template <int NS>
__global__ void k_snapshot(const double *__restrict__ st, long stride, int B, double *__restrict__ snap)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < 3; i++) snap[(long) i * stride + b] = st[S::eidx(L::OFF_VEC + 9 + i, b)];
  for (int i = 0; i < 4; i++) snap[(long) (3 + i) * stride + b] = st[S::eidx(L::OFF_QUAT + i, b)];
}

// T1 = T0 * (t, q)   (rbis_fovis_update.cpp:219-223)
static __global__ void k_compose(const double *__restrict__ snap, long stride, int B, const double *__restrict__ t,
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
__global__ void k_summary(const double *__restrict__ st, int B, double *__restrict__ out)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double s_ll = 0, s_abs = 0, qdev = 0, nonfin = 0;
  if (b < B) {
    s_ll = st[S::eidx(L::OFF_LL, b)];
    double qn = 0;
    for (int i = 0; i < NS + 4; i++) {
      const double v = st[S::eidx(i, b)];
      s_abs += fabs(v);
      if (!isfinite(v)) nonfin += 1;
      if (i >= NS) qn += v * v;
    }
    for (int i = 0; i < L::NP; i++)
      if (!isfinite(st[S::eidx(L::OFF_P + i, b)])) nonfin += 1;
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
// quaternion), wave 1 = role P (passive omega/accel panels) for the SAME 64 filters = one tile; each role owns a
// contiguous range of the tile's rows (Slots<NS>::ROW_SPLIT); one LDS hand-off + one barrier.
// This is the 21-state hot kernel (231 packed entries do not fit one lane) and the default mapping for n = 15.
// No lane returns before the barrier: lanes past the batch end work on the zero-initialised padding filters of the
// last tile and on bounds-checked (zero) inputs.
// inputs of the fused second update (CORR != NoCorr): z2 [M][B], R2 diagonal ([M][B], or broadcast values in rb2 when
// r2 == nullptr), quaternion measurement [4][B] (ORIENT), mask2 [B] (0 = the handler returned NULL for this filter)
struct CorrArgs {
  const double *z2 = nullptr, *r2 = nullptr, *qm2 = nullptr;
  const uint8_t *mask2 = nullptr;
  double rb2[6] = { 0, 0, 0, 0, 0, 0 };
  // zbc: ONE measurement for every filter (PB_HOST_BROADCAST): z and the quaternion travel as kernel arguments too
  double zb2[6] = { 0, 0, 0, 0, 0, 0 }, qb2[4] = { 1, 0, 0, 0 };
  int zbc = 0;
  // rfull: a FULL per-filter R, [m*m][B] column-major (pronto::indexed_measurement_t's R_effective); r2 / rb2 are unused then
  const double *rfull = nullptr;
};
// LegOdoCommon's lin_rot_rate list (rbis_legodo_common.cpp:66-67): velocity AND angular velocity -- the angular-velocity
// rows are pass-through states of the other role, so the two-role kernels below do not take it.
struct IdxVelOmega {
  static constexpr Idx<6> value = { { 3, 4, 5, 0, 1, 2 } };
};

// Stand-alone indexed update, one lane per filter, COMPILE-TIME index list anywhere in the state (15 states: the whole
// filter lives in registers, as in k_step): one coalesced round trip of the state and the in-register update of the fused
// step (measurement_update) instead of the generic k_update's run-time column gather, which reads the measured columns a
// second time.  Diagonal (per filter or one for all) or full per-filter R, skip mask, broadcast z: CorrArgs.
template <int NS, int M, class IDXT, int MH = MH_DEFAULT>
__global__ __launch_bounds__(64, 1) void k_update_lane(const double *st, double *sto, int B, Consts k, CorrArgs ca)
{
  using L = Lay<NS>;
  constexpr Idx<M> idx = IDXT::value;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + threadIdx.x;
  if (b >= (unsigned) B) return;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, threadIdx.x);
  const bool upd = (ca.mask2 == nullptr) || (ca.mask2[b] != 0);  // 0 = handler returned NULL for this filter
  // the measurement first (the only loads that are never cache-resident), then the state rows
  const rsrc_t rz = mkbuf(ca.z2, ca.zbc ? 0u : (unsigned) M * B8);
  const rsrc_t rr = mkbuf(ca.rfull ? ca.rfull : ca.r2, ca.rfull ? (unsigned) (M * M) * B8 : (ca.r2 ? (unsigned) M * B8 : 0u));
  double z[M], R[M * (M + 1) / 2];
#pragma unroll
  for (int i = 0; i < M; i++) {
    z[i] = ca.zbc ? ca.zb2[i] : ldg(rz, i * B8, bo);
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double r;
      if (ca.rfull) r = ldg(rr, (j * M + i) * B8, bo);
      else if (i != j) r = 0.0;
      else r = ca.r2 ? ldg(rr, i * B8, bo) : ca.rb2[i];
      R[pk(i, j)] = upd ? r : (i == j ? 1.0 : 0.0);  // benign R for skipped filters (their block may hold anything)
    }
  }
  io.template need<0, Slots<NS>::NROW>();
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = io.ld(L::OFF_QUAT + i);
  ll = io.ld(L::OFF_LL);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = io.ld(L::OFF_P + i);
  double resid[M], S[M * (M + 1) / 2];
#pragma unroll
  for (int i = 0; i < M; i++) {
    resid[i] = upd ? z[i] - x[idx.v[i]] : 0.0;                                   // rbis.cpp:170
#pragma unroll
    for (int j = 0; j <= i; j++) S[pk(i, j)] = P[pk(idx.v[i], idx.v[j])] + R[pk(i, j)];  // rbis.cpp:134-135
  }
  measurement_update<NS, M>(x, q, P, ll, resid, S, IDXT{}, k, [&io](int pi, double v) { io.st(L::OFF_P + pi, v); }, upd);
#pragma unroll
  for (int i = 0; i < NS; i++) io.st(L::OFF_VEC + i, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) io.st(L::OFF_QUAT + i, q[i]);
  io.st(L::OFF_LL, ll);
}

// Column c (wave-uniform) of the packed covariance into column KK of W, with the residual and row KK of S = R + P[idx, idx]:
// a chain of scalar compares over the compile-time candidates CC, CC + 1, ... (one branch taken, NS register moves).
template <int NS, int M, int KK, int CC, bool ORIENT>
__device__ __forceinline__ void pick_column(int c, const double (&P)[NS * (NS + 1) / 2], const double (&x)[NS], const double (&zz)[M],
                                            const double (&dq)[3], bool upd, double (&W)[NS][M], double (&resid)[M],
                                            double (&S)[M * (M + 1) / 2])
{
  if constexpr (CC < NS) {
    if (c == CC) {
      // (a distinct marker per branch: otherwise the identical branch bodies are merged into ONE body that loads through a
      // selected address, which pins the whole covariance in scratch memory)
      asm volatile("; column %0 -> %1" ::"n"(CC), "n"(KK));
#pragma unroll
      for (int i = 0; i < NS; i++) W[i][KK] = P[pk(i, CC)];
      double r = zz[KK] - x[CC];                                    // rbis.cpp:170
      if constexpr (ORIENT && CC >= 6 && CC <= 8) r = dq[CC - 6];   // rbis.cpp:206-208
      resid[KK] = upd ? r : 0.0;
#pragma unroll
      for (int j = 0; j < KK; j++) S[pk(KK, j)] += W[CC][j];        // P[idx_KK, idx_j]: row CC of an earlier column
      S[pk(KK, KK)] += P[pk(CC, CC)];
      asm volatile("; column %0 -> %1 done" ::"n"(CC), "n"(KK));  // (common-tail sinking stops here)
    } else {
      pick_column<NS, M, KK, CC + 1, ORIENT>(c, P, x, zz, dq, upd, W, resid, S);
    }
  }
}

// The generic update for 15 states: RUN-TIME index list (any indices, m = 1..6, diagonal / broadcast / full R, orientation
// residual, skip mask -- RBISIndexedMeasurement / RBISIndexedPlusOrientationMeasurement::updateFilter,
// rbis_update_interface.cpp:54-107) on the same one-lane in-register scheme.  The index list is a kernel argument, i.e.
// wave-uniform: each measured column is picked by a scalar branch over the 15 compile-time candidates (15 register moves
// taken, nothing gathered from memory), then measurement_update_cols runs as for a compile-time list -- so the result is
// bit-identical to k_update_lane's for the same list.  21 states do not fit one lane's registers: k_update stays for them.
template <int NS, int M, bool ORIENT, int MH = MH_DEFAULT>
__global__ __launch_bounds__(64, 1) void k_update_lane_rt(const double *st, double *sto, int B, IdxArg<M> idx,
                                                          const double *__restrict__ z, const double *__restrict__ R,
                                                          int rkind, DiagArg<M> rb, const double *__restrict__ qmeas,
                                                          const uint8_t *__restrict__ mask, Consts k)
{
  using L = Lay<NS>;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + threadIdx.x;
  if (b >= (unsigned) B) return;
  const bool upd = (mask == nullptr) || (mask[b] != 0);  // 0 = handler returned NULL for this filter
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, threadIdx.x);
  const rsrc_t rz = mkbuf(z, (unsigned) M * B8);
  const rsrc_t rR = mkbuf(R, rkind == PB_R_DIAG ? (unsigned) M * B8 : (rkind == PB_R_FULL ? (unsigned) (M * M) * B8 : 0u));
  const rsrc_t rq = mkbuf(qmeas, ORIENT ? 4u * B8 : 0u);
  double zz[M], S[M * (M + 1) / 2], qm[4] = { 1.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int i = 0; i < M; i++) {
    zz[i] = ldg(rz, i * B8, bo);
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double r;
      if (rkind == PB_R_DIAG_BROADCAST) r = (i == j) ? rb.v[i] : 0.0;
      else if (rkind == PB_R_DIAG) r = (i == j) ? ldg(rR, i * B8, bo) : 0.0;
      else r = ldg(rR, (j * M + i) * B8, bo);
      S[pk(i, j)] = upd ? r : (i == j ? 1.0 : 0.0);  // benign R for skipped filters (their R block may hold anything)
    }
  }
  if constexpr (ORIENT) {
#pragma unroll
    for (int i = 0; i < 4; i++) qm[i] = ldg(rq, i * B8, bo);
  }
  io.template need<0, Slots<NS>::NROW>();
  double x[NS], q[4], ll, P[L::NP];
#pragma unroll
  for (int i = 0; i < NS; i++) x[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = io.ld(L::OFF_QUAT + i);
  ll = io.ld(L::OFF_LL);
#pragma unroll
  for (int i = 0; i < L::NP; i++) P[i] = io.ld(L::OFF_P + i);
  double dq[3] = { 0.0, 0.0, 0.0 };
  if constexpr (ORIENT) subtract_quats(qm, q, dq);  // rbis.cpp:199-205
  // the measured columns, the measured states and P[idx, idx], by scalar branches on the (uniform) indices
  double W[NS][M], resid[M];
  static_for<M>([&](auto KK) { pick_column<NS, M, decltype(KK)::value, 0, ORIENT>(idx.v[decltype(KK)::value], P, x, zz, dq, upd, W, resid, S); });
  measurement_update_cols<NS, M>(x, q, P, ll, resid, S, W, k, [&io](int pi, double v) { io.st(L::OFF_P + pi, v); }, upd);
#pragma unroll
  for (int i = 0; i < NS; i++) io.st(L::OFF_VEC + i, x[i]);
#pragma unroll
  for (int i = 0; i < 4; i++) io.st(L::OFF_QUAT + i, q[i]);
  io.st(L::OFF_LL, ll);
}

template <int NS, bool UPDATE, int MH = MH_DEFAULT, class CORR = NoCorr, bool PREDICT = true>
__global__ __launch_bounds__(128, 1) void k_step_coop(const double *st, double *sto, int B,
                                                      const double *__restrict__ imu, const double *__restrict__ lo,
                                                      const uint8_t *__restrict__ mask, double qg, double qa,
                                                      double qbg, double qba, Consts k, CorrArgs ca, StepBcast bc = StepBcast())
{
  __shared__ double xch[(UPDATE || CORR::M > 0) ? CoopX<NS, CORR>::NXCH : 1][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  // half tiles (Consts::half_tiles, batches that leave workgroup slots empty): workgroup 2 t takes filters 0-31 of tile t, 2 t + 1
  // filters 32-63; lanes l and l + 32 mirror each other, so nothing else in the kernel knows
  const unsigned wgi = xcd_workgroup(k);
  const unsigned tile = k.half_tiles ? (wgi >> 1) : wgi;
  const unsigned lane = k.half_tiles ? ((threadIdx.x & 31u) | ((wgi & 1u) << 5)) : (threadIdx.x & 63u);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA, true> io(st, sto, tile, lane);
  const rsrc_t ri = mkbuf(imu, PREDICT ? 7u * B8 : 0u);
  const rsrc_t rl = mkbuf(lo, UPDATE ? 6u * B8 : 0u);
  StepInputs in;
  if (PREDICT && (bc.on & 1)) {  // one IMU message for every filter: kernel arguments (wave-uniform branch)
#pragma unroll
    for (int i = 0; i < 3; i++) { in.gyro[i] = bc.imu[i]; in.accel[i] = bc.imu[3 + i]; }
    in.dt = bc.imu[6];
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      in.gyro[i] = PREDICT ? ldg(ri, i * B8, bo) : 0.0;
      in.accel[i] = PREDICT ? ldg(ri, (3 + i) * B8, bo) : 0.0;
    }
    in.dt = PREDICT ? ldg(ri, 6u * B8, bo) : 0.0;
  }
  if (PREDICT && (bc.on & 2)) {
#pragma unroll
    for (int i = 0; i < 3; i++) { in.z[i] = bc.lo[i]; in.rd[i] = bc.lo[3 + i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      in.z[i] = UPDATE ? ldg(rl, i * B8, bo) : 0.0;
      in.rd[i] = UPDATE ? ldg(rl, (3 + i) * B8, bo) : 1.0;
    }
  }
  in.upd = UPDATE && (b < (unsigned) B) && (mask == nullptr || mask[b] != 0);
  in.qg = qg; in.qa = qa; in.qbg = qbg; in.qba = qba;
  if (PREDICT && k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    in.qg = ldg(rq, 0u, bo); in.qa = ldg(rq, B8, bo); in.qbg = ldg(rq, 2u * B8, bo); in.qba = ldg(rq, 3u * B8, bo);
  }
  CorrInputs cin;
  if constexpr (CORR::M > 0) {
    const rsrc_t rz = mkbuf(ca.z2, (unsigned) CORR::M * B8);
    const rsrc_t rr = mkbuf(ca.r2, ca.r2 ? (unsigned) CORR::M * B8 : 0u);
    const rsrc_t rq2 = mkbuf(ca.qm2, CORR::ORIENT ? 4u * B8 : 0u);
#pragma unroll
    for (int i = 0; i < CORR::M; i++) {
      cin.z[i] = ca.zbc ? ca.zb2[i] : ldg(rz, i * B8, bo);
      cin.rd[i] = ca.r2 ? ldg(rr, i * B8, bo) : ca.rb2[i];
    }
    if (ca.rfull != nullptr) {  // full R (wave-uniform branch): diagonal + strictly-lower part
      const rsrc_t rf = mkbuf(ca.rfull, (unsigned) (CORR::M * CORR::M) * B8);
#pragma unroll
      for (int i = 0; i < CORR::M; i++) {
        cin.rd[i] = ldg(rf, (unsigned) (i * CORR::M + i) * B8, bo);
#pragma unroll
        for (int j = 0; j < i; j++) cin.ro[i * (i - 1) / 2 + j] = ldg(rf, (unsigned) (j * CORR::M + i) * B8, bo);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) cin.qm[i] = CORR::ORIENT ? (ca.zbc ? ca.qb2[i] : ldg(rq2, i * B8, bo)) : 0.0;
    cin.upd = (b < (unsigned) B) && (ca.mask2 == nullptr || ca.mask2[b] != 0);
    // A stand-alone update that NO filter of this tile takes, working in place: nothing to load, nothing to store.  (Both waves
    // of the tile read the same 64 mask bytes, so they leave together -- no barrier is left waiting.)  This is what makes the
    // per-filter choice between two updates (RBISEitherUpdate: LegOdoCommon's pos_and_lin_rate / lin_rate fall-back) cost one
    // state round trip, not two: the half no filter takes returns at once.
    if constexpr (!PREDICT && !UPDATE) {
      if (st == sto && ca.mask2 != nullptr && __ballot(cin.upd) == 0ull) return;
    }
  }
  auto ld = [&io](int comp) { return io.ld(comp); };
  auto stf = [&io](int comp, double v) { io.st(comp, v); };
  auto sync = []() { __syncthreads(); };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  if (role == 0) {
    io.template need<0, Slots<NS>::ROW_SPLIT>();
    coop_role_core<NS, UPDATE, CORR, PREDICT>(ld, stf, [lane](int s, double v) { xch[s][lane] = v; }, xrd, sync, in, k, cin);
  } else {
    io.template need<Slots<NS>::ROW_SPLIT, Slots<NS>::NROW>();
    coop_role_passive<NS, UPDATE, CORR, PREDICT>(ld, stf, xrd, sync, in, k, cin);
  }
}

// The 21-state hot step on FOUR cooperating waves per 64 filters (rbis_quad.hpp): <= 256 registers per role, so two
// workgroups (8 waves) share a CU and one tile's loads overlap another's arithmetic and stores.  Same inputs, same
// posterior (to rounding: the c-b coupling enters P_cc as one additive term instead of inside the row operations) and the
// same bytes as k_step_coop<21>.  No lane returns before the barriers (see k_step_coop).
template <bool UPDATE, int MH = MH_DEFAULT>
__global__ __launch_bounds__(256, 2) void k_step_quad(const double *st, double *sto, int B,
                                                      const double *__restrict__ imu, const double *__restrict__ lo,
                                                      const uint8_t *__restrict__ mask, double qg, double qa,
                                                      double qbg, double qba, Consts k, StepBcast bc)
{
  using SL = Slots<21>;
  __shared__ double xch[Quad::NXCH][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<21, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, lane);
  // every role fetches its own copy of the inputs INSIDE its branch: nothing but addresses is live across the dispatch
  auto inputs = [&](bool meas) {
    const rsrc_t ri = mkbuf(imu, 7u * B8);
    const rsrc_t rl = mkbuf(lo, UPDATE ? 6u * B8 : 0u);
    StepInputs in;
    if (bc.on & 1) {  // one IMU message for every filter: kernel arguments (wave-uniform branch)
#pragma unroll
      for (int i = 0; i < 3; i++) { in.gyro[i] = bc.imu[i]; in.accel[i] = bc.imu[3 + i]; }
      in.dt = bc.imu[6];
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        in.gyro[i] = ldg(ri, i * B8, bo);
        in.accel[i] = ldg(ri, (3 + i) * B8, bo);
      }
      in.dt = ldg(ri, 6u * B8, bo);
    }
    if (bc.on & 2) {
#pragma unroll
      for (int i = 0; i < 3; i++) { in.z[i] = bc.lo[i]; in.rd[i] = bc.lo[3 + i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        in.z[i] = (UPDATE && meas) ? ldg(rl, i * B8, bo) : 0.0;
        in.rd[i] = (UPDATE && meas) ? ldg(rl, (3 + i) * B8, bo) : 1.0;
      }
    }
    in.upd = UPDATE && (b < (unsigned) B) && (mask == nullptr || mask[b] != 0);
    in.qg = qg; in.qa = qa; in.qbg = qbg; in.qba = qba;
    if (k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
      const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
      in.qg = ldg(rq, 0u, bo); in.qa = ldg(rq, B8, bo); in.qbg = ldg(rq, 2u * B8, bo); in.qba = ldg(rq, 3u * B8, bo);
    }
    return in;
  };
  auto ld = [&io](int comp) { return io.ld(comp); };
  auto stf = [&io](int comp, double v) { io.st(comp, v); };
  auto sync = []() { __syncthreads(); };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  auto xwr = [lane](int s, double v) { xch[s][lane] = v; };
  // the sensor blocks are requested FIRST: they are the only loads of the step that are never cache-resident (a new block
  // every message), and returns are in order per wave
  if (role == 0) {
    const StepInputs in = inputs(true);
    io.template need<SL::QROW[0], SL::QROW[1]>();
    quad_role_cc<UPDATE>(ld, stf, xwr, xrd, sync, in, k);
  } else if (role == 1) {
    const StepInputs in = inputs(false);
    io.template need<SL::QROW[1], SL::QROW[2]>();
    quad_role_cb<UPDATE>(ld, stf, xwr, xrd, sync, in, k);
  } else if (role == 2) {
    const StepInputs in = inputs(false);
    io.template need<SL::QROW[2], SL::QROW[3]>();
    quad_role_passive<UPDATE, 0, 0, true>(ld, stf, xwr, xrd, sync, in, k);
  } else {
    const StepInputs in = inputs(false);
    io.template need<SL::QROW[3], SL::QROW[4]>();
    quad_role_passive<UPDATE, 1, 0, true>(ld, stf, xwr, xrd, sync, in, k);
  }
}

// Stand-alone indexed (+ orientation) update of a 21-state batch on the four-wave mapping (rbis_quad.hpp, quad_upd_*): the
// handlers' index lists as compile-time c-state indices, diagonal R, one barrier, one state round trip at two waves per SIMD.
template <class CORR, int MH = MH_DEFAULT>
__global__ __launch_bounds__(256, 2) void k_update_quad(const double *st, double *sto, int B, Consts k, CorrArgs ca)
{
  using SL = Slots<21>;
  __shared__ double xch[QuadU<CORR>::NXCH][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<21, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, lane);
  // an update that no filter of this tile takes, in place: all four waves see the same mask bytes and leave together (k_step_coop)
  if (st == sto && ca.mask2 != nullptr && __ballot(b < (unsigned) B && ca.mask2[b < (unsigned) B ? b : 0u] != 0) == 0ull) return;
  auto inputs = [&](bool meas) {
    CorrInputs cin;
    const rsrc_t rz = mkbuf(ca.z2, (unsigned) CORR::M * B8);
    const rsrc_t rr = mkbuf(ca.r2, ca.r2 ? (unsigned) CORR::M * B8 : 0u);
    const rsrc_t rq2 = mkbuf(ca.qm2, CORR::ORIENT ? 4u * B8 : 0u);
#pragma unroll
    for (int i = 0; i < CORR::M; i++) {
      cin.z[i] = meas ? (ca.zbc ? ca.zb2[i] : ldg(rz, i * B8, bo)) : 0.0;
      cin.rd[i] = meas ? (ca.r2 ? ldg(rr, i * B8, bo) : ca.rb2[i]) : 1.0;
    }
    if (meas && ca.rfull != nullptr) {  // full R (wave-uniform branch): diagonal + strictly-lower part
      const rsrc_t rf = mkbuf(ca.rfull, (unsigned) (CORR::M * CORR::M) * B8);
#pragma unroll
      for (int i = 0; i < CORR::M; i++) {
        cin.rd[i] = ldg(rf, (unsigned) (i * CORR::M + i) * B8, bo);
#pragma unroll
        for (int j = 0; j < i; j++) cin.ro[i * (i - 1) / 2 + j] = ldg(rf, (unsigned) (j * CORR::M + i) * B8, bo);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) cin.qm[i] = (CORR::ORIENT && meas) ? (ca.zbc ? ca.qb2[i] : ldg(rq2, i * B8, bo)) : 0.0;
    cin.upd = (b < (unsigned) B) && (ca.mask2 == nullptr || ca.mask2[b] != 0);
    return cin;
  };
  auto ld = [&io](int comp) { return io.ld(comp); };
  auto stf = [&io](int comp, double v) { io.st(comp, v); };
  auto sync = []() { __syncthreads(); };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  auto xwr = [lane](int s, double v) { xch[s][lane] = v; };
  if (role == 0) {  // (the measurement blocks are requested first, see k_step_quad)
    const CorrInputs cin = inputs(true);
    io.template need<SL::QROW[0], SL::QROW[1]>();
    quad_upd_cc<CORR>(ld, stf, xwr, xrd, sync, cin, k);
  } else if (role == 1) {
    const CorrInputs cin = inputs(false);
    io.template need<SL::QROW[1], SL::QROW[2]>();
    quad_upd_cb<CORR>(ld, stf, xwr, xrd, sync, cin, k);
  } else if (role == 2) {
    const CorrInputs cin = inputs(false);
    io.template need<SL::QROW[2], SL::QROW[3]>();
    quad_upd_passive<CORR, 0>(ld, stf, xwr, xrd, sync, cin, k);
  } else {
    const CorrInputs cin = inputs(false);
    io.template need<SL::QROW[3], SL::QROW[4]>();
    quad_upd_passive<CORR, 1>(ld, stf, xwr, xrd, sync, cin, k);
  }
}

// Time-fused replay on the cooperative mapping: T consecutive predict + leg-odometry steps per launch with each role's part
// of the state resident in ITS registers; only the 104 B/filter of inputs stream from HBM per step and the posterior is
// written once per launch.  Per step the two roles trade what the other needs of the state vector through LDS (role C
// gives v chi Delta [biases] quat, role P gives omega accel: the process blocks linearise about the whole prior state),
// then run exactly the bodies of k_step_coop with loads and stores redirected to registers.  Two barriers per step.
// No per-message posterior: NOT the plugin path (see pb_replay_legodo_fused); accounting 104 + 2*state/T bytes per step.
// WRITE-THROUGH (pb_replay_legodo_checkpointed): every step's posterior is ALSO stored into its checkpoint slot -- the forward
// pass of the delayed-measurement history and of the smoother keeps every posterior (mav_state_est.cpp:50-70,98-189) -- while the
// state stays in the roles' registers: per step one slot is written and nothing is read, where the per-step path reads the
// previous slot and writes the next (268 MB per step through a 256 MB cache for 64k 21-state filters).
struct SlotOut {
  double *base = nullptr;       // first checkpoint slot to write (NULL: no write-through)
  size_t stride = 0;            // doubles per slot
};
template <int NS>
__global__ __launch_bounds__(128, 1) void k_replay_coop(double *st, int B, int T, const double *__restrict__ imu,
                                                        const double *__restrict__ lo, const uint8_t *__restrict__ mask,
                                                        double qg, double qa, double qbg, double qba, Consts k, SlotOut so)
{
  using L = Lay<NS>;
  using C = Coop<NS>;
  using SL = Slots<NS>;
  __shared__ double xch[C::NXCH][64];
  __shared__ double xst[NS + 4][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = blockIdx.x;
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, 0, 0, true> io(st, st, tile, lane);
  double q4[4] = { qg, qa, qbg, qba };
  if (k.qblk != nullptr) {
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
#pragma unroll
    for (int i = 0; i < 4; i++) q4[i] = ldg(rq, (unsigned) i * B8, bo);
  }
  auto sync = []() { __syncthreads(); };
  auto xw = [lane](int s, double v) { xch[s][lane] = v; };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  // inputs of step t (both roles need the IMU block; role C the measurement)
  auto inputs = [&](int t) {
    const rsrc_t ri = mkbuf(imu + (size_t) t * 7 * B, 7u * B8);
    const rsrc_t rl = mkbuf(lo + (size_t) t * 6 * B, 6u * B8);
    StepInputs in;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      in.gyro[i] = ldg(ri, i * B8, bo);
      in.accel[i] = ldg(ri, (3 + i) * B8, bo);
      in.z[i] = ldg(rl, i * B8, bo);
      in.rd[i] = ldg(rl, (3 + i) * B8, bo);
    }
    in.dt = ldg(ri, 6u * B8, bo);
    in.upd = (b < (unsigned) B) && (mask == nullptr || mask[(size_t) t * B + b] != 0);
    in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
    return in;
  };
  // The two roles run SEPARATE loops (the same two barriers per iteration in each): with one loop around a role branch
  // every component of both roles would be live across the back edge.
  if (role == 0) {
    double V[L::NC];  // canonical components; a role only ever touches its own ones plus the other's state vector
    io.template need<0, SL::ROW_SPLIT>();
    static_for<SL::NSLOT>([&](auto I) {
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
      if constexpr (comp >= 0 && SL::T.role2[slot] == 0) V[comp] = io.ld(comp);
    });
    auto ld = [&V](int comp) { return V[comp]; };
    auto stf = [&V](int comp, double v) { V[comp] = v; };
    for (int t = 0; t < T; t++) {
      const StepInputs in = inputs(t);
#pragma unroll
      for (int i = 0; i < C::NSC; i++) xst[C::fullc(i)][lane] = V[L::OFF_VEC + C::fullc(i)];
#pragma unroll
      for (int i = 0; i < 4; i++) xst[NS + i][lane] = V[L::OFF_QUAT + i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 6; i++) V[L::OFF_VEC + C::fullp(i)] = xst[C::fullp(i)][lane];
      coop_role_core<NS, true>(ld, stf, xw, xrd, sync, in, k);
      // (no third barrier: this role overwrites the hand-off only behind the next state-exchange barrier, which role P
      // reaches after it has finished reading; the exchange slots of the two roles are disjoint)
      if (so.base != nullptr) {  // write-through: this role's components of the posterior of step t
        TileIO<NS, 0, MemHint<MH_STREAM_NT>::SA, true> ios(st, so.base + (size_t) t * so.stride, tile, lane);
        static_for<SL::NSLOT>([&](auto I) {
          constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
          if constexpr (comp >= 0 && SL::T.role2[slot] == 0) ios.st(comp, V[comp]);
        });
      }
    }
    static_for<SL::NSLOT>([&](auto I) {
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
      if constexpr (comp >= 0 && SL::T.role2[slot] == 0) io.st(comp, V[comp]);
    });
  } else {
    double V[L::NC];
    io.template need<SL::ROW_SPLIT, SL::NROW>();
    static_for<SL::NSLOT>([&](auto I) {
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
      if constexpr (comp >= 0 && SL::T.role2[slot] == 1) V[comp] = io.ld(comp);
    });
    auto ld = [&V](int comp) { return V[comp]; };
    auto stf = [&V](int comp, double v) { V[comp] = v; };
    for (int t = 0; t < T; t++) {
      const StepInputs in = inputs(t);
#pragma unroll
      for (int i = 0; i < 6; i++) xst[C::fullp(i)][lane] = V[L::OFF_VEC + C::fullp(i)];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < C::NSC; i++) V[L::OFF_VEC + C::fullc(i)] = xst[C::fullc(i)][lane];
#pragma unroll
      for (int i = 0; i < 4; i++) V[L::OFF_QUAT + i] = xst[NS + i][lane];
      coop_role_passive<NS, true>(ld, stf, xrd, sync, in, k);
      if (so.base != nullptr) {
        TileIO<NS, 0, MemHint<MH_STREAM_NT>::SA, true> ios(st, so.base + (size_t) t * so.stride, tile, lane);
        static_for<SL::NSLOT>([&](auto I) {
          constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
          if constexpr (comp >= 0 && SL::T.role2[slot] == 1) ios.st(comp, V[comp]);
        });
      }
    }
    static_for<SL::NSLOT>([&](auto I) {
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];
      if constexpr (comp >= 0 && SL::T.role2[slot] == 1) io.st(comp, V[comp]);
    });
  }
}

// Time-fused replay of a 21-state batch on the FOUR-wave mapping: each role keeps the components it owns (Slots<21>::QROW
// row ranges) in ITS registers for T steps and runs the bodies of k_step_quad with loads and stores redirected to them.  Per
// step the owners of the state vector trade it through LDS (role PW: v chi Delta quat; role CB: biases, omega; role PA:
// accel -- every role linearises about the whole prior state), three barriers per step; the next step's sensor block is
// requested before this step's arithmetic.  OCC = waves per SIMD the register budget is cut for (2: 256 registers per role
// and 452 B of scratch, two workgroups per CU; 1: no scratch, one workgroup per CU: the faster one, see pb_step.hip).
template <int OCC>
__global__ __launch_bounds__(256, OCC) void k_replay_quad(double *st, int B, int T, const double *__restrict__ imu,
                                                        const double *__restrict__ lo, const uint8_t *__restrict__ mask,
                                                        double qg, double qa, double qbg, double qba, Consts k, SlotOut so)
{
  constexpr int NS = 21;
  using L = Lay<NS>;
  using SL = Slots<NS>;
  __shared__ double xch[Quad::NXCH][64];
  __shared__ double xst[NS + 4][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = blockIdx.x;
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, 0, 0> io(st, st, tile, lane);
  double q4[4] = { qg, qa, qbg, qba };
  if (k.qblk != nullptr) {
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
#pragma unroll
    for (int i = 0; i < 4; i++) q4[i] = ldg(rq, (unsigned) i * B8, bo);
  }
  auto sync = []() { __syncthreads(); };
  auto xw = [lane](int s, double v) { xch[s][lane] = v; };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  auto inputs = [&](int t, bool meas) {
    const rsrc_t ri = mkbuf(imu + (size_t) t * 7 * B, 7u * B8);
    const rsrc_t rl = mkbuf(lo + (size_t) t * 6 * B, 6u * B8);
    StepInputs in;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      in.gyro[i] = ldg(ri, i * B8, bo);
      in.accel[i] = ldg(ri, (3 + i) * B8, bo);
      in.z[i] = meas ? ldg(rl, i * B8, bo) : 0.0;
      in.rd[i] = meas ? ldg(rl, (3 + i) * B8, bo) : 1.0;
    }
    in.dt = ldg(ri, 6u * B8, bo);
    in.upd = (b < (unsigned) B) && (mask == nullptr || mask[(size_t) t * B + b] != 0);
    in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
    return in;
  };
  // state-vector slots of xst: x[i] at i, quat at NS + i.  Owner of each entry in the four-wave mapping:
  //   role 1 (CB): x[0..2] omega, x[15..20] biases;  role 2 (PW): x[3..11] v chi Delta, quat;  role 3 (PA): x[12..14] accel
  // The four roles run SEPARATE loops (the same three barriers per iteration in each), see k_replay_coop.
#define PB_REPLAY_QUAD_ROLE(ROLE_ID, MEAS, OWN_EXPR, BODY)                                                            \
  {                                                                                                                   \
    double V[L::NC];                                                                                                  \
    io.template need<SL::QROW[ROLE_ID], SL::QROW[ROLE_ID + 1]>();                                                     \
    static_for<SL::NSLOT>([&](auto I) {                                                                               \
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];                                            \
      if constexpr (comp >= 0 && slot >= SL::T.nq[ROLE_ID] - (SL::QROW[ROLE_ID + 1] - SL::QROW[ROLE_ID]) * 2 &&       \
                    slot < SL::T.nq[ROLE_ID])                                                                         \
        V[comp] = io.ld(comp);                                                                                        \
    });                                                                                                               \
    auto ld = [&V](int comp) { return V[comp]; };                                                                     \
    auto stf = [&V](int comp, double v) { V[comp] = v; };                                                             \
    auto own = [](int i) { return OWN_EXPR; };                                                                        \
    StepInputs nxt = inputs(0, MEAS);                                                                                 \
    for (int t = 0; t < T; t++) {                                                                                     \
      const StepInputs in = nxt;                                                                                      \
      if (t + 1 < T) nxt = inputs(t + 1, MEAS);                                                                       \
      static_for<NS + 4>([&](auto I) {                                                                                \
        constexpr int i = decltype(I)::value;                                                                         \
        if (own(i)) xst[i][lane] = V[i < NS ? L::OFF_VEC + i : L::OFF_QUAT + (i - NS)];                               \
      });                                                                                                             \
      __syncthreads();                                                                                                \
      static_for<NS + 4>([&](auto I) {                                                                                \
        constexpr int i = decltype(I)::value;                                                                         \
        if (!own(i)) V[i < NS ? L::OFF_VEC + i : L::OFF_QUAT + (i - NS)] = xst[i][lane];                              \
      });                                                                                                             \
      BODY;                                                                                                           \
      if (so.base != nullptr) { /* write-through: this role's rows of the posterior of step t */                      \
        TileIO<NS, 0, MemHint<MH_STREAM_NT>::SA> ios(st, so.base + (size_t) t * so.stride, tile, lane);               \
        static_for<SL::NSLOT>([&](auto I) {                                                                           \
          constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];                                        \
          if constexpr (comp >= 0 && slot >= SL::T.nq[ROLE_ID] - (SL::QROW[ROLE_ID + 1] - SL::QROW[ROLE_ID]) * 2 &&   \
                        slot < SL::T.nq[ROLE_ID])                                                                     \
            ios.st(comp, V[comp]);                                                                                    \
        });                                                                                                           \
      }                                                                                                               \
    }                                                                                                                 \
    static_for<SL::NSLOT>([&](auto I) {                                                                               \
      constexpr int slot = decltype(I)::value, comp = SL::T.comp_of[slot];                                            \
      if constexpr (comp >= 0 && slot >= SL::T.nq[ROLE_ID] - (SL::QROW[ROLE_ID + 1] - SL::QROW[ROLE_ID]) * 2 &&       \
                    slot < SL::T.nq[ROLE_ID])                                                                         \
        io.st(comp, V[comp]);                                                                                         \
    });                                                                                                               \
  }
  if (role == 0) PB_REPLAY_QUAD_ROLE(0, true, false, (quad_role_cc<true, false, 0, false>(ld, stf, xw, xrd, sync, in, k)))
  else if (role == 1) PB_REPLAY_QUAD_ROLE(1, false, (i < 3 || (i >= 15 && i < NS)), (quad_role_cb<true>(ld, stf, xw, xrd, sync, in, k)))
  else if (role == 2) PB_REPLAY_QUAD_ROLE(2, false, ((i >= 3 && i < 12) || i >= NS), (quad_role_passive<true, 0>(ld, stf, xw, xrd, sync, in, k)))
  else PB_REPLAY_QUAD_ROLE(3, false, (i >= 12 && i < 15), (quad_role_passive<true, 1>(ld, stf, xw, xrd, sync, in, k)))
#undef PB_REPLAY_QUAD_ROLE
}

// PB_HOST_BROADCAST inputs: dst [rows][B] <- one value per row (pronto_batch.hip stage_in)
struct RowVals {
  static constexpr int MAX = 36;  // the largest block of one call: a full 6 x 6 measurement covariance
  double v[MAX];
};
static __global__ void k_fill_rows(double *__restrict__ dst, int rows, int B, RowVals vals)
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
__global__ void k_window_nll(const double *__restrict__ st, int B, IdxArg<M> idx,
                             const double *__restrict__ tvec, const double *__restrict__ tquat, double *__restrict__ out,
                             double *__restrict__ err_out)
{
  using L = Lay<NS>;
  using S = Slots<NS>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double q[4], tq[4], dchi[3];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    q[i] = st[S::eidx(L::OFF_QUAT + i, b)];
    tq[i] = tquat[(long) i * B + b];
  }
  subtract_quats(q, tq, dchi);
  if (err_out != nullptr) {
    for (int i = 0; i < NS; i++) {
      double e = st[S::eidx(L::OFF_VEC + i, b)] - tvec[(long) i * B + b];
      if (i >= 6 && i <= 8) e = dchi[i - 6];
      err_out[(long) i * B + b] = e;
    }
  }
  double e[M], Sm[M * (M + 1) / 2], d[M];
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    const int ii = idx.v[kk];
    double v = st[S::eidx(L::OFF_VEC + ii, b)] - tvec[(long) ii * B + b];
    if (ii >= 6 && ii <= 8) v = (ii == 6) ? dchi[0] : (ii == 7 ? dchi[1] : dchi[2]);
    e[kk] = v;
#pragma unroll
    for (int j = 0; j <= kk; j++) Sm[pk(kk, j)] = st[S::eidx(L::OFF_P + pk(ii, idx.v[j]), b)];
  }
  ldlt<M>(Sm, d);
  double det = 1.0, maha = 0.0, y[M];
#pragma unroll
  for (int kk = 0; kk < M; kk++) {
    double s = e[kk];
#pragma unroll
    for (int j = 0; j < kk; j++) s -= Sm[pk(kk, j)] * y[j];
    y[kk] = s;
    det *= d[kk];
    maha += s * s / d[kk];
  }
  const double logdet = log(det);
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
// (the cascade kernel itself is k_notch_counts, rbis_frontend.hpp: one lane per (filter, axis), an optional per-filter packet count)

// Counter calibration: a plain copy with EXACTLY the access pattern of the step kernels (buffer_load/store_dwordx4,
// 16 bytes per lane, one tile per wave, all loads of a chunk before its stores), so that rocprofv3's FETCH_SIZE /
// WRITE_SIZE can be scaled on a known byte count, and the copy ceiling of this access pattern measured.
static __global__ __launch_bounds__(64, 1) void k_calib_copy(const double *__restrict__ src, double *__restrict__ dst, int B,
                                                      int nrow)
{
  const unsigned tile = blockIdx.x;
  if (tile * 64u >= (unsigned) B) return;
  const unsigned tb = (unsigned) nrow * 1024u;
  const rsrc_t ri = mkbuf(reinterpret_cast<const char *>(src) + (size_t) tile * tb, tb);
  const rsrc_t ro = mkbuf(reinterpret_cast<char *>(dst) + (size_t) tile * tb, tb);
  const unsigned vo = threadIdx.x * 16u;
  for (int r0 = 0; r0 < nrow; r0 += 35) {
    d2_t v[35];
#pragma unroll
    for (int i = 0; i < 35; i++) v[i] = (r0 + i < nrow) ? ldg2(ri, (unsigned) (r0 + i) * 1024u, vo) : d2_t{ 0, 0 };
#pragma unroll
    for (int i = 0; i < 35; i++)
      if (r0 + i < nrow) stg2(ro, (unsigned) (r0 + i) * 1024u, vo, v[i]);
  }
}

}  // namespace pb
