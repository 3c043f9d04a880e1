// rbis_smooth_wide.hpp -- the RTS smoother step for 15 states (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) with
// ONE LANE PER FILTER, four role waves per 64-filter tile and ONE wave per SIMD: 512 registers per lane (round 5).
//
//   G      = P_k Ad^T (P^-_{k+1})^-1      P^s_k = P_k + G (P^s_{k+1} - P^-_{k+1}) G^T      x^s_k = x_k (+) G (x^s_{k+1} (-) x^-_{k+1})
//
// k_smooth_lane<15> (rbis_smooth_lane.hpp) is a LATENCY CHAIN, not a throughput problem (profiles/r05_smoother_chain.txt: a tile alone on an
// idle GPU takes 73 k cycles, half of them in the loop that hands the rows of M = G D round two at a time -- two of four roles make a
// row while the others wait, 16 barriers -- and every checkpoint is asked for when it is needed, four exposed round trips to memory).
// With 256 registers a role cannot hold its rows of G AND of M; with 512 it can, and the chain becomes:
//   0. every global load of the tile is issued at the top: the role's columns of P^-, of P^s_{k+1} and of P_k (each checkpoint is read
//      ONCE; D = P^s - P^- is formed column by column in registers as soon as both are there);
//   1. P^- = L diag(d) L^T, right-looking, unpivoted, published column by column into LDS [packed entry][lane] (as k_smooth_lane);
//   2. right-hand sides Ad P_k in registers, both substitutions out of the LDS factor: the role's rows of G;
//   3. D takes the factor's place; every role makes ALL its rows of M = G D in one sweep over D (8 multiply-adds per LDS read);
//   4. the rows of M are published in two halves into the same LDS (8 rows x 15 = 120 entries = the factor's place) and
//      P^s[r][c] = P_k[r][c] + G[r] . M[c] is finished from the role's registers: P_k(r, c) is the copy loaded in step 0.
// 20 + 6 barriers; LDS (n (n + 1) / 2 + 3 n + 1) doubles per lane = 85 KB per tile; one tile per CU (the registers decide).
// Role ownership, the LDS layout, the read fences and the stand-in column are k_smooth_lane's; so is the arithmetic of steps 1-2.
#pragma once

#include <hip/hip_runtime.h>

#include "rbis_device.hpp"
#include "rbis_kernels.hpp"
#include "rbis_smooth_lane.hpp"

namespace pb {

// rows of the factor / of D per group of LDS reads (the next group's reads wait for this group's arithmetic: WFENCE), columns of the
// final product per group
#ifndef SMW_NR
#define SMW_NR 4   // role waves per tile: 4 (one per SIMD, 512 registers) or 8 (two per SIMD, 256 registers)
#endif
#ifndef SMW_SUB_G
#define SMW_SUB_G 2
#endif
#ifndef SMW_M_G
#define SMW_M_G 2
#endif
#ifndef SMW_FIN_G
#define SMW_FIN_G 1
#endif


// ---- LDS reads as an explicit pipeline -------------------------------------------------------------------------------------------
// One wave per SIMD has nobody to hide an LDS round trip behind, and the backend schedules `read, wait, use, read, wait, use` once the
// accumulators fill the 256 architectural registers (the M sweep: 60 exposed round trips).  lds_stream reads a compile-time list of
// entries G at a time into two buffers with ds_read_b64 of its own (single reads: the LDS serves two of them in half the time of the
// paired ds_read2st64_b64 the backend prefers), the reads of group g + 1 issued BEFORE the wait for group g (a counted s_waitcnt: LDS
// operations of a wave return in order; a scalar load the backend may have in flight can only make the wait longer, never shorter).
// `use(k, value)` is called for k = 0 .. N-1 in order with k a compile-time constant; `pin()` after every group: it names what the
// group's arithmetic wrote (lane_pin), which keeps that arithmetic in front of the next group's reads -- the backend would otherwise
// let all the reads of the list go first and park their values in accumulation registers.
template <int OFF>
__device__ __forceinline__ void lds_rd_b64(double &d, int byte_base)
{
  static_assert(OFF >= 0 && OFF < 65536, "immediate offset of an LDS instruction");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(byte_base), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void lds_wait8(double (&b)[8])
{
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               : "n"(CNT));
}
template <int N, class EntryOf, class Use, class Pin>
__device__ __forceinline__ void lds_stream(int byte_base0, int byte_base1, Use &&use, Pin &&pin)
{
  constexpr int G = 8, NG = (N + G - 1) / G;
  double buf[2][G];
#pragma unroll
  for (int j = 0; j < G; j++) buf[0][j] = buf[1][j] = 0.0;
  auto issue = [&](auto GG) {
    constexpr int g = decltype(GG)::value;
    static_for<G>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value, kq = g * G + j;
      if constexpr (kq < N) {
        constexpr int e = EntryOf::at(kq);
        if constexpr (e < 128) lds_rd_b64<e * 512>(buf[g & 1][j], byte_base0);
        else lds_rd_b64<(e - 128) * 512>(buf[g & 1][j], byte_base1);
      }
    });
  };
  issue(std::integral_constant<int, 0>{});
  static_for<NG>([&](auto GG) {
    constexpr int g = decltype(GG)::value;
    constexpr int next = (g + 1 < NG) ? ((N - (g + 1) * G < G) ? N - (g + 1) * G : G) : 0;
    if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
    lds_wait8<next>(buf[g & 1]);
    static_for<G>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value, kq = g * G + j;
      if constexpr (kq < N) use(std::integral_constant<int, kq>{}, buf[g & 1][j]);
    });
    pin();
  });
}
// entry lists
struct SmwLowerStrict {  // k-th (i, m), m < i, row by row: the forward substitution's order
  static constexpr int row(int k) { return pk_row(k) + 1; }
  static constexpr int col(int k) { return pk_col(k); }
  static constexpr int at(int k) { return pk(row(k), col(k)); }
};
template <int NS>
struct SmwBackward {  // k-th (m, i), i = n-2 .. 0, m = i+1 .. n-1: the backward substitution's order
  static constexpr int ci(int k) { int i = NS - 2; while (k >= NS - 1 - i) { k -= NS - 1 - i; i--; } return i; }
  static constexpr int cm(int k) { int i = NS - 2; while (k >= NS - 1 - i) { k -= NS - 1 - i; i--; } return i + 1 + k; }
  static constexpr int at(int k) { return pk(cm(k), ci(k)); }
};
struct SmwDiag { static constexpr int at(int k) { return pk(k, k); } };
struct SmwPacked { static constexpr int at(int k) { return k; } };
template <int BASE>
struct SmwRun { static constexpr int at(int k) { return BASE + k; } };
template <int NS, int ROWS>
struct SmwFinal {  // ROWS rows of NS entries, two rows at a time, the pair's entries interleaved (row a, j), (row a + 1, j); a last single row plain
  static constexpr int pair(int k) { return k / (2 * NS); }
  static constexpr bool single(int k) { return 2 * pair(k) + 1 >= ROWS; }
  static constexpr int row(int k) { return single(k) ? 2 * pair(k) : 2 * pair(k) + (k % (2 * NS)) % 2; }
  static constexpr int col(int k) { return single(k) ? k % (2 * NS) : (k % (2 * NS)) / 2; }
  static constexpr int at(int k) { return row(k) * NS + col(k); }
  static constexpr bool last_of_pair(int k) { return single(k) ? (k % (2 * NS)) == NS - 1 : (k % (2 * NS)) == 2 * NS - 1; }
};

template <int NS>
struct SmoothWideCfg {
  using L = Lay<NS>;
  using SL = Slots<NS>;
  static constexpr int NR = SMW_NR;                  // role waves per tile
  static constexpr int NCOL = (NS + NR - 1) / NR;    // columns / gain rows per role
  static constexpr int NP = L::NP;
  static constexpr int HALF = NP / NS;               // rows of M per publish (n = 15: 8)
  static constexpr int O_X = NP;                     // behind the factor / D / M: residual [n], dx [n]
  static constexpr int O_S = NP + 2 * NS;            // the filtered state vector [n] and log-likelihood, parked by the last role until the end
  static constexpr int PER = NP + 3 * NS + 1;        // doubles per lane
  static constexpr int THREADS = 64 * NR;
  static constexpr size_t LDS_BYTES = sizeof(double) * PER * 64;
  static_assert(NS == 15, "15 states only: 21 states need 231 entries for D and have no room for half of M (rbis_smooth_lane.hpp stays)");
  static_assert(2 * HALF >= NS && HALF * NS <= NP && HALF % NR == 0, "two publishes cover every row of M inside the factor's place");
  static constexpr int off_of(int comp) { return (SL::T.slot_of[comp] / 2) * 128 + (SL::T.slot_of[comp] % 2); }
  static constexpr int NSP = (NS + 3) & ~3;
  struct Tab {
    int col[NR][NCOL][NSP];   // col[w][t][i] = offset of P(i, column w + NR t) -- by symmetry also of P(row w + NR t, i)
  };
  static constexpr Tab make()
  {
    Tab t{};
    for (int w = 0; w < NR; w++)
      for (int c = 0; c < NCOL; c++) {
        const int j = (w + NR * c < NS) ? w + NR * c : NS - 1;
        for (int i = 0; i < NSP; i++) t.col[w][c][i] = off_of(L::OFF_P + pk(i < NS ? i : NS - 1, j));
      }
    return t;
  }
};
template <int NS>
__constant__ const typename SmoothWideCfg<NS>::Tab smooth_wide_tab = SmoothWideCfg<NS>::make();

#ifdef SML_TIMELINE
#define SMW_T(i) do { if (stamp) tl[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMW_T(i) do { } while (0)
#endif

template <int NS>
__global__ __launch_bounds__(SmoothWideCfg<NS>::THREADS, SmoothWideCfg<NS>::NR / 4) void k_smooth_wide(const double *__restrict__ next_pred, const double *__restrict__ next_sm,
                                                                             const double *cur, double *out, int B, int ntiles, double dt, Consts k)
{
  using L = Lay<NS>;
  using SL = Slots<NS>;
  using C = SmoothWideCfg<NS>;
  constexpr int NR = C::NR, NCOL = C::NCOL, O_X = C::O_X, HALF = C::HALF;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int w0 = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  int w = w0;   // (made opaque again at the top of every tile: see the loop)
  // Global accesses are buffer instructions: a descriptor per checkpoint AND TILE (scalar registers), the component's offset in the
  // scalar offset, the lane's 16-byte column in ONE vector register for every access -- no address registers and no vector arithmetic
  // per access (with 64-bit per-lane addresses the 200 loads at the top of a tile held two registers each just for their addresses).
  const unsigned lane_b = (unsigned) lane * 16u;
  struct TileBufs { rsrc_t np, ns, cu, out; };
  auto bufs_of = [&](int tile) {
    const long off = (long) tile * SL::TILE_DOUBLES;
    return TileBufs{ mkbuf(next_pred + off, SL::TILE_BYTES), mkbuf(next_sm + off, SL::TILE_BYTES), mkbuf(cur + off, SL::TILE_BYTES),
                     mkbuf(out + off, SL::TILE_BYTES) };
  };
  TileBufs tbuf = bufs_of((int) blockIdx.x);
  const auto &tab = smooth_wide_tab<NS>;
  // entry e of this lane's filter: lds[sb + e * 64]; two bases so that every access keeps an immediate offset (rbis_smooth_lane.hpp)
  int sb = lane, sb1 = lane + 128 * 64;
  asm volatile("" : "+v"(sb), "+v"(sb1));
  // the same two bases as LDS byte addresses, for lds_stream (dynamic LDS starts at 0: no static LDS in this kernel)
  const int bb0 = lane * 8, bb1 = lane * 8 + 128 * 512;
#define WS (lds + sb)
#define WE(e) (*(((e) < 128) ? (lds + sb + (e) * 64) : (lds + sb1 + ((e) - 128) * 64)))
#define WFENCE(after) asm volatile("" : "+v"(sb), "+v"(sb1) : "v"(after) : "memory")
  auto ldc = [&](rsrc_t src, int comp) { return ldg(src, (unsigned) C::off_of(comp) * 8u, lane_b); };
  auto ld_col = [&](rsrc_t src, int t, int i0, double (&v)[NS]) {  // rows i0 .. n-1 of column t of this role (the others: 0)
    int o[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) o[i] = tab.col[w][t][i];
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = (i >= i0) ? ldg(src, (unsigned) o[i] * 8u, lane_b) : 0.0;
  };

#ifdef SML_TIMELINE
  unsigned long long tl[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif

  // The workgroup is PERSISTENT: tiles blockIdx.x, + gridDim.x, ... (one workgroup per CU fills the registers anyway).  What the
  // factorisation starts from -- the role's columns of P^- -- is requested one tile AHEAD (behind step 4, when the registers of D are
  // free), so a tile begins to compute at once; its other two checkpoints are requested at its top and arrive behind the
  // factorisation.  Without this every CU asks for its whole tile at the same moment and then computes with the memory idle.
  double a[NCOL][NS];
#pragma unroll
  for (int t = 0; t < NCOL; t++) ld_col(tbuf.np, t, NR * t, a[t]);
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  tbuf = bufs_of(tile);
  // The role index is made opaque per tile: what depends on it (table offsets, LDS indices of the role's columns, ~300 scalars) would
  // otherwise be computed ONCE in front of the loop and kept -- in scalar registers the kernel does not have (spilled to vector lanes
  // and from there to scratch).
  w = w0;
  asm volatile("" : "+s"(w));
  int cidx[NCOL], cc[NCOL];  // this role's columns (gain rows); the stand-in mirrors column n - 1
#pragma unroll
  for (int t = 0; t < NCOL; t++) {
    cidx[t] = w + NR * t;
    cc[t] = cidx[t] < NS ? cidx[t] : NS - 1;
  }
  const bool has_last = cidx[NCOL - 1] < NS;  // (wave-uniform) the role's last column slot is a real column
  const bool active = (long) tile * 64 + lane < B;
  lds_barrier();  // the previous tile's readers of the LDS are done
#ifdef SML_TIMELINE
  const bool stamp = tile == SML_TIMELINE;
#endif
  SMW_T(0);

  // The prefetched columns are TAKEN here, in front of this tile's loads: the counter of outstanding memory operations has 6 bits, and
  // a wait for these (old) loads placed behind 100 younger ones can only be written as "at most 62 outstanding" -- it then waits for
  // half of the tile's fresh loads as well (measured: 12 k cycles at the top of every tile).
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int i = NR * t; i < NS; i++) lane_pin(a[t][i]);
  SMW_T(11);

  // ---- 0. the loads of the tile that the factorisation can hide: what the last role needs for the residual and the state update
  //         (FIRST: loads return in order, and it is the one that consumes early), the filtered state for Ad and the role's columns of
  //         P_k (whole: the right-hand side needs the column, step 7 its part left of the diagonal).  The columns of P^s_{k+1} are asked
  //         for behind the factorisation, when the registers of `a` are free ----
  double dcol[NCOL][NS], p[NCOL][NS], a0[NCOL][NS];
  // (the state vectors are shared out: role w asks for components w, w + NR, ... of the three checkpoints -- all of them on one role
  //  were 54 loads in flight on top of its columns, which the backend serialised into 15 round trips to memory for want of registers)
  double rqs[4], rqp[4], rvs[NCOL], rvp[NCOL], xc[NCOL], llc = 0.0;   // (raw: consumed behind the factorisation, nothing waits here)
  static_for<NR>([&](auto WW) {
    constexpr int ww = decltype(WW)::value;
    if (w == ww) {
#pragma unroll
      for (int t = 0; t < NCOL; t++) {
        const int i = ww + NR * t;   // (compile-time inside this branch)
        if (i < NS) {
          rvs[t] = ldc(tbuf.ns, L::OFF_VEC + i);
          rvp[t] = ldc(tbuf.np, L::OFF_VEC + i);
          xc[t] = ldc(tbuf.cu, L::OFF_VEC + i);
        } else {
          rvs[t] = rvp[t] = xc[t] = 0.0;
        }
      }
    }
  });
  if (w == NR - 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      rqs[i] = ldc(tbuf.ns, L::OFF_QUAT + i);
      rqp[i] = ldc(tbuf.np, L::OFF_QUAT + i);
    }
    llc = ldc(tbuf.cu, L::OFF_LL);
  }
  double wv[3], vv[3], q[4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    wv[i] = ldc(tbuf.cu, L::OFF_VEC + i);
    vv[i] = ldc(tbuf.cu, L::OFF_VEC + 3 + i);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = ldc(tbuf.cu, L::OFF_QUAT + i);
#pragma unroll
  for (int t = 0; t < NCOL; t++) ld_col(tbuf.cu, t, 0, p[t]);
  // the uncorrected P^- for D = P^s - P^- (rbis.cpp:256): a copy, the factorisation works on `a`
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int i = 0; i < NS; i++) a0[t][i] = a[t][i];

  SMW_T(12);
  // ---- 1. P^- = L diag(d) L^T (k_smooth_lane's step 1) ----
  {
    double inv_prev = 0.0;
    static_for<NS>([&](auto KK) {
      constexpr int kk = decltype(KK)::value;
      if (w == kk % NR) {  // owner of column kk
        constexpr int t = kk / NR;
        const double d = a[t][kk];
        const double inv = (fabs(d) > 5.562684646268003e-309) ? 1.0 / d : 0.0;
        WE(pk(kk, kk)) = (kk == NS - 1) ? inv : d;
#pragma unroll
        for (int i = kk + 1; i < NS; i++) WE(pk(i, kk)) = a[t][i] * inv;
        inv_prev = inv;
      }
      lds_barrier();
      if constexpr (kk == 0) SMW_T(1);
      if constexpr (kk == NS - 1) SMW_T(2);
      // the diagonal slot of column kk-1 held d for the downdates of step kk-1; every role is past them now: it becomes 1/d
      if constexpr (kk > 0)
        if (w == (kk - 1) % NR) WE(pk(kk - 1, kk - 1)) = inv_prev;
      if constexpr (kk + 1 < NS) {
        const double dk = WE(pk(kk, kk));
        double tc[NCOL];
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (kk < NR * (t + 1) - 1) tc[t] = WS[pk_s(cc[t], kk) * 64] * dk;
#pragma unroll
        for (int i = kk + 1; i < NS; i++) {
          const double lik = WE(pk(i, kk));
#pragma unroll
          for (int t = 0; t < NCOL; t++)
            if (kk < NR * (t + 1) - 1 && i >= NR * t) {
              a[t][i] = fma(-lik, tc[t], a[t][i]);
              lane_pin(a[t][i]);  // (downdated NOW, not when the column is published)
            }
        }
      }
    });
  }

  // residual x^s (-) x^- (rbis.cpp:258-261) for step 3 and the filtered state for step 8, parked in LDS (registers are scarce from here on)
#pragma unroll
  for (int t = 0; t < NCOL; t++)
    if (cidx[t] < NS) {
      if (cidx[t] < 6 || cidx[t] > 8) WS[(O_X + cidx[t]) * 64] = rvs[t] - rvp[t];
      WS[(C::O_S + cidx[t]) * 64] = xc[t];
    }
  if (w == NR - 1) {
    double dchi[3];
    subtract_quats(rqs, rqp, dchi);
#pragma unroll
    for (int i = 0; i < 3; i++) WE(O_X + 6 + i) = dchi[i];
    WE(C::O_S + NS) = llc;
  }
#pragma unroll
  for (int t = 0; t < NCOL; t++) ld_col(tbuf.ns, t, NR * t, dcol[t]);

  // ---- 2. right-hand sides: columns cc[t] of Ad P_k, Ad = I + dt Ac about the filtered state (rbis.cpp:12-35, 236-239) ----
  double z[NCOL][NS];
  {
    double R[9];
    quat_to_rot(q, R);
    const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };
#pragma unroll
    for (int t = 0; t < NCOL; t++) {
#pragma unroll
      for (int i = 0; i < NS; i++) z[t][i] = p[t][i];
      const double pv[3] = { p[t][3], p[t][4], p[t][5] }, pc[3] = { p[t][6], p[t][7], p[t][8] };
      const double wxpv[3] = { wv[1] * pv[2] - wv[2] * pv[1], wv[2] * pv[0] - wv[0] * pv[2], wv[0] * pv[1] - wv[1] * pv[0] };
      const double gxpc[3] = { gb[1] * pc[2] - gb[2] * pc[1], gb[2] * pc[0] - gb[0] * pc[2], gb[0] * pc[1] - gb[1] * pc[0] };
      const double wxpc[3] = { wv[1] * pc[2] - wv[2] * pc[1], wv[2] * pc[0] - wv[0] * pc[2], wv[0] * pc[1] - wv[1] * pc[0] };
      const double vxpc[3] = { vv[1] * pc[2] - vv[2] * pc[1], vv[2] * pc[0] - vv[0] * pc[2], vv[0] * pc[1] - vv[1] * pc[0] };
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double av = -wxpv[i] + gxpc[i], ac = -wxpc[i];
        const double ad = R[3 * i] * (pv[0] - vxpc[0]) + R[3 * i + 1] * (pv[1] - vxpc[1]) + R[3 * i + 2] * (pv[2] - vxpc[2]);
        z[t][3 + i] = fma(dt, av, z[t][3 + i]);
        z[t][6 + i] = fma(dt, ac, z[t][6 + i]);
        z[t][9 + i] = fma(dt, ad, z[t][9 + i]);
      }
    }
  }
  lds_barrier();  // the factor is complete (the last reciprocal pivots were written behind the last barrier of step 1)
  SMW_T(3);

  // ---- 3. (P^-) X = Ad P_k out of the LDS factor: z[t][:] becomes row cidx[t] of G ----
  auto pin_z = [&]() {
#pragma unroll
    for (int t = 0; t < NCOL; t++)
#pragma unroll
      for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
  };
  lds_stream<NS *(NS - 1) / 2, SmwLowerStrict>(bb0, bb1, [&](auto KQ, double l) {
    constexpr int kq = decltype(KQ)::value, i = SmwLowerStrict::row(kq), mm = SmwLowerStrict::col(kq);
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
  }, pin_z);
  lds_stream<NS, SmwDiag>(bb0, bb1, [&](auto KQ, double inv) {
    constexpr int i = decltype(KQ)::value;
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] *= inv;
  }, pin_z);
  lds_stream<NS *(NS - 1) / 2, SmwBackward<NS>>(bb0, bb1, [&](auto KQ, double l) {
    constexpr int kq = decltype(KQ)::value, i = SmwBackward<NS>::ci(kq), mm = SmwBackward<NS>::cm(kq);
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
  }, pin_z);
  // dx = G resid (rbis.cpp:263): this role's entries
  double dxv[NCOL];
#pragma unroll
  for (int t = 0; t < NCOL; t++) dxv[t] = 0.0;
  lds_stream<NS, SmwRun<O_X>>(bb0, bb1, [&](auto KQ, double r) {
    constexpr int i = decltype(KQ)::value;
#pragma unroll
    for (int t = 0; t < NCOL; t++) dxv[t] = fma(z[t][i], r, dxv[t]);
  }, [&]() {
#pragma unroll
    for (int t = 0; t < NCOL; t++) lane_pin(dxv[t]);
  });
#pragma unroll
  for (int t = 0; t < NCOL; t++) {
    lane_pin(dxv[t]);
#pragma unroll
    for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
  }
  lds_barrier();  // factor and residual are dead
  SMW_T(4);

  // ---- 4. D = P^s - P^- (the uncorrected P^-, rbis.cpp:256) takes the factor's place, column by column from the registers of step 0
  //         (an entry above the role's diagonal is the mirror image of another role's entry: the same bits, written twice) ----
#pragma unroll
  for (int t = 0; t < NCOL; t++) {
    if (cidx[t] < NS) WS[(O_X + NS + cidx[t]) * 64] = dxv[t];
#pragma unroll
    for (int i = NR * t; i < NS; i++) WS[pk_s(i, cc[t]) * 64] = dcol[t][i] - a0[t][i];
  }
  {  // the NEXT tile's columns of P^- (the last tile asks for its own again: the same instruction stream for every tile)
    const int ntile = (tile + (int) gridDim.x < ntiles) ? tile + (int) gridDim.x : tile;
    const rsrc_t npn = mkbuf(next_pred + (long) ntile * SL::TILE_DOUBLES, SL::TILE_BYTES);
#pragma unroll
    for (int t = 0; t < NCOL; t++) ld_col(npn, t, NR * t, a[t]);
  }
  lds_barrier();
  SMW_T(5);

  // ---- 6. M = G D: ALL rows of the role in one sweep over the symmetric D (every entry read once, 2 NCOL multiply-adds each) ----
  double m[NCOL][NS];
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int j = 0; j < NS; j++) m[t][j] = 0.0;
  // in passes of two rows: with all four the accumulators and G fill the architectural registers and nothing is left to read ahead into
  auto sweep = [&](auto T0, auto NT) {
    constexpr int t0 = decltype(T0)::value, nt = decltype(NT)::value;
    lds_stream<C::NP, SmwPacked>(bb0, bb1, [&](auto KQ, double d) {
      constexpr int kq = decltype(KQ)::value, i = pk_row(kq), j = pk_col(kq);
#pragma unroll
      for (int t = t0; t < t0 + nt; t++) {
        m[t][j] = fma(z[t][i], d, m[t][j]);
        if (i != j) m[t][i] = fma(z[t][j], d, m[t][i]);
      }
    }, [&]() {
#pragma unroll
      for (int t = t0; t < t0 + nt; t++)
#pragma unroll
        for (int j = 0; j < NS; j++) lane_pin(m[t][j]);
    });
  };
  static_assert(NCOL == 4, "two passes of two rows");
  sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
  if (has_last) sweep(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
  else sweep(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int j = 0; j < NS; j++) lane_pin(m[t][j]);

  // ---- 7. P^s[r][c] = P_k[r][c] + G[r] . M[c] for c <= r: the rows of M in two halves through the place of D ----
  static_for<2>([&](auto HH) {
    constexpr int h = decltype(HH)::value;
    lds_barrier();  // D (h = 0) / the first half of M (h = 1) is dead
    if constexpr (h == 0) SMW_T(6); else SMW_T(9);
#pragma unroll
    for (int t = 0; t < NCOL; t++) {
      // row cidx[t] is in half h for the column slots with HALF h <= NR t + w < HALF (h + 1); HALF is a multiple of NR
      if ((t * NR) / HALF == h && cidx[t] < NS) {
#pragma unroll
        for (int j = 0; j < NS; j++) WS[((cidx[t] - HALF * h) * NS + j) * 64] = m[t][j];
      }
    }
    lds_barrier();
    if constexpr (h == 0) SMW_T(8); else SMW_T(10);
    // the offsets of this half's stores, fetched HERE (a scalar load in the middle of lds_stream's pipeline would stall it: one counter)
    int so[HALF][NCOL];
#pragma unroll
    for (int t = 0; t < NCOL; t++)
#pragma unroll
      for (int cq = 0; cq < HALF; cq++) so[cq][t] = smooth_wide_tab<NS>.col[w][t][HALF * h + cq];   // (8 neighbours: wide loads)
#pragma unroll
    for (int t = 0; t < NCOL; t++)
#pragma unroll
      for (int cq = 0; cq < HALF; cq++)
        if (HALF * h + cq < NS && t >= (HALF * h + cq) / NR) asm volatile("" : "+s"(so[cq][t]));   // (all in flight together, THEN pinned)
    // ONE stream over the half: columns two at a time, their rows of M interleaved entry by entry (2 x up to NCOL sums in flight: a
    // multiply-add that waits for its predecessor costs twice its issue slot); the role's rows r = w + NR t at or below the column.
    // Compile time: t >= c / NR; run time: c <= r.  A pair's sums are stored as soon as its last entry is in.
    constexpr int rows_h = (HALF * (h + 1) <= NS) ? HALF : NS - HALF * h;
    using FE = SmwFinal<NS, rows_h>;
    double acc[HALF][NCOL];
#pragma unroll
    for (int cq = 0; cq < HALF; cq++)
#pragma unroll
      for (int t = 0; t < NCOL; t++) acc[cq][t] = (cq < rows_h) ? p[t][HALF * h + cq < NS ? HALF * h + cq : 0] : 0.0;
    lds_stream<rows_h * NS, FE>(bb0, bb1, [&](auto KQ, double mv) {
      constexpr int kq = decltype(KQ)::value, cq = FE::row(kq), j = FE::col(kq), c = HALF * h + cq;
#pragma unroll
      for (int t = 0; t < NCOL; t++)
        if (t >= c / NR) acc[cq][t] = fma(z[t][j], mv, acc[cq][t]);
      if constexpr (FE::last_of_pair(kq)) {
#pragma unroll
        for (int cs = (cq & ~1); cs <= cq; cs++)
#pragma unroll
          for (int t = 0; t < NCOL; t++)
            if (t >= (HALF * h + cs) / NR) {
              lane_pin(acc[cs][t]);
              if (active && cidx[t] < NS && HALF * h + cs <= cidx[t]) stg(tbuf.out, (unsigned) so[cs][t] * 8u, lane_b, acc[cs][t]);
            }
      }
    }, [&]() {
#pragma unroll
      for (int cq = 0; cq < rows_h; cq++)
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (t >= (HALF * h + cq) / NR) lane_pin(acc[cq][t]);
    });
  });
  // ---- 5. state: cur.addState(RBIS(dx)) (rbis.cpp:263-265), by the LAST role, at the end (nobody waits for it at a barrier) ----
  if (w == NR - 1) {
    double dchi[3] = { WE(O_X + NS + 6), WE(O_X + NS + 7), WE(O_X + NS + 8) };
    double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
    fold_chi(dchi, dq, k.chi_tol);  // RBIS(vec) constructor
    double chi[3], qq[4], qo[4];
#pragma unroll
    for (int i = 0; i < 3; i++) chi[i] = WE(C::O_S + 6 + i) + dchi[i];
#pragma unroll
    for (int i = 0; i < 4; i++) qq[i] = q[i];
    fold_chi(chi, qq, k.chi_tol);
    quat_mul(qq, dq, qo);
    double xo[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) xo[i] = (i >= 6 && i <= 8) ? chi[i - 6] : WE(C::O_S + i) + WE(O_X + NS + i);
    const double ll = WE(C::O_S + NS);
    if (active) {
#pragma unroll
      for (int i = 0; i < NS; i++) stg(tbuf.out, (unsigned) C::off_of(L::OFF_VEC + i) * 8u, lane_b, xo[i]);
#pragma unroll
      for (int i = 0; i < 4; i++) stg(tbuf.out, (unsigned) C::off_of(L::OFF_QUAT + i) * 8u, lane_b, qo[i]);
      stg(tbuf.out, (unsigned) C::off_of(L::OFF_LL) * 8u, lane_b, ll);
    }
  }

#ifdef SML_TIMELINE
  if (stamp) tl[7] = __builtin_amdgcn_s_memtime();
  if (stamp && lane == 0)
    for (int i = 0; i < 16; i++) sml_tl[w][i] = tl[i];
#endif
  }  // tiles
}

#undef SMW_T
#undef WS
#undef WE
#undef WFENCE

}  // namespace pb
