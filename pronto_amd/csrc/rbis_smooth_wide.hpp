// rbis_smooth_wide.hpp -- the RTS smoother step for 15 states (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) with
// ONE LANE PER FILTER, four role waves per 64-filter tile, ONE wave per SIMD (512 registers per lane) and a PERSISTENT workgroup per CU
// that walks the tiles (round 5).
//
//   G      = P_k Ad^T (P^-_{k+1})^-1      P^s_k = P_k + G (P^s_{k+1} - P^-_{k+1}) G^T      x^s_k = x_k (+) G (x^s_{k+1} (-) x^-_{k+1})
//
// k_smooth_lane<15> (rbis_smooth_lane.hpp) is a LATENCY CHAIN, not a throughput problem (profiles/r05_smoother_chain.txt: a tile alone on an
// idle GPU takes 73 k cycles, half of them in the loop that hands the rows of M = G D round two at a time -- two of four roles make a
// row while the others wait, 16 barriers -- and every checkpoint is asked for when it is needed, four exposed round trips to memory).
// With 256 registers a role cannot hold its rows of G AND of M; with 512 it can, and the chain becomes:
//   0. data movement by whole 16-byte ROWS of the tile (role w moves rows w, w + 4, ...: 18 loads per checkpoint and role instead of 40-60
//      8-byte ones -- a wave may have 64 memory instructions in flight, and the texture path handles a half-used line as slowly as a full
//      one), redistributed through LDS [entry][lane]; each checkpoint is read ONCE, the posterior is stored by whole rows;
//      the rows of P^- of the NEXT tile are requested while this one computes (the factorisation can start at once), P_k and P^s_{k+1}
//      of this tile arrive behind the factorisation;
//   1. P^- = L diag(d) L^T, right-looking, unpivoted, published column by column into LDS in place (as k_smooth_lane);
//   2. right-hand sides Ad P_k in registers, both substitutions out of the LDS factor: the role's rows of G;
//   3. D = P^s - P^- takes the factor's place; every role makes ALL its rows of M = G D (two sweeps of two rows over D);
//   4. the rows of M are published in two halves into the same LDS (8 rows x 15 = 120 entries = the factor's place) and
//      P^s[r][c] = P_k[r][c] + G[r] . M[c] is finished from the role's registers into the output staging area.
// LDS reads of the long phases are an explicit pipeline (lds_stream).  LDS per tile: 307 entries of 512 bytes = 154 KB; barriers order
// LDS only (lds_barrier).  Role ownership, the [entry][lane] layout and the stand-in column are k_smooth_lane's, so is the arithmetic.
#pragma once

#include <hip/hip_runtime.h>

#include "rbis_device.hpp"
#include "rbis_kernels.hpp"
#include "rbis_smooth_lane.hpp"

namespace pb {

// cache-policy bits of the row loads / stores (rbis_kernels.hpp: 1 = sc0, 2 = nt, 16 = sc1).  Every checkpoint is read once and the
// posterior written once: non-temporal both ways, 85.7 us against 87.4-88.5 at 64k filters (loads only: 90.4, stores only: 86.7,
// nt loads + sc1 stores: 87.9), no worse at 4k-128k filters and in pb_smooth_log
#ifndef SMW_LOAD_AUX
#define SMW_LOAD_AUX 2
#endif
#ifndef SMW_STORE_AUX
#define SMW_STORE_AUX 2
#endif
#ifndef SMW_DEPTH
#define SMW_DEPTH 1   // groups of LDS reads in flight beside the one consumed, in the long phases (2: 91.0 us against 88.1, 3: 91.4 -- registers)
#endif

template <int NS>
struct SmoothWideCfg {
  using L = Lay<NS>;
  using SL = Slots<NS>;
  static constexpr int NR = 4;                       // role waves per tile, one per SIMD
  static constexpr int NCOL = (NS + NR - 1) / NR;    // columns / gain rows per role
  static constexpr int NP = L::NP;
  static constexpr int HALF = NP / NS;               // rows of M per publish (n = 15: 8)
  // LDS entries (64 doubles each):
  static constexpr int O_A = 0;                      // P^- -> its factor -> D -> half of M
  static constexpr int O_B = NP;                     // [covariance entries | vec, quat, ll] of the filtered checkpoint, then of the posterior
  static constexpr int NST = NS + 5;                 // vec, quat, ll of one checkpoint, canonical order
  static constexpr int NB = NP + NST;
  static constexpr int O_SN = O_B + NB;              // vec, quat of x^s_{k+1}; the residual takes the place of its vec
  static constexpr int O_SP = O_SN + NST;            // vec, quat of x^-_{k+1}; dx takes its place
  static constexpr int O_X = O_SN;                   // residual [n]
  static constexpr int O_DX = O_SP;                  // dx [n]
  static constexpr int DUMMY = O_SP + NST;           // where the halves of a row that nobody wants go
  static constexpr int PER = DUMMY + 1;
  static constexpr int THREADS = 64 * NR;
  static constexpr size_t LDS_BYTES = sizeof(double) * PER * 64;
  static_assert(NS == 15, "15 states only: 21 states need 231 entries for D alone (rbis_smooth_lane.hpp stays)");
  static_assert(2 * HALF >= NS && HALF * NS <= NP && HALF % NR == 0, "two publishes cover every row of M inside the factor's place");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
  static constexpr int off_of(int comp) { return (SL::T.slot_of[comp] / 2) * 128 + (SL::T.slot_of[comp] % 2); }
  static constexpr int RU = (SL::NROW + NR - 1) / NR;   // rows per role
  // the order in which a role asks for its rows of P^s_{k+1}: the four that carry the state vector first (the residual is made during
  // the factorisation), then the rest
  static constexpr int ns_order(int k) { return k == 0 ? 0 : k == 1 ? 1 : k == 2 ? RU - 2 : k == 3 ? RU - 1 : k - 2; }
  // Where the two halves of row w + NR u go in LDS, for each use of a row (ONE table row per use and role: 36 neighbouring words,
  // fetched with a few wide scalar loads):
  enum { PUT_NP = 0, PUT_CUR = 1, PUT_NS_STATE = 2, PUT_D = 3, GET_OUT = 4, NPUT = 5 };
  struct Tab {
    int put[NPUT][NR][RU][2];
  };
  static constexpr Tab make()
  {
    Tab t{};
    for (int w = 0; w < NR; w++)
      for (int u = 0; u < RU; u++)
        for (int h = 0; h < 2; h++) {
          const int r2 = w + NR * u;
          const int comp = (r2 < SL::NROW) ? SL::T.comp_of[2 * r2 + h] : -1;
          const int ep = (comp >= L::OFF_P) ? comp - L::OFF_P : -1;           // packed covariance entry
          const int es = (comp >= 0 && comp < L::OFF_P) ? comp : -1;          // canonical state component (vec i: i; quat j: n + j; ll: n + 4)
          t.put[PUT_NP][w][u][h] = ep >= 0 ? O_A + ep : es >= 0 ? O_SP + es : DUMMY;        // P^- into the factor's place, x^- beside it
          t.put[PUT_CUR][w][u][h] = ep >= 0 ? O_B + ep : es >= 0 ? O_B + NP + es : DUMMY;    // the filtered checkpoint
          t.put[PUT_NS_STATE][w][u][h] = es >= 0 ? O_SN + es : DUMMY;                        // x^s_{k+1}
          t.put[PUT_D][w][u][h] = ep >= 0 ? O_A + ep : DUMMY;                                // D = P^s - P^-
          t.put[GET_OUT][w][u][h] = ep >= 0 ? O_B + ep : es >= 0 ? O_B + NP + es : DUMMY;    // the posterior, from the staging area
        }
    return t;
  }
  // every covariance entry and every state component of a checkpoint has exactly one place in each use of the rows (compile-time check
  // of the table against the tile layout of rbis_device.hpp)
  static constexpr bool tab_ok()
  {
    const Tab t = make();
    for (int use = 0; use < NPUT; use++) {
      int hits[PER] = {};
      for (int w = 0; w < NR; w++)
        for (int u = 0; u < RU; u++)
          for (int h = 0; h < 2; h++) {
            const int e = t.put[use][w][u][h];
            if (e < 0 || e >= PER) return false;
            hits[e]++;
          }
      const int pbase = (use == PUT_NP || use == PUT_D) ? O_A : O_B;
      const int sbase = (use == PUT_NP) ? O_SP : (use == PUT_NS_STATE) ? O_SN : O_B + NP;
      if (use != PUT_NS_STATE)
        for (int e = 0; e < NP; e++)
          if (hits[pbase + e] != 1) return false;
      if (use != PUT_D)
        for (int c = 0; c < NST; c++)
          if (hits[sbase + c] != 1) return false;
      if (hits[DUMMY] != 2 * NR * RU - (use != PUT_NS_STATE ? NP : 0) - (use != PUT_D ? NST : 0)) return false;
    }
    return true;
  }
  static_assert(tab_ok(), "row tables of the smoother do not match the tile layout");
};
template <int NS>
__constant__ const typename SmoothWideCfg<NS>::Tab smooth_wide_tab = SmoothWideCfg<NS>::make();

#ifdef SML_TIMELINE
#define SMW_T(i) do { if (stamp) tl[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMW_T(i) do { } while (0)
#endif

template <int NS>
__global__ __launch_bounds__(SmoothWideCfg<NS>::THREADS, 1) void k_smooth_wide(const double *__restrict__ next_pred, const double *__restrict__ next_sm,
                                                                             const double *cur, double *out, int B, int ntiles, double dt, Consts k)
{
  using L = Lay<NS>;
  using SL = Slots<NS>;
  using C = SmoothWideCfg<NS>;
  constexpr int NR = C::NR, NCOL = C::NCOL, O_X = C::O_X, O_DX = C::O_DX, O_B = C::O_B, HALF = C::HALF, RU = C::RU, NP = C::NP;
  static_assert(L::OFF_VEC == 0 && L::OFF_QUAT == NS && L::OFF_LL == NS + 4, "the staging area keeps vec, quat, ll in canonical order");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int w0 = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  int w = w0;   // (made opaque again at the top of every tile: see the loop)
  // Global accesses are buffer instructions: a descriptor per checkpoint AND TILE (scalar registers), the row / component in the scalar
  // offset, the lane's 16-byte column in ONE vector register for every access -- no address registers, no vector arithmetic per access.
  // A row past the tile's last (role 2 / 3, u = 17) is outside the descriptor: it loads zeros and its store is dropped.
  const unsigned lane_b = (unsigned) lane * 16u;
  struct TileBufs { rsrc_t np, ns, cu, out; };
  auto bufs_of = [&](int tile) {
    const long off = (long) tile * SL::TILE_DOUBLES;
    return TileBufs{ mkbuf(next_pred + off, SL::TILE_BYTES), mkbuf(next_sm + off, SL::TILE_BYTES), mkbuf(cur + off, SL::TILE_BYTES),
                     mkbuf(out + off, SL::TILE_BYTES) };
  };
  TileBufs tbuf = bufs_of((int) blockIdx.x);
  // entry e of this lane's filter: lds[sb + e * 64]; three bases so that every access keeps an immediate offset (rbis_smooth_lane.hpp)
  int sb = lane, sb1 = lane + 128 * 64, sb2 = lane + 256 * 64;
  asm volatile("" : "+v"(sb), "+v"(sb1), "+v"(sb2));
  const LdsBases bb{ lane * 8, lane * 8 + 128 * 512, lane * 8 + 256 * 512 };   // (dynamic LDS starts at 0: no static LDS in this kernel)
#define WS (lds + sb)
#define WE(e) (*(((e) < 128) ? (lds + sb + (e) * 64) : ((e) < 256) ? (lds + sb1 + ((e) - 128) * 64) : (lds + sb2 + ((e) - 256) * 64)))
  auto ld_rows = [&](rsrc_t src, d2_t (&r)[RU]) {   // this role's rows of one checkpoint
#pragma unroll
    for (int u = 0; u < RU; u++) r[u] = ldg2<SMW_LOAD_AUX>(src, (unsigned) (w + NR * u) * 1024u, lane_b);
  };
  // ... into LDS [entry][lane] by one of the table's uses
  auto put_rows = [&](const d2_t (&r)[RU], int use) {
    int e[RU][2];
#pragma unroll
    for (int u = 0; u < RU; u++)
#pragma unroll
      for (int h = 0; h < 2; h++) e[u][h] = smooth_wide_tab<NS>.put[use][w][u][h];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      WS[e[u][0] * 64] = r[u].x;
      WS[e[u][1] * 64] = r[u].y;
    }
  };

  auto put_state_rows = [&](const d2_t (&r)[RU], int use) {   // ... of the four rows that carry state components
    constexpr int us[4] = { 0, 1, RU - 2, RU - 1 };
    int e[4][2];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int h = 0; h < 2; h++) e[q][h] = smooth_wide_tab<NS>.put[use][w][us[q]][h];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      WS[e[q][0] * 64] = r[us[q]].x;
      WS[e[q][1] * 64] = r[us[q]].y;
    }
  };

#ifdef SML_TIMELINE
  unsigned long long tl[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif

  // The workgroup is PERSISTENT: tiles blockIdx.x, + gridDim.x, ... (one workgroup per CU fills the registers anyway).  A burst of
  // row loads BLOCKS the wave that issues it until the CU's memory path has taken them (36 loads of 1 KB per wave: 12 k cycles), so the
  // loads are spread: the rows of P^- and of the filtered checkpoint of the NEXT tile are requested while this one multiplies (steps 6
  // and 7), the rows of P^s_{k+1} of THIS tile during its first factorisation steps (first needed in its ninth).
  d2_t ar[RU], cr[RU], nr[RU];
  ld_rows(tbuf.np, ar);   // (the first tile's filtered checkpoint follows during its factorisation: a burst of 36 row loads here would hold the wave for 12 k cycles)
  bool first = true;
  auto ld_row1 = [&](rsrc_t src, d2_t (&r)[RU], int u) { r[u] = ldg2<SMW_LOAD_AUX>(src, (unsigned) (w + NR * u) * 1024u, lane_b); };
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  tbuf = bufs_of(tile);
  const int ntile = (tile + (int) gridDim.x < ntiles) ? tile + (int) gridDim.x : tile;   // (the last tile asks for its own rows again)
  const rsrc_t npn = mkbuf(next_pred + (long) ntile * SL::TILE_DOUBLES, SL::TILE_BYTES), cun = mkbuf(cur + (long) ntile * SL::TILE_DOUBLES, SL::TILE_BYTES);
  // The tile's 3 RU row loads, one or two at a time at points spread over the whole tile (the CU's memory path takes ~10 bytes per cycle:
  // a load issued into a full queue blocks its wave): 0 .. RU-1 the rows of P^s_{k+1} of THIS tile (two in front of each of the first nine
  // barriers of the factorisation), then the NEXT tile's rows of P^- (end of the factorisation, every other group of the substitutions)
  // and of its filtered checkpoint (every third group of steps 6 and 7)
  auto mem_tick = [&](auto KT) {
    constexpr int kt = decltype(KT)::value;
    if constexpr (kt >= 0 && kt < RU) ld_row1(tbuf.ns, nr, C::ns_order(kt));
    else if constexpr (kt >= RU && kt < 2 * RU) ld_row1(npn, ar, kt - RU);
    else if constexpr (kt >= 2 * RU && kt < 3 * RU) ld_row1(cun, cr, kt - 2 * RU);
  };
  auto tick_every = [&](auto EVERY, auto BASE, auto GG) {   // group g of a stream: an operation when g % EVERY == 0, numbered from BASE
    constexpr int g = decltype(GG)::value, base = decltype(BASE)::value, ev = decltype(EVERY)::value;
    if constexpr (g % ev == 0) mem_tick(std::integral_constant<int, base + g / ev>{});
  };
  // The role index is made opaque per tile: what depends on it (table entries, LDS indices of the role's columns, ~300 scalars) would
  // otherwise be computed ONCE in front of the loop and kept -- in scalar registers the kernel does not have.
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
#ifdef SML_TIMELINE
  const bool stamp = tile == SML_TIMELINE;
#endif
  lds_barrier();  // the previous tile's readers of the LDS are done
  SMW_T(0);

  // ---- 0. the prefetched P^- into the factor's place, x^- beside it (the filtered checkpoint follows in step 1) ----
  put_rows(ar, C::PUT_NP);
  SMW_T(11);
  lds_barrier();  // P^- is in LDS
  SMW_T(12);
  // the role's columns of P^-, rows at or below NR t (what is above the column's own diagonal is never used: those entries belong to
  // another role's column and may already hold its factor)
  double a[NCOL][NS];
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int i = 0; i < NS; i++) a[t][i] = (i >= NR * t) ? WS[pk_s(i, cc[t]) * 64] : 0.0;

  // ---- 1. P^- = L diag(d) L^T (k_smooth_lane's step 1); a row of P^s_{k+1} is requested in front of every barrier, the filtered
  //         checkpoint goes into LDS in front of the ninth ----
  {
    double inv_prev = 0.0;
    auto publish = [&](auto CK) {   // column ck by its owner: d (1 / d for the last) on the diagonal, l_i,ck = a_i,ck / d below
      constexpr int ck = decltype(CK)::value, t = ck / NR;
      const double d = a[t][ck];
      const double inv = (fabs(d) > 5.562684646268003e-309) ? 1.0 / d : 0.0;
      WE(pk(ck, ck)) = (ck == NS - 1) ? inv : d;
#pragma unroll
      for (int i = ck + 1; i < NS; i++) WE(pk(i, ck)) = a[t][i] * inv;
      inv_prev = inv;
    };
    if (w == 0) publish(std::integral_constant<int, 0>{});
    static_for<NS>([&](auto KK) {
      constexpr int kk = decltype(KK)::value;
      // (column kk was published by its owner during step kk - 1 -- column 0 in front of the loop)
      if constexpr (kk < 6) {   // the workgroup's first tile has no predecessor that asked for its filtered checkpoint: three rows per step
        if (first) {
#pragma unroll
          for (int u = 3 * kk; u < 3 * kk + 3; u++) ld_row1(tbuf.cu, cr, u);
        }
      }
      if constexpr (kk < 9) {   // two rows of P^s_{k+1}
        mem_tick(std::integral_constant<int, 2 * kk>{});
        mem_tick(std::integral_constant<int, 2 * kk + 1>{});
      }
      if constexpr (kk == 11) {   // D = P^s - P^- (the uncorrected P^-, rbis.cpp:256) row by row in registers: the rows of the checkpoints are the same rows
#pragma unroll
        for (int u = 0; u < RU; u++) nr[u] = nr[u] - ar[u];
      }
      if constexpr (kk >= 12) mem_tick(std::integral_constant<int, RU + kk - 12>{});   // the next tile's P^- (this tile's rows are used up)
      if constexpr (kk == 8) put_rows(cr, C::PUT_CUR);   // P_k and x_k (prefetched; the last rows were asked for at the end of the previous tile)
      if constexpr (kk == 7) put_state_rows(nr, C::PUT_NS_STATE);   // x^s_{k+1} beside x^- (its rows were the first four asked for)
      if constexpr (kk == 9) {
        // residual x^s (-) x^- (rbis.cpp:258-261) for step 3, in the place of x^s: role w its components w, w + NR, ..., the last role the
        // attitude part
      #pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (cidx[t] < NS && (cidx[t] < 6 || cidx[t] > 8)) WS[(O_X + cidx[t]) * 64] = WS[(C::O_SN + cidx[t]) * 64] - WS[(C::O_SP + cidx[t]) * 64];
        if (w == NR - 1) {
          double rqs[4], rqp[4], dchi[3];
      #pragma unroll
          for (int i = 0; i < 4; i++) {
            rqs[i] = WE(C::O_SN + NS + i);
            rqp[i] = WE(C::O_SP + NS + i);
          }
          subtract_quats(rqs, rqp, dchi);
      #pragma unroll
          for (int i = 0; i < 3; i++) WE(O_X + 6 + i) = dchi[i];
        }
      }
      lds_barrier();
      if constexpr (kk == 0) SMW_T(1);
      if constexpr (kk == NS - 1) SMW_T(2);
      // the diagonal slot of column kk-1 held d for the downdates of step kk-1; every role is past them now: it becomes 1/d
      if constexpr (kk > 0)
        if (w == (kk - 1) % NR) WE(pk(kk - 1, kk - 1)) = inv_prev;
      if constexpr (kk + 1 < NS) {
        // LOOK-AHEAD: the owner of column kk + 1 downdates THAT column first and publishes it at once, so that its reciprocal, scaling
        // and LDS writes run while the other roles (and then itself) downdate the rest; the barrier of step kk + 1 finds the column there
        constexpr int t1 = (kk + 1) / NR;
        const bool nxt = (w == (kk + 1) % NR);
        const double dk = WE(pk(kk, kk));
        double tc[NCOL], lik[NS];
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (kk < NR * (t + 1) - 1) tc[t] = WS[pk_s(cc[t], kk) * 64] * dk;
#pragma unroll
        for (int i = kk + 1; i < NS; i++) lik[i] = WE(pk(i, kk));
        if (nxt) {
#pragma unroll
          for (int i = kk + 1; i < NS; i++) {
            a[t1][i] = fma(-lik[i], tc[t1], a[t1][i]);
            lane_pin(a[t1][i]);
          }
          publish(std::integral_constant<int, kk + 1>{});
        }
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (kk < NR * (t + 1) - 1) {
            if (t != t1 || !nxt) {
#pragma unroll
              for (int i = kk + 1; i < NS; i++)
                if (i >= NR * t) {
                  a[t][i] = fma(-lik[i], tc[t], a[t][i]);
                  lane_pin(a[t][i]);  // (downdated NOW, not when the column is published)
                }
            }
          }
      }
    });
  }

  // ---- 2. right-hand sides: columns cc[t] of Ad P_k, Ad = I + dt Ac about the filtered state (rbis.cpp:12-35, 236-239) ----
  double z[NCOL][NS], p[NCOL][NS];
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int i = 0; i < NS; i++) p[t][i] = WS[(O_B + pk_s(i, cc[t])) * 64];
  double wv[3], vv[3], q[4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    wv[i] = WE(O_B + NP + i);
    vv[i] = WE(O_B + NP + 3 + i);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = WE(O_B + NP + NS + i);
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
  auto pin_z = [&](auto) {
#pragma unroll
    for (int t = 0; t < NCOL; t++)
#pragma unroll
      for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
  };
  lds_stream<NS *(NS - 1) / 2, SmwLowerByColumn<NS>, SMW_DEPTH>(bb, [&](auto KQ, double l) {
    constexpr int kq = decltype(KQ)::value, i = SmwLowerByColumn<NS>::row(kq), mm = SmwLowerByColumn<NS>::col(kq);
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
  }, [&](auto GG) {
    pin_z(GG);
    tick_every(std::integral_constant<int, 2>{}, std::integral_constant<int, RU + 3>{}, GG);   // 14 groups: 7 rows
  });
  lds_stream<NS, SmwDiag>(bb, [&](auto KQ, double inv) {
    constexpr int i = decltype(KQ)::value;
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] *= inv;
  }, pin_z);
  lds_stream<NS *(NS - 1) / 2, SmwUpperByColumn<NS>, SMW_DEPTH>(bb, [&](auto KQ, double l) {
    constexpr int kq = decltype(KQ)::value, i = SmwUpperByColumn<NS>::ci(kq), mm = SmwUpperByColumn<NS>::cm(kq);
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
  }, [&](auto GG) {
    pin_z(GG);
    tick_every(std::integral_constant<int, 2>{}, std::integral_constant<int, RU + 10>{}, GG);   // 14 groups: 7 rows
  });
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
  lds_barrier();  // the factor is dead
  SMW_T(4);

  // ---- 4. D = P^s - P^- (the uncorrected P^-, rbis.cpp:256) takes the factor's place, row by row: the rows of the checkpoints are the
  //         same rows ----
  put_rows(nr, C::PUT_D);
  // dx = G resid (rbis.cpp:263): this role's entries, into the place of x^- (read for the last time by the residual, in step 1)
  {
    double dxv[NCOL];
#pragma unroll
    for (int t = 0; t < NCOL; t++) dxv[t] = 0.0;
    lds_stream<NS, SmwRun<O_X>>(bb, [&](auto KQ, double r) {
      constexpr int i = decltype(KQ)::value;
#pragma unroll
      for (int t = 0; t < NCOL; t++) dxv[t] = fma(z[t][i], r, dxv[t]);
    }, [&](auto) {
#pragma unroll
      for (int t = 0; t < NCOL; t++) lane_pin(dxv[t]);
    });
#pragma unroll
    for (int t = 0; t < NCOL; t++)
      if (cidx[t] < NS) WS[(O_DX + cidx[t]) * 64] = dxv[t];   // (x^- is read by the residual only, and every role is past it)
  }
  lds_barrier();  // D and dx are in LDS
  SMW_T(5);
  // ---- 6. M = G D: ALL rows of the role, in passes of two rows over the symmetric D (with all four the accumulators and G fill the
  //         architectural registers and nothing is left to read ahead into) ----
  double m[NCOL][NS];
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int j = 0; j < NS; j++) m[t][j] = 0.0;
  auto sweep = [&](auto T0, auto NT) {
    constexpr int t0 = decltype(T0)::value, nt = decltype(NT)::value;
    lds_stream<NP, SmwByDiagonal<NS>, SMW_DEPTH>(bb, [&](auto KQ, double d) {
      constexpr int kq = decltype(KQ)::value, i = SmwByDiagonal<NS>::row(kq), j = SmwByDiagonal<NS>::col(kq);
#pragma unroll
      for (int t = t0; t < t0 + nt; t++) {
        m[t][j] = fma(z[t][i], d, m[t][j]);
        if (i != j) m[t][i] = fma(z[t][j], d, m[t][i]);
      }
    }, [&](auto GG) {
#pragma unroll
      for (int t = t0; t < t0 + nt; t++)
#pragma unroll
        for (int j = 0; j < NS; j++) lane_pin(m[t][j]);
      tick_every(std::integral_constant<int, 3>{}, std::integral_constant<int, RU + 17 + (t0 == 0 ? 0 : 5)>{}, GG);   // (15 groups per pass: 5 rows each)
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

  // ---- 7. P^s[r][c] = P_k[r][c] + G[r] . M[c] for c <= r: the rows of M in two halves through the place of D, the results into the
  //         staging area (the place of P^s_{k+1}, read for the last time in step 4) ----
  int rbase[NCOL];   // packed index of (row cidx[t], column 0)
#pragma unroll
  for (int t = 0; t < NCOL; t++) rbase[t] = cidx[t] * (cidx[t] + 1) / 2;
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
    // ONE stream over the half: columns two at a time, their rows of M interleaved entry by entry (2 x up to NCOL sums in flight: a
    // multiply-add that waits for its predecessor costs twice its issue slot); the role's rows r = w + NR t at or below the column.
    // Compile time: t >= c / NR; run time: c <= r.  A pair's sums go to the staging area as soon as its last entry is in.
    constexpr int rows_h = (HALF * (h + 1) <= NS) ? HALF : NS - HALF * h;
    using FE = SmwFinal<NS, rows_h>;
    double acc[HALF][NCOL];
#pragma unroll
    for (int cq = 0; cq < HALF; cq++)
#pragma unroll
      for (int t = 0; t < NCOL; t++) acc[cq][t] = (cq < rows_h && t >= (HALF * h + cq) / NR) ? WS[(O_B + rbase[t] + HALF * h + cq) * 64] : 0.0;   // P_k(r, c): its owner's own entry
    lds_stream<rows_h * NS, FE, SMW_DEPTH>(bb, [&](auto KQ, double mv) {
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
              if (cidx[t] < NS && HALF * h + cs <= cidx[t]) WS[(O_B + rbase[t] + HALF * h + cs) * 64] = acc[cs][t];
            }
      }
    }, [&](auto GG) {
#pragma unroll
      for (int cq = 0; cq < rows_h; cq++)
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (t >= (HALF * h + cq) / NR) lane_pin(acc[cq][t]);
      tick_every(std::integral_constant<int, 3>{}, std::integral_constant<int, RU + 27 + (h == 0 ? 0 : 5)>{}, GG);   // (15 / 14 groups: 5 rows each; 3 RU = 54 in all)
    });
  });
  // ---- 8. state: cur.addState(RBIS(dx)) (rbis.cpp:263-265), by the LAST role (it has a stand-in instead of a fourth row of M to make), in
  //         place behind P_k: that is where the staging area keeps the state ----
  if (w == NR - 1) {
    double dchi[3] = { WE(O_DX + 6), WE(O_DX + 7), WE(O_DX + 8) };
    double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
    fold_chi(dchi, dq, k.chi_tol);  // RBIS(vec) constructor
    double chi[3], qq[4], qo[4];
#pragma unroll
    for (int i = 0; i < 3; i++) chi[i] = WE(O_B + NP + 6 + i) + dchi[i];
#pragma unroll
    for (int i = 0; i < 4; i++) qq[i] = WE(O_B + NP + NS + i);
    fold_chi(chi, qq, k.chi_tol);
    quat_mul(qq, dq, qo);
#pragma unroll
    for (int i = 0; i < NS; i++) WE(O_B + NP + i) = (i >= 6 && i <= 8) ? chi[i - 6] : WE(O_B + NP + i) + WE(O_DX + i);
#pragma unroll
    for (int i = 0; i < 4; i++) WE(O_B + NP + NS + i) = qo[i];
    // (the log-likelihood stays where the filtered checkpoint's row put it)
  }


  lds_barrier();
  SMW_T(13);
  // ---- 9. the posterior by whole rows ----
  {
    int e[RU][2];
#pragma unroll
    for (int u = 0; u < RU; u++)
#pragma unroll
      for (int h = 0; h < 2; h++) e[u][h] = smooth_wide_tab<NS>.put[C::GET_OUT][w][u][h];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      const d2_t v = { WS[e[u][0] * 64], WS[e[u][1] * 64] };
      if (active) stg2<SMW_STORE_AUX>(tbuf.out, (unsigned) (w + NR * u) * 1024u, lane_b, v);
    }
  }

#ifdef SML_TIMELINE
  if (stamp) tl[7] = __builtin_amdgcn_s_memtime();
  if (stamp && lane == 0)
    for (int i = 0; i < 16; i++) sml_tl[w][i] = tl[i];
#endif
  first = false;
  }  // tiles
}

#undef SMW_T
#undef WS
#undef WE

}  // namespace pb
