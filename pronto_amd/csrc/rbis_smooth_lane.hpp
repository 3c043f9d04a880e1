// rbis_smooth_lane.hpp -- the RTS smoother step (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) with ONE LANE
// PER FILTER and the dense n x n work split over NR role waves of a 64-filter tile (round 4; replaces k_smooth_reg as the default).
//
//   G      = P_k Ad^T (P^-_{k+1})^-1      P^s_k = P_k + G (P^s_{k+1} - P^-_{k+1}) G^T      x^s_k = x_k (+) G (x^s_{k+1} (-) x^-_{k+1})
//
// k_smooth_reg (rbis_smooth.hpp) gives a filter to 16 / 32 lanes: every step of the factorisation is a cross-lane round trip (publish
// the pivot row, read it back), 2 500-3 800 VALU instructions per wave of 4 / 2 filters, 0.19 / 0.11 of the HBM roofline for three
// rounds.  Here a lane owns a filter as in the forward step, so every index is a compile-time constant and nothing crosses lanes; what
// one lane cannot hold (P^-, Ad P_k, D and G are n x n each) is split over the waves of the workgroup and LDS:
//   * role w owns the columns c = w, w + NR, ... of the factorisation, the same columns of the right-hand side Ad P_k and therefore
//     the same ROWS of the gain G -- in registers;
//   * P^- = L diag(d) L^T, right-looking, NOT pivoted (P^- is SPD; the reference's .ldlt() pivots on the diagonal, the results
//     differ by rounding: tests <= 1e-9 against the oracle): the owner of column k publishes l_ik into LDS (where the factor has
//     to end up anyway), one barrier, every role downdates its own columns.  The factor lives in LDS [packed entry][64 lanes]:
//     a lane reads ITS filter's entry, 512 contiguous bytes per wave, no bank conflict, and every entry read feeds NCOL FMAs;
//   * the substitutions run out of that LDS factor on the role's NCOL right-hand sides: G rows come out in registers;
//   * D = P^s - P^- takes the factor's place in LDS; the rows of M = G D are made by their owners, CH at a time, and handed round
//     through a small exchange region: P^s[r][c] = P_k[r][c] + G[r] . M[c] is complete as soon as M[c] is there, and is stored at once.
// One instruction stream for all roles (the role enters as scalar offsets and wave-uniform branches), so the instruction cache
// holds one copy.  Columns past n (n is not a multiple of NR) are stand-ins that mirror column n-1 and never store.
// LDS per 64-filter tile: (n (n + 1) / 2 + CH n) doubles per lane = 77 KB (n = 15, two workgroups per CU), 150 KB (n = 21).
#pragma once

#include <hip/hip_runtime.h>

#include "rbis_device.hpp"
#include "rbis_lds_stream.hpp"

// attribution builds (scripts/smooth_attribution.sh lane...): parts compiled OUT, results are garbage, only the time means something
#if !defined(PB_EXPERIMENTS) && (defined(SML_SKIP_FACT) || defined(SML_SKIP_RHS) || defined(SML_SKIP_SUBST) || defined(SML_SKIP_D) || \
                                 defined(SML_SKIP_M) || defined(SML_SKIP_FINAL) || defined(SML_SKIP_CHUNKS) || defined(SML_NO_PKLOAD) || defined(SML_NO_PSTORE) || defined(SML_TIMELINE) || \
                                 defined(SML_WIDE) || defined(SML_SINGLE_READS) || defined(SML_FINAL_SPLIT))
#error "the SML_* attribution flags need -DPB_EXPERIMENTS as well"
#endif

namespace pb {

#ifndef SM_LANE_NR15
#define SM_LANE_NR15 4
#endif
#ifndef SM_LANE_CH15
#define SM_LANE_CH15 2
#endif

template <int NS>
struct SmoothLaneCfg {
  using L = Lay<NS>;
  using SL = Slots<NS>;
  static constexpr int NR = (NS <= 16) ? SM_LANE_NR15 : 8;   // role waves per tile
  static constexpr int NCOL = (NS + NR - 1) / NR;        // columns / gain rows per role
#ifndef SM_LANE_CH21
#define SM_LANE_CH21 3
#endif
  static constexpr int CH = (NS <= 16) ? SM_LANE_CH15 : SM_LANE_CH21;  // rows of M per exchange (<= NR: at most one per role)
  static constexpr int NP = L::NP;
  static constexpr int O_X = NP;                         // exchange region: residual + dx first, then CH rows of M
  static constexpr int PER = O_X + CH * NS;              // doubles per lane
  static constexpr int THREADS = 64 * NR;
  static constexpr int WAVES_PER_SIMD = 2;               // 8 waves per CU either way: 256 registers per lane
  static constexpr size_t LDS_BYTES = sizeof(double) * PER * 64;
  static_assert(CH >= 2 && CH <= NR, "the exchange region holds the residual and dx; one M row per role and chunk");
  // component -> offset (doubles) of lane 0's copy inside a tile (rbis_device.hpp: slot s sits in row s / 2, half s % 2)
  static constexpr int off_of(int comp) { return (SL::T.slot_of[comp] / 2) * 128 + (SL::T.slot_of[comp] % 2); }
  // What depends on the role is read from constant memory with wide scalar loads, ONE table row per use: col[w][t][i] = offset of
  // P(i, column w + NR t) -- by symmetry also of P(row w + NR t, i).
  static constexpr int NSP = (NS + 3) & ~3;
  // D = P^s - P^- is staged by whole 16-byte rows of the checkpoints (two components per lane and load, every byte of a line used):
  // role w takes rows w, w + NR, ...; drow[w][u] = { packed entry of the row's first component, of its second } or -1 (state
  // vector, quaternion, log-likelihood, padding: not part of D)
  static constexpr int RU = (SL::NROW + NR - 1) / NR;
  struct Tab {
    int col[NR][NCOL][NSP];
    int drow[NR][RU][2];
  };
  static constexpr Tab make()
  {
    Tab t{};
    for (int w = 0; w < NR; w++) {
      for (int c = 0; c < NCOL; c++) {
        const int j = (w + NR * c < NS) ? w + NR * c : NS - 1;
        for (int i = 0; i < NSP; i++) t.col[w][c][i] = off_of(L::OFF_P + pk(i < NS ? i : NS - 1, j));
      }
      for (int u = 0; u < RU; u++)
        for (int h = 0; h < 2; h++) {
          const int r2 = w + NR * u;
          const int comp = (r2 < SL::NROW) ? SL::T.comp_of[2 * r2 + h] : -1;
          t.drow[w][u][h] = (comp >= L::OFF_P) ? comp - L::OFF_P : -1;
        }
    }
    return t;
  }
};
template <int NS>
__constant__ const typename SmoothLaneCfg<NS>::Tab smooth_lane_tab = SmoothLaneCfg<NS>::make();

// End of a group of LDS reads.  The backend gathers independent LDS reads at the top of a block whatever scheduling barriers sit
// between them (all of D: 240 registers, spilled), so the NEXT group's reads are made to depend on this group's arithmetic: the
// lane's LDS index passes through an empty asm that also names a result of the group.  The memory clobber makes a later read of
// the same entry a real read instead of a value kept in registers since its first use (the backward substitution reads what the
// forward one read).
__device__ __forceinline__ void lane_fence(int &sb, double after)
{
  asm volatile("" : "+v"(sb) : "v"(after) : "memory");
}
__device__ __forceinline__ void lane_fence3(int &sb, int &sb1, int &sb2, double after)   // the same for all three bases of SE()
{
  asm volatile("" : "+v"(sb), "+v"(sb1), "+v"(sb2) : "v"(after) : "memory");
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a fence over every address space: the backend puts
// `s_waitcnt vmcnt(0)` in front of every s_barrier, so each of the kernel's 20-30 barriers also waited for every global load in flight
// (checkpoints requested ahead of their use) and every global store behind it (the posterior's entries: a full round trip to memory per
// exchange of rows).  Global memory needs no ordering between the waves of a tile here: every wave reads its checkpoints' entries
// itself and stores entries nobody reads in this launch.
__device__ __forceinline__ void lds_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The value is COMPUTED here: without this the backend sinks arithmetic below the next barrier to its first use, and what it was
// computed from (LDS reads that cannot follow it across the barrier) stays live or is spilled (rbis_coop.hpp, pb_pin).
__device__ __forceinline__ void lane_pin(double &v) { asm volatile("" : "+v"(v)); }

#ifdef SML_TIMELINE  // attribution (-DPB_EXPERIMENTS -DSML_TIMELINE=<tile>): shader-clock stamps of one tile's roles, taken right behind
                     // barriers (where the wave has just waited for its LDS traffic anyway), written out at the end; the kernel's time is unchanged
__device__ unsigned long long sml_tl[8][16];
#define SML_T(i) tl[i] = __builtin_amdgcn_s_memtime()
#else
#define SML_T(i) do { } while (0)
#endif

__device__ __forceinline__ int pk_s(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <int NS>
__global__ __launch_bounds__(SmoothLaneCfg<NS>::THREADS, SmoothLaneCfg<NS>::WAVES_PER_SIMD) void k_smooth_lane(
    const double *__restrict__ next_pred, const double *__restrict__ next_sm, const double *cur, double *out, int B, double dt, Consts k)
{
  using L = Lay<NS>;
  using SL = Slots<NS>;
  using C = SmoothLaneCfg<NS>;
  constexpr int NR = C::NR, NCOL = C::NCOL, CH = C::CH, O_X = C::O_X;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const long tb = (long) blockIdx.x * SL::TILE_DOUBLES + lane * 2;
  const bool active = (long) blockIdx.x * 64 + lane < B;
  const auto &tab = smooth_lane_tab<NS>;
  int sb = lane;  // entry e of this lane's filter: lds[sb + e * 64] (sb passes through lane_fence)
#define S (lds + sb)
  // An LDS instruction's immediate offset reaches 64 KB = 128 entries of this [entry][lane] layout, and a paired read
  // (ds_read2st64_b64) 256: beyond, the backend made an address for every single access (n = 21: 870 v_add_u32 and 897 unpaired
  // ds_read_b64 among 10 083 instructions).  Two more bases, 128 and 256 entries in, opaque to the optimiser like sb itself:
  // SE(e) with a compile-time e picks the base that leaves an immediate offset.
  int sb1 = lane + 128 * 64, sb2 = lane + 256 * 64;
  asm volatile("" : "+v"(sb1), "+v"(sb2));
#define SE(e) (*(((e) < 128) ? (lds + sb + (e) * 64) : ((e) < 256) ? (lds + sb1 + ((e) - 128) * 64) : (lds + sb2 + ((e) - 256) * 64)))
#define LANE_FENCE(after) lane_fence3(sb, sb1, sb2, after)
  // SER(e): a READ of the factor / of D / of an exchanged row.  The backend pairs neighbouring reads into ds_read2st64_b64, which the LDS
  // serves at half the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table).  -DSML_SINGLE_READS makes them volatile LDS accesses,
  // which are not paired: 107.8 / 316.0 us against 100.6 / 291.6 paired (profiles/r05_smoother_attribution.txt) -- the read rate of
  // the LDS is not what bounds this kernel, and the ordering volatile imposes costs more than the pairing.  Paired is the default.
#ifdef SML_SINGLE_READS
#define SER(e) (*(volatile __attribute__((address_space(3))) double *) (__attribute__((address_space(3))) double *) &SE(e))   // (a plain volatile pointer would be a FLAT access)
#else
#define SER(e) SE(e)
#endif
#ifndef SML_NT
#define SML_NT 0   // (round 5 experiment) non-temporal global loads and stores
#endif
  auto gld = [&](const double *pp) { return SML_NT ? __builtin_nontemporal_load(pp) : *pp; };
  auto gst = [&](double *pp, double v) { if (SML_NT) __builtin_nontemporal_store(v, pp); else *pp = v; };
  auto ldc = [&](const double *src, int comp) { return gld(src + tb + C::off_of(comp)); };  // compile-time component only
  // column t of this role from one of the checkpoints: the offsets come in with wide scalar loads, the n loads go out back to back
  auto ld_col = [&](const double *src, int t, int i0, double (&v)[NS]) {  // rows i0 .. n-1 (the others: 0)
    int o[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) o[i] = tab.col[w][t][i];
#ifdef SML_WIDE   // attribution (round 5): the SAME loads as whole 16-byte rows -- every byte of every 128-byte line the instruction touches is
                  // fetched by it, where the 8-byte access at a 16-byte stride uses half of each.  Results are garbage.
#pragma unroll
    for (int i = 0; i < NS; i++) {
      if (i >= i0) {
        const d2_t r = *reinterpret_cast<const d2_t *>(src + tb + (o[i] & ~1));
        v[i] = r.x + r.y;
      } else {
        v[i] = 0.0;
      }
    }
#else
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = (i >= i0) ? gld(src + tb + o[i]) : 0.0;
#endif
  };

  int cidx[NCOL], cc[NCOL];  // this role's columns (gain rows); stand-ins mirror column n - 1
#pragma unroll
  for (int t = 0; t < NCOL; t++) {
    cidx[t] = w + NR * t;
    cc[t] = cidx[t] < NS ? cidx[t] : NS - 1;
  }

#ifdef SML_TIMELINE
  unsigned long long tl[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, ta = 0;
#endif
  SML_T(0);
  // ---- 0. residual x^s (-) x^- (rbis.cpp:258-261), by the last role: its loads are requested here, the arithmetic (an atan2) and the
  //         hand-over through the exchange region come behind the factorisation, whose first barrier would otherwise wait for them ----
  double rqs[4], rqp[4], rv[NS];
  if (w == NR - 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      rqs[i] = ldc(next_sm, L::OFF_QUAT + i);
      rqp[i] = ldc(next_pred, L::OFF_QUAT + i);
    }
#pragma unroll
    for (int i = 0; i < NS; i++) rv[i] = ldc(next_sm, L::OFF_VEC + i) - ldc(next_pred, L::OFF_VEC + i);
  }

  // (round 5 experiment, OFF) the role's columns of P_k and the filtered state for step 2 asked for HERE, so that they arrive behind the
  // factorisation (the barriers order LDS only, nothing waits for them on the way): SML_EARLY_PK = how many of the role's 3 columns
  // (21 states only).  Measured at 64k filters: 0: 291.6 us, 1: 303.2 (20 B of scratch), 2: 311.3 (60 B), 3: 339.4 (276 B) -- the
  // kernel has no registers left for them.
#ifndef SML_EARLY_PK
#define SML_EARLY_PK 0
#endif
  constexpr int NEARLY = (NS == 21) ? SML_EARLY_PK : 0;
  constexpr bool EARLY_PK = NEARLY > 0;
  double pke[EARLY_PK ? NEARLY : 1][EARLY_PK ? NS : 1], ske[10];
  // ---- 1. P^- = L diag(d) L^T ----
  {
    double a[NCOL][NS];
    // SML_ROWTOP (21 states): P^- comes in by whole 16-byte rows (role w: rows w, w + NR, ...; the table of step 4) into the factor's place,
    // and every role takes its columns from there -- 17 full-line loads per role instead of 39 eight-byte ones that use half of every
    // line they touch.  Measured: 292.1 against 291.1 us, the tile's first barrier still comes 14 k cycles in -- every CU asks for its
    // tile's 129 KB at the same moment (four dispatch rounds in lockstep): the burst is bound by the memory, not by the form of the loads,
    // and this kernel has neither registers nor LDS to ask earlier.  Off.
#ifndef SML_ROWTOP
#define SML_ROWTOP 0
#endif
    constexpr bool ROWTOP = (NS == 21) && SML_ROWTOP;
    if constexpr (ROWTOP) {
      d2_t rr[C::RU];
#pragma unroll
      for (int u = 0; u < C::RU; u++) {
        const int r2 = (w + NR * u < SL::NROW) ? w + NR * u : SL::NROW - 1;
        rr[u] = *reinterpret_cast<const d2_t *>(next_pred + tb + r2 * 128);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < C::RU; u++) {
        const int e0 = tab.drow[w][u][0], e1 = tab.drow[w][u][1];
        if (e0 >= 0) S[e0 * 64] = rr[u].x;
        if (e1 >= 0) S[e1 * 64] = rr[u].y;
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < NCOL; t++)
#pragma unroll
        for (int i = 0; i < NS; i++) a[t][i] = (i >= NR * t) ? S[pk_s(i, cc[t]) * 64] : 0.0;   // (above the column's diagonal: never used)
    } else {
#pragma unroll
      for (int t = 0; t < NCOL; t++) ld_col(next_pred, t, NR * t, a[t]);  // column w + NR t: rows above NR t are above its diagonal for every role
    }
    if constexpr (EARLY_PK) {
#pragma unroll
      for (int i = 0; i < 6; i++) ske[i] = ldc(cur, L::OFF_VEC + i);
#pragma unroll
      for (int i = 0; i < 4; i++) ske[6 + i] = ldc(cur, L::OFF_QUAT + i);
#pragma unroll
      for (int t = 0; t < NEARLY; t++) ld_col(cur, t, 0, pke[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NS == 21) {  // rbis.cpp:244-251: a bias block whose variance is < 1e-11 is replaced by I in the factorised matrix
      bool fix_g = false, fix_a = false;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        fix_g = fix_g | ((ROWTOP ? SE(pk(15 + i, 15 + i)) : ldc(next_pred, L::OFF_P + pk(15 + i, 15 + i))) < .00000000001);
        fix_a = fix_a | ((ROWTOP ? SE(pk(18 + i, 18 + i)) : ldc(next_pred, L::OFF_P + pk(18 + i, 18 + i))) < .00000000001);
      }
#pragma unroll
      for (int t = 0; t < NCOL; t++)
#pragma unroll
        for (int i = 15; i < 21; i++) {
          const bool gblk = i < 18, in_blk = gblk ? (cc[t] >= 15 && cc[t] < 18) : (cc[t] >= 18);
          a[t][i] = ((gblk ? fix_g : fix_a) && in_blk) ? (i == cc[t] ? 1.0 : 0.0) : a[t][i];
        }
    }
    double inv_prev = 0.0;
    static_for<NS>([&](auto KK) {
      constexpr int kk = decltype(KK)::value;
#ifdef SML_SKIP_FACT
      if (kk == 0) S[0] = a[0][0] + a[NCOL - 1][NS - 1];
      return;
#endif
      if (w == kk % NR) {  // owner of column kk
        constexpr int t = kk / NR;
        const double d = a[t][kk];
        const double inv = (fabs(d) > 5.562684646268003e-309) ? 1.0 / d : 0.0;
        SE(pk(kk, kk)) = (kk == NS - 1) ? inv : d;
#pragma unroll
        for (int i = kk + 1; i < NS; i++) SE(pk(i, kk)) = a[t][i] * inv;
        inv_prev = inv;
      }
      lds_barrier();
      if constexpr (kk == 0) SML_T(1);
      if constexpr (kk == NS - 1) SML_T(2);
      // the diagonal slot of column kk-1 held d for the downdates of step kk-1; every role is past them now: it becomes 1/d
      if constexpr (kk > 0)
        if (w == (kk - 1) % NR) SE(pk(kk - 1, kk - 1)) = inv_prev;
      if constexpr (kk + 1 < NS) {
        const double dk = SER(pk(kk, kk));
        // column slot t holds a column c in [NR t, NR t + NR): it is finished once kk >= NR (t + 1) - 1, and its rows above NR t
        // are above the diagonal -- neither is touched (compile-time bounds; what remains above the diagonal is never read)
        double tc[NCOL];
#pragma unroll
        for (int t = 0; t < NCOL; t++)
          if (kk < NR * (t + 1) - 1) tc[t] = S[pk_s(cc[t], kk) * 64] * dk;
#pragma unroll
        for (int i = kk + 1; i < NS; i++) {
          const double lik = SER(pk(i, kk));
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

  if (w == NR - 1) {
    double dchi[3];
    subtract_quats(rqs, rqp, dchi);
#pragma unroll
    for (int i = 0; i < NS; i++) SE(O_X + i) = (i >= 6 && i <= 8) ? dchi[i - 6] : rv[i];
  }

  // ---- 2. right-hand sides: columns cc[t] of Ad P_k, Ad = I + dt Ac about the filtered state (rbis.cpp:12-35, 236-239) ----
  double z[NCOL][NS];
  {
    double wv[3], v[3], q[4], R[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      wv[i] = EARLY_PK ? ske[i] : ldc(cur, L::OFF_VEC + i);
      v[i] = EARLY_PK ? ske[3 + i] : ldc(cur, L::OFF_VEC + 3 + i);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = EARLY_PK ? ske[6 + i] : ldc(cur, L::OFF_QUAT + i);
    quat_to_rot(q, R);
    const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };
#pragma unroll
    for (int t = 0; t < NCOL; t++) {
      double p[NS];
      if (t < NEARLY) {   // (compile-time inside the unrolled loop)
#pragma unroll
        for (int i = 0; i < NS; i++) p[i] = pke[t < NEARLY ? t : 0][i];
      } else {
        ld_col(cur, t, 0, p);
      }
#pragma unroll
      for (int i = 0; i < NS; i++) z[t][i] = p[i];
      const double pv[3] = { p[3], p[4], p[5] }, pc[3] = { p[6], p[7], p[8] };
      // v rows: -w x p_v + g_b x p_chi [- v x p_bg - p_ba];  chi rows: -w x p_chi [- p_bg];  Delta rows: R p_v - R (v x p_chi)
      const double wxpv[3] = { wv[1] * pv[2] - wv[2] * pv[1], wv[2] * pv[0] - wv[0] * pv[2], wv[0] * pv[1] - wv[1] * pv[0] };
      const double gxpc[3] = { gb[1] * pc[2] - gb[2] * pc[1], gb[2] * pc[0] - gb[0] * pc[2], gb[0] * pc[1] - gb[1] * pc[0] };
      const double wxpc[3] = { wv[1] * pc[2] - wv[2] * pc[1], wv[2] * pc[0] - wv[0] * pc[2], wv[0] * pc[1] - wv[1] * pc[0] };
      const double vxpc[3] = { v[1] * pc[2] - v[2] * pc[1], v[2] * pc[0] - v[0] * pc[2], v[0] * pc[1] - v[1] * pc[0] };
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double av = -wxpv[i] + gxpc[i], ac = -wxpc[i];
        if constexpr (NS == 21) {
          const double pbg[3] = { p[15], p[16], p[17] };
          const double vxpbg = (i == 0) ? v[1] * pbg[2] - v[2] * pbg[1] : (i == 1 ? v[2] * pbg[0] - v[0] * pbg[2] : v[0] * pbg[1] - v[1] * pbg[0]);
          av += -vxpbg - p[18 + i];
          ac += -pbg[i];
        }
        const double ad = R[3 * i] * (pv[0] - vxpc[0]) + R[3 * i + 1] * (pv[1] - vxpc[1]) + R[3 * i + 2] * (pv[2] - vxpc[2]);
        z[t][3 + i] = fma(dt, av, z[t][3 + i]);
        z[t][6 + i] = fma(dt, ac, z[t][6 + i]);
        z[t][9 + i] = fma(dt, ad, z[t][9 + i]);
      }
    }
  }
  lds_barrier();  // the factor is complete (the last reciprocal pivots were written behind the last barrier of step 1)
  SML_T(3);

  // (round 5 experiment) the first half of the rows of step 4 requested HERE, in front of the substitutions, so that they arrive behind
  // them (the barriers order LDS only): SML_EARLY_D, 21 states.  Measured: 304.5 us against 291.5 (100 B of scratch): off.
#ifndef SML_EARLY_D
#define SML_EARLY_D 0
#endif
  constexpr bool EARLY_D = (NS == 21) && SML_EARLY_D;
  constexpr int RH_E = (C::RU + 1) / 2;
  d2_t dse[EARLY_D ? RH_E : 1], dpe[EARLY_D ? RH_E : 1];
  if constexpr (EARLY_D) {
#pragma unroll
    for (int v = 0; v < RH_E; v++) {
      const int r2 = (v < C::RU && w + NR * v < SL::NROW) ? w + NR * v : SL::NROW - 1;
      dse[v] = *reinterpret_cast<const d2_t *>(next_sm + tb + r2 * 128);
      dpe[v] = *reinterpret_cast<const d2_t *>(next_pred + tb + r2 * 128);
    }
  }
  // ---- 3. (P^-) X = Ad P_k out of the LDS factor: z[t][:] becomes row cidx[t] of G ----
  // SML_STREAM (21 states): the reads as lds_stream's explicit pipeline of SINGLE ds_read_b64 (rbis_lds_stream.hpp) -- with 8 roles every
  // entry of the factor is read 8 times, and the paired ds_read2st64_b64 the backend makes of neighbouring reads moves the same bytes in
  // twice the LDS time (8 x 462 reads x 8 cycles per pair = 15 k of the substitutions' 17 k cycles).  Measured: the substitutions take
  // 15.1 k instead of 17.2 k cycles and step 4 waits 2 k longer for its rows: 290.7 against 291.1 us per step.  Off.
#ifndef SML_STREAM
#define SML_STREAM 0
#endif
  constexpr bool STREAM = (NS == 21) && SML_STREAM;
  const LdsBases lbb{ lane * 8, lane * 8 + 128 * 512, lane * 8 + 256 * 512 };   // (dynamic LDS starts at 0: no static LDS in this kernel)
#ifndef SML_SKIP_SUBST
  if constexpr (STREAM) {
    auto pin_z = [&](auto) {
#pragma unroll
      for (int t = 0; t < NCOL; t++)
#pragma unroll
        for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
    };
    lds_stream<NS *(NS - 1) / 2, SmwLowerByColumn<NS>>(lbb, [&](auto KQ, double l) {
      constexpr int kq = decltype(KQ)::value, i = SmwLowerByColumn<NS>::row(kq), mm = SmwLowerByColumn<NS>::col(kq);
#pragma unroll
      for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
    }, pin_z);
    lds_stream<NS, SmwDiag>(lbb, [&](auto KQ, double inv) {
      constexpr int i = decltype(KQ)::value;
#pragma unroll
      for (int t = 0; t < NCOL; t++) z[t][i] *= inv;
    }, pin_z);
    lds_stream<NS *(NS - 1) / 2, SmwUpperByColumn<NS>>(lbb, [&](auto KQ, double l) {
      constexpr int kq = decltype(KQ)::value, i = SmwUpperByColumn<NS>::ci(kq), mm = SmwUpperByColumn<NS>::cm(kq);
#pragma unroll
      for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][mm], z[t][i]);
    }, pin_z);
  } else {
#pragma unroll
  for (int i = 1; i < NS; i++)
#pragma unroll
    for (int m = 0; m < i; m++) {
      const double l = SER(pk(i, m));
#pragma unroll
      for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][m], z[t][i]);
      if (m == i - 1 && (i & 1)) LANE_FENCE(z[0][i >= 2 ? i - 2 : 0]);  // (row i - 2: one group of reads may run ahead)
    }
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const double inv = SER(pk(i, i));
#pragma unroll
    for (int t = 0; t < NCOL; t++) z[t][i] *= inv;
  }
  LANE_FENCE(z[0][NS - 1]);
#pragma unroll
  for (int i = NS - 2; i >= 0; i--)
#pragma unroll
    for (int m = i + 1; m < NS; m++) {
      const double l = SER(pk(m, i));
#pragma unroll
      for (int t = 0; t < NCOL; t++) z[t][i] = fma(-l, z[t][m], z[t][i]);
      if (m == NS - 1 && (i & 1)) LANE_FENCE(z[0][i + 2 < NS ? i + 2 : NS - 1]);
    }
  }
#endif
  // dx = G resid (rbis.cpp:263): this role's entries
  double dxv[NCOL];
#pragma unroll
  for (int t = 0; t < NCOL; t++) dxv[t] = 0.0;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const double r = SER(O_X + i);
#pragma unroll
    for (int t = 0; t < NCOL; t++) dxv[t] = fma(z[t][i], r, dxv[t]);
  }
#pragma unroll
  for (int t = 0; t < NCOL; t++) {
    lane_pin(dxv[t]);
#pragma unroll
    for (int i = 0; i < NS; i++) lane_pin(z[t][i]);
  }
  lds_barrier();  // factor and residual are dead
  SML_T(4);

  // ---- 4. D = P^s - P^- (the uncorrected P^-, rbis.cpp:256) takes the factor's place; dx goes behind the residual ----
#pragma unroll
  for (int t = 0; t < NCOL; t++)
    if (cidx[t] < NS) S[(O_X + NS + cidx[t]) * 64] = dxv[t];
  {
    // in two passes: all rows of both checkpoints in flight at once are 68-72 doubles next to the 60 of G
    constexpr int RH = (C::RU + 1) / 2;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      d2_t ds[RH], dp[RH];
#pragma unroll
      for (int v = 0; v < RH; v++) {
        const int u = pass * RH + v;
        const int r2 = (u < C::RU && w + NR * u < SL::NROW) ? w + NR * u : SL::NROW - 1;
        if (EARLY_D && pass == 0) {
          ds[v] = dse[EARLY_D ? v : 0];
          dp[v] = dpe[EARLY_D ? v : 0];
        } else {
          ds[v] = *reinterpret_cast<const d2_t *>(next_sm + tb + r2 * 128);
          dp[v] = *reinterpret_cast<const d2_t *>(next_pred + tb + r2 * 128);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < RH; v++) {
        const int u = pass * RH + v;
        if (u < C::RU) {
          const int e0 = tab.drow[w][u][0], e1 = tab.drow[w][u][1];
          if (e0 >= 0) S[e0 * 64] = ds[v].x - dp[v].x;
          if (e1 >= 0) S[e1 * 64] = ds[v].y - dp[v].y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  lds_barrier();
  SML_T(5);

  // ---- 5. state: cur.addState(RBIS(dx)) (rbis.cpp:263-265), by the LAST role: the first CH roles make the first rows of M in step 6
  //         meanwhile (with role 0 doing both, every other wave of the tile waited for two fold_chi at the first barrier of step 6) ----
  if (w == NR - 1) {
    double dchi[3] = { SE(O_X + NS + 6), SE(O_X + NS + 7), SE(O_X + NS + 8) };
    double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
    fold_chi(dchi, dq, k.chi_tol);  // RBIS(vec) constructor
    double chi[3], qq[4], o[4];
#pragma unroll
    for (int i = 0; i < 3; i++) chi[i] = ldc(cur, L::OFF_VEC + 6 + i) + dchi[i];
#pragma unroll
    for (int i = 0; i < 4; i++) qq[i] = ldc(cur, L::OFF_QUAT + i);
    fold_chi(chi, qq, k.chi_tol);
    quat_mul(qq, dq, o);
    const double ll = ldc(cur, L::OFF_LL);
    double xo[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) xo[i] = (i >= 6 && i <= 8) ? chi[i - 6] : ldc(cur, L::OFF_VEC + i) + SE(O_X + NS + i);
    if (active) {
#pragma unroll
      for (int i = 0; i < NS; i++) out[tb + C::off_of(L::OFF_VEC + i)] = xo[i];
#pragma unroll
      for (int i = 0; i < 4; i++) out[tb + C::off_of(L::OFF_QUAT + i)] = o[i];
      out[tb + C::off_of(L::OFF_LL)] = ll;
    }
  }

  // ---- 6. P^s = P_k + G D G^T, CH rows of M = G D at a time ----
#ifdef SML_SKIP_CHUNKS
  if (B > 0) return;
#endif
  double pkn[NCOL][CH];
  int pon[NCOL][CH];
#pragma unroll
  for (int t = 0; t < NCOL; t++)
#pragma unroll
    for (int q = 0; q < CH; q++) {
      pon[t][q] = tab.col[w][t][q];
      pkn[t][q] = cur[tb + pon[t][q]];
    }
#pragma unroll 1
  for (int c0 = 0; c0 < NS; c0 += CH) {
    // the owner of row c makes M[c][:] = G[c][:] D in registers (every entry of the symmetric D is read once)
    double m[NS];
    int mcc = -1;
    // P_k(r, c) of the NEXT chunk's columns for the role's rows are requested now: a load waits for every older store as well
    // (one counter, in-order return), so a load issued behind this chunk's stores would expose their latency in every iteration
    double pkv[NCOL][CH];
    int po[NCOL][CH];
#pragma unroll
    for (int t = 0; t < NCOL; t++)
#pragma unroll
      for (int q = 0; q < CH; q++) {
        pkv[t][q] = pkn[t][q];
        po[t][q] = pon[t][q];
        pon[t][q] = tab.col[w][t][(c0 + CH + q < NS) ? c0 + CH + q : NS - 1];
#ifdef SML_NO_PKLOAD
        pkn[t][q] = 1.0;
#elif defined(SML_WIDE)
        {
          const d2_t r = *reinterpret_cast<const d2_t *>(cur + tb + (pon[t][q] & ~1));
          pkn[t][q] = r.x + r.y;
        }
#else
        pkn[t][q] = gld(cur + tb + pon[t][q]);
#endif
      }
#pragma unroll
    for (int t = 0; t < NCOL; t++) {
      const int rel = cidx[t] - c0;
#ifdef SML_SKIP_M
      if (rel >= 0 && rel < CH && cidx[t] < NS) {
        mcc = rel;
#pragma unroll
        for (int j = 0; j < NS; j++) m[j] = z[t][j];
      }
      continue;
#endif
      if (rel >= 0 && rel < CH && cidx[t] < NS) {
        mcc = rel;
        double hook = z[t][0];
#pragma unroll
        for (int j = 0; j < NS; j++) m[j] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) {
            const double d = SER(pk(i, j));
            m[j] = fma(z[t][i], d, m[j]);
            if (i != j) m[i] = fma(z[t][j], d, m[i]);
            if (j == i && (i & 1)) {  // end of a group of two rows of D
              LANE_FENCE(hook);
              hook = m[0];
              lane_pin(hook);
            }
          }
      }
    }
    if (mcc >= 0) {
#pragma unroll
      for (int j = 0; j < NS; j++) lane_pin(m[j]);
    }
    lds_barrier();  // the readers of the previous chunk (and of dx) are done
#ifdef SML_TIMELINE
    ta = __builtin_amdgcn_s_memtime();
    if (c0 == 0) tl[6] = ta; else tl[9] += ta - tl[10];
#endif
    if (mcc >= 0) {
#pragma unroll
      for (int j = 0; j < NS; j++) S[(O_X + mcc * NS + j) * 64] = m[j];
    }
    lds_barrier();
#ifdef SML_TIMELINE
    tl[10] = __builtin_amdgcn_s_memtime();
    tl[8] += tl[10] - ta;
#endif
#ifdef SML_SKIP_FINAL
    continue;
#endif
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int c = c0 + q;
      if (c < NS && c <= cidx[NCOL - 1]) {  // (wave-uniform) some row of this role is at or below the diagonal of column c
        double mr[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) mr[j] = SER(O_X + q * NS + j);
#pragma unroll
        for (int t = 0; t < NCOL; t++) {
          const int r = cidx[t];
          if (r < NS && c <= r) {
#ifdef SML_FINAL_SPLIT   // attribution (round 5): the sum in three interleaved parts -- a multiply-add that waits for its predecessor costs twice its slot
            double acc = pkv[t][q], acc1 = 0.0, acc2 = 0.0;
#pragma unroll
            for (int j = 0; j < NS; j += 3) {
              acc = fma(z[t][j], mr[j], acc);
              if (j + 1 < NS) acc1 = fma(z[t][j + 1], mr[j + 1], acc1);
              if (j + 2 < NS) acc2 = fma(z[t][j + 2], mr[j + 2], acc2);
            }
            acc += acc1 + acc2;
#else
            double acc = pkv[t][q];
#pragma unroll
            for (int j = 0; j < NS; j++) acc = fma(z[t][j], mr[j], acc);
#endif
#ifdef SML_NO_PSTORE
            if (acc == 1.2345e300) out[tb + po[t][q]] = acc;
#elif defined(SML_WIDE)
            if (active) *reinterpret_cast<d2_t *>(out + tb + (po[t][q] & ~1)) = d2_t{ acc, acc };
#else
            if (active) gst(out + tb + po[t][q], acc);
#endif
          }
        }
      }
    }
  }
#ifdef SML_TIMELINE
  tl[7] = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == SML_TIMELINE && lane == 0)
    for (int i = 0; i < 16; i++) sml_tl[w][i] = tl[i];
#endif
}

#undef S
#undef SE
#undef SER
#undef LANE_FENCE

}  // namespace pb
