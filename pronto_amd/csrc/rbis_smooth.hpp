// rbis_smooth.hpp -- RTS smoother step (ekfSmoothingStep, state-estimator/src/mav_state_est/rbis.cpp:234-266) on gfx950.
//
//   L      = P_k Ad^T (P^-_{k+1})^-1            (Ad about the filtered state at k; bias blocks of P^- replaced by I
//   P^s_k  = P_k + L (P^s_{k+1} - P^-_{k+1}) L^T    when their variance is < 1e-11, rbis.cpp:244-251)
//   x^s_k  = x_k (+) L (x^s_{k+1} (-) x^-_{k+1})
//
// Unlike the forward step this IS dense n x n work (an SPD solve with n right-hand sides and two dense products,
// ~40 kflop per 21-state filter against ~8 KB of state: ~5 flop/B, at the fp64 balance point), so one lane per filter
// is the wrong shape: a GROUP of G = 16 / 32 lanes owns one filter, lane r owns matrix row r (k_smooth_reg below).
// The reference calls Eigen's .ldlt(): the kernel keeps Eigen's diagonal pivoting (same pivot order keeps parity at
// 1e-15 instead of cond(P^-) * eps).
// The factorisation and the substitutions are VALU + LDS; the two n x n products run on the matrix pipe
// (v_mfma_f64_16x16x4_f64: one 16 x 16 tile per 15-state filter, 2 x 2 tiles with k padded to 24 per 21-state filter, step 5): its fp64 rate equals the vector rate on MI355X, the
// gain is the VALU / LDS work and the row registers it takes away, not arithmetic throughput.
// Build flags SM_SKIP_* / SM_NO_* / SM_COPY_ONLY / SM_EMPTY compile parts out for scripts/smooth_attribution.sh (timing only).
#pragma once

#if !defined(PB_EXPERIMENTS) && (defined(SM_EMPTY) || defined(SM_NO_LOAD) || defined(SM_NO_STORE) || defined(SM_COPY_ONLY) || \
                                 defined(SM_SKIP_QUAT) || defined(SM_SKIP_FACT) || defined(SM_SKIP_SUBST) || defined(SM_SKIP_PROD) || \
                                 defined(SM_OCC3))
#error "the SM_* attribution flags compile parts of the smoother OUT (garbage results): they need -DPB_EXPERIMENTS as well"
#endif

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "rbis_device.hpp"

namespace pb {

// Orders this wave's LDS traffic: the hardware executes one wave's DS instructions in issue order, so all that is needed
// is that the compiler keeps them in program order across this point.
__device__ __forceinline__ void group_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// =================================================================================================================
// k_smooth_reg<NS>: the same step with the factorisation in REGISTERS and every run-time index turned into an LDS
// ADDRESS.  This is the kernel pb_smooth_step launches; k_smooth_step above is kept as the A/B reference
// (PRONTO_BATCH_SMOOTH_LDS=1).
//
// The LDS kernel is a chain of dependent LDS read-modify-write round trips (PMC: 65 % of wave time waiting, ~180 k
// cycles per wave) at 1.75 waves per SIMD, and its row-per-lane global loads touch 16 component rows x 4 filters per
// instruction, each 128-byte line fetched by four workgroups on four XCDs.  Here:
//   * G = 16 (n = 15) or 32 (n = 21) lanes own one filter, lane r owns matrix row r, 256-thread workgroups of 16 / 8
//     filters; every checkpoint is staged through LDS with filter-fastest global accesses;
//   * pivoted LDL^T without swaps: rows stay in their lanes' registers.  Step kk: butterfly arg-max of the remaining
//     |A_ii| (DPP row rotations; Eigen's rule, ties to the smallest CURRENT position so that the pivot sequence is
//     Eigen's with its swaps), the pivot lane writes its row to LDS, every lane reads it back (broadcast), takes
//     l_r = A[p][r] / d and updates its row.  One LDS round trip per step, nothing is read-modify-written in LDS;
//   * no run-time register index anywhere: x is put in pivot order by reading LDS at address piv[kk]; L is published as
//     the unit lower-triangular factor in pivot order (lane r writes row mypos), so forward and backward substitution
//     are compile-time triangular loops over read-only broadcast reads; the gain row goes back to row order the same way;
//   * P^s_row = P_row + (g_row D) G^T with D (packed) and the gain G read as LDS broadcasts.
// Three compiler behaviours had to be fenced off, each worth hundreds of registers (the first build: 512 + 2 KB scratch):
// select chains over a register array are turned into a load through a selected POINTER (array pinned in scratch);
// values that only feed a store inside `if (row)` are sunk into that block below everything (their LDS operands stay
// live); LDS reads of later steps are hoisted / kept alive for a later re-read (step_fence).
// =================================================================================================================
// workgroup size (attribution builds: -DSM_THREADS=128 / 64 -- fewer filters per workgroup, the same waves per CU)
#ifndef SM_THREADS
#define SM_THREADS 256
#endif
template <int NS>
struct SmoothRegCfg {
  static constexpr int G = (NS <= 16) ? 16 : 32;  // lanes per filter
  static constexpr int THREADS = SM_THREADS, F = THREADS / G;
  static constexpr int NC = Lay<NS>::NC, PITCH = NC | 1;  // odd pitch: conflict-free filter-fastest staging
  // Strides = 2 (mod 4) doubles, i.e. 4 (mod 8) dwords, everywhere a stride separates things that are accessed together:
  //  * row pitch PG of the n x n scratch matrices: even, so rows are 16-byte aligned and are read with ds_read_b128
  //    (16 B per lane in 4 LDS cycles; the compiler's ds_read2_b64 for the same bytes takes 16), and NOT a multiple of
  //    16 dwords, so the lanes of a group reading or writing one COLUMN (lane = row) fall on distinct banks -- a pitch
  //    of 16 doubles for n = 15 made every such access a 16-way conflict (PMC: half of all LDS cycles);
  //  * per-filter strides: the 64/G filters of one wave read their rows as 16-byte broadcasts at the same time; with
  //    4 (mod 8) dwords between them their 4-bank groups never coincide.
  static constexpr int bank_stride(int x) { return (x % 4 == 2) ? x : x + ((6 - x % 4) % 4); }
  static constexpr int KPAD = (NS <= 16) ? 16 : 24;  // columns of an operand row: the k range of the matrix products, zero-padded
  static constexpr int PG = bank_stride(KPAD), MATP = NS * PG;
  static constexpr int U_PER = bank_stride((PITCH + 1 > MATP) ? PITCH + 1 : MATP);  // staging, then L, x, L, L^T, gain^T
  static constexpr int U_DOUBLES = F * U_PER;
  static constexpr int D_PER = bank_stride(MATP);  // D = P^s - P^- (full, row pitch PG)
  // small per-filter buffer of four RBW-wide slots: pivot row of the current step (entry NS = dummy for the padding
  // lanes), reciprocal pivots, residual, dx
  static constexpr int RBW = (NS + 2) & ~1, RB_INV = RBW, RB_RES = 2 * RBW, RB_DX = 3 * RBW, RB = bank_stride(4 * RBW);
#ifdef SM_OCC3
  static constexpr int LDS_DOUBLES = F * RB + U_DOUBLES;
#else
  static constexpr int LDS_DOUBLES = F * RB + U_DOUBLES + F * D_PER;
#endif
};

// 16-byte LDS access: p must be an even number of doubles from the (16-byte aligned) start of LDS
__device__ __forceinline__ d2_t lds_ld2(const double *p)
{
  return *reinterpret_cast<const d2_t *>(__builtin_assume_aligned(p, 16));
}
__device__ __forceinline__ void lds_st2(double *p, double a, double b)
{
  d2_t v = { a, b };
  *reinterpret_cast<d2_t *>(__builtin_assume_aligned(p, 16)) = v;
}

// v_mfma_f64_16x16x4_f64 (one 16 x 16 x 4 product per wave; maps checked by scripts/mfma_f64_probe.hip): lane l supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15] and receives D[(l >> 4) + 4 v][l & 15] in result register v = 0..3.
typedef double d4_t __attribute__((ext_vector_type(4)));
// Column c of an n <= 16 row sits at position mfma_pos(c): the four k = h, h + 4, h + 8, h + 12 that lane group h = l >> 4
// supplies to the four MFMAs of a 16-deep product are then 32 contiguous bytes (two ds_read_b128).
__host__ __device__ constexpr int mfma_pos(int c) { return 4 * (c % 4) + c / 4; }
// 21 states: k is padded to 24 = 6 slices of 4, lane group h supplies k = h + 4 s, s = 0..5: 48 contiguous bytes
__host__ __device__ constexpr int mfma_pos6(int c) { return 6 * (c % 4) + c / 4; }
__host__ __device__ constexpr int mfma_col6(int p) { return 4 * (p % 6) + p / 6; }  // column held at position p

// End of one unrolled step: the scheduling barrier keeps later steps' LDS reads from being hoisted to the top, the
// memory clobber makes a re-read in a later phase a real LDS read instead of a value kept in a register since its first use.
__device__ __forceinline__ void step_fence()
{
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// DPP row rotation inside each 16-lane row (mov_dpp has no "old" operand: every lane has a valid source)
template <int S>
__device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x120 + S, 0xF, 0xF, true); }
template <int S>
__device__ __forceinline__ double row_ror_d(double v)
{
  return __hiloint2double(row_ror_i<S>(__double2hiint(v)), row_ror_i<S>(__double2loint(v)));
}
// Pivot search of one factorisation step over the G lanes of a group (Eigen's rule: the largest remaining |A_ii|, ties to
// the smallest CURRENT position).  Two all-reduces over row rotations instead of one (value, key) butterfly with a compare /
// select chain per stage: the maximum of |d| (two DPP moves + v_max_f64 per stage), then the minimum key among the lanes
// that hold exactly that value (v_min_u32 with a DPP operand).  The key carries the sign of the winner's d in its lowest
// bit, so every lane can rebuild the signed pivot without another exchange.
template <int S>
__device__ __forceinline__ double max_stage(double v) { return fmax(v, row_ror_d<S>(v)); }
template <int S>
__device__ __forceinline__ unsigned min_stage(unsigned v)
{
  const unsigned o = (unsigned) row_ror_i<S>((int) v);
  return o < v ? o : v;
}

// 32-lane groups (21 states): the last stage joins the two 16-lane rows of a group.  gfx950's v_permlane16_swap exchanges
// the odd rows of one register with the even rows of another: applied to two copies of v it leaves the even row's value
// in both rows of one result and the odd row's value in both rows of the other -- a VALU instruction where __shfl_xor(v, 16)
// is an LDS-crossbar round trip (ds_bpermute) on the factorisation's critical path.
__device__ __forceinline__ double rowpair_max(double v)
{
  const unsigned hi = (unsigned) __double2hiint(v), lo = (unsigned) __double2loint(v);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  return fmax(__hiloint2double((int) h[0], (int) l[0]), __hiloint2double((int) h[1], (int) l[1]));
}
__device__ __forceinline__ unsigned rowpair_min(unsigned v)
{
  const auto x = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return x[0] < x[1] ? x[0] : x[1];
}

// SM_OCC3 (attribution build, garbage results): what a THIRD workgroup per CU would buy with this instruction stream -- the D
// buffer aliases the gain buffer (LDS per workgroup 77.6 -> 43 KB for 15 states) and the register budget is the 168 of three
// waves per SIMD (whatever does not fit goes to scratch): an upper bound for the restructuring DESIGN.md 4 describes.
#ifdef SM_OCC3
#define PB_SMOOTH_WG_PER_CU 3
#else
#define PB_SMOOTH_WG_PER_CU 2
#endif
// PIVOT = true: Eigen's diagonal pivoting (the reference calls .ldlt(): same pivot sequence, parity with the oracle at 1e-15).
// PIVOT = false: NO pivot search -- P^- (with the bias-block fix) is symmetric positive definite, for which the unpivoted
// factorisation is backward stable and its accuracy, like Cholesky's, is governed by the condition of the diagonally SCALED
// matrix (the correlations), not by the spread of the variances (1e-10 for a bias, 0.25 for a position): step kk's pivot is row
// kk, known at compile time, so the two cross-lane reductions per step (8-10 DPP stages on the critical path), the position
// bookkeeping and the permutation of the right-hand side disappear.  Results differ from the pivoted path by rounding
// (tests: <= 1e-9 against the oracle; observed 1e-13).
template <int NS, bool PIVOT = true>
__global__ __launch_bounds__(SM_THREADS, PB_SMOOTH_WG_PER_CU) void k_smooth_reg(const double *__restrict__ next_pred,
                                                    const double *__restrict__ next_sm,
                                                    const double *__restrict__ cur, double *__restrict__ out,
                                                    int B, double dt, Consts k)
{
  using L = Lay<NS>;
  using SL = Slots<NS>;
  using C = SmoothRegCfg<NS>;
  constexpr int G = C::G, F = C::F, PITCH = C::PITCH, PG = C::PG, MATP = C::MATP;
  constexpr bool MFMA = (NS <= 16);  // the two n x n products of step 5 on v_mfma_f64_16x16x4 (one 16 x 16 tile per filter)
  constexpr bool MFMA21 = !MFMA;     // 21 states: the same on 2 x 2 tiles of 16 x 16 with k padded to 24 (two filters per wave)
  static_assert(!MFMA || (G == 16 && PG >= 16), "one filter per 16-lane group, rows padded to 16 columns");
  static_assert(NS < 24 && C::U_DOUBLES >= F * PITCH + C::THREADS, "buffer slots");
  extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef SM_EMPTY  // workgroup dispatch cost alone: same registers, same LDS request, no work
  if (B > 0) return;
#endif
#ifdef SM_SKEW  // attribution: the second workgroup of every CU in the first dispatch round starts SM_SKEW x 3.4 us late, so that
                // the two workgroups of a CU are out of phase (one stages while the other computes) instead of in lock-step
  if (blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < SM_SKEW; i++) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef SM_OCC3
  double *U = lds + F * C::RB, *DP = U;
#else
  double *U = lds + F * C::RB, *DP = U + C::U_DOUBLES;  // [small buffers | staging / x / L / gain | D packed]
#endif
  const int t = threadIdx.x;
  const int f = t / G, r = t % G;        // compute mapping: filter slot f, matrix row r
  const int sf = t % F, sc = t / F;      // staging mapping: filter fastest
  const long b0 = (long) blockIdx.x * F;
  const long sb = (b0 + sf < B) ? b0 + sf : (long) B - 1;  // slots past the batch end shadow the last filter
  const bool row = r < NS;               // padding lanes mirror row NS-1 and write only dummy slots
  const int rr = row ? r : NS - 1;
  double *Uf = U + f * PITCH;            // this filter in the staging layout
  double *Lf = U + f * C::U_PER;         // this filter's n x n scratch (row pitch PG)
  double *Df = DP + f * C::D_PER;        // this filter's D = P^s - P^-
  double *Rf = lds + f * C::RB;          // this filter's small buffer
  // checkpoint -> LDS, filter fastest: thread (sf, sc) moves storage rows sc, sc+G, ... of filter sf, 16 bytes each (the F
  // filters of a workgroup sit side by side in one tile: F x 16 contiguous bytes per row), and scatters the two
  // components of a row to their canonical places in the staging layout
  const long srow0 = (sb >> 6) * SL::TILE_DOUBLES + (sb & 63) * 2;
  // The three checkpoints are requested from memory up front (`issue`) and scattered into the staging layout one after
  // the other (`commit`): the kernel is latency-bound, and with load -> scatter -> barrier per checkpoint the
  // second and third global-load latencies were exposed one after the other.  21 states have no registers to spare
  // (254 VGPRs): only the first two are in flight together there.
  constexpr int NST = (SL::NROW + G - 1) / G;  // storage rows per staging thread
  auto issue = [&](const double *src, d2_t (&v)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int r2 = sc + i * G;
#ifdef SM_NO_LOAD
      v[i] = d2_t{ 1.0 + 0.001 * r2 + 1e-6 * threadIdx.x, 0.5 + 0.002 * r2 };
#else
      v[i] = (r2 < SL::NROW) ? *reinterpret_cast<const d2_t *>(src + srow0 + (long) r2 * 128) : d2_t{ 0.0, 0.0 };
#endif
    }
  };
  auto commit = [&](const d2_t (&v)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int r2 = sc + i * G;
      if (r2 < SL::NROW) {
        const int c0 = SL::T.comp_of[2 * r2], c1 = SL::T.comp_of[2 * r2 + 1];
        U[sf * PITCH + c0] = v[i].x;
        if (c1 >= 0) U[sf * PITCH + c1] = v[i].y;
      }
    }
  };
  constexpr bool PREFETCH_CUR = (NS <= 16);
  d2_t vp[NST], vs[NST], vc[NST];
  issue(next_pred, vp);
  issue(next_sm, vs);
  if constexpr (PREFETCH_CUR) issue(cur, vc);
  int poff[NS];                          // packed offsets of this lane's row
#pragma unroll
  for (int j = 0; j < NS; j++) poff[j] = L::OFF_P + pk_rt(rr, j);

  // ---- 1. operands through LDS: P^- row, D = P^s - P^- (packed, LDS), residual, P_k row, prior state ----
  double am[NS], prow[NS];
  commit(vp);
  __syncthreads();
  double qp[4];
#pragma unroll
  for (int j = 0; j < NS; j++) am[j] = Uf[poff[j]];
#pragma unroll
  for (int i = 0; i < 4; i++) qp[i] = Uf[L::OFF_QUAT + i];
  const double xpr = Uf[L::OFF_VEC + rr];
  double dg = Uf[L::OFF_P + pk_rt(rr, rr)];  // own diagonal entry, carried through the factorisation
  // rbis.cpp:244-251 replaces the bias-bias blocks of P^- by I (when a variance is < 1e-11) ONLY in the matrix that is
  // factorised; D = P^s - P^- below uses the uncorrected entries (rbis.cpp:256), kept here
  double am_raw[NS == 21 ? 6 : 1] = { 0 };
  if constexpr (NS == 21) {
#pragma unroll
    for (int j = 0; j < 6; j++) am_raw[j] = am[15 + j];
  }
  if constexpr (NS == 21) {  // bias-block fix
    bool fix_g = false, fix_a = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      fix_g = fix_g | (Uf[L::OFF_P + pk(15 + i, 15 + i)] < .00000000001);
      fix_a = fix_a | (Uf[L::OFF_P + pk(18 + i, 18 + i)] < .00000000001);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      am[15 + j] = (fix_g & (rr >= 15) & (rr < 18)) ? ((rr - 15 == j) ? 1.0 : 0.0) : am[15 + j];
      am[18 + j] = (fix_a & (rr >= 18) & (rr < 21)) ? ((rr - 18 == j) ? 1.0 : 0.0) : am[18 + j];
    }
    dg = ((fix_g & (rr >= 15) & (rr < 18)) | (fix_a & (rr >= 18) & (rr < 21))) ? 1.0 : dg;
  }
  __syncthreads();
  commit(vs);
  if constexpr (!PREFETCH_CUR) issue(cur, vc);
  __syncthreads();
  {
    // D = P^s - P^- with the UNCORRECTED P^- (the bias fix applies to the factorised matrix only): every lane its full
    // row, bitwise symmetric because both (r,j) and (j,r) subtract the same two packed entries
    double *const drow = row ? Df + rr * PG : Rf;  // padding lanes: the pivot-row buffer is not in use yet
    auto pminus = [&](int j) { return (NS == 21 && j >= 15) ? am_raw[j - 15] : am[j]; };
    if constexpr (MFMA) {  // columns in mfma_pos order (pairs (j, j + 4) are neighbours there); column 15 = 0
      auto dval = [&](int j) { return j < NS ? Uf[poff[j < NS ? j : 0]] - pminus(j < NS ? j : 0) : 0.0; };
      static_for<8>([&](auto JJ) {
        constexpr int j = (decltype(JJ)::value % 4) + 8 * (decltype(JJ)::value / 4);
        lds_st2(drow + mfma_pos(j), dval(j), dval(j + 4));
      });
    } else {  // 21 states: mfma_pos6 order, columns 21..23 = 0
      auto dval = [&](int j) { return j < NS ? Uf[poff[j < NS ? j : 0]] - pminus(j < NS ? j : 0) : 0.0; };
      static_for<12>([&](auto JJ) {
        constexpr int j = (decltype(JJ)::value % 4) + 8 * (decltype(JJ)::value / 4);
        lds_st2(drow + mfma_pos6(j), dval(j), dval(j + 4));
      });
    }
    double qs[4], dchi[3];
#pragma unroll
    for (int i = 0; i < 4; i++) qs[i] = Uf[L::OFF_QUAT + i];
#ifdef SM_SKIP_QUAT
    dchi[0] = qs[1] - qp[1]; dchi[1] = qs[2] - qp[2]; dchi[2] = qs[3] - qp[3];
#else
    subtract_quats(qs, qp, dchi);  // chi = Log(q^-^-1 q^s)   (rbis.cpp:259-261)
#endif
    double res = Uf[L::OFF_VEC + rr] - xpr;
    if (rr >= 6 && rr <= 8) res = (rr == 6) ? dchi[0] : (rr == 7 ? dchi[1] : dchi[2]);
    Rf[row ? C::RB_RES + rr : NS] = res;
  }
  __syncthreads();
  commit(vc);
  __syncthreads();
#ifdef SM_COPY_ONLY
  if (b0 + sf < B) {
#pragma unroll 4
    for (int r2 = sc; r2 < SL::NROW; r2 += G) {
      const int c0 = SL::T.comp_of[2 * r2], c1 = SL::T.comp_of[2 * r2 + 1];
      const d2_t v2 = { U[sf * PITCH + c0] + am[0], c1 >= 0 ? U[sf * PITCH + c1] : 0.0 };
      *reinterpret_cast<d2_t *>(out + srow0 + (long) r2 * 128) = v2;
    }
  }
  return;
#endif
  double w[3], v[3], q[4];
#pragma unroll
  for (int j = 0; j < NS; j++) prow[j] = Uf[poff[j]];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    w[i] = Uf[L::OFF_VEC + i];
    v[i] = Uf[L::OFF_VEC + 3 + i];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = Uf[L::OFF_QUAT + i];
  const double xcur = Uf[L::OFF_VEC + rr], llcur = Uf[L::OFF_LL];
  // 15 states: P_k of the four filters of this wave in the MFMA result layout (the accumulator of step 5)
  const int mh = (t >> 4) & 3, mj = (r < NS) ? r : NS - 1, fw = f & ~3;
  d4_t acc[MFMA ? 4 : 1];
  if constexpr (MFMA) {
#pragma unroll
    for (int ff = 0; ff < 4; ff++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = (mh + 4 * v < NS) ? mh + 4 * v : NS - 1;  // row / column 15 do not exist: never stored
        acc[ff][v] = U[(fw + ff) * PITCH + L::OFF_P + pk_rt(i, mj)];
      }
  }
  double chi_cur[3];
#pragma unroll
  for (int i = 0; i < 3; i++) chi_cur[i] = Uf[L::OFF_VEC + 6 + i];
  __syncthreads();  // the staging buffer is free from here

  // ---- 2. pivoted LDL^T of P^-: rows in registers, the pivot row goes round through LDS ----
  int piv[NS];
  constexpr bool KEEP_L = (NS <= 16);
  double lreg[KEEP_L ? NS : 1];
  int mypos = 0;                 // pivot position of this lane's row
  double *const mine = row ? Lf + rr * PG : Rf + NS;  // this lane's row of the n x n scratch (padding lanes: dummy slot)
  const int one = row ? 1 : 0;
  {
    bool done = !row;            // padding lanes are never candidates
    int pos = r;                 // current position of this row under Eigen's swaps (tie-break only)
    static_for<NS>([&](auto KK) {
      constexpr int kk = decltype(KK)::value;
#ifdef SM_SKIP_FACT
      piv[kk] = kk; if (r == kk) mypos = kk; if constexpr (KEEP_L) lreg[kk] = 0.0; Rf[C::RB_INV + kk] = 1.0; return;
#endif
      if constexpr (!PIVOT) {
        const bool is_k = row && (r == kk);
        piv[kk] = kk;
        if (is_k) {
#pragma unroll
          for (int j = 0; j + 1 < NS; j += 2) lds_st2(Rf + j, am[j], am[j + 1]);
          if (NS & 1) Rf[NS - 1] = am[NS - 1];
          Rf[C::RB_INV + kk] = (fabs(dg) > 5.562684646268003e-309) ? 1.0 / dg : 0.0;
          mypos = kk;
        }
        group_sync();
        const double inv_k = Rf[C::RB_INV + kk];
        const double c_k = Rf[rr];                                 // A[kk][r]
        const double l_k = (row && r > kk) ? c_k * inv_k : 0.0;    // rows above the pivot are done, the pivot row keeps itself
#pragma unroll
        for (int j = 0; j < NS; j += 2) {
          const d2_t pr = lds_ld2(Rf + j);
          am[j] = fma(-l_k, pr.x, am[j]);
          if (j + 1 < NS) am[j + 1] = fma(-l_k, pr.y, am[j + 1]);
        }
        dg = fma(-l_k, c_k, dg);
        if constexpr (KEEP_L) lreg[kk] = l_k;
        else mine[one * kk] = l_k;
        group_sync();
        step_fence();
        return;
      }
      // candidates: |d| >= 0; rows pivoted earlier and the padding lanes: -1 (never the maximum while a row remains)
      const double ad = done ? -1.0 : fabs(dg);
      double mx = max_stage<1>(ad);
      mx = max_stage<2>(mx);
      mx = max_stage<4>(mx);
      mx = max_stage<8>(mx);
      if constexpr (G == 32) mx = rowpair_max(mx);
      unsigned key = (!done & (ad == mx)) ? (unsigned) ((pos * 32 + r) * 2 + (dg < 0.0 ? 1 : 0)) : (0x40000000u + 2u * (unsigned) r);
      key = min_stage<1>(key);
      key = min_stage<2>(key);
      key = min_stage<4>(key);
      key = min_stage<8>(key);
      if constexpr (G == 32) key = rowpair_min(key);
      const double cs = (key & 1u) ? -mx : mx;  // the pivot, signed
      const int p = (int) (key >> 1) & 31;      // pivot row (= lane of the group), identical in all lanes of the group
      const bool is_p = (r == p);
      piv[kk] = p;
      const double inv = (fabs(cs) > 5.562684646268003e-309) ? 1.0 / cs : 0.0;  // Eigen's solve() tolerance: 1/highest
      // Eigen swaps position kk with the pivot's position: the row sitting at kk inherits the pivot's old position
      if (!done && pos == kk) pos = (int) (key >> 6);
      if (is_p) {
#pragma unroll
        for (int j = 0; j + 1 < NS; j += 2) lds_st2(Rf + j, am[j], am[j + 1]);
        if (NS & 1) Rf[NS - 1] = am[NS - 1];
        Rf[C::RB_INV + kk] = inv;
        mypos = kk;
      }
      group_sync();
      const double c = Rf[rr];                                   // A[p][r] (= A[r][p])
      const double l = (done || is_p) ? 0.0 : c * inv;
#pragma unroll
      for (int j = 0; j < NS; j += 2) {                          // A[r][j] -= l_r A[p][j]
        const d2_t pr = lds_ld2(Rf + j);
        am[j] = fma(-l, pr.x, am[j]);
        if (j + 1 < NS) am[j + 1] = fma(-l, pr.y, am[j + 1]);
      }
      dg = fma(-l, c, dg);
      // L[r][kk]: 0 for the pivot row itself and for rows pivoted earlier.  15 states: kept in registers (there is
      // room); 21 states: parked in this lane's row of the LDS scratch (42 more live registers would spill)
      if constexpr (KEEP_L) lreg[kk] = l;
      else mine[one * kk] = l;
      done = done || is_p;
      group_sync();
      step_fence();
    });
  }

  // ---- 3. right-hand side: x = column r of T = Ad P_k = Ad (row r of P_k)^T, Ad = I + dt Ac (rbis.cpp:12-35) ----
  double z[NS];
  {
    double R[9];
    quat_to_rot(q, R);
    const double gb[3] = { -k.g * R[6], -k.g * R[7], -k.g * R[8] };
    const double pv[3] = { prow[3], prow[4], prow[5] }, pc[3] = { prow[6], prow[7], prow[8] };
#pragma unroll
    for (int i = 0; i < NS; i++) z[i] = prow[i];
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
      z[3 + i] = fma(dt, av, z[3 + i]);
      z[6 + i] = fma(dt, ac, z[6 + i]);
      z[9 + i] = fma(dt, ad, z[9 + i]);
    }
  }

  // ---- 4. A y = x.  (Publishing stores are unconditional with a selected address -- padding lanes write a dummy slot --
  //         because whatever only feeds a store inside `if (row)` is sunk into that block, see step 6.) ----
  double lrow[NS];               // this lane's L row, kept while the scratch carries x
#pragma unroll
  for (int kk = 0; kk < NS; kk++) lrow[kk] = KEEP_L ? lreg[kk] : mine[one * kk];
  group_sync();
#pragma unroll
  for (int i = 0; i < NS; i++) mine[one * i] = z[i];
  group_sync();
#pragma unroll
  for (int kk = 0; kk < NS; kk++) z[kk] = Lf[rr * PG + piv[kk]];   // x in pivot order
  group_sync();
  {
    double *const lp = row ? Lf + mypos * PG : Rf + NS;
#pragma unroll
    for (int kk = 0; kk < NS; kk++) lp[one * kk] = lrow[kk];       // L in pivot order: unit lower triangular, by rows
  }
  group_sync();
  static_for<NS>([&](auto KK) {  // forward: z[kk] -= sum_{m<kk} L[kk][m] z[m], row kk read 16 bytes at a time
    constexpr int kk = decltype(KK)::value;
#ifdef SM_SKIP_SUBST
    return;
#endif
    double s = z[kk];
#pragma unroll
    for (int m = 0; m < kk; m += 2) {
      const d2_t lv = lds_ld2(Lf + kk * PG + m);
      s = fma(-lv.x, z[m], s);
      if (m + 1 < kk) s = fma(-lv.y, z[m + 1], s);
    }
    z[kk] = s;
    if constexpr (kk % 4 == 3 || kk == NS - 1) step_fence();
  });
#pragma unroll
  for (int kk = 0; kk < NS; kk++) z[kk] *= Rf[C::RB_INV + kk];
  group_sync();  // every lane of the group is done reading L by rows
  {
    const int col = row ? mypos : NS;  // padding lanes: the pad column of an even pitch, or the dummy slot
    double *const lt = (row || PG > NS) ? Lf + col : Rf + NS;
    const int step = (row || PG > NS) ? PG : 0;
#pragma unroll
    for (int kk = 0; kk < NS; kk++) lt[step * kk] = lrow[kk];       // L^T by rows: Lt[kk][m] = L[m][kk]
  }
  group_sync();
  static_for<NS>([&](auto KR) {  // backward: z[kk] -= sum_{m>kk} L[m][kk] z[m]
    constexpr int kk = NS - 1 - decltype(KR)::value;
#ifdef SM_SKIP_SUBST
    return;
#endif
    double s = z[kk];
#pragma unroll
    for (int m = (kk + 1) & ~1; m < NS; m += 2) {
      const d2_t lv = lds_ld2(Lf + kk * PG + m);
      if (m > kk) s = fma(-lv.x, z[m], s);
      if (m + 1 < NS) s = fma(-lv.y, z[m + 1], s);
    }
    z[kk] = s;
    if constexpr (kk % 4 == 0) step_fence();
  });
  group_sync();  // every lane of the group is done reading L^T
  // z[kk] = ((P^-)^-1 T[:,r])_{piv[kk]} = G[r][piv[kk]]; published TRANSPOSED (Gt[i][r] = G[r][i]) so that step 5
  // reads a row, and read back in row order for this lane's own gain row
  double gain[NS];
  if constexpr (MFMA) {
    // 15 states: by rows, columns in mfma_pos order, column 15 = 0 (operand of the matrix instructions of step 5)
    double *const gr = row ? Lf + rr * PG : Rf + NS;
#pragma unroll
    for (int kk = 0; kk < NS; kk++) gr[one * (((piv[kk] & 3) << 2) | (piv[kk] >> 2))] = z[kk];
    gr[one * 15] = 0.0;
    group_sync();
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const d2_t gv = lds_ld2(Lf + rr * PG + c);
      // position c holds column (c % 4) * 4 + c / 4 ... inverse of mfma_pos: column = 4 * (c % 4) + c / 4 as well
      if (mfma_pos(c) < NS) gain[mfma_pos(c)] = gv.x;
      if (mfma_pos(c + 1) < NS) gain[mfma_pos(c + 1)] = gv.y;
    }
  } else {
    // 21 states: by rows, columns in mfma_pos6 order, columns 21..23 (positions 11, 17, 23) = 0
    double *const gr = row ? Lf + rr * PG : Rf + NS;
#pragma unroll
    for (int kk = 0; kk < NS; kk++) gr[one * (6 * (piv[kk] & 3) + (piv[kk] >> 2))] = z[kk];
    gr[one * 11] = 0.0;
    gr[one * 17] = 0.0;
    gr[one * 23] = 0.0;
    group_sync();
#pragma unroll
    for (int c = 0; c < 24; c += 2) {
      const d2_t gv = lds_ld2(Lf + rr * PG + c);
      if (mfma_col6(c) < NS) gain[mfma_col6(c)] = gv.x;
      if (mfma_col6(c + 1) < NS) gain[mfma_col6(c + 1)] = gv.y;
    }
  }
  step_fence();
  // dx = G resid (rbis.cpp:263).  21 states: ahead of the products, so that the gain row is dead under them and its scratch
  // row can take the product; 15 states: after them (measured 1 % faster there)
  double dx = 0.0;
  auto state_delta = [&]() {
#pragma unroll
    for (int a = 0; a < NS; a++) dx = fma(gain[a], Rf[C::RB_RES + a], dx);
    Rf[row ? C::RB_DX + rr : NS] = dx;
  };
  if constexpr (MFMA21) {
    state_delta();
    step_fence();
  }

  // ---- 5. P^s_row = P_row + (g D) G^T: row bcol of D (symmetric) and row bcol of Gt, 16 bytes at a time ----
  if constexpr (MFMA) {
    // 15 states on the matrix pipe: F = D G^T, then P^s = P_k + G F, one 16 x 16 x 16 product = four v_mfma_f64_16x16x4 each,
    // for the four filters of this wave (independent chains).  Lane (h, j) supplies D[j][h + 4 s] and G[j][h + 4 s]
    // (s = 0..3: contiguous in mfma_pos order), F's result register s IS the B operand of k-slice s of the second product.
#ifndef SM_SKIP_PROD
    d4_t gq[4], dq[4];
#pragma unroll
    for (int ff = 0; ff < 4; ff++) {
      const double *const gp = U + (fw + ff) * C::U_PER + mj * PG + 4 * mh, *const dp = DP + (fw + ff) * C::D_PER + mj * PG + 4 * mh;
      const d2_t g0 = lds_ld2(gp), g1 = lds_ld2(gp + 2), d0 = lds_ld2(dp), d1 = lds_ld2(dp + 2);
      gq[ff] = d4_t{ g0.x, g0.y, g1.x, g1.y };
      dq[ff] = d4_t{ d0.x, d0.y, d1.x, d1.y };
    }
    d4_t fq[4];
#pragma unroll
    for (int ff = 0; ff < 4; ff++) fq[ff] = d4_t{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
      for (int ff = 0; ff < 4; ff++) fq[ff] = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[ff][s4], gq[ff][s4], fq[ff], 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
      for (int ff = 0; ff < 4; ff++) acc[ff] = __builtin_amdgcn_mfma_f64_16x16x4f64(gq[ff][s4], fq[ff][s4], acc[ff], 0, 0, 0);
#endif
  } else {
#ifndef SM_SKIP_PROD
    // 21 states on the matrix pipe: 2 x 2 tiles of 16 x 16, k padded to 24 = 6 slices.  F = D G^T (4 tiles x 6 MFMAs), then the
    // lower tiles (0,0), (1,0), (1,1) of G F (3 x 6): F's tile (tk, tj) has row 16 tk + h + 4 v in result register v of lane
    // (h, j) -- the B operand of k-slice (tk, v) -- and the matching A operand G[i][16 tk + h + 4 v] is slice 4 tk + v of the
    // six this lane group holds.  The product goes back through this filter's scratch (its gain is in registers by now)
    // into the row layout and is added to the P_k row there.
    const int ml = t & 63, mh2 = ml >> 4, mi = ml & 15, fw2 = f & ~1;
    const int r0 = mi, r1 = (16 + mi < NS) ? 16 + mi : NS - 1;  // rows 21..31 do not exist: finite stand-ins, never stored
#pragma unroll
    for (int ff = 0; ff < 2; ff++) {
      double *const Gb = U + (fw2 + ff) * C::U_PER;
      const double *const Db = DP + (fw2 + ff) * C::D_PER;
      double gq[2][6], dq[2][6];
#pragma unroll
      for (int ti = 0; ti < 2; ti++) {
        const int rw = ti ? r1 : r0;
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          const d2_t gv = lds_ld2(Gb + rw * PG + 6 * mh2 + c), dv = lds_ld2(Db + rw * PG + 6 * mh2 + c);
          gq[ti][c] = gv.x; gq[ti][c + 1] = gv.y;
          dq[ti][c] = dv.x; dq[ti][c + 1] = dv.y;
        }
      }
      d4_t fq[2][2], oq[3];
#pragma unroll
      for (int tk = 0; tk < 2; tk++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) fq[tk][tj] = d4_t{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int s6 = 0; s6 < 6; s6++)
#pragma unroll
        for (int tk = 0; tk < 2; tk++)
#pragma unroll
          for (int tj = 0; tj < 2; tj++)
            fq[tk][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[tk][s6], gq[tj][s6], fq[tk][tj], 0, 0, 0);
#pragma unroll
      for (int o = 0; o < 3; o++) oq[o] = d4_t{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int s6 = 0; s6 < 6; s6++)
#pragma unroll
        for (int o = 0; o < 3; o++) {
          const int ti = (o + 1) / 2, tj = o / 2;  // (0,0), (1,0), (1,1)
          oq[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(gq[ti][s6], fq[s6 / 4][tj][s6 % 4], oq[o], 0, 0, 0);
        }
      group_sync();  // this wave's operand reads of the filter's scratch are done
#pragma unroll
      for (int o = 0; o < 3; o++) {
        const int ti = (o + 1) / 2, tj = o / 2;
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int orow = 16 * ti + mh2 + 4 * v, ocol = 16 * tj + mi;
          double *const dst = (orow < NS && ocol <= orow) ? Gb + orow * PG + ocol : Rf + NS;
          *dst = oq[o][v];
        }
      }
    }
    group_sync();
#pragma unroll
    for (int m = 0; m < NS; m += 2) {  // this lane's row of G D G^T (entries m <= row are the ones stored and used)
      const d2_t pv = lds_ld2(Lf + rr * PG + m);
      prow[m] += pv.x;
      if (m + 1 < NS) prow[m + 1] += pv.y;
    }
    step_fence();
#endif
  }
  // ---- 6. state: cur.addState(RBIS(dx))  (rbis.cpp:263-265) ----
  if constexpr (!MFMA21) state_delta();
  __syncthreads();  // all waves are done with L / G: the buffer becomes the output staging area
  // Unconditional stores with a selected ADDRESS (entries this lane does not own go to a private dummy slot behind the
  // staging layout): with the stores inside `if (row && m <= rr)` the compiler sinks all the multiply-adds of step 5
  // into that block, below the barrier, and keeps every LDS operand they need alive (or spilled) until then.
  const int dummy = F * PITCH + t;
  if constexpr (MFMA) {  // lane (h, j) holds P^s[h + 4 v][j] of the wave's four filters: the lower triangle goes out
#pragma unroll
    for (int ff = 0; ff < 4; ff++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = mh + 4 * v;
        U[(i < NS && r <= i) ? (fw + ff) * PITCH + L::OFF_P + i * (i + 1) / 2 + r : dummy] = acc[ff][v];
      }
  } else {
#pragma unroll
    for (int m = 0; m < NS; m++) U[(row && m <= rr) ? f * PITCH + L::OFF_P + rr * (rr + 1) / 2 + m : dummy] = prow[m];  // (m <= rr: pk(rr, m))
  }
  U[(row && !(rr >= 6 && rr <= 8)) ? f * PITCH + L::OFF_VEC + rr : dummy] = xcur + dx;
  if (r == 0) {
    double dchi[3] = { Rf[C::RB_DX + 6], Rf[C::RB_DX + 7], Rf[C::RB_DX + 8] };
    double dq[4] = { 1.0, 0.0, 0.0, 0.0 };
#ifndef SM_SKIP_QUAT
    fold_chi(dchi, dq, k.chi_tol);  // RBIS(vec) constructor
#endif
    double chi[3];
#pragma unroll
    for (int i = 0; i < 3; i++) chi[i] = chi_cur[i] + dchi[i];
    double qq[4] = { q[0], q[1], q[2], q[3] };
#ifndef SM_SKIP_QUAT
    fold_chi(chi, qq, k.chi_tol);
#endif
    double o[4];
    quat_mul(qq, dq, o);
#pragma unroll
    for (int i = 0; i < 3; i++) Uf[L::OFF_VEC + 6 + i] = chi[i];
#pragma unroll
    for (int i = 0; i < 4; i++) Uf[L::OFF_QUAT + i] = o[i];
    Uf[L::OFF_LL] = llcur;
  }
  __syncthreads();
  if (b0 + sf < B) {
#pragma unroll 4
    for (int r2 = sc; r2 < SL::NROW; r2 += G) {
      const int c0 = SL::T.comp_of[2 * r2], c1 = SL::T.comp_of[2 * r2 + 1];
      const d2_t v2 = { U[sf * PITCH + c0], c1 >= 0 ? U[sf * PITCH + c1] : 0.0 };
#ifdef SM_NO_STORE
      if (v2.x == 1.2345e300) *reinterpret_cast<d2_t *>(out + srow0 + (long) r2 * 128) = v2;
#else
      *reinterpret_cast<d2_t *>(out + srow0 + (long) r2 * 128) = v2;
#endif
    }
  }
}

}  // namespace pb
