// rbis_lds_stream.hpp -- LDS reads as an explicit, software-pipelined stream of ds_read_b64 (round 5): used by the smoother kernels
// (rbis_smooth_wide.hpp, rbis_smooth_lane.hpp).  Device code only.
#pragma once

#include <hip/hip_runtime.h>

#include "rbis_device.hpp"

namespace pb {

// ---- LDS reads as an explicit pipeline -------------------------------------------------------------------------------------------
// One wave per SIMD has nobody to hide an LDS round trip behind, and the backend schedules `read, wait, use, read, wait, use` once the
// accumulators fill the 256 architectural registers (the M sweep: 60 exposed round trips).  lds_stream reads a compile-time list of
// entries G at a time into D + 1 buffers with ds_read_b64 of its own (single reads: the LDS serves two of them in half the time of the
// paired ds_read2st64_b64 the backend prefers), the reads of groups g + 1 .. g + D issued BEFORE the wait for group g (a counted s_waitcnt: LDS
// operations of a wave return in order; a scalar load the backend may have in flight can only make the wait longer, never shorter).
// `use(k, value)` is called for k = 0 .. N-1 in order with k a compile-time constant; `pin(g)` after every group g: it names what the
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
struct LdsBases { int b0, b1, b2; };   // byte addresses of this lane's entries 0, 128, 256 ([entry][lane] layout: 512 bytes per entry)
template <int N, class EntryOf, int D = 1, class Use, class Pin>
__device__ __forceinline__ void lds_stream(LdsBases bb, Use &&use, Pin &&pin)
{
  // D groups are in flight while one is consumed (D = 2 where a group's arithmetic is shorter than an LDS round trip)
  constexpr int G = 8, NG = (N + G - 1) / G;
  double buf[D + 1][G];
#pragma unroll
  for (int q = 0; q <= D; q++)
#pragma unroll
    for (int j = 0; j < G; j++) buf[q][j] = 0.0;
  auto size_of = [](int g) constexpr { return g < NG ? ((N - g * G < G) ? N - g * G : G) : 0; };
  auto issue = [&](auto GG) {
    constexpr int g = decltype(GG)::value;
    static_for<G>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value, kq = g * G + j;
      if constexpr (kq < N) {
        constexpr int e = EntryOf::at(kq);
        if constexpr (e < 128) lds_rd_b64<e * 512>(buf[g % (D + 1)][j], bb.b0);
        else if constexpr (e < 256) lds_rd_b64<(e - 128) * 512>(buf[g % (D + 1)][j], bb.b1);
        else lds_rd_b64<(e - 256) * 512>(buf[g % (D + 1)][j], bb.b2);
      }
    });
  };
  static_for<D>([&](auto GG) {
    if constexpr (decltype(GG)::value < NG) issue(GG);
  });
  static_for<NG>([&](auto GG) {
    constexpr int g = decltype(GG)::value;
    if constexpr (g + D < NG) issue(std::integral_constant<int, g + D>{});
    constexpr int younger_all = size_of(g + 1) + (D >= 2 ? size_of(g + 2) : 0) + (D >= 3 ? size_of(g + 3) : 0);
    constexpr int younger = younger_all > 15 ? 15 : younger_all;   // (the counter's field has 4 bits: waiting for one read more is harmless)
    lds_wait8<younger>(buf[g % (D + 1)]);
    static_for<G>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value, kq = g * G + j;
      if constexpr (kq < N) use(std::integral_constant<int, kq>{}, buf[g % (D + 1)][j]);
    });
    pin(std::integral_constant<int, g>{});
  });
}
// entry lists
struct SmwDiag { static constexpr int at(int k) { return pk(k, k); } };
// Orders in which CONSECUTIVE entries touch different accumulators (a multiply-add that waits for its predecessor's result costs twice
// its issue slot, and one wave per SIMD has nothing else to issue meanwhile):
template <int NS>
struct SmwByDiagonal {  // the lower triangle diagonal by diagonal: (d, 0), (d + 1, 1), ... for d = 0 .. n-1
  static constexpr int dg(int k) { int d = 0; while (k >= NS - d) { k -= NS - d; d++; } return d; }
  static constexpr int col(int k) { int d = 0; while (k >= NS - d) { k -= NS - d; d++; } return k; }
  static constexpr int row(int k) { return col(k) + dg(k); }
  static constexpr int at(int k) { return pk(row(k), col(k)); }
};
template <int NS>
struct SmwLowerByColumn {  // k-th (i, m), m < i, column by column: z_i -= l_im z_m for i = m+1 .. n-1, m = 0 .. n-2
  static constexpr int col(int k) { int m = 0; while (k >= NS - 1 - m) { k -= NS - 1 - m; m++; } return m; }
  static constexpr int row(int k) { int m = 0; while (k >= NS - 1 - m) { k -= NS - 1 - m; m++; } return m + 1 + k; }
  static constexpr int at(int k) { return pk(row(k), col(k)); }
};
template <int NS>
struct SmwUpperByColumn {  // k-th (m, i), i < m: z_i -= l_mi z_m for i = m-1 .. 0, m = n-1 .. 1
  static constexpr int cm(int k) { int m = NS - 1; while (k >= m) { k -= m; m--; } return m; }
  static constexpr int ci(int k) { int m = NS - 1; while (k >= m) { k -= m; m--; } return m - 1 - k; }
  static constexpr int at(int k) { return pk(cm(k), ci(k)); }
};
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


}  // namespace pb
