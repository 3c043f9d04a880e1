// pb_smooth.hip -- launcher of the RTS smoother kernel (rbis_smooth.hpp); see pb_ctx.hpp.
#include "pb_ctx.hpp"
#include "rbis_smooth.hpp"
#include "rbis_smooth_lane.hpp"

int pbk_smooth_step(pb_ctx *c, const double *np_, const double *ns_, const double *cu, double *out, double dt)
{
#ifdef PB_EXPERIMENTS  // attribution builds only (scripts/smooth_attribution.sh): extra dynamic LDS to force fewer workgroups per CU
  static const size_t pad = [] {
    const char *e = getenv("PRONTO_SMOOTH_LDS_PAD");
    const long v = e ? atol(e) : 0;
    return (size_t) (v < 0 ? 0 : (v > 65536 ? 65536 : v));
  }();
#else
  constexpr size_t pad = 0;
#endif
  // PRONTO_SMOOTH_PIVOT=1: Eigen's diagonal pivoting in the factorisation of P^- (the reference's .ldlt(); parity with the oracle at
  // 1e-15); default: no pivot search (P^- is SPD; rbis_smooth.hpp)
  static const bool pivot = getenv("PRONTO_SMOOTH_PIVOT") && getenv("PRONTO_SMOOTH_PIVOT")[0] == '1';
  // PRONTO_SMOOTH_KERNEL=reg: the 16 / 32-lanes-per-filter kernel of rounds 2-4 (rbis_smooth.hpp); default: one lane per filter,
  // the dense work split over role waves (rbis_smooth_lane.hpp)
  static const bool reg_kernel = pivot || (getenv("PRONTO_SMOOTH_KERNEL") && !strcmp(getenv("PRONTO_SMOOTH_KERNEL"), "reg"));
  // 15 states: k_smooth_wide (rbis_smooth_wide.hpp, pb_smooth_wide.hip: persistent workgroups, one wave per SIMD with 512 registers, data
  // movement by whole rows) unless PRONTO_SMOOTH_KERNEL=lane asks for k_smooth_lane (two tiles per CU, 256 registers: the default until
  // the second half of round 5, and still the kernel for 21 states)
  static const bool lane15 = getenv("PRONTO_SMOOTH_KERNEL") && !strcmp(getenv("PRONTO_SMOOTH_KERNEL"), "lane");
  if (!c->smooth_attr) {  // more than the default 64 KB of dynamic LDS per workgroup
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_lane<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) (SmoothLaneCfg<15>::LDS_BYTES + pad)));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_lane<21>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) SmoothLaneCfg<21>::LDS_BYTES));
    const int l15 = (int) (pad + sizeof(double) * SmoothRegCfg<15>::LDS_DOUBLES), l21 = (int) (pad + sizeof(double) * SmoothRegCfg<21>::LDS_DOUBLES);
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_reg<15, true>), hipFuncAttributeMaxDynamicSharedMemorySize, l15));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_reg<21, true>), hipFuncAttributeMaxDynamicSharedMemorySize, l21));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_reg<15, false>), hipFuncAttributeMaxDynamicSharedMemorySize, l15));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_reg<21, false>), hipFuncAttributeMaxDynamicSharedMemorySize, l21));
    c->smooth_attr = true;
  }
  if (!reg_kernel) {
    const dim3 grid((unsigned) ((c->B + 63) / 64));
    // (the pad of the attribution builds: past the kernel's own LDS, never touched -- it only keeps a second workgroup off the CU)
    if (c->ns == 15 && !lane15) return pbk_smooth_wide(c, np_, ns_, cu, out, dt);
    if (c->ns == 15) k_smooth_lane<15><<<grid, SmoothLaneCfg<15>::THREADS, SmoothLaneCfg<15>::LDS_BYTES + pad, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
    else k_smooth_lane<21><<<grid, SmoothLaneCfg<21>::THREADS, SmoothLaneCfg<21>::LDS_BYTES, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
  } else if (c->ns == 15) {
    using S = SmoothRegCfg<15>;
    const dim3 grid((unsigned) ((c->B + S::F - 1) / S::F));
    const size_t ldsb = pad + sizeof(double) * S::LDS_DOUBLES;
    if (pivot) k_smooth_reg<15, true><<<grid, S::THREADS, ldsb, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
    else k_smooth_reg<15, false><<<grid, S::THREADS, ldsb, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
  } else {
    using S = SmoothRegCfg<21>;
    const dim3 grid((unsigned) ((c->B + S::F - 1) / S::F));
    const size_t ldsb = pad + sizeof(double) * S::LDS_DOUBLES;
    if (pivot) k_smooth_reg<21, true><<<grid, S::THREADS, ldsb, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
    else k_smooth_reg<21, false><<<grid, S::THREADS, ldsb, c->stream>>>(np_, ns_, cu, out, c->B, dt, c->k);
  }
  LAUNCHCHK(c);
#ifdef SML_TIMELINE  // attribution build: print the stamps of launch 60 (n = 15) and 310 (n = 21) of scripts/smooth_rate.py
  {
    static int calls = 0;
    if (++calls == 60 || calls == 310) {
      unsigned long long h[8][16];
      (void) hipStreamSynchronize(c->stream);
      (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(sml_tl), sizeof(h));
      const int nr = c->ns == 15 ? SmoothLaneCfg<15>::NR : SmoothLaneCfg<21>::NR;
      for (int w = 0; w < nr; w++) {
        const double t0 = (double) h[0][0];
        fprintf(stderr, "timeline n=%d tile %d role %d [k cycles from role 0's start] first factor barrier %.1f, last %.1f, rhs done %.1f, subst done %.1f, "
                        "D staged %.1f, first chunk barrier %.1f, end %.1f; in the chunks: publish phases %.1f, final+M phases %.1f\n", c->ns, SML_TIMELINE, w,
                (h[w][1] - t0) / 1e3, (h[w][2] - t0) / 1e3, (h[w][3] - t0) / 1e3, (h[w][4] - t0) / 1e3, (h[w][5] - t0) / 1e3, (h[w][6] - t0) / 1e3,
                (h[w][7] - t0) / 1e3, h[w][8] / 1e3, h[w][9] / 1e3);

      }
    }
  }
#endif
  return PB_OK;
}
