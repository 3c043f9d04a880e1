// pb_smooth_wide.hip -- launcher of k_smooth_wide (rbis_smooth_wide.hpp), the RTS smoother step for 15 states.  A translation unit of its
// own because it is compiled with -mllvm -disable-machine-licm (Makefile): the kernel's body is one loop over the workgroup's tiles, and
// the backend's loop-invariant code motion lifts ~60 constants of the attitude arithmetic in front of it, holds them in registers the
// kernel does not have and spills them to scratch (372 bytes per lane; 162 vs 139 us per step when that was measured).
#include "pb_ctx.hpp"
#include "rbis_smooth_wide.hpp"

int pbk_smooth_wide(pb_ctx *c, const double *np_, const double *ns_, const double *cu, double *out, double dt)
{
  if (c->ns != 15) return fail(c, PB_ERR_ARG, "pbk_smooth_wide: 15 states only");
  if (!c->smooth_wide_attr) {  // more than the default 64 KB of dynamic LDS per workgroup
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_wide<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) SmoothWideCfg<15>::LDS_BYTES));
    c->smooth_wide_attr = true;
  }
  // persistent: one workgroup per CU walks the tiles (PRONTO_SMOOTH_GRID: workgroups, for experiments)
  static const int grid_env = getenv("PRONTO_SMOOTH_GRID") ? atoi(getenv("PRONTO_SMOOTH_GRID")) : 0;
  const int ntiles = (c->B + 63) / 64, nwg = grid_env > 0 ? grid_env : c->n_cu;
  k_smooth_wide<15><<<dim3((unsigned) (ntiles < nwg ? ntiles : nwg)), SmoothWideCfg<15>::THREADS, SmoothWideCfg<15>::LDS_BYTES, c->stream>>>(np_, ns_, cu, out, c->B, ntiles, dt,
                                                                                                                                       c->k);
  LAUNCHCHK(c);
#ifdef SML_TIMELINE  // attribution build: print the stamps of launch 60 of scripts/smooth_rate.py
  {
    static int calls = 0;
    if (++calls == 60) {
      unsigned long long h[8][16];
      (void) hipStreamSynchronize(c->stream);
      (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(sml_tl), sizeof(h));
      for (int w = 0; w < SmoothWideCfg<15>::NR; w++) {
        const double t0 = (double) h[0][0];
        auto T = [&](int i) { return (h[w][i] - t0) / 1e3; };
        fprintf(stderr, "timeline wide n=15 tile %d role %d [k cycles from role 0's start] P^- rows in LDS %.1f, barrier behind them %.1f, factorised %.1f (first barrier %.1f), "
                        "rhs done %.1f, substituted %.1f, D and dx in LDS %.1f, M made %.1f, first half published %.1f, its products done %.1f, second half published %.1f, "
                        "posterior staged %.1f, end %.1f\n", SML_TIMELINE, w, T(11), T(12), T(2), T(1), T(3), T(4), T(5), T(6), T(8), T(9), T(10), T(13), T(7));
      }
    }
  }
#endif
  return PB_OK;
}
