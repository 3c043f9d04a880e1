// pb_update_rt21.hip -- launcher of the generic (run-time index list) update of a 21-state batch: k_update_quad_rt<M, MH>
// (rbis_quad_rt.hpp).  The kernel is expensive to compile (a 21-way column pick per measurement and wave), so the six m are
// spread over three objects built in parallel: -DPB_UPD_MSET=0 (m = 1, 2, 3), 1 (m = 4, 5), 2 (m = 6); see pb_ctx.hpp.
#include "pb_ctx.hpp"
#include "rbis_quad_rt.hpp"

template <int M>
static int launch_rt21(pb_ctx *c, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                       const uint8_t *mask)
{
  IdxArg<M> ia;
  DiagArg<M> da;
  for (int i = 0; i < M; i++) {
    if (idx[i] < 0 || idx[i] > 20) return fail(c, PB_ERR_ARG, "update: index %d out of range", idx[i]);
    ia.v[i] = idx[i];
    da.v[i] = rb ? rb[i] : 0.0;
  }
  double *out = update_target(c);
  // (two cache policies are built: non-temporal streaming falls back to the default one)
  if (c->mem_hint == MH_STORE_SC1)
    k_update_quad_rt<M, MH_STORE_SC1><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
  else
    k_update_quad_rt<M, MH_DEFAULT><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}

#if PB_UPD_MSET == 0
int pbk_update21_m123(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                      const uint8_t *mask)
{
  if (m == 1) return launch_rt21<1>(c, idx, z, R, rkind, rb, qm, mask);
  if (m == 2) return launch_rt21<2>(c, idx, z, R, rkind, rb, qm, mask);
  return launch_rt21<3>(c, idx, z, R, rkind, rb, qm, mask);
}
#elif PB_UPD_MSET == 1
int pbk_update21_m45(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                     const uint8_t *mask)
{
  if (m == 4) return launch_rt21<4>(c, idx, z, R, rkind, rb, qm, mask);
  return launch_rt21<5>(c, idx, z, R, rkind, rb, qm, mask);
}
#else
int pbk_update21_m6(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                    const uint8_t *mask)
{
  static const int lin_rot_rate[6] = { 3, 4, 5, 0, 1, 2 };  // rbis_legodo_common.cpp:66-67
  if (!c->generic_update && memcmp(idx, lin_rot_rate, sizeof lin_rot_rate) == 0) {
    DiagArg<6> da;
    for (int i = 0; i < 6; i++) da.v[i] = rb ? rb[i] : 0.0;
    double *out = update_target(c);
    if (c->mem_hint == MH_STORE_SC1)
      k_update_quad_list<MH_STORE_SC1, 3, 4, 5, 0, 1, 2><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, z, R, rkind, da, qm, mask, c->k);
    else
      k_update_quad_list<MH_DEFAULT, 3, 4, 5, 0, 1, 2><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, z, R, rkind, da, qm, mask, c->k);
    LAUNCHCHK(c);
    update_done(c, out);
    return PB_OK;
  }
  return launch_rt21<6>(c, idx, z, R, rkind, rb, qm, mask);
}
int pbk_update21(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                 const uint8_t *mask)
{
  if (m >= 1 && m <= 3) return pbk_update21_m123(c, m, idx, z, R, rkind, rb, qm, mask);
  if (m == 4 || m == 5) return pbk_update21_m45(c, m, idx, z, R, rkind, rb, qm, mask);
  if (m == 6) return pbk_update21_m6(c, m, idx, z, R, rkind, rb, qm, mask);
  return fail(c, PB_ERR_ARG, "update: m must be 1..6");
}
#endif
