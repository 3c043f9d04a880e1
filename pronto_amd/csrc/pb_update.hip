// pb_update.hip -- launcher of the generic (run-time index list) update of a 15-state batch: k_update_lane_rt<15, M, ORIENT, MH>
// (one lane, the filter in registers) for m <= 4, k_update_coop_rt<M, MH> (two waves per tile, rbis_quad_rt.hpp) for m = 5, 6
// where the one-lane kernel spills; 21 states: pb_update_rt21.hip.  See pb_ctx.hpp.
#include "pb_ctx.hpp"
#include "rbis_quad_rt.hpp"

template <int NS, int M, int MH>
static void launch_update_mh(pb_ctx *c, const IdxArg<M> &ia, const DiagArg<M> &da, const double *z, const double *R,
                             int rkind, const double *qm, const uint8_t *mask)
{
  double *out = update_target(c);
  if constexpr (M >= 5) {  // five / six gathered columns beside the covariance do not fit one lane: two waves per tile
    // (two cache policies are built: non-temporal streaming falls back to the default one)
    if (MH == MH_STORE_SC1)
      k_update_coop_rt<M, MH_STORE_SC1><<<nblk(c->B), 128, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
    else
      k_update_coop_rt<M, MH_DEFAULT><<<nblk(c->B), 128, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
  } else {  // the whole 15-state filter fits one lane's registers: no column gather (k_update_lane_rt)
    if (qm)
      k_update_lane_rt<NS, M, true, MH><<<nblk(c->B), 64, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
    else
      k_update_lane_rt<NS, M, false, MH><<<nblk(c->B), 64, 0, c->stream>>>(c->st, out, c->B, ia, z, R, rkind, da, qm, mask, c->k);
  }
  update_done(c, out);
}

template <int NS, int M>
static void launch_update_m(pb_ctx *c, const int *idx, const double *z, const double *R, int rkind, const double *rb,
                            const double *qm, const uint8_t *mask)
{
  IdxArg<M> ia;
  DiagArg<M> da;
  for (int i = 0; i < M; i++) {
    ia.v[i] = idx[i];
    da.v[i] = rb ? rb[i] : 0.0;
  }
  switch (c->mem_hint) {
  case MH_STORE_SC1: launch_update_mh<NS, M, MH_STORE_SC1>(c, ia, da, z, R, rkind, qm, mask); break;
  case MH_STREAM_NT: launch_update_mh<NS, M, MH_STREAM_NT>(c, ia, da, z, R, rkind, qm, mask); break;
  default: launch_update_mh<NS, M, MH_DEFAULT>(c, ia, da, z, R, rkind, qm, mask); break;
  }
}

template <int NS>
static int launch_update(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind,
                         const double *rb, const double *qm, const uint8_t *mask)
{
  switch (m) {
    case 1: launch_update_m<NS, 1>(c, idx, z, R, rkind, rb, qm, mask); break;
    case 2: launch_update_m<NS, 2>(c, idx, z, R, rkind, rb, qm, mask); break;
    case 3: launch_update_m<NS, 3>(c, idx, z, R, rkind, rb, qm, mask); break;
    case 4: launch_update_m<NS, 4>(c, idx, z, R, rkind, rb, qm, mask); break;
    case 5: launch_update_m<NS, 5>(c, idx, z, R, rkind, rb, qm, mask); break;
    case 6: launch_update_m<NS, 6>(c, idx, z, R, rkind, rb, qm, mask); break;
    default: return fail(c, PB_ERR_ARG, "update: m must be 1..6");
  }
  LAUNCHCHK(c);
  return PB_OK;
}


int pbk_update15(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb,
                 const double *qm, const uint8_t *mask)
{
  return launch_update<15>(c, m, idx, z, R, rkind, rb, qm, mask);
}
