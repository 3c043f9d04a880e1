// pb_update_ct.hip -- stand-alone indexed (+ orientation) updates whose index list is one the handlers actually produce
// (compile-time core indices, diagonal R): 15 states on the two-role cooperative mapping (k_step_coop with PREDICT = false,
// rbis_coop.hpp), 21 states on the four-wave mapping (k_update_quad, rbis_quad.hpp) -- one coalesced round trip of the
// state, no column gather -- instead of the generic run-time-index kernel k_update.  LegOdoCommon's lin_rot_rate list reaches
// the pass-through angular-velocity states: 15 states take it on the one-lane in-register kernel k_update_lane.  See pb_ctx.hpp.
#include "pb_ctx.hpp"

template <int NS, int MH, class CORR>
static void launch_ct(pb_ctx *c, double *out, const CorrArgs &ca)
{
  k_step_coop<NS, false, MH, CORR, false><<<nblk(c->B), 128, 0, c->stream>>>(c->st, out, c->B, nullptr, nullptr, nullptr, 0.0, 0.0,
                                                                             0.0, 0.0, c->k, ca);
}
template <class CORR>
static void launch_ct_mh(pb_ctx *c, double *out, const CorrArgs &ca)
{
  if (c->ns == 15) {
    switch (c->mem_hint) {
    case MH_STORE_SC1: launch_ct<15, MH_STORE_SC1, CORR>(c, out, ca); break;
    case MH_STREAM_NT: launch_ct<15, MH_STREAM_NT, CORR>(c, out, ca); break;
    default: launch_ct<15, MH_DEFAULT, CORR>(c, out, ca); break;
    }
  } else if (c->quad21) {
    switch (c->mem_hint) {
    case MH_STORE_SC1: k_update_quad<CORR, MH_STORE_SC1><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    case MH_STREAM_NT: k_update_quad<CORR, MH_STREAM_NT><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    default: k_update_quad<CORR, MH_DEFAULT><<<nblk(c->B), 256, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    }
  } else if constexpr (CORR::M <= 4) {  // PRONTO_BATCH_QUAD21=0: the two-role mapping (its six-row variants spill: not built)
    switch (c->mem_hint) {
    case MH_STORE_SC1: launch_ct<21, MH_STORE_SC1, CORR>(c, out, ca); break;
    case MH_STREAM_NT: launch_ct<21, MH_STREAM_NT, CORR>(c, out, ca); break;
    default: launch_ct<21, MH_DEFAULT, CORR>(c, out, ca); break;
    }
  }
}

static bool same(const int *idx, int m, std::initializer_list<int> l)
{
  if ((int) l.size() != m) return false;
  int i = 0;
  for (int v : l)
    if (idx[i++] != v) return false;
  return true;
}

// returns PB_OK after a launch, -1 when this (idx, R kind, orientation) combination has no compile-time kernel
int pbk_update_ct(pb_ctx *c, int m, const int *idx, const double *z, const double *r2, const double *rb2, const double *qm,
                  const uint8_t *mask, const double *zb, const double *qb, const double *rfull)
{
  CorrArgs ca;
  ca.z2 = z; ca.r2 = r2; ca.qm2 = qm; ca.mask2 = mask; ca.rfull = rfull;
  if (rb2)
    for (int i = 0; i < m; i++) ca.rb2[i] = rb2[i];
  if (zb) {  // one measurement for every filter: host values as kernel arguments
    ca.zbc = 1;
    for (int i = 0; i < m; i++) ca.zb2[i] = zb[i];
    if (qb)
      for (int i = 0; i < 4; i++) ca.qb2[i] = qb[i];
  }
  const bool orient = qm != nullptr || qb != nullptr;
  int which = -1;
  if (same(idx, m, { 3, 4, 5 })) which = 0;
  else if (same(idx, m, { 9, 10, 11 })) which = 1;
  else if (same(idx, m, { 9, 10, 11, 3, 4, 5 })) which = 2;
  else if (orient && same(idx, m, { 9, 10, 11, 6, 7, 8 })) which = 3;
  else if (orient && same(idx, m, { 9, 10, 11, 8 })) which = 4;
  else if (orient && same(idx, m, { 3, 4, 5, 8 })) which = 5;
  else if (orient && same(idx, m, { 8 })) which = 6;
  else if (!orient && same(idx, m, { 8, 9, 10, 11 })) which = 7;          // GPF pos_yaw
  else if (!orient && same(idx, m, { 6, 7, 8, 9, 10, 11 })) which = 8;    // GPF pos_chi
  else if (!orient && same(idx, m, { 11 })) which = 9;                    // GPF z_only
  else if (!orient && c->ns == 15 && same(idx, m, { 3, 4, 5, 0, 1, 2 })) which = 10;  // LegOdoCommon lin_rot_rate
  if (which < 0) return -1;
  // two-role mapping only (A/B switch): 21 states with six measurement rows spill there; the generic kernel takes them
  if (c->ns == 21 && !c->quad21 && m > 4) return -1;
  double *out = update_target(c);
  switch (which) {
  case 0: launch_ct_mh<CorrVel>(c, out, ca); break;
  case 1: launch_ct_mh<CorrPos>(c, out, ca); break;
  case 2: launch_ct_mh<CorrPosVel>(c, out, ca); break;
  case 3: launch_ct_mh<CorrPosOrient>(c, out, ca); break;
  case 4: launch_ct_mh<CorrPosYaw>(c, out, ca); break;
  case 5: launch_ct_mh<CorrVelYaw>(c, out, ca); break;
  case 6: launch_ct_mh<CorrYaw>(c, out, ca); break;
  case 7: launch_ct_mh<CorrGpfYawPos>(c, out, ca); break;
  case 8: launch_ct_mh<CorrGpfChiPos>(c, out, ca); break;
  case 9: launch_ct_mh<CorrGpfZ>(c, out, ca); break;
  default:  // a list that reaches the pass-through states: the one-lane in-register kernel (15 states only)
    switch (c->mem_hint) {
    case MH_STORE_SC1: k_update_lane<15, 6, IdxVelOmega, MH_STORE_SC1><<<nblk(c->B), 64, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    case MH_STREAM_NT: k_update_lane<15, 6, IdxVelOmega, MH_STREAM_NT><<<nblk(c->B), 64, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    default: k_update_lane<15, 6, IdxVelOmega, MH_DEFAULT><<<nblk(c->B), 64, 0, c->stream>>>(c->st, out, c->B, c->k, ca); break;
    }
    break;
  }
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}
