// pb_step.hip -- launchers of the step kernels (k_step, k_step_coop, k_replay_fused); see pb_ctx.hpp.
#include "pb_ctx.hpp"

template <bool UPDATE, int MH>
static void launch_step_mh(pb_ctx *c, double *out, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
                           const StepBcast &bc)
{
  const int B = c->B;
  if (c->ns == 15 && c->coop15) {
    Consts kk = c->k;
    kk.half_tiles = c->half15 ? 1 : 0;   // (this launch only: every other kernel that takes c->k keeps whole tiles)
    k_step_coop<15, UPDATE, MH><<<nblk(B) * (c->half15 ? 2 : 1), 128, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], kk, CorrArgs(), bc);
  } else if (c->ns == 15) {
    k_step<15, UPDATE, MH><<<(B + PB_STEP_BLOCK - 1) / PB_STEP_BLOCK, PB_STEP_BLOCK, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, bc);
  } else if (c->quad21) {
    // n = 21: 231 packed covariance entries do not fit one lane's registers; four cooperating waves per tile at two waves
    // per SIMD (rbis_quad.hpp): one launch, one state round trip.
    k_step_quad<UPDATE, MH><<<nblk(B), 256, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, bc);
  } else {
    // the two-wave cooperative kernel at one wave per SIMD (rbis_coop.hpp); PRONTO_BATCH_QUAD21=0
    k_step_coop<21, UPDATE, MH><<<nblk(B), 128, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, CorrArgs(), bc);
  }
}

template <bool UPDATE>
static int launch_step(pb_ctx *c, const double *imu, const double *lo, const uint8_t *mask, const double q[4], const StepBcast &bc)
{
  double *out = update_target(c);
  int rc = PB_OK;
  imu = pbk_idle_prepare(c, imu, &rc);
  if (rc) return rc;
  switch (c->mem_hint) {  // cache policy of the state round trip, chosen in pb_create from the state size
  case MH_STORE_SC1: launch_step_mh<UPDATE, MH_STORE_SC1>(c, out, imu, lo, mask, q, bc); break;
  case MH_STREAM_NT: launch_step_mh<UPDATE, MH_STREAM_NT>(c, out, imu, lo, mask, q, bc); break;
  default: launch_step_mh<UPDATE, MH_DEFAULT>(c, out, imu, lo, mask, q, bc); break;
  }
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}


// The fused step on the filters [b0, b0 + nb) only -- b0 and the batch multiples of 64 (whole tiles), the posterior in place, the inputs
// still blocks [rows][B] of the WHOLE batch (the kernels take B as the row stride; the block is addressed by shifting every pointer).
// coop15 / mem_hint are the caller's: a block that fits the memory-side cache wants the kernel and the cache policy of ITS size.
template <int MH>
static void launch_range_mh(pb_ctx *c, double *st, const double *imu, const double *lo, const uint8_t *mask, const double q[4], int nb, bool coop15)
{
  const int B = c->B;
  const StepBcast bc = StepBcast();
  if (c->ns == 15 && coop15) k_step_coop<15, true, MH><<<nblk(nb), 128, 0, c->stream>>>(st, st, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, CorrArgs(), bc);
  else if (c->ns == 15) k_step<15, true, MH><<<(nb + PB_STEP_BLOCK - 1) / PB_STEP_BLOCK, PB_STEP_BLOCK, 0, c->stream>>>(st, st, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, bc);
  else if (c->quad21) k_step_quad<true, MH><<<nblk(nb), 256, 0, c->stream>>>(st, st, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, bc);
  else k_step_coop<21, true, MH><<<nblk(nb), 128, 0, c->stream>>>(st, st, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, CorrArgs(), bc);
}
int pbk_step_range(pb_ctx *c, const double *imu, const double *lo, const uint8_t *mask, const double q[4], long b0, int nb, bool coop15, int mem_hint)
{
  if (c->st != c->st_base || c->out_slot >= 0 || (c->B & 63) || (b0 & 63) || (nb & 63) || b0 + nb > c->B)
    return fail(c, PB_ERR_STATE, "pbk_step_range: whole tiles of a head that lives in the context's own array");
  const size_t tile_doubles = c->state_doubles / (size_t) (c->stride / 64);
  double *st = c->st + (size_t) (b0 / 64) * tile_doubles;
  const double *imu_b = imu + b0, *lo_b = lo + b0;
  const uint8_t *mask_b = mask ? mask + b0 : nullptr;
  switch (mem_hint) {
  case MH_STORE_SC1: launch_range_mh<MH_STORE_SC1>(c, st, imu_b, lo_b, mask_b, q, nb, coop15); break;
  case MH_STREAM_NT: launch_range_mh<MH_STREAM_NT>(c, st, imu_b, lo_b, mask_b, q, nb, coop15); break;
  default: launch_range_mh<MH_DEFAULT>(c, st, imu_b, lo_b, mask_b, q, nb, coop15); break;
  }
  LAUNCHCHK(c);
  return PB_OK;
}

int pbk_step(pb_ctx *c, bool update, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
             const StepBcast *bcast)
{
  const StepBcast bc = bcast ? *bcast : StepBcast();
  return update ? launch_step<true>(c, imu, lo, mask, q, bc) : launch_step<false>(c, imu, nullptr, nullptr, q, bc);
}

int pbk_step_leg(pb_ctx *c, const double *imu, const StepBcast *bcast, const double q[4], const LegIn &lin, int64_t utime, const LegMeasPar &mp,
                 double *lo_out, uint8_t *mask_out)
{
  // the two-wave 15-state mapping (the default up to 393 216 filters) and the four-wave 21-state mapping.  Mode lin_rate with
  // leg_estimate's world constraint switched on needs the stand-alone kernel (mode pos_and_lin_rate tracks it inside the pair kernel).
  if ((c->ns == 15 && !c->coop15) || (c->ns == 21 && !c->quad21) || (c->leg_par.world_constraint && mp.mode != 2)) return -1;
  // PRONTO_BATCH_LEG21_TWO=1: round 3's two launches for 21 states with per-filter joint blocks (A/B runs)
  static const bool two_launches21 = getenv("PRONTO_BATCH_LEG21_TWO") && getenv("PRONTO_BATCH_LEG21_TWO")[0] == '1';
  if (c->ns == 21 && lin.kind == 1 && two_launches21 && mp.mode == 0) return -1;
  const StepBcast bc = bcast ? *bcast : StepBcast();
  LegStepArgs la{ c->legd, c->legi, c->stride, utime, mp.r_v2, mp.r_v2_uncertain, lo_out, mask_out, mp };
  double *out = update_target(c);
  int rc = PB_OK;
  imu = pbk_idle_prepare(c, imu, &rc);
  if (rc) return rc;
  if (c->ns == 15) pbk_step_leg15(c, out, imu, q, bc, lin, la);
  else pbk_step_leg21(c, out, imu, q, bc, lin, la);
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}

int pbk_replay_fused(pb_ctx *c, int T, const double *imu, const double *lo, const uint8_t *mask, const double q[4], int slot0)
{
  SlotOut so;
  if (slot0 >= 0) {
    so.base = c->hist + (size_t) slot0 * c->state_doubles;
    so.stride = c->state_doubles;
  }
  // PRONTO_BATCH_REPLAY_ONELANE=1: the first, one-lane-per-filter version (15 states only; A/B runs)
  static const bool one_lane = getenv("PRONTO_BATCH_REPLAY_ONELANE") && getenv("PRONTO_BATCH_REPLAY_ONELANE")[0] == '1';
  if (c->ns == 15 && one_lane && slot0 < 0)
    k_replay_fused<15><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->B, T, imu, lo, mask, q[0], q[1], q[2], q[3], c->k);
  else if (c->ns == 15)
    k_replay_coop<15><<<nblk(c->B), 128, 0, c->stream>>>(c->st, c->B, T, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, so);
  else if (c->quad21)
    // four waves per tile, register budget of ONE wave per SIMD: 3.3e9 steps/s at 64k filters, T = 32; cut for two waves
    // per SIMD it carries 452 B of scratch and measured 2.4e9; the two-wave kernel below 1.7e9
    k_replay_quad<1><<<nblk(c->B), 256, 0, c->stream>>>(c->st, c->B, T, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, so);
  else
    k_replay_coop<21><<<nblk(c->B), 128, 0, c->stream>>>(c->st, c->B, T, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, so);
  LAUNCHCHK(c);
  return PB_OK;
}

// predict + leg-odometry update + one more (orientation) update in ONE state round trip on the cooperative kernel
template <int NS, int MH, class CORR>
static void launch_corr(pb_ctx *c, double *out, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
                        const CorrArgs &ca, const StepBcast &bc)
{
  k_step_coop<NS, true, MH, CORR><<<nblk(c->B), 128, 0, c->stream>>>(c->st, out, c->B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k, ca, bc);
}
template <int NS, class CORR>
static void launch_corr_mh(pb_ctx *c, double *out, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
                           const CorrArgs &ca, const StepBcast &bc)
{
  switch (c->mem_hint) {
  case MH_STORE_SC1: launch_corr<NS, MH_STORE_SC1, CORR>(c, out, imu, lo, mask, q, ca, bc); break;
  case MH_STREAM_NT: launch_corr<NS, MH_STREAM_NT, CORR>(c, out, imu, lo, mask, q, ca, bc); break;
  default: launch_corr<NS, MH_DEFAULT, CORR>(c, out, imu, lo, mask, q, ca, bc); break;
  }
}

int pbk_step_correct(pb_ctx *c, int corr_kind, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
                     const double *z2, const double *r2, const double *rb2, const double *qm2, const uint8_t *mask2,
                     const StepBcast *bcast, const double *zb, const double *qb)
{
  if (c->ns == 21) {
    // 21 states: role C's 15 x 15 sub-matrix already fills the register file (256 VGPR + 242 AGPR for the plain step); a
    // second update stage in the same kernel spills 550-690 bytes per lane and measured SLOWER than two launches.  Same
    // arithmetic as two launches: the fused step, then the correction alone on the cooperative update kernel.
    int rc = pbk_step(c, true, imu, lo, mask, q, bcast);
    if (rc) return rc;
    static const int idx_po[6] = { 9, 10, 11, 6, 7, 8 }, idx_py[4] = { 9, 10, 11, 8 };
    const int m2 = (corr_kind == PB_CORR_POS_ORIENT) ? 6 : 4;
    const int *idx2 = (corr_kind == PB_CORR_POS_ORIENT) ? idx_po : idx_py;
    const int slot = pb_head_slot(c);  // a checkpointed step: the correction lands in the same slot
    if (slot >= 0) c->out_slot = slot;
    rc = pbk_update_ct(c, m2, idx2, z2, r2, rb2, qm2, mask2, zb, qb);
    if (rc >= 0) return rc;
    if (!z2 || !qm2) return fail(c, PB_ERR_ARG, "pb_step_legodo_correct: no kernel takes this correction as arguments; stage the blocks");
    return pbk_update21(c, m2, idx2, z2, r2 ? r2 : rb2, r2 ? PB_R_DIAG : PB_R_DIAG_BROADCAST, r2 ? nullptr : rb2, qm2, mask2);
  }
  CorrArgs ca;
  ca.z2 = z2; ca.r2 = r2; ca.qm2 = qm2; ca.mask2 = mask2;
  const int m2 = (corr_kind == PB_CORR_POS_ORIENT) ? 6 : 4;
  if (rb2)
    for (int i = 0; i < m2; i++) ca.rb2[i] = rb2[i];
  if (zb) {  // one correction for every filter: kernel arguments
    ca.zbc = 1;
    for (int i = 0; i < m2; i++) ca.zb2[i] = zb[i];
    for (int i = 0; i < 4; i++) ca.qb2[i] = qb[i];
  }
  const StepBcast bc = bcast ? *bcast : StepBcast();
  double *out = update_target(c);
  int rc = PB_OK;
  imu = pbk_idle_prepare(c, imu, &rc);
  if (rc) return rc;
  if (corr_kind == PB_CORR_POS_ORIENT) launch_corr_mh<15, CorrPosOrient>(c, out, imu, lo, mask, q, ca, bc);
  else launch_corr_mh<15, CorrPosYaw>(c, out, imu, lo, mask, q, ca, bc);
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}
