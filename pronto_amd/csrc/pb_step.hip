// pb_step.hip -- launchers of the step kernels (k_step, k_step_coop, k_replay_fused); see pb_ctx.hpp.
#include "pb_ctx.hpp"

template <bool UPDATE, int MH>
static void launch_step_mh(pb_ctx *c, double *out, const double *imu, const double *lo, const uint8_t *mask, const double q[4])
{
  const int B = c->B;
  if (c->ns == 15 && c->coop15) {
    k_step_coop<15, UPDATE, MH><<<nblk(B), 128, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k);
  } else if (c->ns == 15) {
    k_step<15, UPDATE, MH><<<(B + PB_STEP_BLOCK - 1) / PB_STEP_BLOCK, PB_STEP_BLOCK, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k);
  } else {
    // n = 21: 231 packed covariance entries do not fit one lane's registers; the step runs on the two-wave
    // cooperative kernel (rbis_coop.hpp): one launch, one state round trip.
    k_step_coop<21, UPDATE, MH><<<nblk(B), 128, 0, c->stream>>>(c->st, out, B, imu, lo, mask, q[0], q[1], q[2], q[3], c->k);
  }
}

template <bool UPDATE>
static int launch_step(pb_ctx *c, const double *imu, const double *lo, const uint8_t *mask, const double q[4])
{
  double *out = update_target(c);
  switch (c->mem_hint) {  // cache policy of the state round trip, chosen in pb_create from the state size
  case MH_STORE_SC1: launch_step_mh<UPDATE, MH_STORE_SC1>(c, out, imu, lo, mask, q); break;
  case MH_STREAM_NT: launch_step_mh<UPDATE, MH_STREAM_NT>(c, out, imu, lo, mask, q); break;
  default: launch_step_mh<UPDATE, MH_DEFAULT>(c, out, imu, lo, mask, q); break;
  }
  LAUNCHCHK(c);
  update_done(c, out);
  return PB_OK;
}


int pbk_step(pb_ctx *c, bool update, const double *imu, const double *lo, const uint8_t *mask, const double q[4])
{
  return update ? launch_step<true>(c, imu, lo, mask, q) : launch_step<false>(c, imu, nullptr, nullptr, q);
}

int pbk_replay_fused(pb_ctx *c, int T, const double *imu, const double *lo, const uint8_t *mask, const double q[4])
{
  k_replay_fused<15><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->B, T, imu, lo, mask, q[0], q[1], q[2], q[3], c->k);
  LAUNCHCHK(c);
  return PB_OK;
}
