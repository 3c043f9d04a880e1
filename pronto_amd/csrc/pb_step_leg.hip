// pb_step_leg.hip -- launchers of the pair kernels (rbis_legstep.hpp): IMU step + leg odometry + its update in one kernel.
// Built as two objects (-DPB_LEG_NS=15 | 21: k_step_leg | k_step_quad_leg) so that `make -j` compiles them side by side;
// each holds the three measurement modes (SIX = 0 lin_rate, 1 lin_rot_rate, 2 pos_and_lin_rate) x the three cache policies.
#include "pb_ctx.hpp"

#ifndef PB_LEG_NS
#error "PB_LEG_NS = 15 | 21"
#endif
#ifndef PB_LEG_EARLY_SIX2
#define PB_LEG_EARLY_SIX2 12   // panel rows role P requests ahead of the odometry in mode pos_and_lin_rate (it holds the whole state vector there)
#endif

template <int MH, int SIX>
static void launch_step_leg(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin,
                            const LegStepArgs &la)
{
  // (the division of the leg work between the waves and the number of panel rows requested ahead of the odometry are template
  // parameters with measured defaults, rbis_legstep.hpp; -DPB_EXPERIMENTS builds the alternatives: PRONTO_BATCH_LEGPLAN / _LEGEARLY)
#define LEG_ARGS c->st, out, c->B, imu, q[0], q[1], q[2], q[3], c->k, bc, c->leg_par, lin, c->leg_chain, la
#if PB_LEG_NS == 15
#ifdef PB_EXPERIMENTS
  static const int plan = getenv("PRONTO_BATCH_LEGPLAN") ? atoi(getenv("PRONTO_BATCH_LEGPLAN")) : -1;
  static const int early = getenv("PRONTO_BATCH_LEGEARLY") ? atoi(getenv("PRONTO_BATCH_LEGEARLY")) : -1;
  if (MH == MH_STORE_SC1 && SIX == 0 && plan == 1) { k_step_leg<15, MH, 1, 0><<<nblk(c->B), 128, 0, c->stream>>>(LEG_ARGS); return; }
  if (MH == MH_STORE_SC1 && SIX == 0 && plan == 4) { k_step_leg<15, MH, 4, 12><<<nblk(c->B), 128, 0, c->stream>>>(LEG_ARGS); return; }
  if (MH == MH_STORE_SC1 && SIX == 0 && early == 0) { k_step_leg<15, MH, 0, 0><<<nblk(c->B), 128, 0, c->stream>>>(LEG_ARGS); return; }
#endif
  k_step_leg<15, MH, 0, (SIX == 2 ? PB_LEG_EARLY_SIX2 : 12), SIX><<<nblk(c->B), 128, 0, c->stream>>>(LEG_ARGS);
#else
#ifdef PB_EXPERIMENTS
  static const int plan = getenv("PRONTO_BATCH_LEGPLAN") ? atoi(getenv("PRONTO_BATCH_LEGPLAN")) : -1;
  static const int early = getenv("PRONTO_BATCH_LEGEARLY") ? atoi(getenv("PRONTO_BATCH_LEGEARLY")) : -1;
  if (MH == MH_STORE_SC1 && SIX == 0 && plan == 0) { k_step_quad_leg<MH, 0><<<nblk(c->B), 256, 0, c->stream>>>(LEG_ARGS); return; }
  if (MH == MH_STORE_SC1 && SIX == 0 && plan == 2) { k_step_quad_leg<MH, 2><<<nblk(c->B), 256, 0, c->stream>>>(LEG_ARGS); return; }
  if (MH == MH_STORE_SC1 && SIX == 0 && early == 8) { k_step_quad_leg<MH, 3, 8><<<nblk(c->B), 256, 0, c->stream>>>(LEG_ARGS); return; }
#endif
  k_step_quad_leg<MH, 3, 0, SIX><<<nblk(c->B), 256, 0, c->stream>>>(LEG_ARGS);
#endif
#undef LEG_ARGS
}

template <int SIX>
static void launch_step_leg_mh(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin,
                               const LegStepArgs &la)
{
  switch (c->mem_hint) {
  case MH_STORE_SC1: launch_step_leg<MH_STORE_SC1, SIX>(c, out, imu, q, bc, lin, la); break;
  case MH_STREAM_NT: launch_step_leg<MH_STREAM_NT, SIX>(c, out, imu, q, bc, lin, la); break;
  default: launch_step_leg<MH_DEFAULT, SIX>(c, out, imu, q, bc, lin, la); break;
  }
}

#if PB_LEG_NS == 15
int pbk_step_leg15(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin, const LegStepArgs &la)
#else
int pbk_step_leg21(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin, const LegStepArgs &la)
#endif
{
  switch (la.mp.mode) {
  case 1: launch_step_leg_mh<1>(c, out, imu, q, bc, lin, la); break;
  case 2: launch_step_leg_mh<2>(c, out, imu, q, bc, lin, la); break;
  default: launch_step_leg_mh<0>(c, out, imu, q, bc, lin, la); break;
  }
  return PB_OK;
}
