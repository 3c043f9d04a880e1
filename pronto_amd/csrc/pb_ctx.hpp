// pb_ctx.hpp -- the context behind the C ABI and the launchers shared by the translation units of libpronto_batch.so
// (the kernels are instantiated in several .hip files so that they compile in parallel: pb_step.hip the step kernels,
// pb_update15.hip / pb_update21.hip the generic update kernels, pb_smooth.hip the smoother, pronto_batch.hip the rest).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/pronto_batch.h"
#include "rbis_kernels.hpp"
#include "rbis_legodo.hpp"
#include "rbis_legstep.hpp"
#include "rbis_jointfilt.hpp"

using namespace pb;

#define PB_MAX_FENCES 16

struct pb_ctx {
  int ns = 0, B = 0, dev = 0, nsnap = 0, nc = 0;  // nc: canonical components (Lay<NS>::NC)
  long stride = 0;      // batch rounded up to whole 64-filter tiles
  size_t state_doubles = 0;  // one state array / checkpoint slot: stride * Slots<NS>::NSLOT (tiled layout)
  hipStream_t stream = nullptr, own_stream = nullptr;
  double *st = nullptr;       // the HEAD posterior: st_base, or a checkpoint slot an update wrote its posterior into
  double *st_base = nullptr;  // the context's own state array
  int out_slot = -1;          // pb_set_output_slot: where the next update writes (then that slot is the head)
  double *snaps = nullptr, *d_small = nullptr;
  double *hist = nullptr;  // posterior checkpoint slots (pb_history_reserve)
  int nhist = 0;
  double *legd = nullptr;   // leg odometry state (pb_legodo_init): [NLD][stride] doubles ...
  int64_t *legi = nullptr;  // ... and [NLI][stride] 64-bit integers (rbis_legodo.hpp)
  LegPar leg_par;
  int64_t *leg_ut = nullptr;      // per-filter message times / validity of the NEXT odometry call (pb_legodo_set_message_times),
  uint8_t *leg_valid = nullptr;   // device [B] each; consumed by that call
  bool leg_ut_on = false, leg_valid_on = false;
  const int64_t *leg_ut_ext = nullptr;    // PB_DEVICE times / validity are read in place: the caller's arrays, not copies
  const uint8_t *leg_valid_ext = nullptr;
  LegMeasPar leg_meas;            // pb_legodo_set_measurement_mode: which of LegOdoCommon's measurements the odometry calls write
  LegChain *leg_chain = nullptr;  // forward-kinematics chain table (pb_legodo_set_chain), device copy ...
  LegChain leg_chain_h;           // ... and the host copy (PB_HOST_BROADCAST joint states are reduced to chain angles on the host)
  int leg_chain_rows = 0;         // rows a joint-position block must have (highest row the chain reads + 1)
  int32_t *leg_nc = nullptr;      // controller contact counts [2][B] (pb_legodo_set_control_contacts), used when leg_nc_dev
  int leg_nc_h[2] = { -1, -1 };   // ... or ONE pair for every filter (-1: none received yet)
  bool leg_nc_dev = false;
  // joint-position filters in front of the kinematics (pb_joint_filter_init / pb_joint_filter, rbis_jointfilt.hpp)
  JfPar jf_par;
  bool jf_ready = false, jf_first = true;
  int jf_input = -1;              // -1 no message yet, 0 per-robot blocks (state in HBM), 1 one robot (state on the host)
  int jf_head = 0;                // low-pass window slot of the oldest sample
  double jf_tlast = 0;            // SimpleKalmanFilter::tlast_
  float *jf_ring = nullptr;       // [JF_TAPS][nf][B] floats
  double *jf_kst = nullptr;       // [JF_KSTATE][nf][B]
  std::vector<float> jf_ring_h;   // [JF_TAPS][nf]   (one robot)
  std::vector<double> jf_kst_h;   // [JF_KSTATE][nf]
  double *leg_lo = nullptr;       // measurement block + mask between the two kernels of pb_step_legodo_joints' fallback path
  double *notch = nullptr;  // IMU notch cascade state [36][stride] (pb_imu_notch_init)
  NotchCoef notch_coef;
  bool notch_ready = false;
  // IMU front end per filter (pb_ins_body_block, rbis_frontend.hpp): last body-frame sample [6][stride], previous message time [stride]
  double *ins_last = nullptr;
  int64_t *ins_prev_ut = nullptr;
  const uint8_t *imu_valid_next = nullptr;   // pb_set_imu_valid: one-shot, taken by the next call that takes an IMU step ...
  const uint8_t *imu_valid_cur = nullptr;    // ... and held here for the duration of that call
  double *imu_keep = nullptr;                // [7][stride]: the IMU block that call's step kernel reads instead (pbk_idle_prepare)
  // chunked uploads (pb_upload_async): fences recorded on the main stream, one event for "the uploads issued so far"
  hipEvent_t fence[PB_MAX_FENCES] = {};
  int n_fences = 0;
  hipEvent_t ev_upload = nullptr;
  void *stage = nullptr;
  size_t stage_bytes = 0;
  // PB_HOST inputs: two staging buffers filled on a copy stream, so that the copy of message k+1 overlaps the kernels
  // of message k (with pinned source buffers, pb_host_alloc, the DMA runs at link rate)
  void *in_stage[2] = { nullptr, nullptr };
  size_t in_stage_bytes[2] = { 0, 0 };
  int in_idx = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_consumed[2] = { nullptr, nullptr }, ev_copied = nullptr;
  Consts k{ 9.80665, 1e-6 };
  int64_t utime = 0;
  bool have_state = false;
  bool half15 = false;  // the 15-state fused step with two workgroups per tile (pb_create: batches that leave half the workgroup slots empty; PRONTO_BATCH_HALF=0/1)
  bool coop15 = false;  // PRONTO_BATCH_COOP15=1: run the 15-state step on the two-wave cooperative kernel (A/B switch)
  int mem_hint = 0;     // MH_* cache policy of the step kernels' state round trip (PRONTO_BATCH_MEMHINT=0/1/2 forces it)
  // pb_run_legodo's cache-blocked order (filter range outer, time inner) for states beyond the memory-side cache: filters per block
  // (whole tiles; 0 = the whole batch per launch), and the kernel / cache policy that block size wants
  int run_block = 0, run_block_hint = 0;
  bool run_block_coop15 = false;
  bool quad21 = true;   // PRONTO_BATCH_QUAD21=0: run the 21-state step on the two-wave kernel instead of the four-wave one (A/B)
  bool generic_update = false;  // PRONTO_BATCH_GENERIC_UPDATE=1: every stand-alone update on the run-time-index kernel (A/B, tests)
  bool smooth_attr = false;  // dynamic-LDS limit of the smoother kernels raised on this device
  bool smooth_wide_attr = false;
  int n_cu = 256;            // compute units of the device (grid of the persistent smoother kernel)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  char err[512] = { 0 };
};

inline thread_local char g_create_err[512] = "";

inline int fail(pb_ctx *c, int code, const char *fmt, ...)
{
  char *dst = c ? c->err : g_create_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(c, call)                                                                               \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) return fail((c), PB_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

inline int nblk(int n) { return (n + 63) / 64; }

// Where an update writes its posterior.  Normally in place.  With pb_set_output_slot the posterior goes straight into a
// checkpoint slot (a checkpoint per update without a copy: the step moves the same bytes either way).  If the head IS a
// checkpoint slot and no output slot was named, the update writes back into the context's own array, so a saved
// posterior is never modified.
int detach_head(pb_ctx *c, bool keep_contents);

inline double *update_target(pb_ctx *c)
{
  if (c->out_slot >= 0) return c->hist + (size_t) c->out_slot * c->state_doubles;
  return (c->st != c->st_base) ? c->st_base : c->st;
}
inline void update_done(pb_ctx *c, double *target)
{
  c->st = target;
  c->out_slot = -1;
}


#define LAUNCHCHK(c) HIPCHK((c), hipGetLastError())

// in front of the ONE step launch of a call that was given a mask (pb_set_imu_valid): the IMU block with the samples of the filters
// WITHOUT a message replaced by what reproduces their angular-velocity / acceleration entries (pronto_batch.hip); the block itself
// when there is no mask
const double *pbk_idle_prepare(pb_ctx *c, const double *imu_dev, int *rc_out);

// ---- launchers defined in the other translation units ----
// pb_step.hip: predict (update = false) or predict + leg-odometry update on the kernel pb_create picked
int pbk_step(pb_ctx *c, bool update, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
             const StepBcast *bcast = nullptr);  // bcast: one message for every filter, as kernel arguments
// the fused step on the filters [b0, b0 + nb) only (whole tiles, in place; pb_run_legodo's cache-blocked order)
int pbk_step_range(pb_ctx *c, const double *imu, const double *lo, const uint8_t *mask, const double q[4], long b0, int nb, bool coop15, int mem_hint);
// IMU step + leg odometry (from `lin`) + its lin_rate update in ONE kernel; -1 = this context has no such kernel (run
// pb_legodo_update* ahead of pbk_step instead)
// mp.mode 1 / 2: LegOdoCommon's six-row measurements instead (lo_out [12][B], mask_out [2][B] as pb_legodo_set_measurement_mode)
int pbk_step_leg(pb_ctx *c, const double *imu, const StepBcast *bcast, const double q[4], const LegIn &lin, int64_t utime, const LegMeasPar &mp,
                 double *lo_out, uint8_t *mask_out);
// pb_step_leg.hip, one object per state size: the pair kernels' launchers
int pbk_step_leg15(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin, const LegStepArgs &la);
int pbk_step_leg21(pb_ctx *c, double *out, const double *imu, const double q[4], const StepBcast &bc, const LegIn &lin, const LegStepArgs &la);
// slot0 >= 0: write-through -- the posterior of step t also goes to checkpoint slot slot0 + t (pb_replay_legodo_checkpointed)
int pbk_replay_fused(pb_ctx *c, int T, const double *imu, const double *lo, const uint8_t *mask, const double q[4], int slot0 = -1);
// predict + leg-odometry update + a second (orientation) update in one state round trip; corr_kind = enum pb_corr
int pbk_step_correct(pb_ctx *c, int corr_kind, const double *imu, const double *lo, const uint8_t *mask, const double q[4],
                     const double *z2, const double *r2, const double *rb2, const double *qm2, const uint8_t *mask2,
                     const StepBcast *bcast = nullptr, const double *zb = nullptr, const double *qb = nullptr);
// pb_update15.hip / pb_update21.hip: generic indexed (+ orientation, qm != NULL) update, m = 1..6
int pbk_update15(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb,
                 const double *qm, const uint8_t *mask);
int pbk_update21(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb,
                 const double *qm, const uint8_t *mask);
// (pb_update_rt21.hip is built as three objects, by m)
int pbk_update21_m123(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                      const uint8_t *mask);
int pbk_update21_m45(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                     const uint8_t *mask);
int pbk_update21_m6(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind, const double *rb, const double *qm,
                    const uint8_t *mask);
// pb_update_ct.hip: the same update on the cooperative mapping when idx is one of the handlers' lists and R is diagonal
// (r2 = [m][B] device diagonal or NULL with rb2 = m broadcast values); -1 = no such kernel, use pbk_update15/21
// zb / qb: HOST values of a measurement that is the same for every filter (kernel arguments instead of device blocks)
int pbk_update_ct(pb_ctx *c, int m, const int *idx, const double *z, const double *r2, const double *rb2, const double *qm,
                  const uint8_t *mask, const double *zb = nullptr, const double *qb = nullptr, const double *rfull = nullptr);
// pb_smooth.hip
int pbk_smooth_step(pb_ctx *c, const double *next_pred, const double *next_sm, const double *cur, double *out, double dt);
int pbk_smooth_wide(pb_ctx *c, const double *next_pred, const double *next_sm, const double *cur, double *out, double dt);   // 15 states (pb_smooth_wide.hip)
