// rbis_legstep.hpp -- ONE kernel per IMU + joint-state (or foot-state) message pair: what the reference does per pair of
// messages in LegOdoHandler::processMessage and the two updateFilter calls around it
//   rbis_legodo_update.cpp:206-280   head pose -> leg_estimate::updateOdometry -> LegOdoCommon::createMeasurement (lin_rate)
//   rbis_update_interface.cpp:30-95  RBISIMUProcessStep::updateFilter, RBISIndexedMeasurement::updateFilter
// in one round trip of the filter state.  Round 2 needed two launches for it (k_legodo, then the fused step reading the
// measurement block k_legodo had written).
//
// 15 states, two waves per tile (rbis_coop.hpp).  Role P, the wave that owns the passive panels, has the shorter chain
// before the hand-off barrier and already reads the whole prior state for its process blocks, so IT runs the odometry
// FIRST, before it requests its panel rows (the odometry's registers are dead by then):
//   role P: prior x, quat | leg state | joint / foot inputs -> orientation after the IMU step (ins_update_state on a copy)
//           -> forward kinematics, contact logic, pelvis increment -> leg state stored, (z, R, valid) to LDS -> barrier L
//           -> panel rows requested -> the unchanged passive role (its barrier: the Kalman factors from role C)
//   role C: rows requested -> predict (state, P_cc) -> barrier L -> S = R + P_vv ... the unchanged core role
// The kernel is HBM-bound like the plain step; the odometry adds its state (136 B read + 136 B written per filter) and its
// inputs (a float joint block, or nothing for a broadcast message) and removes the 49-byte measurement block.
#pragma once

#include "rbis_kernels.hpp"
#include "rbis_legodo.hpp"
#include "rbis_quad.hpp"

namespace pb {

#if defined(__HIPCC__)
struct LegStepArgs {
  double *legd;
  int64_t *legi;
  long stride;
  int64_t utime;
  double r2, r2_uncertain;
  double *lo_out;     // [6][B] or NULL: the measurement, kept for a later re-application of this update (history replay)
  uint8_t *mask_out;  // [B] (with lo_out)
  LegMeasPar mp;      // SIX != 0: the six-row modes' variances; lo_out [12][B], mask_out [2][B] as pb_legodo_set_measurement_mode
};

// The odometry wave's measurement for the step roles, shared by the two pair kernels.  SIX == 0: lin_rate.  SIX == 1 / 2:
// LegOdoCommon's lin_rot_rate / pos_and_lin_rate (leg_measurement6) split into the velocity block (zv, rv, valid_v) and the
// other block (z2, r2, on2): mode 2's per-filter fall-back to lin_rate is the velocity block alone.
template <int SIX>
struct LegBlocks {
  double zv[3], rv, z2[3], r2;
  bool valid_v, on2;
};
template <int SIX>
__device__ __forceinline__ void leg_blocks(const Pose &delta, double status, const double (&position)[3], bool position_ok, int64_t ut,
                                           int64_t prev, const LegStepArgs &la, unsigned b, int B, LegBlocks<SIX> &o)
{
  if constexpr (SIX == 0) {
    LegMeas m;
    leg_measurement(delta, status, ut, prev, la.r2, la.r2_uncertain, m);
#pragma unroll
    for (int i = 0; i < 3; i++) { o.zv[i] = m.z[i]; o.z2[i] = 0.0; }
    o.rv = m.r; o.r2 = 1.0;
    o.valid_v = m.valid; o.on2 = false;
    if (la.lo_out != nullptr && b < (unsigned) B) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        la.lo_out[(long) i * B + b] = m.z[i];
        la.lo_out[(long) (3 + i) * B + b] = m.r;
      }
      la.mask_out[b] = m.valid ? 1 : 0;
    }
  } else {
    LegMeasPar mp = la.mp;
    mp.mode = SIX;
    LegMeas6 m;
    leg_measurement6(delta, status, position, position_ok, ut, prev, mp, m);
    constexpr int V = (SIX == 1) ? 0 : 3, O = (SIX == 1) ? 3 : 0;   // rows of the velocity block / of the other block
#pragma unroll
    for (int i = 0; i < 3; i++) { o.zv[i] = m.z[V + i]; o.z2[i] = m.z[O + i]; }
    o.rv = m.r[V]; o.r2 = m.r[O];
    o.valid_v = m.valid6 || m.valid3;
    o.on2 = m.valid6;
    if (la.lo_out != nullptr && b < (unsigned) B) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        la.lo_out[(long) i * B + b] = m.z[i];
        la.lo_out[(long) (6 + i) * B + b] = m.r[i];
      }
      la.mask_out[b] = m.valid6 ? 1 : 0;
      if (SIX == 2) la.mask_out[(long) B + b] = m.valid3 ? 1 : 0;
    }
  }
}

// PLAN (per-filter joint blocks only; who does what in front of barrier L):
//   0  role C: forward kinematics of the left leg | role P: right leg, then (barrier F) contact logic + pelvis integration
//   1  role C: forward kinematics of BOTH legs (it waits for its 29 rows anyway) | role P: contact logic -- Schmitt triggers and the
//      walking-phase classifier read the foot forces only -- in parallel, then (barrier F) the pelvis integration alone
//   4  as 0 with the contact logic in FRONT of barrier F (behind the right leg's kinematics)
// EARLY: panel rows role P requests BEFORE the odometry (the rest behind it): as many as the odometry's registers leave room for.
// Measured at 64k filters with per-filter joint blocks (one box, min of 3 runs, `scripts/leg_ab.sh`): PLAN 0 / EARLY 0 27.3 us,
// PLAN 0 / EARLY 12 26.6 us (foot poses 24.2 -> 23.9), EARLY 20 spills (37 us); PLAN 1 31.2 us -- role C is the wave whose rows
// the memory system is busy with first, a second leg's kinematics in it delays everything behind barrier F.
// SIX: LegOdoCommon's six-row modes in the same kernel (rbis_coop.hpp, coop_role_core): 1 lin_rot_rate, 2 pos_and_lin_rate (with
// leg_estimate's world constraint: the pelvis position is measured, so the odometry wave needs the head POSITION after the IMU
// step too -- the whole state propagate instead of the quaternion's).
// PB_LEG_PRIO (experiment, round 5): the kernel starts with ~3 us of kinematics per tile, and at 64k filters every tile of the GPU starts
// at the same moment (one dispatch round: 4 workgroups per CU) -- the memory sits idle while all of them compute, then all of them load.
// With a raised priority for HALF of the waves that share a SIMD, those finish their kinematics in half the time and start loading
// while the others compute.  1: odd workgroups; 2: bit 8 of the workgroup index; 3: the odd hardware wave slots.
#ifndef PB_LEG_PRIO
#define PB_LEG_PRIO 0
#endif
__device__ __forceinline__ void leg_prio_raise()
{
#if PB_LEG_PRIO == 1
  if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(3);
#elif PB_LEG_PRIO == 2
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(3);
#elif PB_LEG_PRIO == 3
  if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(3);   // HW_REG_HW_ID, WAVE_ID
#endif
}
__device__ __forceinline__ void leg_prio_drop()
{
#if PB_LEG_PRIO != 0
  __builtin_amdgcn_s_setprio(0);
#endif
}

template <int NS, int MH, int PLAN = 0, int EARLY = 12, int SIX = 0>
__global__ __launch_bounds__(128, 2) void k_step_leg(const double *st, double *sto, int B, const double *__restrict__ imu, double qg,
                                                     double qa, double qbg, double qba, Consts k, StepBcast bc, LegPar par, LegIn lin,
                                                     const LegChain *__restrict__ chain, LegStepArgs la)
{
  static_assert(NS == 15, "21 states: k_step_quad_leg");
  using L = Lay<NS>;
  using CORR = typename std::conditional<SIX == 2, CorrPos, NoCorr>::type;
  using CX = CoopX<NS, CORR>;
  __shared__ double xch[CX::NXCH_LEG][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<NS, MemHint<MH>::LA, MemHint<MH>::SA, true> io(st, sto, tile, lane);
  const rsrc_t ri = mkbuf(imu, 7u * B8);
  StepInputs in;
  if (bc.on & 1) {  // one IMU message for every filter: kernel arguments (wave-uniform branch)
#pragma unroll
    for (int i = 0; i < 3; i++) { in.gyro[i] = bc.imu[i]; in.accel[i] = bc.imu[3 + i]; }
    in.dt = bc.imu[6];
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      in.gyro[i] = ldg(ri, i * B8, bo);
      in.accel[i] = ldg(ri, (3 + i) * B8, bo);
    }
    in.dt = ldg(ri, 6u * B8, bo);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) { in.z[i] = 0.0; in.rd[i] = 1.0; }
  in.upd = b < (unsigned) B;
  in.qg = qg; in.qa = qa; in.qbg = qbg; in.qba = qba;
  if (k.qblk != nullptr) {  // per-filter process noise (wave-uniform branch)
    const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
    in.qg = ldg(rq, 0u, bo); in.qa = ldg(rq, B8, bo); in.qbg = ldg(rq, 2u * B8, bo); in.qba = ldg(rq, 3u * B8, bo);
  }
  auto ld = [&io](int comp) { return io.ld(comp); };
  auto stf = [&io](int comp, double v) { io.st(comp, v); };
  auto sync = []() { __syncthreads(); };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  // Per-filter joint blocks: the forward kinematics of the LEFT leg is role C's -- it has nothing to do until its rows arrive --
  // the right leg's is role P's; one more barrier (F) hands the left foot over.  (A broadcast joint state arrives as foot poses.)
  const bool split_fk = lin.kind == 1;  // wave-uniform
  const long bl_ = b < (unsigned) B ? (long) b : (long) B - 1;  // lanes past the batch read the last robot's inputs, store nothing
  leg_prio_raise();
  if (role == 0) {
    io.template need<0, Slots<NS>::ROW_SPLIT>();
    if (split_fk) {
#pragma unroll
      for (int side = 0; side < (PLAN == 1 ? 2 : 1); side++) {
        Pose T;
        leg_fk_side(lin, chain, side, bl_, (long) B, T);
#pragma unroll
        for (int i = 0; i < 3; i++) xch[CX::XCH_FOOT + 7 * side + i][lane] = T.t[i];
#pragma unroll
        for (int i = 0; i < 4; i++) xch[CX::XCH_FOOT + 7 * side + 3 + i][lane] = T.q[i];
      }
      __syncthreads();  // barrier F
    }
    leg_prio_drop();
    coop_role_core<NS, true, CORR, true, true, SIX>(ld, stf, [lane](int s, double v) { xch[s][lane] = v; }, xrd, sync, in, k);
  } else {
    // ---- the odometry, on the prior state this role reads anyway ----
    if constexpr (EARLY > 0) {
      io.template need<Slots<NS>::ROW_SPLIT, Slots<NS>::ROW_SPLIT + EARLY>();
      reload_fence();
    }
    constexpr bool WC = (SIX == 2);
    LegState s;
    leg_load(s, la.legd, la.legi, la.stride, (long) b, WC);     // (the state arrays are padded to whole tiles)
    // what the head orientation after this pair's IMU step depends on: chi and the quaternion -- WC: the whole state vector
    double xs[WC ? NS : 9], wq[4];
#pragma unroll
    for (int i = WC ? 0 : 6; i < (WC ? NS : 9); i++) xs[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
    for (int i = 0; i < 4; i++) wq[i] = io.ld(L::OFF_QUAT + i);
    Pose fl_, fr_, delta;
    float zl, zr;
    int ncl, ncr, cs;
    int64_t prev = 0;
    double classification;
    const int64_t ut = lin.utimes != nullptr ? lin.utimes[bl_] : la.utime;     // independent segments: this filter's own message time
    const bool msg_ok = lin.valid == nullptr || lin.valid[bl_] != 0;           // ... or no message for it at all
    if (split_fk) {
      if (PLAN != 1) leg_fk_side(lin, chain, 1, bl_, (long) B, fr_);
      leg_inputs_rest(lin, bl_, (long) B, zl, zr, ncl, ncr);
      if (PLAN != 0) cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);  // (needs no foot pose)
      __syncthreads();  // barrier F
#pragma unroll
      for (int i = 0; i < 3; i++) fl_.t[i] = xch[CX::XCH_FOOT + i][lane];
#pragma unroll
      for (int i = 0; i < 4; i++) fl_.q[i] = xch[CX::XCH_FOOT + 3 + i][lane];
      if (PLAN == 1) {
#pragma unroll
        for (int i = 0; i < 3; i++) fr_.t[i] = xch[CX::XCH_FOOT + 7 + i][lane];
#pragma unroll
        for (int i = 0; i < 4; i++) fr_.q[i] = xch[CX::XCH_FOOT + 10 + i][lane];
      } else if (PLAN == 0) {
        cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);
      }
    } else {
      leg_inputs(lin, chain, bl_, (long) B, fl_, fr_, zl, zr, ncl, ncr);
      cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);
    }
    // world_to_body_ = the head AFTER this pair's IMU step
    double wpos[3] = { 0.0, 0.0, 0.0 };
    if constexpr (WC) {
      ins_update_state<NS>(xs, wq, in.gyro, in.accel, in.dt, k);
#pragma unroll
      for (int i = 0; i < 3; i++) wpos[i] = xs[9 + i];
    } else {
      const double chi[3] = { xs[6], xs[7], xs[8] }, bg[3] = { 0.0, 0.0, 0.0 };
      ins_update_quat<NS>(chi, bg, wq, in.gyro, in.dt, k);
    }
    LegPar lp = par;
    lp.world_constraint = WC ? 1 : 0;
    double position[3];
    bool position_ok;
    double status = leg_integrate(s, lp, cs, classification, fl_, fr_, wq, delta, wpos, position, position_ok);
    if (leg_zero_velocity(s, status)) {  // odo_delta.setIdentity(); odo_position.setIdentity() (rbis_legodo_update.cpp:266-267)
      pose_identity(delta);
      position[0] = position[1] = position[2] = 0.0;
    }
    if (b < (unsigned) B && msg_ok) leg_store(s, la.legd, la.legi, la.stride, (long) b, WC);
    if (!msg_ok) status = -1.0;
    LegBlocks<SIX> m;
    leg_blocks<SIX>(delta, status, position, position_ok, ut, prev, la, b, B, m);
#pragma unroll
    for (int i = 0; i < 3; i++) xch[CX::XCH_LEG + i][lane] = m.zv[i];
    xch[CX::XCH_LEG + 3][lane] = m.rv;
    xch[CX::XCH_LEG + 4][lane] = m.valid_v ? 1.0 : 0.0;
    if constexpr (SIX == 2) {
#pragma unroll
      for (int i = 0; i < 3; i++) xch[CX::XCH_LEG + 5 + i][lane] = m.z2[i];
      xch[CX::XCH_LEG + 8][lane] = m.r2;
      xch[CX::XCH_LEG + 9][lane] = m.on2 ? 1.0 : 0.0;
    }
    leg_prio_drop();
    if constexpr (SIX != 1) __syncthreads();  // barrier L (SIX == 1: inside the passive role, behind its omega stage)
    else reload_fence();                       // (keeps the panel loads below the odometry, as the barrier does)
    CorrInputs cin2;
    SixIn six;
#pragma unroll
    for (int i = 0; i < 3; i++) six.z[i] = m.z2[i];
    six.r = m.r2;
    six.on = cin2.upd = in.upd && m.on2;
    in.upd = in.upd && m.valid_v;
    io.template need<Slots<NS>::ROW_SPLIT, Slots<NS>::NROW>();
    if constexpr (SIX == 0) coop_role_passive<NS, true, NoCorr, true>(ld, stf, xrd, sync, in, k);
    else coop_role_passive_x<NS, true, CORR, true, SIX>(ld, stf, [lane](int s_, double v) { xch[s_][lane] = v; }, xrd, sync, in, k, cin2, six);
  }
}

// The same for 21 states on the four-wave mapping (rbis_quad.hpp).  Role PW owns the state vector and the quaternion and
// propagates them before barrier A anyway; it runs the odometry first, on the prior state it reads for its process blocks,
// and publishes (z, R, valid) before that barrier -- role CC reads them behind it: no extra barrier.
// PLAN (per-filter joint blocks only):
//   0  forward kinematics: left leg in role CC, right leg in role CB; role PW contact logic + integration behind barrier F
//   2  forward kinematics: left leg in role CB, right leg in role PA (the lightest role; CC, the heaviest, does none); role PW
//      runs the contact logic in parallel and only the pelvis integration behind barrier F
//   3  as 2 with the left leg in role CC instead of CB (CB has the most arithmetic in front of barrier A).  The default since role
//      CC's own propagation is pinned in front of barrier A (pb_pin): 46.6 / 48.1 us against PLAN 2's 48.6 / 49.5 us (joint state /
//      + efforts, 64k filters, one box, min of 2 interleaved runs), PLAN 0 53.1 / 54.4 us.  (The figures below are older: before the pin.)
// Measured (same runs): PLAN 2 51.2-51.4 us in ONE kernel, PLAN 0 54.6 us, round 3's two launches 54.2-55.7 us; panel rows of role PW
// requested ahead of the odometry (EARLY 8 / 16) change nothing (53.1 / 51.8 us).
template <int MH, int PLAN = 3, int EARLY = 0, int SIX = 0>
__global__ __launch_bounds__(256, 2) void k_step_quad_leg(const double *st, double *sto, int B, const double *__restrict__ imu, double qg,
                                                          double qa, double qbg, double qba, Consts k, StepBcast bc, LegPar par, LegIn lin,
                                                          const LegChain *__restrict__ chain, LegStepArgs la)
{
  constexpr int NS = 21;
  using L = Lay<NS>;
  using SL = Slots<21>;
  __shared__ double xch[Quad::nxch_leg(SIX)][64];
  const int role = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = xcd_workgroup(k);
  const unsigned b = tile * 64u + lane;
  const unsigned bo = b * 8u, B8 = (unsigned) B * 8u;
  TileIO<21, MemHint<MH>::LA, MemHint<MH>::SA> io(st, sto, tile, lane);
  // (each role requests its inputs inside its own branch: hoisted above the role switch they cost the four-wave kernel its
  // registers -- 428 bytes of scratch here, as rbis_kernels.hpp k_step_quad found before)
  auto inputs = [&]() {
    const rsrc_t ri = mkbuf(imu, 7u * B8);
    StepInputs in;
    if (bc.on & 1) {
#pragma unroll
      for (int i = 0; i < 3; i++) { in.gyro[i] = bc.imu[i]; in.accel[i] = bc.imu[3 + i]; }
      in.dt = bc.imu[6];
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        in.gyro[i] = ldg(ri, i * B8, bo);
        in.accel[i] = ldg(ri, (3 + i) * B8, bo);
      }
      in.dt = ldg(ri, 6u * B8, bo);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { in.z[i] = 0.0; in.rd[i] = 1.0; }
    in.upd = b < (unsigned) B;
    in.qg = qg; in.qa = qa; in.qbg = qbg; in.qba = qba;
    if (k.qblk != nullptr) {
      const rsrc_t rq = mkbuf(k.qblk, 4u * B8);
      in.qg = ldg(rq, 0u, bo); in.qa = ldg(rq, B8, bo); in.qbg = ldg(rq, 2u * B8, bo); in.qba = ldg(rq, 3u * B8, bo);
    }
    return in;
  };
  auto ld = [&io](int comp) { return io.ld(comp); };
  auto stf = [&io](int comp, double v) { io.st(comp, v); };
  auto sync = []() { __syncthreads(); };
  auto xrd = [lane](int s) { return xch[s][lane]; };
  auto xwr = [lane](int s, double v) { xch[s][lane] = v; };
  // per-filter joint blocks: role CC runs the left leg's forward kinematics and role CB the right leg's while their rows are
  // on the way; barrier F hands both feet to role PW (all four waves take part in it)
  const bool split_fk = lin.kind == 1;  // wave-uniform
  const long bl_ = b < (unsigned) B ? (long) b : (long) B - 1;
  auto fk_to_lds = [&](int side) {
    Pose T;
    leg_fk_side(lin, chain, side, bl_, (long) B, T);
#pragma unroll
    for (int i = 0; i < 3; i++) xch[Quad::X_FOOT + 7 * side + i][lane] = T.t[i];
#pragma unroll
    for (int i = 0; i < 4; i++) xch[Quad::X_FOOT + 7 * side + 3 + i][lane] = T.q[i];
  };
  if (role == 0) {
    const StepInputs in = inputs();
    io.template need<SL::QROW[0], SL::QROW[1]>();
    if (split_fk) {
      if (PLAN == 0 || PLAN == 3) fk_to_lds(0);
      __syncthreads();  // barrier F
    }
    quad_role_cc<true, true, SIX>(ld, stf, xwr, xrd, sync, in, k);
  } else if (role == 1) {
    const StepInputs in = inputs();
    io.template need<SL::QROW[1], SL::QROW[2]>();
    if (split_fk) {
      if (PLAN != 3) fk_to_lds(PLAN == 0 ? 1 : 0);
      __syncthreads();
    }
    quad_role_cb<true, SIX>(ld, stf, xwr, xrd, sync, in, k);
  } else if (role == 2) {
    StepInputs in = inputs();
    if constexpr (EARLY > 0) {
      io.template need<SL::QROW[2], SL::QROW[2] + EARLY>();
      reload_fence();
    }
    constexpr bool WC = (SIX == 2);
    LegState s;
    leg_load(s, la.legd, la.legi, la.stride, (long) b, WC);
    double xs[NS], wq[4];   // (WC: the whole state vector, for the head position after the IMU step; else chi and the gyro bias)
#pragma unroll
    for (int i = 0; i < NS; i++)
      if (WC || (i >= 6 && i < 9) || (i >= 15 && i < 18)) xs[i] = io.ld(L::OFF_VEC + i);
#pragma unroll
    for (int i = 0; i < 4; i++) wq[i] = io.ld(L::OFF_QUAT + i);
    Pose fl_, fr_, delta;
    float zl, zr;
    int ncl, ncr, cs;
    int64_t prev = 0;
    double classification;
    const int64_t ut = lin.utimes != nullptr ? lin.utimes[bl_] : la.utime;     // independent segments: this filter's own message time
    const bool msg_ok = lin.valid == nullptr || lin.valid[bl_] != 0;           // ... or no message for it at all
    if (split_fk) {
      leg_inputs_rest(lin, bl_, (long) B, zl, zr, ncl, ncr);
      if (PLAN != 0) cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);  // (needs no foot pose)
      __syncthreads();  // barrier F
#pragma unroll
      for (int i = 0; i < 3; i++) { fl_.t[i] = xch[Quad::X_FOOT + i][lane]; fr_.t[i] = xch[Quad::X_FOOT + 7 + i][lane]; }
#pragma unroll
      for (int i = 0; i < 4; i++) { fl_.q[i] = xch[Quad::X_FOOT + 3 + i][lane]; fr_.q[i] = xch[Quad::X_FOOT + 10 + i][lane]; }
      if (PLAN == 0) cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);
    } else {
      leg_inputs(lin, chain, bl_, (long) B, fl_, fr_, zl, zr, ncl, ncr);
      cs = leg_contacts(s, par, ut, zl, zr, ncl, ncr, classification, prev);
    }
    double wpos[3] = { 0.0, 0.0, 0.0 };
    if constexpr (WC) {
      ins_update_state<NS>(xs, wq, in.gyro, in.accel, in.dt, k);
#pragma unroll
      for (int i = 0; i < 3; i++) wpos[i] = xs[9 + i];
    } else {
      const double chi[3] = { xs[6], xs[7], xs[8] }, bg[3] = { xs[15], xs[16], xs[17] };
      ins_update_quat<NS>(chi, bg, wq, in.gyro, in.dt, k);
    }
    LegPar lp = par;
    lp.world_constraint = WC ? 1 : 0;
    double position[3];
    bool position_ok;
    double status = leg_integrate(s, lp, cs, classification, fl_, fr_, wq, delta, wpos, position, position_ok);
    if (leg_zero_velocity(s, status)) {
      pose_identity(delta);
      position[0] = position[1] = position[2] = 0.0;
    }
    if (b < (unsigned) B && msg_ok) leg_store(s, la.legd, la.legi, la.stride, (long) b, WC);
    if (!msg_ok) status = -1.0;
    LegBlocks<SIX> m;
    leg_blocks<SIX>(delta, status, position, position_ok, ut, prev, la, b, B, m);
#pragma unroll
    for (int i = 0; i < 3; i++) xch[Quad::X_LEG + i][lane] = m.zv[i];
    xch[Quad::X_LEG + 3][lane] = m.rv;
    xch[Quad::X_LEG + 4][lane] = m.valid_v ? 1.0 : 0.0;
    if constexpr (SIX == 2) {
#pragma unroll
      for (int i = 0; i < 3; i++) xch[Quad::X_LEG + 5 + i][lane] = m.z2[i];
      xch[Quad::X_LEG + 8][lane] = m.r2;
      xch[Quad::X_LEG + 9][lane] = m.on2 ? 1.0 : 0.0;
    }
    SixIn six;
#pragma unroll
    for (int i = 0; i < 3; i++) six.z[i] = m.z2[i];
    six.r = m.r2;
    six.on = in.upd && m.on2;
    in.upd = in.upd && m.valid_v;
    reload_fence();  // the panel rows are requested HERE, not above the odometry (the two do not fit the registers together)
    io.template need<SL::QROW[2], SL::QROW[3]>();
    quad_role_passive<true, 0, SIX>(ld, stf, xwr, xrd, sync, in, k, six);
  } else {
    const StepInputs in = inputs();
    io.template need<SL::QROW[3], SL::QROW[4]>();
    if (split_fk) {
      if (PLAN != 0) fk_to_lds(1);
      __syncthreads();  // barrier F
    }
    quad_role_passive<true, 1, SIX>(ld, stf, xwr, xrd, sync, in, k);
  }
}
#endif

}  // namespace pb
