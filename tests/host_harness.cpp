// host_harness.cpp -- TEST-ONLY: compiles pronto_amd/csrc/rbis_device.hpp (the per-lane arithmetic the HIP kernels
// run) with g++ so that `-m "not gpu"` tests can check that arithmetic against the oracle in a container without
// a GPU.  It is NOT part of the product: libpronto_batch.so contains no host path and pronto_amd/ never loads this.
// The glue below mirrors k_step / k_update in rbis_kernels.hpp.
#include <cstdint>
#include <cstring>

#include "../pronto_amd/csrc/rbis_device.hpp"

using namespace pb;

struct IdxVel { static constexpr Idx<3> value = { { 3, 4, 5 } }; };
struct IdxPosChi { static constexpr Idx<6> value = { { 9, 10, 11, 6, 7, 8 } }; };
struct IdxPosYaw { static constexpr Idx<4> value = { { 9, 10, 11, 8 } }; };

template <int NS>
static void load(const double *st, long stride, int b, double (&x)[NS], double (&q)[4], double &ll,
                 double (&P)[Lay<NS>::NP])
{
  using L = Lay<NS>;
  for (int i = 0; i < NS; i++) x[i] = st[(L::OFF_VEC + i) * stride + b];
  for (int i = 0; i < 4; i++) q[i] = st[(L::OFF_QUAT + i) * stride + b];
  ll = st[L::OFF_LL * stride + b];
  for (int i = 0; i < L::NP; i++) P[i] = st[(L::OFF_P + i) * stride + b];
}
template <int NS>
static void store(double *st, long stride, int b, const double (&x)[NS], const double (&q)[4], double ll,
                  const double (&P)[Lay<NS>::NP])
{
  using L = Lay<NS>;
  for (int i = 0; i < NS; i++) st[(L::OFF_VEC + i) * stride + b] = x[i];
  for (int i = 0; i < 4; i++) st[(L::OFF_QUAT + i) * stride + b] = q[i];
  st[L::OFF_LL * stride + b] = ll;
  for (int i = 0; i < L::NP; i++) st[(L::OFF_P + i) * stride + b] = P[i];
}

template <int NS>
static void step(double *st, long stride, int B, const double *imu, const double *lo, const uint8_t *mask,
                 const double *q4, double g, double tol, int do_update)
{
  Consts k{ g, tol };
  for (int b = 0; b < B; b++) {
    double x[NS], q[4], ll, P[Lay<NS>::NP];
    load<NS>(st, stride, b, x, q, ll, P);
    const double gyro[3] = { imu[b], imu[B + b], imu[2 * B + b] };
    const double accel[3] = { imu[3 * B + b], imu[4 * B + b], imu[5 * B + b] };
    imu_process_step<NS>(x, q, P, gyro, accel, imu[6 * B + b], q4[0], q4[1], q4[2], q4[3], k);
    if (do_update && (!mask || mask[b])) {
      double resid[3], S[6];
      for (int i = 0; i < 3; i++) resid[i] = lo[i * B + b] - x[3 + i];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j <= i; j++) S[pk(i, j)] = P[pk(3 + i, 3 + j)] + (i == j ? lo[(3 + i) * B + b] : 0.0);
      measurement_update<NS, 3>(x, q, P, ll, resid, S, IdxVel{}, k);
    }
    store<NS>(st, stride, b, x, q, ll, P);
  }
}

// compile-time-index orientation update (VO position_orient / scan-match position_yaw)
template <int NS, int M, typename IDXT>
static void update_orient(double *st, long stride, int B, const double *z, const double *rdiag, const double *qm,
                          double g, double tol)
{
  Consts k{ g, tol };
  constexpr Idx<M> idx = IDXT::value;
  for (int b = 0; b < B; b++) {
    double x[NS], q[4], ll, P[Lay<NS>::NP];
    load<NS>(st, stride, b, x, q, ll, P);
    const double qmeas[4] = { qm[b], qm[B + b], qm[2 * B + b], qm[3 * B + b] };
    double dq[3];
    subtract_quats(qmeas, q, dq);
    double resid[M], S[M * (M + 1) / 2];
    for (int i = 0; i < M; i++) {
      const int ii = idx.v[i];
      resid[i] = (ii >= 6 && ii <= 8) ? dq[ii - 6] : z[i * B + b] - x[ii];
    }
    for (int i = 0; i < M; i++)
      for (int j = 0; j <= i; j++) S[pk(i, j)] = P[pk(idx.v[i], idx.v[j])] + (i == j ? rdiag[i * B + b] : 0.0);
    measurement_update<NS, M>(x, q, P, ll, resid, S, IDXT{}, k);
    store<NS>(st, stride, b, x, q, ll, P);
  }
}

extern "C" {
int hh_nc(int ns) { return ns == 15 ? Lay<15>::NC : Lay<21>::NC; }
int hh_pk(int i, int j) { return pk(i, j); }
void hh_step(int ns, double *st, long stride, int B, const double *imu, const double *lo, const uint8_t *mask,
             const double *q4, double g, double tol, int do_update)
{
  if (ns == 15) step<15>(st, stride, B, imu, lo, mask, q4, g, tol, do_update);
  else step<21>(st, stride, B, imu, lo, mask, q4, g, tol, do_update);
}
void hh_update_vo(int ns, double *st, long stride, int B, const double *z, const double *rdiag, const double *qm,
                  double g, double tol)
{
  if (ns == 15) update_orient<15, 6, IdxPosChi>(st, stride, B, z, rdiag, qm, g, tol);
  else update_orient<21, 6, IdxPosChi>(st, stride, B, z, rdiag, qm, g, tol);
}
void hh_update_posyaw(int ns, double *st, long stride, int B, const double *z, const double *rdiag, const double *qm,
                      double g, double tol)
{
  if (ns == 15) update_orient<15, 4, IdxPosYaw>(st, stride, B, z, rdiag, qm, g, tol);
  else update_orient<21, 4, IdxPosYaw>(st, stride, B, z, rdiag, qm, g, tol);
}
}

// ---- two-wave cooperative step (rbis_coop.hpp): the two roles run back to back on a snapshot of the filter's
// column, with a plain array standing in for the LDS hand-off; stores are applied after both roles have read. ----
#include "../pronto_amd/csrc/rbis_coop.hpp"

template <int NS, class CORR = NoCorr>
static void step_coop(double *st, long stride, int B, const double *imu, const double *lo, const uint8_t *mask,
                      const double *q4, double g, double tol, int do_update, const double *z2 = nullptr,
                      const double *rd2 = nullptr, const double *qm2 = nullptr, const uint8_t *mask2 = nullptr)
{
  Consts k{ g, tol };
  constexpr int NC = Lay<NS>::NC;
  for (int b = 0; b < B; b++) {
    double in_col[NC], out_col[NC], xch[CoopX<NS, CORR>::NXCH];
    CorrInputs cin;
    for (int i = 0; i < CORR::M; i++) { cin.z[i] = z2[i * B + b]; cin.rd[i] = rd2[i * B + b]; }
    for (int i = 0; i < 4; i++) cin.qm[i] = (CORR::M > 0) ? qm2[i * B + b] : 0.0;
    cin.upd = CORR::M > 0 && (!mask2 || mask2[b]);
    for (int c = 0; c < NC; c++) in_col[c] = out_col[c] = st[c * stride + b];
    StepInputs in;
    for (int i = 0; i < 3; i++) {
      in.gyro[i] = imu[i * B + b];
      in.accel[i] = imu[(3 + i) * B + b];
      in.z[i] = do_update ? lo[i * B + b] : 0.0;
      in.rd[i] = do_update ? lo[(3 + i) * B + b] : 1.0;
    }
    in.dt = imu[6 * B + b];
    in.upd = do_update && (!mask || mask[b]);
    in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
    auto ld = [&](int c) { return in_col[c]; };
    auto stf = [&](int c, double v) { out_col[c] = v; };
    auto sync = []() {};
    auto xw = [&](int s, double v) { xch[s] = v; };
    auto xr = [&](int s) { return xch[s]; };
    if constexpr (CORR::M > 0) {
      if (do_update == 2) {  // the correction alone: no predict, no leg-odometry update
        coop_role_core<NS, false, CORR, false>(ld, stf, xw, xr, sync, in, k, cin);
        coop_role_passive<NS, false, CORR, false>(ld, stf, xr, sync, in, k, cin);
      } else {
        coop_role_core<NS, true, CORR>(ld, stf, xw, xr, sync, in, k, cin);
        coop_role_passive<NS, true, CORR>(ld, stf, xr, sync, in, k, cin);
      }
    } else if (do_update) {
      coop_role_core<NS, true>(ld, stf, xw, xr, sync, in, k);
      coop_role_passive<NS, true>(ld, stf, xr, sync, in, k);
    } else {
      coop_role_core<NS, false>(ld, stf, xw, xr, sync, in, k);
      coop_role_passive<NS, false>(ld, stf, xr, sync, in, k);
    }
    for (int c = 0; c < NC; c++) st[c * stride + b] = out_col[c];
  }
}

extern "C" void hh_step_coop(int ns, double *st, long stride, int B, const double *imu, const double *lo,
                             const uint8_t *mask, const double *q4, double g, double tol, int do_update)
{
  if (ns == 15) step_coop<15>(st, stride, B, imu, lo, mask, q4, g, tol, do_update);
  else step_coop<21>(st, stride, B, imu, lo, mask, q4, g, tol, do_update);
}

// predict + leg-odometry update + a fused second (orientation) update: kind 0 = position_orient (m = 6), 1 = position_yaw
// (m = 4); mode 1 = all three, mode 2 = the correction alone (stand-alone update on the two-role mapping)
extern "C" void hh_step_coop_correct(int ns, int kind, int mode, double *st, long stride, int B, const double *imu, const double *lo,
                                     const uint8_t *mask, const double *q4, double g, double tol, const double *z2,
                                     const double *rd2, const double *qm2, const uint8_t *mask2)
{
  if (ns == 15 && kind == 0) step_coop<15, CorrPosOrient>(st, stride, B, imu, lo, mask, q4, g, tol, mode, z2, rd2, qm2, mask2);
  else if (ns == 15) step_coop<15, CorrPosYaw>(st, stride, B, imu, lo, mask, q4, g, tol, mode, z2, rd2, qm2, mask2);
  else if (kind == 0) step_coop<21, CorrPosOrient>(st, stride, B, imu, lo, mask, q4, g, tol, mode, z2, rd2, qm2, mask2);
  else step_coop<21, CorrPosYaw>(st, stride, B, imu, lo, mask, q4, g, tol, mode, z2, rd2, qm2, mask2);
}

// ---- LegOdoCommon's six-row measurements inside the step (SIX, rbis_coop.hpp): the two roles as two threads with a real
// barrier (SIX == 1: role C needs role P's omega stage, role P needs role C's factors).  lo12 = z [6][B] | R diagonal [6][B] in
// the order of pb_legodo_set_measurement_mode; masks [2][B]: the six-row update | the three-row (velocity) fall-back of mode 2. ----
#include <pthread.h>

#include <thread>

template <int NS, int SIX>
static void step_coop_six(double *st, long stride, int B, const double *imu, const double *lo12, const uint8_t *masks, const double *q4,
                          double g, double tol)
{
  using CORR = typename std::conditional<SIX == 2, CorrPos, NoCorr>::type;
  Consts k{ g, tol };
  constexpr int NC = Lay<NS>::NC;
  static double in_col[NC], out_col[NC], xch[CoopX<NS, CORR>::NXCH_LEG];
  static StepInputs in;
  static CorrInputs cin;
  static SixIn six;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, 2);
  auto body = [&](int role) {
    for (int b = 0; b < B; b++) {
      if (role == 0) {
        for (int c = 0; c < NC; c++) in_col[c] = out_col[c] = st[c * stride + b];
        const int vrow = SIX == 1 ? 0 : 3;   // rows of the velocity block
        for (int i = 0; i < 3; i++) {
          in.gyro[i] = imu[i * B + b];
          in.accel[i] = imu[(3 + i) * B + b];
          in.z[i] = lo12[(vrow + i) * B + b];
          in.rd[i] = lo12[(6 + vrow + i) * B + b];
        }
        in.dt = imu[6 * B + b];
        in.upd = masks[b] != 0 || (SIX == 2 && masks[B + b] != 0);
        in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
        const int orow = SIX == 1 ? 3 : 0;   // rows of the other block
        for (int i = 0; i < 3; i++) {
          cin.z[i] = six.z[i] = lo12[(orow + i) * B + b];
          cin.rd[i] = lo12[(6 + orow + i) * B + b];
        }
        six.r = lo12[(6 + orow) * B + b];
        cin.upd = six.on = masks[b] != 0;
      }
      pthread_barrier_wait(&bar);
      auto ld = [&](int c) { return in_col[c]; };
      auto stf = [&](int c, double v) { out_col[c] = v; };
      auto sync = [&]() { pthread_barrier_wait(&bar); };
      auto xw = [&](int s, double v) { xch[s] = v; };
      auto xr = [&](int s) { return xch[s]; };
      if (role == 0) coop_role_core<NS, true, CORR, true, false, SIX>(ld, stf, xw, xr, sync, in, k, cin);
      else coop_role_passive_x<NS, true, CORR, true, SIX>(ld, stf, xw, xr, sync, in, k, cin, six);
      pthread_barrier_wait(&bar);
      if (role == 0)
        for (int c = 0; c < NC; c++) st[c * stride + b] = out_col[c];
    }
  };
  std::thread t1(body, 1);
  body(0);
  t1.join();
  pthread_barrier_destroy(&bar);
}

extern "C" void hh_step_coop_six(int ns, int six, double *st, long stride, int B, const double *imu, const double *lo12,
                                 const uint8_t *masks, const double *q4, double g, double tol)
{
  if (ns == 15 && six == 1) step_coop_six<15, 1>(st, stride, B, imu, lo12, masks, q4, g, tol);
  else if (ns == 15) step_coop_six<15, 2>(st, stride, B, imu, lo12, masks, q4, g, tol);
  else if (six == 1) step_coop_six<21, 1>(st, stride, B, imu, lo12, masks, q4, g, tol);
  else step_coop_six<21, 2>(st, stride, B, imu, lo12, masks, q4, g, tol);
}

// ---- four-wave 21-state step (rbis_quad.hpp): the four roles run as four threads with a real barrier (role CB needs
// role CC's factors and role CC needs role CB's H, so no back-to-back order works); a plain array stands in for LDS. ----
#include "../pronto_amd/csrc/rbis_quad.hpp"

template <bool UPDATE>
static void step_quad(double *st, long stride, int B, const double *imu, const double *lo, const uint8_t *mask, const double *q4,
                      double g, double tol)
{
  Consts k{ g, tol };
  constexpr int NC = Lay<21>::NC;
  static double in_col[NC], out_col[NC], xch[Quad::NXCH];
  static StepInputs in;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, 4);
  auto body = [&](int role) {
    for (int b = 0; b < B; b++) {
      if (role == 0) {
        for (int c = 0; c < NC; c++) in_col[c] = out_col[c] = st[c * stride + b];
        for (int i = 0; i < 3; i++) {
          in.gyro[i] = imu[i * B + b];
          in.accel[i] = imu[(3 + i) * B + b];
          in.z[i] = UPDATE ? lo[i * B + b] : 0.0;
          in.rd[i] = UPDATE ? lo[(3 + i) * B + b] : 1.0;
        }
        in.dt = imu[6 * B + b];
        in.upd = UPDATE && (!mask || mask[b]);
        in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
      }
      pthread_barrier_wait(&bar);
      auto ld = [&](int c) { return in_col[c]; };
      auto stf = [&](int c, double v) { out_col[c] = v; };
      auto sync = [&]() { pthread_barrier_wait(&bar); };
      auto xw = [&](int s, double v) { xch[s] = v; };
      auto xr = [&](int s) { return xch[s]; };
      if (role == 0) quad_role_cc<UPDATE>(ld, stf, xw, xr, sync, in, k);
      else if (role == 1) quad_role_cb<UPDATE>(ld, stf, xw, xr, sync, in, k);
      else if (role == 2) quad_role_passive<UPDATE, 0>(ld, stf, xw, xr, sync, in, k);
      else quad_role_passive<UPDATE, 1>(ld, stf, xw, xr, sync, in, k);
      pthread_barrier_wait(&bar);
      if (role == 0)
        for (int c = 0; c < NC; c++) st[c * stride + b] = out_col[c];
    }
  };
  std::thread t1(body, 1), t2(body, 2), t3(body, 3);
  body(0);
  t1.join(); t2.join(); t3.join();
  pthread_barrier_destroy(&bar);
}

extern "C" void hh_step_quad(double *st, long stride, int B, const double *imu, const double *lo, const uint8_t *mask,
                             const double *q4, double g, double tol, int do_update)
{
  if (do_update) step_quad<true>(st, stride, B, imu, lo, mask, q4, g, tol);
  else step_quad<false>(st, stride, B, imu, lo, mask, q4, g, tol);
}

// the six-row leg-odometry modes on the four-wave mapping (SIX, rbis_quad.hpp); arguments as hh_step_coop_six
template <int SIX>
static void step_quad_six(double *st, long stride, int B, const double *imu, const double *lo12, const uint8_t *masks, const double *q4,
                          double g, double tol)
{
  Consts k{ g, tol };
  constexpr int NC = Lay<21>::NC;
  static double in_col[NC], out_col[NC], xch[Quad::NXCH_SIX];
  static StepInputs in;
  static SixIn six;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, 4);
  auto body = [&](int role) {
    for (int b = 0; b < B; b++) {
      if (role == 0) {
        for (int c = 0; c < NC; c++) in_col[c] = out_col[c] = st[c * stride + b];
        const int vrow = SIX == 1 ? 0 : 3, orow = SIX == 1 ? 3 : 0;
        for (int i = 0; i < 3; i++) {
          in.gyro[i] = imu[i * B + b];
          in.accel[i] = imu[(3 + i) * B + b];
          in.z[i] = lo12[(vrow + i) * B + b];
          in.rd[i] = lo12[(6 + vrow + i) * B + b];
          six.z[i] = lo12[(orow + i) * B + b];
        }
        six.r = lo12[(6 + orow) * B + b];
        six.on = masks[b] != 0;
        in.dt = imu[6 * B + b];
        in.upd = masks[b] != 0 || (SIX == 2 && masks[B + b] != 0);
        in.qg = q4[0]; in.qa = q4[1]; in.qbg = q4[2]; in.qba = q4[3];
      }
      pthread_barrier_wait(&bar);
      auto ld = [&](int c) { return in_col[c]; };
      auto stf = [&](int c, double v) { out_col[c] = v; };
      auto sync = [&]() { pthread_barrier_wait(&bar); };
      auto xw = [&](int s, double v) { xch[s] = v; };
      auto xr = [&](int s) { return xch[s]; };
      if (role == 0) quad_role_cc<true, false, SIX>(ld, stf, xw, xr, sync, in, k, six);
      else if (role == 1) quad_role_cb<true, SIX>(ld, stf, xw, xr, sync, in, k);
      else if (role == 2) quad_role_passive<true, 0, SIX>(ld, stf, xw, xr, sync, in, k, six);
      else quad_role_passive<true, 1, SIX>(ld, stf, xw, xr, sync, in, k, six);
      pthread_barrier_wait(&bar);
      if (role == 0)
        for (int c = 0; c < NC; c++) st[c * stride + b] = out_col[c];
    }
  };
  std::thread t1(body, 1), t2(body, 2), t3(body, 3);
  body(0);
  t1.join(); t2.join(); t3.join();
  pthread_barrier_destroy(&bar);
}

extern "C" void hh_step_quad_six(int six, double *st, long stride, int B, const double *imu, const double *lo12, const uint8_t *masks,
                                 const double *q4, double g, double tol)
{
  if (six == 1) step_quad_six<1>(st, stride, B, imu, lo12, masks, q4, g, tol);
  else step_quad_six<2>(st, stride, B, imu, lo12, masks, q4, g, tol);
}

// ---- stand-alone update on the four-wave mapping (rbis_quad.hpp, quad_upd_*): four threads, one barrier ----
template <class CORR>
static void update_quad(double *st, long stride, int B, const double *z2, const double *rd2, const double *qm2, const uint8_t *mask2,
                        double g, double tol)
{
  Consts k{ g, tol };
  constexpr int NC = Lay<21>::NC;
  static double in_col[NC], out_col[NC], xch[QuadU<CORR>::NXCH];
  static CorrInputs cin;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, 4);
  auto body = [&](int role) {
    for (int b = 0; b < B; b++) {
      if (role == 0) {
        for (int c = 0; c < NC; c++) in_col[c] = out_col[c] = st[c * stride + b];
        for (int i = 0; i < CORR::M; i++) { cin.z[i] = z2[i * B + b]; cin.rd[i] = rd2[i * B + b]; }
        for (int i = 0; i < 4; i++) cin.qm[i] = CORR::ORIENT ? qm2[i * B + b] : 0.0;
        cin.upd = (!mask2 || mask2[b]);
      }
      pthread_barrier_wait(&bar);
      auto ld = [&](int c) { return in_col[c]; };
      auto stf = [&](int c, double v) { out_col[c] = v; };
      auto sync = [&]() { pthread_barrier_wait(&bar); };
      auto xw = [&](int s, double v) { xch[s] = v; };
      auto xr = [&](int s) { return xch[s]; };
      if (role == 0) quad_upd_cc<CORR>(ld, stf, xw, xr, sync, cin, k);
      else if (role == 1) quad_upd_cb<CORR>(ld, stf, xw, xr, sync, cin, k);
      else if (role == 2) quad_upd_passive<CORR, 0>(ld, stf, xw, xr, sync, cin, k);
      else quad_upd_passive<CORR, 1>(ld, stf, xw, xr, sync, cin, k);
      pthread_barrier_wait(&bar);
      if (role == 0)
        for (int c = 0; c < NC; c++) st[c * stride + b] = out_col[c];
    }
  };
  std::thread t1(body, 1), t2(body, 2), t3(body, 3);
  body(0);
  t1.join(); t2.join(); t3.join();
  pthread_barrier_destroy(&bar);
}

// kind: 0 vel, 1 pos, 2 pos+vel, 3 pos+orient, 4 pos+yaw, 5 vel+yaw, 6 yaw, 7-9 the GPF's yaw+pos, chi+pos, z (pb_update_ct.hip)
extern "C" void hh_update_quad(int kind, double *st, long stride, int B, const double *z2, const double *rd2, const double *qm2,
                               const uint8_t *mask2, double g, double tol)
{
  switch (kind) {
  case 0: update_quad<CorrVel>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 1: update_quad<CorrPos>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 2: update_quad<CorrPosVel>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 3: update_quad<CorrPosOrient>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 4: update_quad<CorrPosYaw>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 5: update_quad<CorrVelYaw>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 6: update_quad<CorrYaw>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 7: update_quad<CorrGpfYawPos>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  case 8: update_quad<CorrGpfChiPos>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  default: update_quad<CorrGpfZ>(st, stride, B, z2, rd2, qm2, mask2, g, tol); break;
  }
}

// ---- leg kinematic odometry (rbis_legodo.hpp): one call = one joint-state message for B robots ----
#include "../pronto_amd/csrc/rbis_legodo.hpp"
extern "C" {
int hh_leg_nld() { return NLD + NLD_WC; }
int hh_leg_nli() { return NLI; }
// info = primary_foot, leg_odo_init, walking-phase mode, unknown transitions of robot b (the integers are stored packed)
void hh_leg_info(const double *legd, const int64_t *legi, long stride, long b, int64_t *info)
{
  LegState s;
  leg_load(s, legd, legi, stride, b);
  info[0] = s.primary_foot; info[1] = s.leg_odo_init; info[2] = s.mode; info[3] = s.unknown_transitions;
}
void hh_leg_reset(double *legd, int64_t *legi, long stride, int B)
{
  for (int b = 0; b < B; b++) {
    LegState s;
    leg_reset(s);
    leg_store(s, legd, legi, stride, b, true);
  }
}
void hh_leg_set_zero_ticks(double *legd, int64_t *legi, long stride, int B, int ticks)
{
  for (int b = 0; b < B; b++) {
    LegState s;
    leg_load(s, legd, legi, stride, b, true);
    s.zero_ticks = ticks;
    leg_store(s, legd, legi, stride, b, true);
  }
}
// feet [14][B], forces [2][B], wq [4][B]; delta [7][B], status [B], prev [B]
// par9 = schmitt low, high, low delay, high delay, filter_contact_events, standing, total_force, standing_schmitt_level,
// use_controller_input (thresholds rounded to float like pb_legodo_init / pb_legodo_set_contact_mode do); nc = controller
// contact counts {left, right} for every robot
void hh_leg_update(double *legd, int64_t *legi, long stride, int B, int64_t utime, const double *par9, const int *nc,
                   const double *feet, const double *forces, const double *wq, double *delta, double *status, int64_t *prev)
{
  LegPar par;
  par.alt = SchmittPar{ (double) (float) par9[0], (double) (float) par9[1], (int64_t) par9[2], (int64_t) par9[3] };
  par.filter_contact_events = (int) par9[4];
  par.standing = (int) par9[5];
  par.total_force = (float) par9[6];
  par.standing_schmitt_level = (float) par9[7];
  par.use_controller_input = (int) par9[8];
  for (int b = 0; b < B; b++) {
    LegState s;
    leg_load(s, legd, legi, stride, b);
    Pose bl, br, d;
    for (int i = 0; i < 3; i++) { bl.t[i] = feet[i * B + b]; br.t[i] = feet[(7 + i) * B + b]; }
    for (int i = 0; i < 4; i++) { bl.q[i] = feet[(3 + i) * B + b]; br.q[i] = feet[(10 + i) * B + b]; }
    const double w[4] = { wq[b], wq[B + b], wq[2 * B + b], wq[3 * B + b] };
    status[b] = leg_update(s, par, utime, bl, br, (float) forces[b], (float) forces[B + b], nc[0], nc[1], w, d, prev[b]);
    leg_store(s, legd, legi, stride, b);
    for (int i = 0; i < 3; i++) delta[i * B + b] = d.t[i];
    for (int i = 0; i < 4; i++) delta[(3 + i) * B + b] = d.q[i];
  }
}
// the same with the world constraint (LegPar::world_constraint) and the per-robot zero_initial_velocity counter, as k_legodo
// runs them: wpos [3][B] head position; pos [3][B], pos_ok [B] out
void hh_leg_update_wc(double *legd, int64_t *legi, long stride, int B, int64_t utime, const double *par9, const int *nc,
                      const double *feet, const double *forces, const double *wpos, const double *wq, double *delta, double *status,
                      int64_t *prev, double *pos, int *pos_ok)
{
  LegPar par;
  par.alt = SchmittPar{ (double) (float) par9[0], (double) (float) par9[1], (int64_t) par9[2], (int64_t) par9[3] };
  par.filter_contact_events = (int) par9[4];
  par.standing = (int) par9[5];
  par.total_force = (float) par9[6];
  par.standing_schmitt_level = (float) par9[7];
  par.use_controller_input = (int) par9[8];
  par.world_constraint = 1;
  for (int b = 0; b < B; b++) {
    LegState s;
    leg_load(s, legd, legi, stride, b, true);
    Pose bl, br, d;
    for (int i = 0; i < 3; i++) { bl.t[i] = feet[i * B + b]; br.t[i] = feet[(7 + i) * B + b]; }
    for (int i = 0; i < 4; i++) { bl.q[i] = feet[(3 + i) * B + b]; br.q[i] = feet[(10 + i) * B + b]; }
    const double w[4] = { wq[b], wq[B + b], wq[2 * B + b], wq[3 * B + b] }, wp[3] = { wpos[b], wpos[B + b], wpos[2 * B + b] };
    double position[3];
    bool ok;
    status[b] = leg_update(s, par, utime, bl, br, (float) forces[b], (float) forces[B + b], nc[0], nc[1], w, d, prev[b], wp, position, ok);
    if (leg_zero_velocity(s, status[b])) {
      pose_identity(d);
      position[0] = position[1] = position[2] = 0.0;
    }
    leg_store(s, legd, legi, stride, b, true);
    for (int i = 0; i < 3; i++) { delta[i * B + b] = d.t[i]; pos[i * B + b] = position[i]; }
    for (int i = 0; i < 4; i++) delta[(3 + i) * B + b] = d.q[i];
    pos_ok[b] = ok;
  }
}
// forward kinematics of one chain with the device code's quaternion arithmetic: type / origin_xyz_rpy [n][6] / axis [n][3] as
// pb_legodo_set_chain takes them, angle [n] -> t[3], q[4]
void hh_fk(int n, const int *type, const double *origin_xyz_rpy, const double *axis, const double *angle, double *t, double *q)
{
  LegChain ch;
  memset(&ch, 0, sizeof ch);
  ch.n[0] = n;
  for (int j = 0; j < n; j++) leg_chain_entry(ch, 0, j, type[j], j, origin_xyz_rpy + 6 * j, axis + 3 * j, 0.0f);
  Pose T;
  double ang[LEG_MAXJ];
  leg_angles(ch, 0, [&](int j) { return j < n ? angle[j] : 0.0; }, ang);
  leg_fk(ch, 0, ang, [&](int j, int f) { return ch.rec[0][j][f]; }, T);
  for (int i = 0; i < 3; i++) t[i] = T.t[i];
  for (int i = 0; i < 4; i++) q[i] = T.q[i];
}
void hh_sincos_joint(int n, const double *x, double *s, double *c)
{
  for (int i = 0; i < n; i++) sincos_joint(x[i], s[i], c[i]);
}
float hh_torque_adjust(float position, float effort, float gain) { return torque_adjust(position, effort, gain); }
}

// ---- joint-position filters (rbis_jointfilt.hpp): one joint through T messages with the per-context bookkeeping the library
// does around jf_lowpass / jf_kalman (window slot of the oldest sample, first-sample flag, previous time stamp) ----
#include "../pronto_amd/csrc/rbis_jointfilt.hpp"
extern "C" void hh_joint_filter(int mode, int T, const long *utime, const float *x, const float *xdot, double pn_pos, double pn_vel,
                                double r, float *out)
{
  double coef[JF_TAPS];
  jf_lowpass_coeffs(coef);
  float ring[JF_TAPS] = { 0 };
  double s[JF_KSTATE] = { 0, 0, 1, 0, 0, 1 };
  int head = 0;
  double tlast = 0;
  for (int k = 0; k < T; k++) {
    const bool first = k == 0;
    const double t = (double) utime[k] * 1E-6, dt = t - tlast;
    if (mode == JF_LOWPASS) {
      if (first) for (int i = 0; i < JF_TAPS; i++) ring[i] = x[k];
      else ring[head] = x[k];
      out[k] = jf_lowpass(coef, [&](int i) { return ring[(head + 1 + i) % JF_TAPS]; });
      if (!first) head = (head + 1) % JF_TAPS;
    } else {
      if (first) { s[0] = x[k]; s[1] = xdot[k]; out[k] = x[k]; }
      else out[k] = jf_kalman(s, dt, x[k], (float) pn_pos, (float) pn_vel, (float) r);
    }
    tlast = t;
  }
}
