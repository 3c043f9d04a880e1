/*
 * pronto_oracle.c -- CPU restatement of Pronto's RBIS/RBIM EKF hot path (see pronto_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference ships no golden vectors; see header).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Arithmetic is deliberately DENSE and in the reference's operation order (21x21 Ad*P*Ad^T,
 * 21x12 Wc, m x 21 selector C, pivoted LDLT, LU determinant) so that it doubles as the
 * "CPU restatement of reference" baseline of BASELINE.md section 4.
 */
#include "pronto_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define N PO_N
#define IDX(r, c) ((c) * N + (r)) /* column-major 21x21 */

/* eigen_utils constants [NOT IN TREE]; overridable so a maintainer can pin them to their eigen_utils */
static double g_val = 9.80665;
static double chi_tol = 1e-6;

void po_set_constants(double g, double tol) { g_val = g; chi_tol = tol; }
void po_get_constants(double *g, double *tol) { if (g) *g = g_val; if (tol) *tol = chi_tol; }

/* ------------------------------------------------------------------------------------------- */
/* Eigen::Quaterniond primitives (Eigen/src/Geometry/Quaternion.h published algorithms)        */
/* ------------------------------------------------------------------------------------------- */

void po_quat_mul(const double *a, const double *b, double *out)
{
  /* Eigen quat_product: Hamilton product, no normalisation */
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  out[0] = w; out[1] = x; out[2] = y; out[3] = z;
}

static void cross3(const double *a, const double *b, double *o)
{
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

void po_quat_rotate(const double *q, const double *v, double *out)
{
  /* QuaternionBase::_transformVector: uv = 2 * q.vec x v; v + w*uv + q.vec x uv */
  double uv[3], uuv[3];
  cross3(q + 1, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(q + 1, uv, uuv);
  out[0] = v[0] + q[0] * uv[0] + uuv[0];
  out[1] = v[1] + q[0] * uv[1] + uuv[1];
  out[2] = v[2] + q[0] * uv[2] + uuv[2];
}

static void quat_inverse(const double *q, double *qi)
{
  /* QuaternionBase::inverse: conjugate / squaredNorm */
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0) {
    qi[0] = q[0] / n2; qi[1] = -q[1] / n2; qi[2] = -q[2] / n2; qi[3] = -q[3] / n2;
  } else {
    qi[0] = qi[1] = qi[2] = qi[3] = 0;
  }
}

void po_quat_inv_rotate(const double *q, const double *v, double *out)
{
  double qi[4];
  quat_inverse(q, qi);
  po_quat_rotate(qi, v, out);
}

void po_quat_to_rot(const double *q, double *R)
{
  /* QuaternionBase::toRotationMatrix */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void skew_hat(const double *v, double *M /* 3x3 row-major */)
{
  /* eigen_utils::skewHat [NOT IN TREE]: [[0,-z,y],[z,0,-x],[-y,x,0]] */
  M[0] = 0;     M[1] = -v[2]; M[2] = v[1];
  M[3] = v[2];  M[4] = 0;     M[5] = -v[0];
  M[6] = -v[1]; M[7] = v[0];  M[8] = 0;
}

/* ------------------------------------------------------------------------------------------- */
/* eigen_utils::RigidBodyState semantics [NOT IN TREE -- restated, see header]                 */
/* ------------------------------------------------------------------------------------------- */

void po_rbis_zero(po_rbis *s)
{
  memset(s, 0, sizeof(*s));
  s->quat[0] = 1.0;
}

void po_chi_to_quat(po_rbis *s)
{
  double *chi = s->vec + PO_CHI;
  double n = sqrt(chi[0] * chi[0] + chi[1] * chi[1] + chi[2] * chi[2]);
  if (n > chi_tol) {
    /* dquat = AngleAxisd(n, chi/n); quat *= dquat; chi = 0 */
    double ax[3] = { chi[0] / n, chi[1] / n, chi[2] / n };
    double h = 0.5 * n;
    double sh = sin(h), ch = cos(h);
    double dq[4] = { ch, sh * ax[0], sh * ax[1], sh * ax[2] };
    double out[4];
    po_quat_mul(s->quat, dq, out);
    memcpy(s->quat, out, sizeof(out));
    chi[0] = chi[1] = chi[2] = 0.0;
  }
}

void po_rbis_from_vec(po_rbis *s, const double *vec)
{
  memcpy(s->vec, vec, sizeof(s->vec));
  s->quat[0] = 1; s->quat[1] = s->quat[2] = s->quat[3] = 0;
  s->utime = 0;
  po_chi_to_quat(s);
}

void po_add_state(po_rbis *s, const po_rbis *d)
{
  double out[4];
  for (int i = 0; i < N; i++) s->vec[i] += d->vec[i];
  po_chi_to_quat(s);
  po_quat_mul(s->quat, d->quat, out);
  memcpy(s->quat, out, sizeof(out));
}

static double mod2pi(double a)
{
  /* libbot bot_mod2pi: wrap into [-pi, pi] */
  const double twopi = 2.0 * M_PI;
  a = fmod(a + M_PI, twopi);
  if (a < 0) a += twopi;
  return a - M_PI;
}

void po_subtract_quats(const double *q1, const double *q2, double *out3)
{
  /* eigen_utils::subtractQuats: AngleAxisd(q2.inverse()*q1); angle wrapped; axis*angle.
   * AngleAxis(quaternion) per Eigen 3.3: angle = 2*atan2(|vec|, |w|), axis = vec/(+-|vec|). */
  double q2i[4], qr[4];
  quat_inverse(q2, q2i);
  po_quat_mul(q2i, q1, qr);
  double n = sqrt(qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
  if (n != 0.0) {
    double angle = 2.0 * atan2(n, fabs(qr[0]));
    if (qr[0] < 0) n = -n;
    angle = mod2pi(angle);
    out3[0] = qr[1] / n * angle;
    out3[1] = qr[2] / n * angle;
    out3[2] = qr[3] / n * angle;
  } else {
    out3[0] = out3[1] = out3[2] = 0.0; /* angle 0 about x */
  }
}

/* ------------------------------------------------------------------------------------------- */
/* rbis.cpp                                                                                    */
/* ------------------------------------------------------------------------------------------- */

static void set_block3(po_rbim *A, int r0, int c0, const double *M /* row-major 3x3 */, double scale)
{
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A->m[IDX(r0 + i, c0 + j)] = scale * M[3 * i + j];
}

void po_get_imu_linearization(const po_rbis *state, po_rbim *Ac)
{
  /* rbis.cpp:12-35 */
  static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  double omega_hat[9], vb_hat[9], R[9], gb[3], gb_hat[9], RV[9];
  const double gvec[3] = { 0, 0, -g_val };
  memset(Ac, 0, sizeof(*Ac));
  skew_hat(state->vec + PO_ANGVEL, omega_hat);
  skew_hat(state->vec + PO_VEL, vb_hat);
  po_quat_to_rot(state->quat, R);
  po_quat_inv_rotate(state->quat, gvec, gb);
  skew_hat(gb, gb_hat);

  set_block3(Ac, PO_VEL, PO_VEL, omega_hat, -1.0);  /* :20 */
  set_block3(Ac, PO_VEL, PO_CHI, gb_hat, 1.0);      /* :21 */
  set_block3(Ac, PO_CHI, PO_CHI, omega_hat, -1.0);  /* :24 */
  set_block3(Ac, PO_POS, PO_VEL, R, 1.0);           /* :27 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += (-R[3 * i + k]) * vb_hat[3 * k + j];
      RV[3 * i + j] = s;
    }
  set_block3(Ac, PO_POS, PO_CHI, RV, 1.0);               /* :28 */
  set_block3(Ac, PO_VEL, PO_GYRO_BIAS, vb_hat, -1.0);    /* :31 */
  set_block3(Ac, PO_VEL, PO_ACCEL_BIAS, I3, -1.0);       /* :32 */
  set_block3(Ac, PO_CHI, PO_GYRO_BIAS, I3, -1.0);        /* :33 */
}

void po_ins_update_state(const double *gyro, const double *accel, double dt, po_rbis *state)
{
  /* rbis.cpp:37-75 */
  const double gvec[3] = { 0, 0, -g_val };
  po_rbis d;
  double wxv[3], gb[3], Rv[3];
  for (int i = 0; i < 3; i++) {
    state->vec[PO_ANGVEL + i] = gyro[i] - state->vec[PO_GYRO_BIAS + i];  /* :50 */
    state->vec[PO_ACC + i] = accel[i] - state->vec[PO_ACCEL_BIAS + i];   /* :51 */
  }
  po_rbis_zero(&d);                                                      /* :54 */
  cross3(state->vec + PO_ANGVEL, state->vec + PO_VEL, wxv);
  po_quat_inv_rotate(state->quat, gvec, gb);
  for (int i = 0; i < 3; i++) {
    d.vec[PO_VEL + i] = -wxv[i];                                         /* :55 */
    d.vec[PO_VEL + i] += gb[i] + state->vec[PO_ACC + i];                 /* :56 */
    d.vec[PO_CHI + i] = state->vec[PO_ANGVEL + i];                       /* :58 */
  }
  po_quat_rotate(state->quat, state->vec + PO_VEL, Rv);
  for (int i = 0; i < 3; i++) d.vec[PO_POS + i] = Rv[i];                 /* :59 */
  for (int i = 0; i < N; i++) d.vec[i] *= dt;                            /* :62 */
  po_chi_to_quat(&d);                                                    /* :63 */
  po_add_state(state, &d);                                               /* :69 */
}

static void matmul_nn(const double *A, const double *B, double *C)
{ /* C = A*B, all 21x21 col-major */
  for (int c = 0; c < N; c++)
    for (int r = 0; r < N; r++) {
      double s = 0;
      for (int k = 0; k < N; k++) s += A[IDX(r, k)] * B[IDX(k, c)];
      C[IDX(r, c)] = s;
    }
}
static void matmul_nt(const double *A, const double *B, double *C)
{ /* C = A*B^T */
  for (int c = 0; c < N; c++)
    for (int r = 0; r < N; r++) {
      double s = 0;
      for (int k = 0; k < N; k++) s += A[IDX(r, k)] * B[IDX(c, k)];
      C[IDX(r, c)] = s;
    }
}

void po_ins_update_covariance(double q_gyro, double q_accel, double q_gyro_bias, double q_accel_bias,
                              const po_rbis *state, po_rbim *cov, double dt)
{
  /* rbis.cpp:77-122 */
  enum { NI = 12, GY = 0, AC = 3, GB = 6, AB = 9 };
  static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  po_rbim Ac, Ad, Qd, T1, T2;
  double Wc[N * NI], WQ[N * NI], Qc[NI], vhat[9]; /* Wc col-major 21x12 */
  po_get_imu_linearization(state, &Ac);                                   /* :81 */

  memset(Wc, 0, sizeof(Wc));
  skew_hat(state->vec + PO_VEL, vhat);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Wc[(GY + j) * N + PO_VEL + i] = vhat[3 * i + j];                    /* :93 */
      Wc[(AC + j) * N + PO_VEL + i] = I3[3 * i + j];                      /* :94 */
      Wc[(GY + j) * N + PO_CHI + i] = I3[3 * i + j];                      /* :97 */
      Wc[(GB + j) * N + PO_GYRO_BIAS + i] = I3[3 * i + j];                /* :99 */
      Wc[(AB + j) * N + PO_ACCEL_BIAS + i] = I3[3 * i + j];               /* :100 */
    }
  for (int i = 0; i < 3; i++) {
    Qc[GY + i] = q_gyro; Qc[AC + i] = q_accel; Qc[GB + i] = q_gyro_bias; Qc[AB + i] = q_accel_bias; /* :102-107 */
  }
  /* Ad = I + Ac*dt  (:112-114) */
  for (int i = 0; i < N * N; i++) Ad.m[i] = Ac.m[i] * dt;
  for (int i = 0; i < N; i++) Ad.m[IDX(i, i)] += 1.0;
  /* Qd = Wc*Qc*Wc^T*dt  (:116) */
  for (int k = 0; k < NI; k++)
    for (int r = 0; r < N; r++) WQ[k * N + r] = Wc[k * N + r] * Qc[k];
  for (int c = 0; c < N; c++)
    for (int r = 0; r < N; r++) {
      double s = 0;
      for (int k = 0; k < NI; k++) s += WQ[k * N + r] * Wc[k * N + c];
      Qd.m[IDX(r, c)] = s * dt;
    }
  /* cov = Ad*cov*Ad^T + Qd  (:118) */
  matmul_nn(Ad.m, cov->m, T1.m);
  matmul_nt(T1.m, Ad.m, T2.m);
  for (int i = 0; i < N * N; i++) cov->m[i] = T2.m[i] + Qd.m[i];
  /* :120-121 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      cov->m[IDX(PO_ACC + i, PO_ACC + j)] = q_accel * I3[3 * i + j];
      cov->m[IDX(PO_ANGVEL + i, PO_ANGVEL + j)] = q_gyro * I3[3 * i + j];
    }
}

/* Eigen::LDLT (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked): symmetric pivoting on the
 * largest remaining |diagonal|.  A is m x m col-major (full symmetric copy), overwritten with L (unit
 * lower) and D on the diagonal; perm[k] = transposition applied at step k. */
#define MMAX 21
static void ldlt_factor(int m, double *A, int *perm)
{
  for (int k = 0; k < m; k++) {
    int p = k;
    double big = fabs(A[k * m + k]);
    for (int i = k + 1; i < m; i++)
      if (fabs(A[i * m + i]) > big) { big = fabs(A[i * m + i]); p = i; }
    perm[k] = p;
    if (p != k) { /* symmetric swap of rows/cols k and p on the full matrix */
      for (int c = 0; c < m; c++) { double t = A[c * m + k]; A[c * m + k] = A[c * m + p]; A[c * m + p] = t; }
      for (int r = 0; r < m; r++) { double t = A[k * m + r]; A[k * m + r] = A[p * m + r]; A[p * m + r] = t; }
    }
    /* A(k,k) -= sum_j L(k,j)^2 D_j ; A(i,k) = (A(i,k) - sum_j L(i,j) D_j L(k,j)) / A(k,k) */
    double temp[MMAX];
    for (int j = 0; j < k; j++) temp[j] = A[j * m + j] * A[j * m + k];
    double d = A[k * m + k];
    for (int j = 0; j < k; j++) d -= A[j * m + k] * temp[j];
    A[k * m + k] = d;
    for (int i = k + 1; i < m; i++) {
      double s = A[k * m + i];
      for (int j = 0; j < k; j++) s -= A[j * m + i] * temp[j];
      if (fabs(d) > 0) s /= d;
      A[k * m + i] = s;
    }
  }
}

static void ldlt_solve(int m, const double *LD, const int *perm, double *b /* in/out, length m */)
{
  /* x = P^T L^-T D^-1 L^-1 P b */
  const double tol = 1.0 / 1.7976931348623157e308;
  for (int k = 0; k < m; k++)
    if (perm[k] != k) { double t = b[k]; b[k] = b[perm[k]]; b[perm[k]] = t; }
  for (int i = 0; i < m; i++)
    for (int j = 0; j < i; j++) b[i] -= LD[j * m + i] * b[j];
  for (int i = 0; i < m; i++) {
    double d = LD[i * m + i];
    if (fabs(d) > tol) b[i] /= d; else b[i] = 0;
  }
  for (int i = m - 1; i >= 0; i--)
    for (int j = i + 1; j < m; j++) b[i] -= LD[i * m + j] * b[j];
  for (int k = m - 1; k >= 0; k--)
    if (perm[k] != k) { double t = b[k]; b[k] = b[perm[k]]; b[perm[k]] = t; }
}

/* MatrixXd::determinant() for dynamic sizes = PartialPivLU::determinant() */
static double lu_det(int m, const double *Ain)
{
  double A[MMAX * MMAX];
  double det = 1.0;
  memcpy(A, Ain, sizeof(double) * m * m);
  for (int k = 0; k < m; k++) {
    int p = k;
    double big = fabs(A[k * m + k]);
    for (int i = k + 1; i < m; i++)
      if (fabs(A[k * m + i]) > big) { big = fabs(A[k * m + i]); p = i; }
    if (p != k) {
      for (int c = 0; c < m; c++) { double t = A[c * m + k]; A[c * m + k] = A[c * m + p]; A[c * m + p] = t; }
      det = -det;
    }
    double piv = A[k * m + k];
    if (piv == 0) return 0.0;
    for (int i = k + 1; i < m; i++) {
      double f = A[k * m + i] / piv;
      A[k * m + i] = f;
      for (int c = k + 1; c < m; c++) A[c * m + i] -= f * A[c * m + k];
    }
  }
  for (int k = 0; k < m; k++) det *= A[k * m + k];
  return det;
}

double po_matrix_measurement_k_dcov(int m, const double *R, const double *C, const po_rbim *cov,
                                    const double *z_resid, po_rbim *dcov, double *K)
{
  /* rbis.cpp:124-143.  C: m x 21 col-major, K: 21 x m col-major */
  double CP[MMAX * N];    /* m x 21 col-major: C*cov */
  double S[MMAX * MMAX], LD[MMAX * MMAX], X[MMAX * N], KC[N * N];
  int perm[MMAX];
  /* C*cov */
  for (int c = 0; c < N; c++)
    for (int r = 0; r < m; r++) {
      double s = 0;
      for (int k = 0; k < N; k++) s += C[k * m + r] * cov->m[IDX(k, c)];
      CP[c * m + r] = s;
    }
  /* S = R; S += C*cov*C^T  (:134-135) */
  for (int c = 0; c < m; c++)
    for (int r = 0; r < m; r++) {
      double s = 0;
      for (int k = 0; k < N; k++) s += CP[k * m + r] * C[k * m + c];
      S[c * m + r] = R[c * m + r] + s;
    }
  memcpy(LD, S, sizeof(double) * m * m);
  ldlt_factor(m, LD, perm);                                               /* :137 */
  /* K^T = Sldlt.solve(C*cov)  (:139): solve column by column */
  for (int c = 0; c < N; c++) {
    double col[MMAX];
    for (int r = 0; r < m; r++) col[r] = CP[c * m + r];
    ldlt_solve(m, LD, perm, col);
    for (int r = 0; r < m; r++) X[c * m + r] = col[r];
  }
  for (int r = 0; r < N; r++)
    for (int k = 0; k < m; k++) K[k * N + r] = X[r * m + k];
  /* cov_delta = K*C*cov, evaluated left to right: (K*C)*cov  (:140) */
  for (int c = 0; c < N; c++)
    for (int r = 0; r < N; r++) {
      double s = 0;
      for (int k = 0; k < m; k++) s += K[k * N + r] * C[c * m + k];
      KC[IDX(r, c)] = s;
    }
  matmul_nn(KC, cov->m, dcov->m);
  /* -log(det S) - r^T S^-1 r  (:142) */
  {
    double sol[MMAX], quad = 0;
    memcpy(sol, z_resid, sizeof(double) * m);
    ldlt_solve(m, LD, perm, sol);
    for (int i = 0; i < m; i++) quad += z_resid[i] * sol[i];
    return -log(lu_det(m, S)) - quad;
  }
}

static double measurement_common(int m, const double *z_resid, const double *R, const int *idx, const po_rbim *cov,
                                 po_rbis *dstate, po_rbim *dcov)
{
  double C[MMAX * N], K[N * MMAX], dx[N];
  memset(C, 0, sizeof(double) * m * N);
  memset(K, 0, sizeof(double) * m * N);
  for (int i = 0; i < m; i++) C[idx[i] * m + i] = 1.0;
  double ll = po_matrix_measurement_k_dcov(m, R, C, cov, z_resid, dcov, K);
  for (int r = 0; r < N; r++) {
    double s = 0;
    for (int k = 0; k < m; k++) s += K[k * N + r] * z_resid[k];
    dx[r] = s;
  }
  po_rbis_from_vec(dstate, dx); /* dstate = RBIS(K * z_resid) */
  return ll;
}

double po_indexed_measurement(int m, const double *z, const double *R, const int *idx, const po_rbis *state,
                              const po_rbim *cov, po_rbis *dstate, po_rbim *dcov)
{
  /* rbis.cpp:160-178 */
  double resid[MMAX];
  for (int i = 0; i < m; i++) resid[i] = z[i] - state->vec[idx[i]];
  return measurement_common(m, resid, R, idx, cov, dstate, dcov);
}

double po_indexed_plus_orientation_measurement(int m, const double *z, const double *quat, const double *R,
                                               const int *idx, const po_rbis *state, const po_rbim *cov,
                                               po_rbis *dstate, po_rbim *dcov)
{
  /* rbis.cpp:189-217 */
  double resid[MMAX], dq[3];
  po_subtract_quats(quat, state->quat, dq);                               /* :199 */
  for (int i = 0; i < m; i++) {
    if (idx[i] >= PO_CHI && idx[i] <= PO_CHI + 2) resid[i] = dq[idx[i] - PO_CHI];  /* :203-205 */
    else resid[i] = z[i] - state->vec[idx[i]];                                      /* :207 */
  }
  return measurement_common(m, resid, R, idx, cov, dstate, dcov);
}

void po_apply_delta(const po_rbis *prior, const po_rbim *prior_cov, const po_rbis *dstate, const po_rbim *dcov,
                    po_rbis *post, po_rbim *post_cov)
{
  /* rbis.cpp:219-227 */
  po_rbis s = *prior;
  po_add_state(&s, dstate);
  for (int i = 0; i < N * N; i++) post_cov->m[i] = prior_cov->m[i] - dcov->m[i];
  *post = s;
}

/* dense LDLT solve with n = 21 right-hand-side matrix, used by the smoother */
void po_ekf_smoothing_step(const po_rbis *next_state_pred, const po_rbim *next_cov_pred, const po_rbis *next_state,
                           const po_rbim *next_cov, double dt, po_rbis *cur_state, po_rbim *cur_cov)
{
  /* rbis.cpp:234-266 */
  po_rbim Ac, Ad, corr, AP, L, D, T1, T2;
  int perm[N];
  po_get_imu_linearization(cur_state, &Ac);
  for (int i = 0; i < N * N; i++) Ad.m[i] = Ac.m[i] * dt;
  for (int i = 0; i < N; i++) Ad.m[IDX(i, i)] += 1.0;
  corr = *next_cov_pred;
  for (int blk = 0; blk < 2; blk++) {
    int o = blk ? PO_ACCEL_BIAS : PO_GYRO_BIAS, any = 0;
    for (int i = 0; i < 3; i++) any |= (next_cov_pred->m[IDX(o + i, o + i)] < .00000000001);
    if (any)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) corr.m[IDX(o + i, o + j)] = (i == j);
  }
  /* L^T = corr.ldlt().solve(Ad*cur_cov) */
  matmul_nn(Ad.m, cur_cov->m, AP.m);
  {
    double LD[N * N];
    memcpy(LD, corr.m, sizeof(LD));
    ldlt_factor(N, LD, perm);
    for (int c = 0; c < N; c++) {
      double col[N];
      for (int r = 0; r < N; r++) col[r] = AP.m[IDX(r, c)];
      ldlt_solve(N, LD, perm, col);
      for (int r = 0; r < N; r++) L.m[IDX(c, r)] = col[r]; /* transpose */
    }
  }
  for (int i = 0; i < N * N; i++) D.m[i] = next_cov->m[i] - next_cov_pred->m[i];
  matmul_nn(L.m, D.m, T1.m);
  matmul_nt(T1.m, L.m, T2.m);
  for (int i = 0; i < N * N; i++) cur_cov->m[i] = cur_cov->m[i] + T2.m[i];
  {
    /* smooth_resid = next_state (-) next_state_pred; quatToChi */
    po_rbis resid = *next_state, innov;
    double qi[4], qr[4], chi[3], dx[N];
    const double ident[4] = { 1, 0, 0, 0 };
    for (int i = 0; i < N; i++) resid.vec[i] -= next_state_pred->vec[i];
    quat_inverse(next_state_pred->quat, qi);
    po_quat_mul(qi, resid.quat, qr);
    po_subtract_quats(qr, ident, chi);
    for (int i = 0; i < 3; i++) resid.vec[PO_CHI + i] = chi[i];
    for (int r = 0; r < N; r++) {
      double s = 0;
      for (int k = 0; k < N; k++) s += L.m[IDX(r, k)] * resid.vec[k];
      dx[r] = s;
    }
    po_rbis_from_vec(&innov, dx);
    po_add_state(cur_state, &innov);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* rbis_update_interface.cpp                                                                   */
/* ------------------------------------------------------------------------------------------- */

void po_imu_process_step(const double *gyro, const double *accel, double dt, double q_gyro, double q_accel,
                         double q_gyro_bias, double q_accel_bias, const po_rbis *prior, const po_rbim *prior_cov,
                         double prior_ll, po_rbis *post, po_rbim *post_cov, double *post_ll)
{
  /* rbis_update_interface.cpp:30-52.  NB :39 linearises the covariance about the PRIOR state. */
  po_rbis prior_copy = *prior; /* post may alias prior */
  po_rbis s = *prior;
  po_rbim c = *prior_cov;
  po_ins_update_state(gyro, accel, dt, &s);
  po_ins_update_covariance(q_gyro, q_accel, q_gyro_bias, q_accel_bias, &prior_copy, &c, dt);
  *post = s;
  *post_cov = c;
  *post_ll = prior_ll;
}

static void diag_to_full(int m, const double *Rdiag, double *R)
{
  memset(R, 0, sizeof(double) * m * m);
  for (int i = 0; i < m; i++) R[i * m + i] = Rdiag[i];
}

void po_indexed_update(int m, const int *idx, const double *z, const double *R, const po_rbis *prior,
                       const po_rbim *prior_cov, double prior_ll, po_rbis *post, po_rbim *post_cov, double *post_ll)
{
  /* rbis_update_interface.cpp:54-95 */
  po_rbis dstate;
  po_rbim dcov;
  double ll = po_indexed_measurement(m, z, R, idx, prior, prior_cov, &dstate, &dcov);
  po_apply_delta(prior, prior_cov, &dstate, &dcov, post, post_cov);
  *post_ll = prior_ll + ll;
}

void po_indexed_orient_update(int m, const int *idx, const double *z, const double *R, const double *quat,
                              const po_rbis *prior, const po_rbim *prior_cov, double prior_ll, po_rbis *post,
                              po_rbim *post_cov, double *post_ll)
{
  /* rbis_update_interface.cpp:97-107 */
  po_rbis dstate;
  po_rbim dcov;
  double ll = po_indexed_plus_orientation_measurement(m, z, quat, R, idx, prior, prior_cov, &dstate, &dcov);
  po_apply_delta(prior, prior_cov, &dstate, &dcov, post, post_cov);
  *post_ll = prior_ll + ll;
}

/* ------------------------------------------------------------------------------------------- */
/* measurement formers                                                                         */
/* ------------------------------------------------------------------------------------------- */

void po_euler_to_quat(double roll, double pitch, double yaw, double *q)
{
  /* pronto_math.cpp:25-50 */
  if (roll == M_PI && pitch == 0 && yaw == 0) { q[0] = 0; q[1] = 1; q[2] = 0; q[3] = 0; return; }
  if (pitch == M_PI && roll == 0 && yaw == 0) { q[0] = 0; q[1] = 0; q[2] = 1; q[3] = 0; return; }
  if (yaw == M_PI && roll == 0 && pitch == 0) { q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 1; return; }
  double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5);
  double sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
  double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
  q[0] = cr * cp * cy + sr * sp * sy;
  q[1] = sr * cp * cy - cr * sp * sy;
  q[2] = cr * sp * cy + sr * cp * sy;
  q[3] = cr * cp * sy - sr * sp * cy;
}

void po_quat_to_euler(const double *q, double *rpy)
{
  /* pronto_math.cpp:53-61 */
  double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  rpy[0] = atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
  rpy[1] = asin(2 * (q0 * q2 - q3 * q1));
  rpy[2] = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
}

void po_delta_as_velocity(const double *t, const double *q, int64_t dt_us, double *t_vel, double *q_vel)
{
  /* pronto_conversions_lcm.hpp:38-87 (the rotation passes through a matrix in the reference; the
   * quaternion here is used directly -- equal up to sign/rounding, and quat_to_euler is sign-invariant) */
  double rpy[3];
  po_quat_to_euler(q, rpy);
  double elapsed = (double) dt_us * 1E-6;
  for (int i = 0; i < 3; i++) t_vel[i] = t[i] / elapsed;
  po_euler_to_quat(rpy[0] / elapsed, rpy[1] / elapsed, rpy[2] / elapsed, q_vel);
}

int po_legodo_create_measurement(int mode, const double *r, const double *pos_t, const double *delta_t,
                                 const double *delta_q, int64_t utime, int64_t prev_utime, int pos_status,
                                 float delta_status, int *idx, double *z, double *Rdiag)
{
  /* rbis_legodo_common.cpp:110-169 with getCovariance :34-88 */
  enum { LIN_RATE = 0, LIN_ROT_RATE = 1, POS_LIN_RATE = 2 };
  double tv[3], qv[4];
  po_delta_as_velocity(delta_t, delta_q, utime - prev_utime, tv, qv);    /* :113 */
  int cur = mode;
  if (cur == POS_LIN_RATE && !pos_status) cur = LIN_RATE;                /* :118-122 */
  int certain = (delta_status < 0.5);                                    /* :124-129 */
  double rv = certain ? r[1] : r[3], ra = certain ? r[2] : r[4];
  if (cur == LIN_ROT_RATE) {
    double rpy[3];
    po_quat_to_euler(delta_q, rpy);                                      /* :142 bot_quat_to_roll_pitch_yaw */
    double el = ((double) utime - prev_utime) / 1000000;
    for (int i = 0; i < 3; i++) {
      idx[i] = PO_VEL + i; idx[3 + i] = PO_ANGVEL + i;
      z[i] = tv[i]; z[3 + i] = rpy[i] / el;
      Rdiag[i] = rv * rv; Rdiag[3 + i] = ra * ra;
    }
    return 6;
  } else if (cur == LIN_RATE) {
    for (int i = 0; i < 3; i++) { idx[i] = PO_VEL + i; z[i] = tv[i]; Rdiag[i] = rv * rv; }
    return 3;
  } else {
    for (int i = 0; i < 3; i++) {
      idx[i] = PO_POS + i; idx[3 + i] = PO_VEL + i;
      z[i] = pos_t[i]; z[3 + i] = tv[i];
      Rdiag[i] = r[0] * r[0]; Rdiag[3 + i] = rv * rv;
    }
    return 6;
  }
}

void po_fovis_compose(const double *pos0, const double *quat0, const double *t, const double *q, double *z3,
                      double *q_meas)
{
  /* rbis_fovis_update.cpp:199-223: t1 = t0_internal * t0t1_vo  (Isometry product) */
  double rt[3];
  po_quat_rotate(quat0, t, rt);
  for (int i = 0; i < 3; i++) z3[i] = pos0[i] + rt[i];
  po_quat_mul(quat0, q, q_meas);
}

/* ------------------------------------------------------------------------------------------- */
/* IMU front end                                                                               */
/* ------------------------------------------------------------------------------------------- */

void po_notch_init(po_notch *f, double notch_freq, double fs)
{
  /* iir_notch.cpp:3-32 */
  double Wo = notch_freq / (fs / 2);
  double BW = Wo;
  double Ab = fabs(10 * log10(.5));
  BW = BW * M_PI;
  Wo = Wo * M_PI;
  double Gb = pow(10, -Ab / 20.);
  double beta = (sqrt(1.0 - Gb * Gb) / Gb) * tan(BW / 2.0);
  double gain = 1 / (1 + beta);
  f->b[0] = gain * 1.0; f->b[1] = gain * (-2.0 * cos(Wo)); f->b[2] = gain * 1;
  f->a[0] = 1.0; f->a[1] = -2 * gain * cos(Wo); f->a[2] = 2 * gain - 1;
  f->x[0] = f->x[1] = f->y[0] = f->y[1] = 0;
}

double po_notch_process(po_notch *f, double input)
{
  /* iir_notch.cpp:34-61: output = [input x0 x1].b - [0 y0 y1].a */
  double xb = input * f->b[0] + f->x[0] * f->b[1] + f->x[1] * f->b[2];
  double ya = 0 * f->a[0] + f->y[0] * f->a[1] + f->y[1] * f->a[2];
  double output = xb - ya;
  f->x[1] = f->x[0]; f->x[0] = input;
  f->y[1] = f->y[0]; f->y[0] = output;
  return output;
}

void po_notch_cascade_init(po_notch *filt9, double notch_freq, double fs)
{
  /* sensor_handlers.cpp:29-42: IIRNotch(notch_freq*pow(2,i), fs), i = 0..2, for x, y, z */
  for (int ax = 0; ax < 3; ax++)
    for (int i = 0; i < 3; i++) po_notch_init(&filt9[ax * 3 + i], notch_freq * pow(2, i), fs);
}

void po_notch_cascade(po_notch *filt9, double *acc3)
{
  /* sensor_handlers.cpp:154-162 */
  for (int i = 0; i < 3; i++)
    for (int ax = 0; ax < 3; ax++) acc3[ax] = po_notch_process(&filt9[ax * 3 + i], acc3[ax]);
}

/* ------------------------------------------------------------------------------------------- */
/* noise identification                                                                        */
/* ------------------------------------------------------------------------------------------- */

void po_noise_id_window(int Nw, const po_rbis *truth, const po_rbim *start_cov, double dt, double q_gyro, double q_accel,
                        po_rbis *err_out, po_rbim *cov_out)
{
  /* noise_id.cpp:19-40 */
  po_rbis rolled = truth[0];
  po_rbim start_window_cov = *start_cov, rolled_cov = *start_cov;
  for (int ii = 0; ii < Nw; ii++) {
    po_ins_update_covariance(q_gyro, q_accel, 0, 0, &rolled, &rolled_cov, dt);      /* :24 */
    po_ins_update_covariance(0, 0, 0, 0, &rolled, &start_window_cov, dt);           /* :25 */
    po_ins_update_state(truth[ii].vec + PO_ANGVEL, truth[ii].vec + PO_ACC, dt, &rolled); /* :26 */
  }
  /* rolled_state.subtractState(truth[N]); quatToChi()  (:37-38) */
  double qi[4], qr[4], chi[3];
  const double ident[4] = { 1, 0, 0, 0 };
  for (int i = 0; i < N; i++) rolled.vec[i] -= truth[Nw].vec[i];
  quat_inverse(truth[Nw].quat, qi);
  po_quat_mul(qi, rolled.quat, qr);
  po_subtract_quats(qr, ident, chi);
  for (int i = 0; i < 3; i++) rolled.vec[PO_CHI + i] = chi[i];
  rolled.quat[0] = 1; rolled.quat[1] = rolled.quat[2] = rolled.quat[3] = 0;
  *err_out = rolled;
  for (int i = 0; i < N * N; i++) cov_out->m[i] = rolled_cov.m[i] - start_window_cov.m[i];  /* :40 */
}

double po_loglike_pieces(int m, const int *idx, const po_rbis *err, const po_rbim *cov, double *logdet, double *maha)
{
  /* noise_id.cpp:52-58: cov_active = cov(idx, idx), error_active = err(idx) */
  double S[MMAX * MMAX], LD[MMAX * MMAX], e[MMAX], sol[MMAX];
  int perm[MMAX];
  for (int c = 0; c < m; c++)
    for (int r = 0; r < m; r++) S[c * m + r] = cov->m[IDX(idx[r], idx[c])];
  for (int i = 0; i < m; i++) e[i] = sol[i] = err->vec[idx[i]];
  memcpy(LD, S, sizeof(double) * m * m);
  ldlt_factor(m, LD, perm);
  ldlt_solve(m, LD, perm, sol);
  double q = 0;
  for (int i = 0; i < m; i++) q += e[i] * sol[i];
  const double ld = log(lu_det(m, S));
  if (logdet) *logdet = ld;
  if (maha) *maha = q;
  return -0.5 * (m * log(2 * M_PI) + ld + q);
}

/* ------------------------------------------------------------------------------------------- */
/* batch drivers                                                                               */
/* ------------------------------------------------------------------------------------------- */

int po_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void gather(const po_batch *s, int b, po_rbis *x, po_rbim *P, double *ll)
{
  int B = s->B;
  for (int i = 0; i < N; i++) x->vec[i] = s->vec[(size_t) i * B + b];
  for (int i = 0; i < 4; i++) x->quat[i] = s->quat[(size_t) i * B + b];
  for (int i = 0; i < N * N; i++) P->m[i] = s->cov[(size_t) i * B + b];
  x->utime = 0;
  *ll = s->ll[b];
}
static void scatter(po_batch *s, int b, const po_rbis *x, const po_rbim *P, double ll)
{
  int B = s->B;
  for (int i = 0; i < N; i++) s->vec[(size_t) i * B + b] = x->vec[i];
  for (int i = 0; i < 4; i++) s->quat[(size_t) i * B + b] = x->quat[i];
  for (int i = 0; i < N * N; i++) s->cov[(size_t) i * B + b] = P->m[i];
  s->ll[b] = ll;
}

/* nthreads <= 0: automatic -- at most 16 (the CPU share of a one-GPU box; omp_get_max_threads() reports the HOST's cores, and a
 * team of hundreds of spinning threads on 16 CPUs turns every call into tens of milliseconds), one thread per 32 filters */
static int pick_threads(int nthreads, int B)
{
  int mx = po_max_threads();
  if (nthreads <= 0) {
    nthreads = mx < 16 ? mx : 16;
    if (nthreads > B / 32 + 1) nthreads = B / 32 + 1;
  }
  if (nthreads > mx) nthreads = mx;
  return nthreads;
}

void po_batch_predict(po_batch *s, const double *imu, const double *q4, int nthreads)
{
  int B = s->B;
  nthreads = pick_threads(nthreads, B);
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; b++) {
    po_rbis x; po_rbim P; double ll;
    double gyro[3], accel[3];
    gather(s, b, &x, &P, &ll);
    for (int i = 0; i < 3; i++) { gyro[i] = imu[(size_t) i * B + b]; accel[i] = imu[(size_t)(3 + i) * B + b]; }
    po_imu_process_step(gyro, accel, imu[(size_t) 6 * B + b], q4[0], q4[1], q4[2], q4[3], &x, &P, ll, &x, &P, &ll);
    scatter(s, b, &x, &P, ll);
  }
}

void po_batch_update_indexed(po_batch *s, int m, const int *idx, const double *z, const double *Rdiag,
                             const double *quat_meas, const uint8_t *mask, int nthreads)
{
  int B = s->B;
  nthreads = pick_threads(nthreads, B);
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; b++) {
    if (mask && !mask[b]) continue;
    po_rbis x; po_rbim P; double ll;
    double zz[MMAX], rd[MMAX], R[MMAX * MMAX];
    gather(s, b, &x, &P, &ll);
    for (int i = 0; i < m; i++) { zz[i] = z[(size_t) i * B + b]; rd[i] = Rdiag[(size_t) i * B + b]; }
    diag_to_full(m, rd, R);
    if (quat_meas) {
      double qm[4];
      for (int i = 0; i < 4; i++) qm[i] = quat_meas[(size_t) i * B + b];
      po_indexed_orient_update(m, idx, zz, R, qm, &x, &P, ll, &x, &P, &ll);
    } else {
      po_indexed_update(m, idx, zz, R, &x, &P, ll, &x, &P, &ll);
    }
    scatter(s, b, &x, &P, ll);
  }
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double po_batch_run_legodo(po_batch *s, int T, const double *imu_stream, const double *lo_stream,
                           const uint8_t *mask_stream, const double *q4, int nthreads)
{
  /* One filter = one sequential replay (the reference's own structure: mav_state_est.cpp:50-70), filters
   * split statically over threads (BASELINE.md section 4 "cpu-dense-Nt"). */
  static const int idx[3] = { PO_VEL, PO_VEL + 1, PO_VEL + 2 };
  int B = s->B;
  nthreads = pick_threads(nthreads, B);
  double t0 = now_s();
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; b++) {
    po_rbis x; po_rbim P; double ll;
    gather(s, b, &x, &P, &ll);
    for (int k = 0; k < T; k++) {
      const double *imu = imu_stream + (size_t) k * 7 * B;
      const double *lo = lo_stream + (size_t) k * 6 * B;
      double gyro[3], accel[3], z[3], rd[3], R[9];
      for (int i = 0; i < 3; i++) {
        gyro[i] = imu[(size_t) i * B + b];
        accel[i] = imu[(size_t)(3 + i) * B + b];
        z[i] = lo[(size_t) i * B + b];
        rd[i] = lo[(size_t)(3 + i) * B + b];
      }
      po_imu_process_step(gyro, accel, imu[(size_t) 6 * B + b], q4[0], q4[1], q4[2], q4[3], &x, &P, ll, &x, &P, &ll);
      if (!mask_stream || mask_stream[(size_t) k * B + b]) {
        diag_to_full(3, rd, R);
        po_indexed_update(3, idx, z, R, &x, &P, ll, &x, &P, &ll);
      }
    }
    scatter(s, b, &x, &P, ll);
  }
  return now_s() - t0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * INS initialisation (sensor_handlers.cpp:283-364)
 * ------------------------------------------------------------------------------------------------------------- */
void po_quat_from_two_vectors(const double *a, const double *b, double *q)
{
  double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  double v0[3], v1[3];
  int i;
  for (i = 0; i < 3; i++) {
    v0[i] = a[i] / na;
    v1[i] = b[i] / nb;
  }
  double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
  if (c < -1.0 + 1e-12) { /* opposite vectors: half a turn about any axis orthogonal to v0 */
    double ax[3] = { 0, 0, 0 };
    int k = (fabs(v0[0]) < fabs(v0[1])) ? (fabs(v0[0]) < fabs(v0[2]) ? 0 : 2) : (fabs(v0[1]) < fabs(v0[2]) ? 1 : 2);
    double e[3] = { 0, 0, 0 };
    e[k] = 1.0;
    ax[0] = v0[1] * e[2] - v0[2] * e[1];
    ax[1] = v0[2] * e[0] - v0[0] * e[2];
    ax[2] = v0[0] * e[1] - v0[1] * e[0];
    double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (c < -1.0) c = -1.0;
    double w2 = (1.0 + c) * 0.5;
    q[0] = sqrt(w2);
    for (i = 0; i < 3; i++) q[1 + i] = ax[i] / n * sqrt(1.0 - w2);
    return;
  }
  double axis[3] = { v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0] };
  double s = sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
  q[0] = s * 0.5;
  for (i = 0; i < 3; i++) q[1 + i] = axis[i] * invs;
}

/* sensor_handlers.cpp:338-351: yaw from the mean magnetometer vector (body frame), horizontal part turned onto +y (ENU) */
void po_ins_init_yaw(const double *mag_vec_sum, int count, const double *quat_in, double *quat_out)
{
  double m_est[3] = { mag_vec_sum[0] / (double) count, mag_vec_sum[1] / (double) count, 0.0 }, qm[4];
  const double unit_y[3] = { 0.0, 1.0, 0.0 };
  po_quat_from_two_vectors(m_est, unit_y, qm);
  po_quat_mul(qm, quat_in, quat_out);
}

void po_ins_init(const double *g_vec_sum, const double *gyro_sum, int count, double max_gyro_bias, const double *quat_in,
                 double *quat_out, double *gyro_bias_est)
{
  double g_est[3], qg[4];
  const double minus_z[3] = { 0.0, 0.0, -1.0 };
  int i, too_big = 0;
  for (i = 0; i < 3; i++) {
    g_est[i] = g_vec_sum[i] / (double) count;
    gyro_bias_est[i] = gyro_sum[i] / (double) count;
    if (fabs(gyro_bias_est[i]) > max_gyro_bias) too_big = 1;
  }
  if (too_big) gyro_bias_est[0] = gyro_bias_est[1] = gyro_bias_est[2] = 0.0;
  po_quat_from_two_vectors(g_est, minus_z, qg);
  po_quat_mul(quat_in, qg, quat_out);
}
