/*
 * viekf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See viekf_oracle.h for the pinning statement ("parity unpinned") and scope.
 *
 * Dense, reference-order fp64 restatement.  Matrices are column-major.
 * Citations are file:line relative to /root/reference.
 */
#include "viekf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define AT(m, ld, i, j) ((m)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

static const double E_Z[3] = {0.0, 0.0, 1.0};
static const double GRAVITY[3] = {0.0, 0.0, 9.80665}; /* include/vi_ekf.h:70-74 */

/* ------------------------------------------------------------------ small dense helpers */

/* C(m x n) = A(m x k) * B(k x n), all column-major with given leading dims */
static void mm(int m, int k, int n, const double *A, int lda, const double *B, int ldb, double *C, int ldc) {
  for (int j = 0; j < n; j++) {
    double *c = &AT(C, ldc, 0, j);
    for (int i = 0; i < m; i++) c[i] = 0.0;
    for (int p = 0; p < k; p++) {
      const double b = AT(B, ldb, p, j);
      const double *a = &AT(A, lda, 0, p);
      for (int i = 0; i < m; i++) c[i] += a[i] * b;
    }
  }
}

/* C(m x n) = A(m x k) * B(n x k)^T */
static void mmT(int m, int k, int n, const double *A, int lda, const double *B, int ldb, double *C, int ldc) {
  for (int j = 0; j < n; j++) {
    double *c = &AT(C, ldc, 0, j);
    for (int i = 0; i < m; i++) c[i] = 0.0;
    for (int p = 0; p < k; p++) {
      const double b = AT(B, ldb, j, p);
      const double *a = &AT(A, lda, 0, p);
      for (int i = 0; i < m; i++) c[i] += a[i] * b;
    }
  }
}

static void skew3(const double *v, double *S /*3x3 col-major*/) { /* src/quat.cpp:55-62 */
  AT(S, 3, 0, 0) = 0.0;   AT(S, 3, 0, 1) = -v[2]; AT(S, 3, 0, 2) = v[1];
  AT(S, 3, 1, 0) = v[2];  AT(S, 3, 1, 1) = 0.0;   AT(S, 3, 1, 2) = -v[0];
  AT(S, 3, 2, 0) = -v[1]; AT(S, 3, 2, 1) = v[0];  AT(S, 3, 2, 2) = 0.0;
}

static void cross3(const double *a, const double *b, double *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double *a) { return sqrt(dot3(a, a)); }

/* general small inverse by LU with partial pivoting (what Eigen's dynamic-size
 * .inverse() does for the rdim x rdim innovation, vi_ekf_meas.cpp:232) */
static void small_inverse(int r, const double *S, double *Sinv) {
  double a[9], b[9];
  for (int j = 0; j < r; j++)
    for (int i = 0; i < r; i++) {
      a[i + j * r] = S[i + j * r];
      b[i + j * r] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < r; c++) {
    int piv = c;
    double best = fabs(a[c + c * r]);
    for (int i = c + 1; i < r; i++)
      if (fabs(a[i + c * r]) > best) { best = fabs(a[i + c * r]); piv = i; }
    if (piv != c)
      for (int j = 0; j < r; j++) {
        double t = a[c + j * r]; a[c + j * r] = a[piv + j * r]; a[piv + j * r] = t;
        t = b[c + j * r]; b[c + j * r] = b[piv + j * r]; b[piv + j * r] = t;
      }
    for (int i = c + 1; i < r; i++) {
      double l = a[i + c * r] / a[c + c * r];
      for (int j = 0; j < r; j++) {
        a[i + j * r] -= l * a[c + j * r];
        b[i + j * r] -= l * b[c + j * r];
      }
    }
  }
  for (int j = 0; j < r; j++)
    for (int i = r - 1; i >= 0; i--) {
      double s = b[i + j * r];
      for (int k = i + 1; k < r; k++) s -= a[i + k * r] * Sinv[k + j * r];
      Sinv[i + j * r] = s / a[i + i * r];
    }
}

/* ------------------------------------------------------------------ quaternions (Hamilton, [w x y z]) */

void vo_q_otimes(const double *a, const double *b, double *out) { /* src/quat.cpp:304-312 */
  double r[4];
  r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  memcpy(out, r, sizeof r);
}

void vo_q_exp(const double *v, double *out) { /* src/quat.cpp:64-80 */
  double nv = norm3(v);
  if (nv > 1e-4) {
    double s = sin(nv / 2.0) / nv;
    out[0] = cos(nv / 2.0);
    out[1] = s * v[0]; out[2] = s * v[1]; out[3] = s * v[2];
  } else {
    double q[4] = {1.0, v[0] / 2.0, v[1] / 2.0, v[2] / 2.0};
    double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) out[i] = q[i] / nq;
  }
}

void vo_q_log(const double *q, double *out) { /* src/quat.cpp:82-98 */
  double nv = norm3(q + 1);
  if (nv < 1e-8) {
    out[0] = out[1] = out[2] = 0.0;
  } else {
    double s = 2.0 * atan2(nv, q[0]) / nv;
    out[0] = s * q[1]; out[1] = s * q[2]; out[2] = s * q[3];
  }
}

void vo_q_boxplus(const double *q, const double *d, double *out) { /* src/quat.cpp:314-317: q (x) exp(d) */
  double e[4];
  vo_q_exp(d, e);
  vo_q_otimes(q, e, out);
}

void vo_q_boxminus(const double *q1, const double *q2, double *out) { /* src/quat.cpp:319-327: log(q2^-1 (x) q1) */
  double inv[4] = {q2[0], -q2[1], -q2[2], -q2[3]};
  double dq[4];
  vo_q_otimes(inv, q1, dq);
  if (dq[0] < 0.0)
    for (int i = 0; i < 4; i++) dq[i] = -dq[i];
  vo_q_log(dq, out);
}

void vo_q_R(const double *q, double *R) { /* src/quat.cpp:226-242 (passive / R_I^b) */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double wx = w * x, wy = w * y, wz = w * z, xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
  AT(R, 3, 0, 0) = 1. - 2. * yy - 2. * zz; AT(R, 3, 0, 1) = 2. * xy + 2. * wz;      AT(R, 3, 0, 2) = 2. * xz - 2. * wy;
  AT(R, 3, 1, 0) = 2. * xy - 2. * wz;      AT(R, 3, 1, 1) = 1. - 2. * xx - 2. * zz; AT(R, 3, 1, 2) = 2. * yz + 2. * wx;
  AT(R, 3, 2, 0) = 2. * xz + 2. * wy;      AT(R, 3, 2, 1) = 2. * yz - 2. * wx;      AT(R, 3, 2, 2) = 1. - 2. * xx - 2. * yy;
}

void vo_q_rota(const double *q, const double *v, double *out) { /* src/quat.cpp:279-283  (= R^T v) */
  double t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
  cross3(q + 1, t, c);
  double r[3] = {v[0] + q[0] * t[0] + c[0], v[1] + q[0] * t[1] + c[1], v[2] + q[0] * t[2] + c[2]};
  memcpy(out, r, sizeof r);
}

void vo_q_rotp(const double *q, const double *v, double *out) { /* src/quat.cpp:286-290  (= R v) */
  double t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= -2.0; t[1] *= -2.0; t[2] *= -2.0;
  cross3(q + 1, t, c);
  double r[3] = {v[0] + q[0] * t[0] - c[0], v[1] + q[0] * t[1] - c[1], v[2] + q[0] * t[2] - c[2]};
  memcpy(out, r, sizeof r);
}

void vo_q_from_two_unit_vectors(const double *u, const double *v, double *out) { /* src/quat.cpp:167-185 */
  double d = dot3(u, v);
  if (d < 1.0) {
    double invs = 1.0 / sqrt(2.0 * (1.0 + d));
    double c[3];
    cross3(u, v, c);
    double q[4] = {0.5 / invs, c[0] * invs, c[1] * invs, c[2] * invs};
    double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) out[i] = q[i] / nq;
  } else {
    out[0] = 1.0; out[1] = out[2] = out[3] = 0.0;
  }
}

static void q_from_euler(double roll, double pitch, double yaw, double *out) { /* src/quat.cpp:150-165 */
  double cp = cos(roll / 2.0), ct = cos(pitch / 2.0), cs = cos(yaw / 2.0);
  double sp = sin(roll / 2.0), st = sin(pitch / 2.0), ss = sin(yaw / 2.0);
  out[0] = cp * ct * cs + sp * st * ss;
  out[1] = sp * ct * cs - cp * st * ss;
  out[2] = cp * st * cs + sp * ct * ss;
  out[3] = cp * ct * ss - sp * st * cs;
}
static double q_roll(const double *q) { return atan2(2.0 * (q[0] * q[1] + q[2] * q[3]), 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2])); }  /* :211-214 */
static double q_pitch(const double *q) { return asin(2.0 * (q[0] * q[2] - q[3] * q[1])); }                                          /* :216-219 */
static double q_yaw(const double *q) { return atan2(2.0 * (q[0] * q[3] + q[1] * q[2]), 1.0 - 2.0 * (q[2] * q[2] + q[3] * q[3])); } /* :221-224 */

/* include/math_helper.h:19-22 : T_zeta(q) = q.doublerota(I_2x3^T) = [rota(e_x) rota(e_y)] (src/quat.cpp:255-263) */
void vo_T_zeta(const double *q, double *T) {
  const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
  vo_q_rota(q, ex, T);
  vo_q_rota(q, ey, T + 3);
}

/* include/math_helper.h:45-48 : exp(T_zeta(q) dq) (x) q */
void vo_q_feat_boxplus(const double *q, const double *dq, double *out) {
  double T[6], v[3], e[4];
  vo_T_zeta(q, T);
  for (int i = 0; i < 3; i++) v[i] = T[i] * dq[0] + T[3 + i] * dq[1];
  vo_q_exp(v, e);
  vo_q_otimes(e, q, out);
}

/* include/math_helper.h:25-43 : q_j [-] q_i */
void vo_q_feat_boxminus(const double *qj, const double *qi, double *out) {
  double zi[3], zj[3], d[3];
  vo_q_rota(qi, E_Z, zi);
  vo_q_rota(qj, E_Z, zj);
  for (int i = 0; i < 3; i++) d[i] = zi[i] - zj[i];
  if (norm3(d) > 1e-8) {
    double s[3], T[6];
    cross3(zi, zj, s);
    double ns = norm3(s);
    for (int i = 0; i < 3; i++) s[i] /= ns;
    double theta = acos(dot3(zi, zj));
    vo_T_zeta(qi, T);
    for (int i = 0; i < 3; i++) s[i] *= theta;
    out[0] = dot3(T, s);
    out[1] = dot3(T + 3, s);
  } else {
    out[0] = out[1] = 0.0;
  }
}

/* ------------------------------------------------------------------ lifecycle */

vo_filter *vo_create(int N) {
  vo_filter *f = (vo_filter *)calloc(1, sizeof(vo_filter));
  f->N = N;
  f->nx = 17 + 5 * N; /* include/vi_ekf.h:47 */
  f->n = 16 + 3 * N;  /* include/vi_ekf.h:48 */
  size_t n = (size_t)f->n, nx = (size_t)f->nx;
  f->feature_ids = (int *)calloc((size_t)(N > 0 ? N : 1), sizeof(int));
  f->x = (double *)calloc(nx, sizeof(double));
  f->xp = (double *)calloc(nx, sizeof(double));
  f->P = (double *)calloc(n * n, sizeof(double));
  f->Qx = (double *)calloc(n * n, sizeof(double));
  f->lambda = (double *)calloc(n, sizeof(double));
  f->Lambda = (double *)calloc(n * n, sizeof(double));
  f->A = (double *)calloc(n * n, sizeof(double));
  f->G = (double *)calloc(n * 6, sizeof(double));
  f->dx = (double *)calloc(n, sizeof(double));
  f->K = (double *)calloc(n * 3, sizeof(double));
  f->H = (double *)calloc(3 * n, sizeof(double));
  f->T1 = (double *)calloc(n * n, sizeof(double));
  f->T2 = (double *)calloc(n * n, sizeof(double));
  f->T3 = (double *)calloc(n * n, sizeof(double));
  f->q_b_u[0] = 1.0;
  f->q_b_c[0] = 1.0;
  return f;
}

void vo_destroy(vo_filter *f) {
  if (!f) return;
  free(f->feature_ids); free(f->x); free(f->xp); free(f->P); free(f->Qx); free(f->lambda); free(f->Lambda);
  free(f->A); free(f->G); free(f->dx); free(f->K); free(f->H); free(f->T1); free(f->T2); free(f->T3);
  free(f);
}

vo_filter *vo_clone(const vo_filter *s) {
  vo_filter *f = vo_create(s->N);
  size_t n = (size_t)s->n, nx = (size_t)s->nx;
  int *ids = f->feature_ids;
  double *x = f->x, *xp = f->xp, *P = f->P, *Qx = f->Qx, *lam = f->lambda, *Lam = f->Lambda, *A = f->A, *G = f->G,
         *dx = f->dx, *K = f->K, *H = f->H, *T1 = f->T1, *T2 = f->T2, *T3 = f->T3;
  *f = *s;
  f->feature_ids = ids; f->x = x; f->xp = xp; f->P = P; f->Qx = Qx; f->lambda = lam; f->Lambda = Lam;
  f->A = A; f->G = G; f->dx = dx; f->K = K; f->H = H; f->T1 = T1; f->T2 = T2; f->T3 = T3;
  memcpy(f->feature_ids, s->feature_ids, sizeof(int) * (size_t)(s->N > 0 ? s->N : 1));
  memcpy(f->x, s->x, nx * sizeof(double));
  memcpy(f->xp, s->xp, nx * sizeof(double));
  memcpy(f->P, s->P, n * n * sizeof(double));
  memcpy(f->Qx, s->Qx, n * n * sizeof(double));
  memcpy(f->lambda, s->lambda, n * sizeof(double));
  memcpy(f->Lambda, s->Lambda, n * n * sizeof(double));
  memcpy(f->A, s->A, n * n * sizeof(double));
  memcpy(f->G, s->G, n * 6 * sizeof(double));
  memcpy(f->dx, s->dx, n * sizeof(double));
  memcpy(f->K, s->K, n * 3 * sizeof(double));
  memcpy(f->H, s->H, 3 * n * sizeof(double));
  return f;
}

/* vi_ekf.cpp:64-99 (init) / :134-150 (load): same population of x0,P0,Qx,lambda, per-slot feature blocks, Lambda */
void vo_init(vo_filter *f, const double *x0, const double *P0, const double *Qx, const double *lambda,
             const double *Qu, const double *P0_feat, const double *Qx_feat, const double *lambda_feat,
             const double *cam_center, const double *focal_len, const double *q_b_c, const double *p_b_c,
             const double *q_b_u, double min_depth, int use_drag_term, int use_partial_update,
             int use_keyframe_reset) {
  int n = f->n;
  memset(f->x, 0, sizeof(double) * (size_t)f->nx);
  memset(f->P, 0, sizeof(double) * (size_t)n * n);
  memset(f->Qx, 0, sizeof(double) * (size_t)n * n);
  memset(f->Qu, 0, sizeof f->Qu);
  memset(f->P0_feat, 0, sizeof f->P0_feat);
  f->len_features = 0;
  f->next_feature_id = 0;
  for (int i = 0; i < VO_xZ; i++) f->x[i] = x0[i];
  for (int i = 0; i < VO_dxZ; i++) {
    AT(f->P, n, i, i) = P0[i];
    AT(f->Qx, n, i, i) = Qx[i];
    f->lambda[i] = lambda[i];
  }
  for (int i = 0; i < 6; i++) f->Qu[i + 6 * i] = Qu[i];
  for (int k = 0; k < 3; k++) f->P0_feat[k + 3 * k] = P0_feat[k];
  for (int i = 0; i < f->N; i++) /* vi_ekf.cpp:76-81 / :139-144 : ALL slots, active or not */
    for (int k = 0; k < 3; k++) {
      int d = VO_dxZ + 3 * i + k;
      AT(f->P, n, d, d) = P0_feat[k];
      AT(f->Qx, n, d, d) = Qx_feat[k];
      f->lambda[d] = lambda_feat[k];
    }
  /* vi_ekf.cpp:83 / :146 : Lambda = 1 l^T + l 1^T - l l^T */
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) AT(f->Lambda, n, i, j) = f->lambda[j] + f->lambda[i] - f->lambda[i] * f->lambda[j];
  f->cam_center[0] = cam_center[0]; f->cam_center[1] = cam_center[1];
  memset(f->cam_F, 0, sizeof f->cam_F); /* vi_ekf.cpp:87-88 */
  AT(f->cam_F, 2, 0, 0) = focal_len[0];
  AT(f->cam_F, 2, 1, 1) = focal_len[1];
  memcpy(f->q_b_c, q_b_c, 4 * sizeof(double));
  memcpy(f->p_b_c, p_b_c, 3 * sizeof(double));
  memcpy(f->q_b_u, q_b_u, 4 * sizeof(double));
  f->min_depth = min_depth;
  f->use_drag_term = use_drag_term;
  f->use_partial_update = use_partial_update;
  f->use_keyframe_reset = use_keyframe_reset;
}

/* ------------------------------------------------------------------ manifold ops on the full state */

void vo_boxplus(const vo_filter *f, const double *x, const double *dx, double *out) { /* vi_ekf_helper.cpp:88-98 */
  for (int i = 0; i < 6; i++) out[VO_xPOS + i] = x[VO_xPOS + i] + dx[VO_dxPOS + i];
  double q[4];
  vo_q_boxplus(x + VO_xATT, dx + VO_dxATT, q);
  memcpy(out + VO_xATT, q, sizeof q);
  for (int i = 0; i < 7; i++) out[VO_xB_A + i] = x[VO_xB_A + i] + dx[VO_dxB_A + i];
  for (int i = 0; i < f->len_features; i++) {
    vo_q_feat_boxplus(x + VO_xZ + 5 * i, dx + VO_dxZ + 3 * i, q);
    memcpy(out + VO_xZ + 5 * i, q, sizeof q);
    out[VO_xZ + 5 * i + 4] = x[VO_xZ + 5 * i + 4] + dx[VO_dxZ + 3 * i + 2];
  }
}

void vo_boxminus(const vo_filter *f, const double *x1, const double *x2, double *out) { /* vi_ekf_helper.cpp:100-111 */
  for (int i = 0; i < 6; i++) out[VO_dxPOS + i] = x1[VO_xPOS + i] - x2[VO_xPOS + i];
  vo_q_boxminus(x1 + VO_xATT, x2 + VO_xATT, out + VO_dxATT);
  for (int i = 0; i < 7; i++) out[VO_dxB_A + i] = x1[VO_xB_A + i] - x2[VO_xB_A + i];
  for (int i = 0; i < f->len_features; i++) {
    vo_q_feat_boxminus(x1 + VO_xZ + 5 * i, x2 + VO_xZ + 5 * i, out + VO_dxZ + 3 * i);
    out[VO_dxZ + 3 * i + 2] = x1[VO_xZ + 5 * i + 4] - x2[VO_xZ + 5 * i + 4];
  }
}

/* ------------------------------------------------------------------ dynamics  (vi_ekf_dyn.cpp:14-135) */

static void set_block(double *M, int ld, int r0, int c0, int m, int n, const double *B /*m x n col-major*/) {
  for (int j = 0; j < n; j++)
    for (int i = 0; i < m; i++) AT(M, ld, r0 + i, c0 + j) = B[i + j * m];
}

void vo_dynamics(vo_filter *f, const double *x, const double *u, int state, int jac) {
  const int n = f->n;
  double *A = f->A, *G = f->G, *dx = f->dx;
  if (state) memset(dx, 0, sizeof(double) * (size_t)n);   /* :16-19 */
  if (jac) {                                              /* :21-25 */
    memset(A, 0, sizeof(double) * (size_t)n * n);
    memset(G, 0, sizeof(double) * (size_t)n * 6);
  }
  const double *vel = x + VO_xVEL;                        /* :27 */
  const double *q_I_b = x + VO_xATT;                      /* :28 */
  double acc[3], omega[3];
  for (int i = 0; i < 3; i++) {                           /* :30-31 */
    acc[i] = u[VO_uA + i] - x[VO_xB_A + i];
    omega[i] = u[VO_uG + i] - x[VO_xB_G + i];
  }
  const double acc_z[3] = {0.0, 0.0, acc[2]};             /* :32-33 */
  const double mu = x[VO_xMU];                            /* :34 */
  double R_I_b[9], gravity_B[3];
  vo_q_R(q_I_b, R_I_b);                                   /* :36 */
  vo_q_rotp(q_I_b, GRAVITY, gravity_B);                   /* :37 */
  const double vel_xy[3] = {vel[0], vel[1], 0.0};         /* :38-39 */
  double w_x_v[3];
  cross3(omega, vel, w_x_v);

  if (state) {                                            /* :42-50 */
    vo_q_rota(q_I_b, vel, dx + VO_dxPOS);
    for (int i = 0; i < 3; i++) {
      if (f->use_drag_term) dx[VO_dxVEL + i] = acc_z[i] + gravity_B[i] - w_x_v[i] - mu * vel_xy[i];
      else dx[VO_dxVEL + i] = acc[i] + gravity_B[i] - w_x_v[i];
      dx[VO_dxATT + i] = omega[i];
    }
  }

  double sk_vel[9], sk_omega[9], sk_g[9];
  skew3(vel, sk_vel);
  skew3(omega, sk_omega);
  skew3(gravity_B, sk_g);

  if (jac) {                                              /* :53-80 */
    double RT[9], B[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) RT[i + 3 * j] = R_I_b[j + 3 * i];
    set_block(A, n, VO_dxPOS, VO_dxVEL, 3, 3, RT);                        /* :55 */
    double nRT[9];
    for (int i = 0; i < 9; i++) nRT[i] = -RT[i];
    mm(3, 3, 3, nRT, 3, sk_vel, 3, B, 3);
    set_block(A, n, VO_dxPOS, VO_dxATT, 3, 3, B);                         /* :56 */
    if (f->use_drag_term) {
      for (int i = 0; i < 9; i++) B[i] = -sk_omega[i];                    /* :59 */
      B[0] += -mu; B[4] += -mu;
      set_block(A, n, VO_dxVEL, VO_dxVEL, 3, 3, B);
      memset(B, 0, sizeof B); B[8] = -1.0;                                /* :60 */
      set_block(A, n, VO_dxVEL, VO_dxB_A, 3, 3, B);
      for (int i = 0; i < 3; i++) AT(A, n, VO_dxVEL + i, VO_dxMU) = -vel_xy[i]; /* :61 */
    } else {
      for (int i = 0; i < 9; i++) B[i] = -sk_omega[i];                    /* :65 */
      set_block(A, n, VO_dxVEL, VO_dxVEL, 3, 3, B);
      memset(B, 0, sizeof B); B[0] = B[4] = B[8] = -1.0;                  /* :66 */
      set_block(A, n, VO_dxVEL, VO_dxB_A, 3, 3, B);
    }
    set_block(A, n, VO_dxVEL, VO_dxATT, 3, 3, sk_g);                      /* :68 */
    for (int i = 0; i < 9; i++) B[i] = -sk_vel[i];
    set_block(A, n, VO_dxVEL, VO_dxB_G, 3, 3, B);                         /* :69 */
    for (int i = 0; i < 9; i++) B[i] = -sk_omega[i];
    set_block(A, n, VO_dxATT, VO_dxATT, 3, 3, B);                         /* :70 */
    memset(B, 0, sizeof B); B[0] = B[4] = B[8] = -1.0;
    set_block(A, n, VO_dxATT, VO_dxB_G, 3, 3, B);                         /* :71 */
    if (f->use_drag_term) {                                               /* :74-77 */
      memset(B, 0, sizeof B); B[8] = -1.0;
    }
    set_block(G, n, VO_dxVEL, VO_uA, 3, 3, B);
    for (int i = 0; i < 9; i++) B[i] = -sk_vel[i];
    set_block(G, n, VO_dxVEL, VO_uG, 3, 3, B);                            /* :78 */
    memset(B, 0, sizeof B); B[0] = B[4] = B[8] = -1.0;
    set_block(G, n, VO_dxATT, VO_uG, 3, 3, B);                            /* :79 */
  }

  /* camera dynamics :83-94 */
  double w_x_p[3], tmp[3], vel_c_i[3], omega_c_i[3];
  cross3(omega, f->p_b_c, w_x_p);
  for (int i = 0; i < 3; i++) tmp[i] = vel[i] + w_x_p[i];
  vo_q_rotp(f->q_b_c, tmp, vel_c_i);
  vo_q_rotp(f->q_b_c, omega, omega_c_i);
  double sk_vel_c[9], sk_p_b_c[9], R_b_c[9];
  skew3(vel_c_i, sk_vel_c);
  skew3(f->p_b_c, sk_p_b_c);
  vo_q_R(f->q_b_c, R_b_c);

  for (int i = 0; i < f->len_features; i++) {                             /* :96-134 */
    const int xZETA = VO_xZ + 5 * i, xRHO = VO_xZ + 5 * i + 4;
    const int dZ = VO_dxZ + 3 * i, dR = VO_dxZ + 3 * i + 2;
    const double *q_zeta = x + xZETA;
    const double rho = x[xRHO];
    double zeta[3], T_z[6], sk_zeta[9];
    vo_q_rota(q_zeta, E_Z, zeta);                                         /* :105 */
    vo_T_zeta(q_zeta, T_z);                                               /* :106 */
    skew3(zeta, sk_zeta);                                                 /* :107 */
    const double rho2 = rho * rho;
    double z_x_v[3], wv[3];
    cross3(zeta, vel_c_i, z_x_v);
    for (int k = 0; k < 3; k++) wv[k] = omega_c_i[k] + rho * z_x_v[k];
    double TzT[6]; /* 2x3 col-major */
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 3; c++) TzT[r + 2 * c] = T_z[c + 3 * r];
    double nTzT[6];
    for (int k = 0; k < 6; k++) nTzT[k] = -TzT[k];

    if (state) {                                                          /* :112-116 */
      dx[dZ + 0] = nTzT[0] * wv[0] + nTzT[2] * wv[1] + nTzT[4] * wv[2];
      dx[dZ + 1] = nTzT[1] * wv[0] + nTzT[3] * wv[1] + nTzT[5] * wv[2];
      dx[dR] = rho2 * dot3(zeta, vel_c_i);
    }
    if (jac) {
      double M23a[6], M23b[6], M33a[9], M33b[9], M33c[9], M22[4], M13a[3], M13b[3];
      /* :121  -rho * T_z^T * skew_zeta * R_b_c */
      double rT[6];
      for (int k = 0; k < 6; k++) rT[k] = -rho * TzT[k];
      mm(2, 3, 3, rT, 2, sk_zeta, 3, M23a, 2);
      mm(2, 3, 3, M23a, 2, R_b_c, 3, M23b, 2);
      set_block(A, n, dZ, VO_dxVEL, 2, 3, M23b);
      /* :122  -T_z^T * (rho * skew_zeta * R_b_c * skew_p_b_c - R_b_c) */
      for (int k = 0; k < 9; k++) M33a[k] = rho * sk_zeta[k];
      mm(3, 3, 3, M33a, 3, R_b_c, 3, M33b, 3);
      mm(3, 3, 3, M33b, 3, sk_p_b_c, 3, M33c, 3);
      for (int k = 0; k < 9; k++) M33c[k] -= R_b_c[k];
      mm(2, 3, 3, nTzT, 2, M33c, 3, M23a, 2);
      set_block(A, n, dZ, VO_dxB_G, 2, 3, M23a);
      set_block(G, n, dZ, VO_uG, 2, 3, M23a);                             /* :131 */
      /* :123  -T_z^T * (skew(omega_c + rho zeta x v_c) + rho * skew_vel_c * skew_zeta) * T_z */
      skew3(wv, M33a);
      for (int k = 0; k < 9; k++) M33b[k] = rho * sk_vel_c[k];
      mm(3, 3, 3, M33b, 3, sk_zeta, 3, M33c, 3);
      for (int k = 0; k < 9; k++) M33a[k] += M33c[k];
      mm(2, 3, 3, nTzT, 2, M33a, 3, M23a, 2);
      mm(2, 3, 2, M23a, 2, T_z, 3, M22, 2);
      set_block(A, n, dZ, dZ, 2, 2, M22);
      /* :124  -T_z^T * zeta.cross(vel_c_i) */
      AT(A, n, dZ + 0, dR) = nTzT[0] * z_x_v[0] + nTzT[2] * z_x_v[1] + nTzT[4] * z_x_v[2];
      AT(A, n, dZ + 1, dR) = nTzT[1] * z_x_v[0] + nTzT[3] * z_x_v[1] + nTzT[5] * z_x_v[2];
      /* :125  rho2 * zeta^T * R_b_c */
      double rz[3] = {rho2 * zeta[0], rho2 * zeta[1], rho2 * zeta[2]};
      mm(1, 3, 3, rz, 1, R_b_c, 3, M13a, 1);
      set_block(A, n, dR, VO_dxVEL, 1, 3, M13a);
      /* :126  rho2 * zeta^T * R_b_c * skew_p_b_c */
      mm(1, 3, 3, M13a, 1, sk_p_b_c, 3, M13b, 1);
      set_block(A, n, dR, VO_dxB_G, 1, 3, M13b);
      set_block(G, n, dR, VO_uG, 1, 3, M13b);                             /* :132 */
      /* :127  rho2 * zeta^T * skew_vel_c * T_z */
      mm(1, 3, 3, rz, 1, sk_vel_c, 3, M13a, 1);
      double M12[2];
      mm(1, 3, 2, M13a, 1, T_z, 3, M12, 1);
      set_block(A, n, dR, dZ, 1, 2, M12);
      /* :128  2 * rho * zeta^T * vel_c_i */
      AT(A, n, dR, dR) = 2 * rho * dot3(zeta, vel_c_i);
    }
  }
}

/* ------------------------------------------------------------------ propagate (vi_ekf.cpp:262-318) */

void vo_fix_depth(vo_filter *f) { /* vi_ekf_helper.cpp:128-156 */
  const int n = f->n;
  for (int i = 0; i < f->len_features; i++) {
    int xR = VO_xZ + 5 * i + 4, dR = VO_dxZ + 3 * i + 2;
    if (f->x[xR] != f->x[xR]) f->x[xR] = 1.0 / (2.0 * f->min_depth);
    if (f->x[xR] < 0.0) {
      double err = 1.0 / (2.0 * f->min_depth) - f->x[xR];
      AT(f->P, n, dR, dR) += err * err;
      f->x[xR] = 1.0 / (2.0 * f->min_depth);
    } else if (f->x[xR] > 1e2) {
      AT(f->P, n, dR, dR) = f->P0_feat[8];
      f->x[xR] = 1.0 / (2.0 * f->min_depth);
    }
  }
}

void vo_propagate(vo_filter *f, const double *u_imu, double dt) {
  const int n = f->n;
  double ub[6];
  vo_q_rota(f->q_b_u, u_imu + VO_uA, ub + VO_uA); /* :265-267 */
  vo_q_rota(f->q_b_u, u_imu + VO_uG, ub + VO_uG);
  vo_dynamics(f, f->x, ub, 1, 1);                 /* :295 */
  /* :301  boxplus(x, dx*dt, x_next) */
  double *sdx = f->T3; /* first n entries used as the scaled step */
  for (int i = 0; i < n; i++) sdx[i] = f->dx[i] * dt;
  vo_boxplus(f, f->x, sdx, f->xp);
  memcpy(f->x, f->xp, sizeof(double) * (size_t)(VO_xZ + 5 * f->len_features));
  /* :302  G = (I + A*dt/2 + A*A*dt*dt/6) * G * dt */
  double *A = f->A, *A2 = f->T1, *M = f->T2;
  mm(n, n, n, A, n, A, n, A2, n);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++)
      AT(M, n, i, j) = ((i == j) ? 1.0 : 0.0) + AT(A, n, i, j) * dt / 2.0 + AT(A2, n, i, j) * dt * dt / 6.0;
  double *Gd = f->T3; /* n x 6 */
  mm(n, n, 6, M, n, f->G, n, Gd, n);
  for (int k = 0; k < n * 6; k++) f->G[k] = Gd[k] * dt;
  /* :303  A = I + A*dt + A*A*dt*dt/2 */
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++)
      AT(A, n, i, j) = ((i == j) ? 1.0 : 0.0) + AT(A, n, i, j) * dt + AT(A2, n, i, j) * dt * dt / 2.0;
  /* :304  P = A*P*A^T + G*Qu*G^T + Qx */
  double *AP = f->T1, *APAT = f->T2;
  mm(n, n, n, A, n, f->P, n, AP, n);
  mmT(n, n, n, AP, n, A, n, APAT, n);
  double *GQ = f->T1; /* n x 6 */
  mm(n, 6, 6, f->G, n, f->Qu, 6, GQ, n);
  double *GQGT = f->T3;
  mmT(n, 6, n, GQ, n, f->G, n, GQGT, n);
  for (size_t k = 0; k < (size_t)n * n; k++) f->P[k] = APAT[k] + GQGT[k] + f->Qx[k];
  vo_fix_depth(f); /* :311 */
}

/* ------------------------------------------------------------------ feature lifecycle (vi_ekf_feat.cpp) */

int vo_global_to_local_feature_id(const vo_filter *f, int global_id) { /* vi_ekf_helper.cpp:114-125 */
  for (int i = 0; i < f->len_features; i++)
    if (f->feature_ids[i] == global_id) return i;
  return -1;
}

int vo_init_feature(vo_filter *f, const double *l, int id, double depth) { /* vi_ekf_feat.cpp:6-47 */
  (void)id; /* the reference ignores the caller's id (:29-30) */
  const int n = f->n;
  if (f->len_features >= f->N) return 0;
  double lc[2] = {l[0] - f->cam_center[0], l[1] - f->cam_center[1]};
  double zeta[3] = {lc[0], lc[1] * (AT(f->cam_F, 2, 1, 1) / AT(f->cam_F, 2, 0, 0)), AT(f->cam_F, 2, 0, 0)};
  double nz = norm3(zeta);
  for (int i = 0; i < 3; i++) zeta[i] /= nz;
  double qzeta[4];
  vo_q_from_two_unit_vectors(E_Z, zeta, qzeta);
  double init_depth = depth;
  if (depth != depth) init_depth = 2.0 * f->min_depth;
  f->feature_ids[f->len_features] = f->next_feature_id;
  f->next_feature_id += 1;
  f->len_features += 1;
  int x_max = VO_xZ + 5 * f->len_features;
  memcpy(f->x + x_max - 5, qzeta, sizeof qzeta);
  f->x[x_max - 1] = 1.0 / init_depth;
  int dx_max = VO_dxZ + 3 * f->len_features;
  for (int j = 0; j < dx_max - 3; j++)
    for (int i = dx_max - 3; i < dx_max; i++) {
      AT(f->P, n, i, j) = 0.0;
      AT(f->P, n, j, i) = 0.0;
    }
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) AT(f->P, n, dx_max - 3 + i, dx_max - 3 + j) = f->P0_feat[i + 3 * j];
  return 1;
}

void vo_clear_feature(vo_filter *f, int id) { /* vi_ekf_feat.cpp:50-73 */
  const int n = f->n, nx = f->nx;
  int local = vo_global_to_local_feature_id(f, id);
  if (local < 0) return;
  int xZETA = VO_xZ + 5 * local, dxZETA = VO_dxZ + 3 * local;
  for (int i = local; i + 1 < f->len_features; i++) f->feature_ids[i] = f->feature_ids[i + 1];
  f->len_features -= 1;
  int dx_max = VO_dxZ + 3 * f->len_features;
  if (local < f->len_features) {
    memmove(f->x + xZETA, f->x + xZETA + 5, sizeof(double) * (size_t)(nx - (xZETA + 5)));
    /* rows up */
    for (int j = 0; j < n; j++)
      for (int i = dxZETA; i < n - 3; i++) AT(f->P, n, i, j) = AT(f->P, n, i + 3, j);
    /* cols left */
    for (int j = dxZETA; j < n - 3; j++)
      for (int i = 0; i < n; i++) AT(f->P, n, i, j) = AT(f->P, n, i, j + 3);
  }
  for (int i = VO_xZ + 5 * f->len_features; i < nx; i++) f->x[i] = 0.0;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++)
      if (i >= dx_max || j >= dx_max) AT(f->P, n, i, j) = 0.0;
}

/* ------------------------------------------------------------------ measurement models (vi_ekf_meas.cpp:281-395) */

void vo_h(const vo_filter *f, int type, const double *x, double *h, double *H, int id) {
  const int n = f->n;
  memset(H, 0, sizeof(double) * 3 * (size_t)n);
  switch (type) {
  case VO_ACC: { /* :281-306 */
    const double *b_a = x + VO_xB_A;
    if (f->use_drag_term) {
      const double *vel = x + VO_xVEL;
      double mu = x[VO_xMU];
      h[0] = -mu * vel[0] + b_a[0];
      h[1] = -mu * vel[1] + b_a[1];
      AT(H, 3, 0, VO_dxVEL + 0) = -mu; AT(H, 3, 1, VO_dxVEL + 1) = -mu;
      AT(H, 3, 0, VO_dxB_A + 0) = 1.0; AT(H, 3, 1, VO_dxB_A + 1) = 1.0;
      AT(H, 3, 0, VO_dxMU) = -vel[0];  AT(H, 3, 1, VO_dxMU) = -vel[1];
    } else {
      double gB[3], ng[3], S[9];
      vo_q_rotp(x + VO_xATT, GRAVITY, gB);
      for (int i = 0; i < 3; i++) { h[i] = b_a[i] - gB[i]; ng[i] = -1.0 * gB[i]; }
      skew3(ng, S);
      for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) AT(H, 3, i, VO_dxATT + j) = S[i + 3 * j];
      for (int i = 0; i < 3; i++) AT(H, 3, i, VO_dxB_A + i) = 1.0;
    }
  } break;
  case VO_ALT: /* :308-315 */
    h[0] = -x[VO_xPOS + 2];
    AT(H, 3, 0, VO_dxPOS + 2) = -1.0;
    break;
  case VO_ATT: /* :317-324 */
    for (int i = 0; i < 4; i++) h[i] = x[VO_xATT + i];
    for (int i = 0; i < 3; i++) AT(H, 3, i, VO_dxATT + i) = 1.0;
    break;
  case VO_POS: /* :326-333 */
    for (int i = 0; i < 3; i++) { h[i] = x[VO_xPOS + i]; AT(H, 3, i, VO_xPOS + i) = 1.0; }
    break;
  case VO_VEL: /* :335-342 */
    for (int i = 0; i < 3; i++) { h[i] = x[VO_xVEL + i]; AT(H, 3, i, VO_dxVEL + i) = 1.0; }
    break;
  case VO_QZETA: { /* :344-352 */
    int i = vo_global_to_local_feature_id(f, id);
    for (int k = 0; k < 4; k++) h[k] = x[VO_xZ + 5 * i + k];
    AT(H, 3, 0, VO_dxZ + 3 * i + 0) = 1.0;
    AT(H, 3, 1, VO_dxZ + 3 * i + 1) = 1.0;
  } break;
  case VO_FEAT: { /* :354-367 */
    int i = vo_global_to_local_feature_id(f, id);
    const double *q_zeta = x + VO_xZ + 5 * i;
    double zeta[3], sk[9], T_z[6];
    vo_q_rota(q_zeta, E_Z, zeta);
    skew3(zeta, sk);
    double ezT = zeta[2];
    vo_T_zeta(q_zeta, T_z);
    /* :363  cam_F * zeta / ezT + cam_center */
    double Fz[2];
    mm(2, 3, 1, f->cam_F, 2, zeta, 3, Fz, 2);
    h[0] = Fz[0] / ezT + f->cam_center[0];
    h[1] = Fz[1] / ezT + f->cam_center[1];
    /* :366  (1/ezT) * cam_F * ((zeta e_z^T)/ezT - I) * sk_zeta * T_z */
    double sF[6], M[9], M23a[6], M23b[6], M22[4];
    for (int k = 0; k < 6; k++) sF[k] = (1.0 / ezT) * f->cam_F[k];
    memset(M, 0, sizeof M);
    for (int r = 0; r < 3; r++) AT(M, 3, r, 2) = (zeta[r] * 1.0) / ezT;
    for (int r = 0; r < 3; r++) AT(M, 3, r, r) -= 1.0;
    mm(2, 3, 3, sF, 2, M, 3, M23a, 2);
    mm(2, 3, 3, M23a, 2, sk, 3, M23b, 2);
    mm(2, 3, 2, M23b, 2, T_z, 3, M22, 2);
    for (int c = 0; c < 2; c++)
      for (int r = 0; r < 2; r++) AT(H, 3, r, VO_dxZ + 3 * i + c) = M22[r + 2 * c];
  } break;
  case VO_DEPTH: { /* :369-377 */
    int i = vo_global_to_local_feature_id(f, id);
    double rho = x[VO_xZ + 5 * i + 4];
    h[0] = 1.0 / rho;
    AT(H, 3, 0, VO_dxZ + 3 * i + 2) = -1.0 / (rho * rho);
  } break;
  case VO_INV_DEPTH: { /* :379-386 */
    int i = vo_global_to_local_feature_id(f, id);
    h[0] = x[VO_xZ + 5 * i + 4];
    AT(H, 3, 0, VO_dxZ + 3 * i + 2) = 1.0;
  } break;
  default: /* PIXEL_VEL is an empty TODO in the reference (:388-395) */
    break;
  }
}

/* ------------------------------------------------------------------ update (vi_ekf_meas.cpp:196-278) */

int vo_update(vo_filter *f, int type, const double *z, int zdim, const double *R, int rdim, int active, int id) {
  const int n = f->n;
  double *H3 = f->H, *K3 = f->K;
  memset(f->zhat, 0, sizeof f->zhat);                 /* :201-203 */
  memset(K3, 0, sizeof(double) * (size_t)n * 3);
  vo_h(f, type, f->x, f->zhat, H3, id);               /* :205 (h_* also zero H) */

  double residual[4] = {0, 0, 0, 0};
  if (type == VO_QZETA) vo_q_feat_boxminus(z, f->zhat, residual);       /* :210-213 */
  else if (type == VO_ATT) vo_q_boxminus(z, f->zhat, residual);         /* :214-217 */
  else for (int i = 0; i < zdim; i++) residual[i] = z[i] - f->zhat[i];  /* :218-221 */

  if (active) { /* :230 */
    const int r = rdim;
    /* H = H_.topRows(r)  (r x n, stored with ld 3) */
    double *HP = f->T1; /* r x n, ld r */
    for (int j = 0; j < n; j++)
      for (int i = 0; i < r; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += AT(H3, 3, i, k) * AT(f->P, n, k, j);
        HP[i + j * r] = s;
      }
    double S[9], Sinv[9];
    for (int j = 0; j < r; j++)
      for (int i = 0; i < r; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += HP[i + k * r] * AT(H3, 3, j, k);
        S[i + j * r] = s + R[i + j * r];
      }
    small_inverse(r, S, Sinv);                        /* :232 */
    double t[3] = {0, 0, 0}, mahal = 0.0;             /* :234 */
    for (int j = 0; j < r; j++)
      for (int i = 0; i < r; i++) t[j] += residual[i] * Sinv[i + j * r];
    for (int j = 0; j < r; j++) mahal += t[j] * residual[j];
    if (mahal > 9.0) return VO_MEAS_GATED;            /* :235-239 */

    /* :241  K = P * H^T * innov */
    double *PHT = f->T1; /* n x r */
    for (int j = 0; j < r; j++)
      for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += AT(f->P, n, i, k) * AT(H3, 3, j, k);
        PHT[i + j * n] = s;
      }
    mm(n, r, r, PHT, n, Sinv, r, K3, n);

    int nan_free = 1;                                 /* :247 */
    for (int k = 0; k < n * 3; k++) if (K3[k] != K3[k]) nan_free = 0;
    for (int k = 0; k < n * 3; k++) if (H3[k] != H3[k]) nan_free = 0;
    if (nan_free) {
      double *dxv = f->dx; /* reuse as the correction vector */
      double *Am = f->A;
      double *Hn = f->T3;  /* r x n, ld r (compact copy of H rows) */
      for (int j = 0; j < n; j++)
        for (int i = 0; i < r; i++) Hn[i + j * r] = AT(H3, 3, i, j);
      if (f->use_partial_update) {                    /* :249-258 */
        for (int i = 0; i < n; i++) {
          double s = 0.0;
          for (int k = 0; k < r; k++) s += (f->lambda[i] * AT(K3, n, i, k)) * residual[k];
          dxv[i] = s;
        }
      } else {                                        /* :262 */
        for (int i = 0; i < n; i++) {
          double s = 0.0;
          for (int k = 0; k < r; k++) s += AT(K3, n, i, k) * residual[k];
          dxv[i] = s;
        }
      }
      vo_boxplus(f, f->x, dxv, f->xp);
      memcpy(f->x, f->xp, sizeof(double) * (size_t)(VO_xZ + 5 * f->len_features));
      /* A = I - K*H */
      mm(n, r, n, K3, n, Hn, r, Am, n);
      for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) AT(Am, n, i, j) = ((i == j) ? 1.0 : 0.0) - AT(Am, n, i, j);
      double *AP = f->T1, *APAT = f->T2;
      mm(n, n, n, Am, n, f->P, n, AP, n);
      mmT(n, n, n, AP, n, Am, n, APAT, n);
      /* K R K^T */
      double KR[3 * 1]; (void)KR;
      double *KRm = f->T1; /* n x r */
      mm(n, r, r, K3, n, R, r, KRm, n);
      double *KRKT = f->T3;
      mmT(n, r, n, KRm, n, K3, n, KRKT, n);
      if (f->use_partial_update) {                    /* :257 */
        for (size_t k = 0; k < (size_t)n * n; k++)
          f->P[k] += f->Lambda[k] * (APAT[k] + KRKT[k] - f->P[k]);
      } else {                                        /* :265 */
        for (size_t k = 0; k < (size_t)n * n; k++) f->P[k] = APAT[k] + KRKT[k];
      }
    }
  }
  vo_fix_depth(f);                                    /* :271 */
  return VO_MEAS_SUCCESS;
}

/* ------------------------------------------------------------------ keyframe reset (vi_ekf_kfr.cpp:56-157, "Dan's way") */

/* edge (optional, 17 doubles): what the reference keeps of the relative pose before it is reset (:58-62,125-126) --
 * t(3) = position, q(4) = from_euler(0,0,yaw), cov_pos(9) = P[pos,pos] column-major, cov_yaw = P(xATT+2, xATT+2).  The
 * composition into the global node pose / covariance (:147-149) uses Xformd of the absent `geometry` library and is left
 * to the caller. */
void vo_keyframe_reset_edge(vo_filter *f, double *edge) {
  const int n = f->n;
  double *q = f->x + VO_xATT;
  double yaw = q_yaw(q), roll = q_roll(q), pitch = q_pitch(q); /* :120-122 */
  if (edge) {
    for (int i = 0; i < 3; i++) edge[i] = f->x[VO_xPOS + i];   /* :61 */
    q_from_euler(0.0, 0.0, yaw, edge + 3);                     /* :125 */
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < 3; i++) edge[7 + i + 3 * j] = AT(f->P, n, VO_dxPOS + i, VO_dxPOS + j);   /* :62 */
    edge[16] = AT(f->P, n, VO_dxATT + 2, VO_dxATT + 2);        /* :126 */
  }
  for (int i = 0; i < 3; i++) f->x[VO_xPOS + i] = 0.0;         /* :65 */
  double qn[4];
  q_from_euler(roll, pitch, 0.0, qn);                          /* :129 */
  memcpy(q, qn, sizeof qn);
  double cp = cos(roll), sp = sin(roll), tt = tan(pitch);      /* :134-136 */
  double *A = f->A;
  memset(A, 0, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; i++) AT(A, n, i, i) = 1.0;
  for (int i = 0; i < 3; i++) AT(A, n, VO_dxPOS + i, VO_dxPOS + i) = 0.0; /* :138 */
  AT(A, n, VO_dxATT + 0, VO_dxATT + 0) = 1;  AT(A, n, VO_dxATT + 0, VO_dxATT + 1) = sp * tt;   AT(A, n, VO_dxATT + 0, VO_dxATT + 2) = cp * tt;
  AT(A, n, VO_dxATT + 1, VO_dxATT + 0) = 0;  AT(A, n, VO_dxATT + 1, VO_dxATT + 1) = cp * cp;   AT(A, n, VO_dxATT + 1, VO_dxATT + 2) = -cp * sp;
  AT(A, n, VO_dxATT + 2, VO_dxATT + 0) = 0;  AT(A, n, VO_dxATT + 2, VO_dxATT + 1) = -cp * sp;  AT(A, n, VO_dxATT + 2, VO_dxATT + 2) = sp * sp;
  double *AP = f->T1, *APAT = f->T2;                           /* :146 */
  mm(n, n, n, A, n, f->P, n, AP, n);
  mmT(n, n, n, AP, n, A, n, APAT, n);
  memcpy(f->P, APAT, sizeof(double) * (size_t)n * n);
}

void vo_keyframe_reset(vo_filter *f) { vo_keyframe_reset_edge(f, 0); }

/* ------------------------------------------------------------------ error predicates (vi_ekf_error.cpp:6-38) */

int vo_nans_in_the_house(const vo_filter *f) {
  int x_max = VO_xZ + 5 * f->len_features, dx_max = VO_dxZ + 3 * f->len_features;
  for (int i = 0; i < x_max; i++) if (f->x[i] != f->x[i]) return 1;
  for (int j = 0; j < dx_max; j++)
    for (int i = 0; i < dx_max; i++) if (AT(f->P, f->n, i, j) != AT(f->P, f->n, i, j)) return 1;
  return 0;
}
int vo_blowing_up(const vo_filter *f) {
  for (int i = 0; i < f->nx; i++) if (f->x[i] > 1e6) return 1;
  for (size_t k = 0; k < (size_t)f->n * f->n; k++) if (f->P[k] > 1e6) return 1;
  return 0;
}
int vo_negative_depth(const vo_filter *f) {
  for (int i = 0; i < f->len_features; i++) if (f->x[VO_xZ + 5 * i + 4] < 0) return 1;
  return 0;
}

/* ------------------------------------------------------------------ "structured" CPU flavour (cpu_baseline only)
 *
 * SURVEY.md 8(d) / BASELINE.md 3 name two CPU baselines: (i) the dense reference-order algorithm above and (ii) the
 * block-sparse / rank-2 formulation the HIP kernels use.  This is (ii): the same results up to rounding (checked against
 * the dense flavour in tests/test_oracle_structured.py), O(n^2) instead of O(n^3) per call.
 *   A = [[A_bb 0],[A_fb blkdiag(A_ff)]]  (vi_ekf_dyn.cpp:55-71,121-128)  =>  Phi, M of vi_ekf.cpp:302-303 have that shape;
 *   a FEAT H has one 2x2 block (vi_ekf_meas.cpp:366)  =>  W = P H^T is two columns of P, and for symmetric P
 *   (I-KH) P (I-KH)^T + K R K^T = P - K W^T  (vi_ekf_meas.cpp:256-257,265).
 */
static void mm3(const double *A, const double *B, double *C) { mm(3, 3, 3, A, 3, B, 3, C, 3); }

void vo_propagate_structured(vo_filter *f, const double *u_imu, double dt) {
  const int n = f->n, N = f->N, nb = 16;
  double ub[6];
  vo_q_rota(f->q_b_u, u_imu + VO_uA, ub + VO_uA);
  vo_q_rota(f->q_b_u, u_imu + VO_uG, ub + VO_uG);
  vo_dynamics(f, f->x, ub, 1, 1);
  double *sdx = f->T3;
  for (int i = 0; i < n; i++) sdx[i] = f->dx[i] * dt;
  vo_boxplus(f, f->x, sdx, f->xp);
  memcpy(f->x, f->xp, sizeof(double) * (size_t)(VO_xZ + 5 * f->len_features));
  const double *A = f->A, *G = f->G;
  /* body blocks */
  double Abb[256], Abb2[256], Phibb[256], Mbb[256];
  for (int j = 0; j < nb; j++) for (int i = 0; i < nb; i++) Abb[i + j * nb] = AT(A, n, i, j);
  mm(nb, nb, nb, Abb, nb, Abb, nb, Abb2, nb);
  for (int j = 0; j < nb; j++) for (int i = 0; i < nb; i++) {
    const double id = (i == j) ? 1.0 : 0.0;
    Phibb[i + j * nb] = id + Abb[i + j * nb] * dt + Abb2[i + j * nb] * dt * dt / 2.0;
    Mbb[i + j * nb] = id + Abb[i + j * nb] * dt / 2.0 + Abb2[i + j * nb] * dt * dt / 6.0;
  }
  /* Gd (n x 6) in T3:  body rows M_bb G_b dt,  feature rows (M_fb G_b + M_ff G_f) dt */
  double *Gd = f->T3;
  double Gb[96];
  for (int k = 0; k < 6; k++) for (int i = 0; i < nb; i++) Gb[i + k * nb] = AT(G, n, i, k);
  for (int k = 0; k < 6; k++) for (int i = 0; i < nb; i++) {
    double s = 0.0;
    for (int c = 0; c < nb; c++) s += Mbb[i + c * nb] * Gb[c + k * nb];
    AT(Gd, n, i, k) = s * dt;
  }
  /* per feature: Phi_fb (3 x 16), Phi_ff (3 x 3) kept in T1: [N][3*16 + 9] */
  double *PF = f->T1;
  for (int ft = 0; ft < N; ft++) {
    const int r0 = nb + 3 * ft;
    double Afb[48], Aff[9], Aff2[9], A2fb[48], t[48];
    for (int c = 0; c < nb; c++) for (int r = 0; r < 3; r++) Afb[r + 3 * c] = AT(A, n, r0 + r, c);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) Aff[r + 3 * c] = AT(A, n, r0 + r, r0 + c);
    mm3(Aff, Aff, Aff2);
    mm(3, nb, nb, Afb, 3, Abb, nb, A2fb, 3);
    mm(3, 3, nb, Aff, 3, Afb, 3, t, 3);
    double *Phifb = PF + (size_t)ft * 57, *Phiff = Phifb + 48;
    double Mfb[48], Mff[9];
    for (int e = 0; e < 48; e++) {
      const double a2 = A2fb[e] + t[e];
      Phifb[e] = Afb[e] * dt + a2 * dt * dt / 2.0;
      Mfb[e] = Afb[e] * dt / 2.0 + a2 * dt * dt / 6.0;
    }
    for (int e = 0; e < 9; e++) {
      const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
      Phiff[e] = id + Aff[e] * dt + Aff2[e] * dt * dt / 2.0;
      Mff[e] = id + Aff[e] * dt / 2.0 + Aff2[e] * dt * dt / 6.0;
    }
    for (int k = 0; k < 6; k++) for (int r = 0; r < 3; r++) {
      double s = 0.0;
      for (int c = 0; c < nb; c++) s += Mfb[r + 3 * c] * Gb[c + k * nb];
      for (int c = 0; c < 3; c++) s += Mff[r + 3 * c] * AT(G, n, r0 + c, k);
      AT(Gd, n, r0 + r, k) = s * dt;
    }
  }
  /* T = Phi P (n x n) in T2:  body rows Phi_bb P[b,:],  feature rows Phi_fb P[b,:] + Phi_ff P[f,:] */
  double *T = f->T2, *P = f->P;
  for (int j = 0; j < n; j++) {
    const double *pc = &AT(P, n, 0, j);
    double *tc = &AT(T, n, 0, j);
    for (int i = 0; i < nb; i++) {
      double s = 0.0;
      for (int c = 0; c < nb; c++) s += Phibb[i + c * nb] * pc[c];
      tc[i] = s;
    }
    for (int ft = 0; ft < N; ft++) {
      const double *Phifb = PF + (size_t)ft * 57, *Phiff = Phifb + 48;
      const int r0 = nb + 3 * ft;
      for (int r = 0; r < 3; r++) {
        double s = 0.0;
        for (int c = 0; c < nb; c++) s += Phifb[r + 3 * c] * pc[c];
        for (int c = 0; c < 3; c++) s += Phiff[r + 3 * c] * pc[r0 + c];
        tc[r0 + r] = s;
      }
    }
  }
  /* P+ = T Phi^T:  body columns T[:,b] Phi_bb^T,  feature columns T[:,b] Phi_fb^T + T[:,f] Phi_ff^T */
  for (int j = 0; j < nb; j++) {
    double *pc = &AT(P, n, 0, j);
    for (int i = 0; i < n; i++) pc[i] = 0.0;
    for (int c = 0; c < nb; c++) {
      const double ph = Phibb[j + c * nb];
      const double *tc = &AT(T, n, 0, c);
      for (int i = 0; i < n; i++) pc[i] += tc[i] * ph;
    }
  }
  for (int ft = 0; ft < N; ft++) {
    const double *Phifb = PF + (size_t)ft * 57, *Phiff = Phifb + 48;
    const int r0 = nb + 3 * ft;
    for (int r = 0; r < 3; r++) {
      double *pc = &AT(P, n, 0, r0 + r);
      for (int i = 0; i < n; i++) pc[i] = 0.0;
      for (int c = 0; c < nb; c++) {
        const double ph = Phifb[r + 3 * c];
        const double *tc = &AT(T, n, 0, c);
        for (int i = 0; i < n; i++) pc[i] += tc[i] * ph;
      }
      for (int c = 0; c < 3; c++) {
        const double ph = Phiff[r + 3 * c];
        const double *tc = &AT(T, n, 0, r0 + c);
        for (int i = 0; i < n; i++) pc[i] += tc[i] * ph;
      }
    }
  }
  /* + Gd Qu Gd^T + Qx */
  double *GQ = f->T1; /* n x 6 (PF is dead) */
  mm(n, 6, 6, Gd, n, f->Qu, 6, GQ, n);
  for (int j = 0; j < n; j++) {
    double *pc = &AT(P, n, 0, j);
    const double *qc = &AT(f->Qx, n, 0, j);
    for (int k = 0; k < 6; k++) {
      const double g = AT(Gd, n, j, k);
      const double *gq = &AT(GQ, n, 0, k);
      for (int i = 0; i < n; i++) pc[i] += gq[i] * g;
    }
    for (int i = 0; i < n; i++) pc[i] += qc[i];
  }
  memcpy(f->G, Gd, sizeof(double) * (size_t)n * 6);
  vo_fix_depth(f);
}

/* active FEAT update, vi_ekf_meas.cpp:196-278 in its rank-2 form */
int vo_update_feat_structured(vo_filter *f, const double *z, const double *R, int id) {
  const int n = f->n;
  double *H3 = f->H, *K3 = f->K, *P = f->P;
  memset(f->zhat, 0, sizeof f->zhat);
  vo_h(f, VO_FEAT, f->x, f->zhat, H3, id);
  const int slot = vo_global_to_local_feature_id(f, id);
  const int j0 = VO_dxZ + 3 * slot;
  const double res[2] = {z[0] - f->zhat[0], z[1] - f->zhat[1]};
  const double h00 = AT(H3, 3, 0, j0), h01 = AT(H3, 3, 0, j0 + 1), h10 = AT(H3, 3, 1, j0), h11 = AT(H3, 3, 1, j0 + 1);
  double *W = f->T1; /* n x 2 */
  const double *p0 = &AT(P, n, 0, j0), *p1 = &AT(P, n, 0, j0 + 1);
  for (int i = 0; i < n; i++) {
    W[i] = p0[i] * h00 + p1[i] * h01;
    W[i + n] = p0[i] * h10 + p1[i] * h11;
  }
  double S[4], Sinv[4];
  S[0] = h00 * W[j0] + h01 * W[j0 + 1] + R[0];
  S[1] = h10 * W[j0] + h11 * W[j0 + 1] + R[1];
  S[2] = h00 * W[j0 + n] + h01 * W[j0 + 1 + n] + R[2];
  S[3] = h10 * W[j0 + n] + h11 * W[j0 + 1 + n] + R[3];
  small_inverse(2, S, Sinv);
  const double t0 = res[0] * Sinv[0] + res[1] * Sinv[1], t1 = res[0] * Sinv[2] + res[1] * Sinv[3];
  if (t0 * res[0] + t1 * res[1] > 9.0) return VO_MEAS_GATED;
  int nan_free = 1;
  for (int i = 0; i < n; i++) {
    const double k0 = W[i] * Sinv[0] + W[i + n] * Sinv[1], k1 = W[i] * Sinv[2] + W[i + n] * Sinv[3];
    K3[i] = k0; K3[i + n] = k1;
    if (k0 != k0 || k1 != k1) nan_free = 0;
  }
  if (h00 != h00 || h01 != h01 || h10 != h10 || h11 != h11) nan_free = 0;
  if (nan_free) {
    double *dxv = f->dx;
    for (int i = 0; i < n; i++) {
      const double l = f->use_partial_update ? f->lambda[i] : 1.0;
      dxv[i] = (l * K3[i]) * res[0] + (l * K3[i + n]) * res[1];
    }
    vo_boxplus(f, f->x, dxv, f->xp);
    memcpy(f->x, f->xp, sizeof(double) * (size_t)(VO_xZ + 5 * f->len_features));
    for (int j = 0; j < n; j++) {
      const double w0 = W[j], w1 = W[j + n], lj = f->lambda[j];
      double *pc = &AT(P, n, 0, j);
      if (f->use_partial_update) {
        for (int i = 0; i < n; i++) {
          const double li = f->lambda[i];
          pc[i] -= (li + lj - li * lj) * (K3[i] * w0 + K3[i + n] * w1);
        }
      } else {
        for (int i = 0; i < n; i++) pc[i] -= K3[i] * w0 + K3[i + n] * w1;
      }
    }
  }
  vo_fix_depth(f);
  return VO_MEAS_SUCCESS;
}

/* ------------------------------------------------------------------ step drivers */


static void run_steps_flavour(vo_filter *f, int steps, const double *u, double dt, const double *z, const int *slot, int M,
                              const double *R, int *results, int structured) {
  for (int s = 0; s < steps; s++) {
    if (structured) vo_propagate_structured(f, u + 6 * (size_t)s, dt);
    else vo_propagate(f, u + 6 * (size_t)s, dt);
    for (int m = 0; m < M; m++) {
      int sl = slot[m];
      int res;
      if (sl < 0 || sl >= f->len_features) res = VO_MEAS_INVALID;
      else if (structured) res = vo_update_feat_structured(f, z + 2 * ((size_t)s * M + m), R, f->feature_ids[sl]);
      else res = vo_update(f, VO_FEAT, z + 2 * ((size_t)s * M + m), 2, R, 2, 1, f->feature_ids[sl]);
      if (results) results[(size_t)s * M + m] = res;
    }
  }
}

void vo_run_steps(vo_filter *f, int steps, const double *u, double dt, const double *z, const int *slot, int M,
                  const double *R, int *results) {
  run_steps_flavour(f, steps, u, dt, z, slot, M, R, results, 0);
}

typedef struct {
  vo_filter **fs; int lo, hi, steps, M, nf, structured;
  const double *u, *z, *R; const int *slot; double dt; int *results;
} mt_job;

static void *mt_worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  for (int k = j->lo; k < j->hi; k++) {
    /* per-filter inputs: u [nf][steps][6], z [nf][steps][M][2], slot [nf][M], results [nf][steps][M] */
    run_steps_flavour(j->fs[k], j->steps, j->u + (size_t)k * j->steps * 6, j->dt,
                      j->z + (size_t)k * j->steps * j->M * 2, j->slot + (size_t)k * j->M, j->M, j->R,
                      j->results ? j->results + (size_t)k * j->steps * j->M : NULL, j->structured);
  }
  return NULL;
}

void vo_run_steps_mt(vo_filter **fs, int nf, int threads, int steps, const double *u, double dt, const double *z,
                     const int *slot, int M, const double *R, int *results) {
  vo_run_steps_mt_flavour(fs, nf, threads, steps, u, dt, z, slot, M, R, results, 0);
}

void vo_run_steps_mt_flavour(vo_filter **fs, int nf, int threads, int steps, const double *u, double dt, const double *z,
                             const int *slot, int M, const double *R, int *results, int structured) {
  if (threads < 1) threads = 1;
  if (threads > nf) threads = nf;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)threads);
  for (int t = 0; t < threads; t++) {
    mt_job j = {fs, (int)((long)nf * t / threads), (int)((long)nf * (t + 1) / threads), steps, M, nf, structured, u, z, R, slot, dt, results};
    jobs[t] = j;
    pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}
