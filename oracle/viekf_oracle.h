/*
 * viekf_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C fp64 restatement of the predict/update hot path of byu-magicc/VI-EKF,
 * in the reference's own DENSE operation order, one filter at a time.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors (test/jac_test.cpp is
 * wall-clock seeded, test/vi_ekf_test.cpp asserts nothing) and cannot be built
 * here (Eigen3, yaml-cpp, gtest and the lib/geometry + lib/multirotor_sim
 * submodules are absent).  The oracle is pinned instead by (i) the five
 * jac_test properties restated with fixed seeds (tests/test_oracle_properties.py),
 * (ii) an independent numpy restatement (oracle/np_twin.py) and (iii) committed
 * fixtures generated from this file (tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Matrices are column-major like Eigen: M(i,j) = m[i + j*ld].
 */
#ifndef VIEKF_ORACLE_H
#define VIEKF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* include/vi_ekf.h:87-111 */
enum { VO_xPOS = 0, VO_xVEL = 3, VO_xATT = 6, VO_xB_A = 10, VO_xB_G = 13, VO_xMU = 16, VO_xZ = 17 };
enum { VO_uA = 0, VO_uG = 3 };
enum { VO_dxPOS = 0, VO_dxVEL = 3, VO_dxATT = 6, VO_dxB_A = 9, VO_dxB_G = 12, VO_dxMU = 15, VO_dxZ = 16 };
/* include/vi_ekf.h:113-124 */
enum { VO_ACC = 0, VO_ALT, VO_ATT, VO_POS, VO_VEL, VO_QZETA, VO_FEAT, VO_PIXEL_VEL, VO_DEPTH, VO_INV_DEPTH, VO_TOTAL_MEAS };
/* include/vi_ekf.h:132-138 */
enum { VO_MEAS_SUCCESS = 0, VO_MEAS_GATED, VO_MEAS_NAN, VO_MEAS_INVALID, VO_MEAS_NEW_FEATURE };

typedef struct vo_filter {
  int N;            /* NUM_FEATURES  (include/vi_ekf.h:39-45, compile-time there, run-time here) */
  int nx;           /* MAX_X  = 17 + 5N  (include/vi_ekf.h:47) */
  int n;            /* MAX_DX = 16 + 3N  (include/vi_ekf.h:48) */
  int len_features;
  int next_feature_id;
  int *feature_ids; /* current_feature_ids_ */
  double *x;        /* x_[i_]  (live slot only; the history ring is host plumbing) */
  double *P;        /* P_[i_]  n x n */
  double *Qx;       /* n x n */
  double Qu[36];    /* 6 x 6 */
  double *lambda;   /* n */
  double *Lambda;   /* n x n */
  double P0_feat[9];
  int use_drag_term, use_partial_update, use_keyframe_reset;
  double min_depth;
  double cam_center[2];
  double cam_F[6];  /* 2 x 3 column-major */
  double q_b_c[4], p_b_c[3], q_b_u[4];
  /* workspace (include/vi_ekf.h:205-214) */
  double *A, *G, *dx, *K, *H, *xp, zhat[4];
  /* scratch for dense products */
  double *T1, *T2, *T3;
} vo_filter;

/* ---- quaternion / manifold helpers (src/quat.cpp, include/math_helper.h) ---- */
void vo_q_otimes(const double *a, const double *b, double *out);
void vo_q_exp(const double *v, double *out);
void vo_q_log(const double *q, double *out);
void vo_q_boxplus(const double *q, const double *d, double *out);
void vo_q_boxminus(const double *q1, const double *q2, double *out);
void vo_q_R(const double *q, double *R /*3x3 col-major*/);
void vo_q_rota(const double *q, const double *v, double *out);
void vo_q_rotp(const double *q, const double *v, double *out);
void vo_q_from_two_unit_vectors(const double *u, const double *v, double *out);
void vo_T_zeta(const double *q, double *T /*3x2 col-major*/);
void vo_q_feat_boxplus(const double *q, const double *dq, double *out);
void vo_q_feat_boxminus(const double *qj, const double *qi, double *out);

/* ---- filter ---- */
vo_filter *vo_create(int num_features);
void vo_destroy(vo_filter *f);
vo_filter *vo_clone(const vo_filter *f);
/* vi_ekf.cpp:64-99 (17-argument init) plus P0_feat / q_b_u as load() sets them (vi_ekf.cpp:134-150) */
void vo_init(vo_filter *f, const double *x0 /*17*/, const double *P0 /*16*/, const double *Qx /*16*/,
             const double *lambda /*16*/, const double *Qu /*6*/, const double *P0_feat /*3*/,
             const double *Qx_feat /*3*/, const double *lambda_feat /*3*/, const double *cam_center /*2*/,
             const double *focal_len /*2*/, const double *q_b_c /*4*/, const double *p_b_c /*3*/,
             const double *q_b_u /*4*/, double min_depth, int use_drag_term, int use_partial_update,
             int use_keyframe_reset);

void vo_boxplus(const vo_filter *f, const double *x, const double *dx, double *out);
void vo_boxminus(const vo_filter *f, const double *x1, const double *x2, double *out);
void vo_dynamics(vo_filter *f, const double *x, const double *u, int state, int jac);
/* numeric core of propagate_state, vi_ekf.cpp:262-318, with dt given (the
 * start_t_/ring/dt bookkeeping at :274-289 is host plumbing).  u is the raw IMU
 * sample: it is rotated by q_b_u here exactly as :265-267 does. */
void vo_propagate(vo_filter *f, const double *u_imu, double dt);
void vo_fix_depth(vo_filter *f);
int  vo_init_feature(vo_filter *f, const double *l, int id, double depth);
void vo_clear_feature(vo_filter *f, int id);
int  vo_global_to_local_feature_id(const vo_filter *f, int global_id);
/* measurement model table (vi_ekf_meas.cpp:281-395); H is 3 x n col-major */
void vo_h(const vo_filter *f, int type, const double *x, double *h /*4*/, double *H, int id);
/* vi_ekf_meas.cpp:196-278; z has zdim entries, R is rdim x rdim col-major */
int  vo_update(vo_filter *f, int type, const double *z, int zdim, const double *R, int rdim, int active, int id);
/* vi_ekf_kfr.cpp:56-157 (Dan's way); N (n x n) receives A_ */
void vo_keyframe_reset(vo_filter *f);
void vo_keyframe_reset_edge(vo_filter *f, double *edge /* 17 or NULL */);
/* vi_ekf_error.cpp:6-38 */
int vo_nans_in_the_house(const vo_filter *f);
int vo_blowing_up(const vo_filter *f);
int vo_negative_depth(const vo_filter *f);

/* batched convenience used by the parity tests and the cpu_baseline timing:
 * runs `steps` hot-path steps (one propagate + `M` sequential active FEAT
 * updates, slots as given) on filter f.  u: [steps][6], z: [steps][M][2],
 * slot: [M], R: 2x2.  results: [steps][M].  */
void vo_run_steps(vo_filter *f, int steps, const double *u, double dt, const double *z,
                  const int *slot, int M, const double *R, int *results);
/* same over `nf` independent filters on `threads` host threads */
void vo_run_steps_mt(vo_filter **fs, int nf, int threads, int steps, const double *u, double dt,
                     const double *z, const int *slot, int M, const double *R, int *results);

/* "structured" CPU flavour: the block-sparse propagate and the rank-2 FEAT update the HIP
 * kernels use, on the CPU -- second cpu_baseline figure of bench.py, checked against the dense flavour above */
void vo_propagate_structured(vo_filter *f, const double *u_imu, double dt);
int  vo_update_feat_structured(vo_filter *f, const double *z, const double *R, int id);
void vo_run_steps_mt_flavour(vo_filter **fs, int nf, int threads, int steps, const double *u, double dt,
                             const double *z, const int *slot, int M, const double *R, int *results, int structured);

#ifdef __cplusplus
}
#endif
#endif
