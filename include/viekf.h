/*
 * viekf.h -- C ABI of the MI355X-native batched VI-EKF core (libviekf_hip.so).
 *
 * The reference has no C ABI or plugin registry: its boundary is the C++ class
 * vi_ekf::VIEKF (reference include/vi_ekf.h:82-338).  Every entry point below
 * names the VIEKF member it replaces (file:line relative to the reference
 * tree).  One viekf_batch holds `batch` independent filters that share one
 * parameter set and one capacity `num_features` (the reference's compile-time
 * NUM_FEATURES, include/vi_ekf.h:39-48, is a run-time value here):
 *     nx = 17 + 5*num_features   (MAX_X)      n = 16 + 3*num_features  (MAX_DX)
 *
 * Array conventions (all caller-owned, never retained past the call):
 *   x   [batch][nx]      state, reference layout include/vi_ekf.h:87-95
 *   P   [batch][n][n]    covariance, column-major per filter like Eigen's dxMatrix
 *   u   [batch][6]       raw IMU sample (acc, gyro) -- rotated by q_b_u inside, as vi_ekf.cpp:265-267
 *   dt  [batch]          propagation interval per filter (t - t_[i_], vi_ekf.cpp:281)
 *   z   [batch][M][2]    pixel measurements;  slot [batch][M] local feature index (-1 = none)
 *   result [batch][M]    viekf_meas_result per measurement (-1 for an empty slot)
 * `where` says whether the pointers of that call are host or device memory
 * (device pointers must belong to the batch's device).  Calls on one batch must
 * be serialised by the caller (same contract as VIEKF_ROS::ekf_mtx_, reference
 * include/vi_ekf_ros.h:65); different batches are independent.
 *
 * Every function returns 0 (VIEKF_OK) or a negative viekf_status; nothing
 * throws or exits across this boundary (the reference's NAN_CHECK -> exit(0),
 * include/vi_ekf.h:29-37, becomes the per-filter status word).
 * There is NO CPU fallback: without a usable HIP device the calls fail with
 * VIEKF_ERR_NO_DEVICE.
 */
#ifndef VIEKF_H
#define VIEKF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIEKF_ABI_VERSION 1

typedef enum viekf_status {
  VIEKF_OK = 0,
  VIEKF_ERR_INVALID = -1,   /* bad argument (null pointer, size, slot out of range ...) */
  VIEKF_ERR_NO_DEVICE = -2, /* no HIP device / device index out of range */
  VIEKF_ERR_HIP = -3,       /* a HIP runtime call failed; see viekf_last_error() */
  VIEKF_ERR_YAML = -4,      /* parameter file missing / key missing / wrong length */
  VIEKF_ERR_UNSUPPORTED = -5
} viekf_status;

typedef enum viekf_mem { VIEKF_HOST = 0, VIEKF_DEVICE = 1 } viekf_mem;

/* reference include/vi_ekf.h:132-138 (same numeric values) */
typedef enum viekf_meas_result {
  VIEKF_MEAS_SKIPPED = -1, /* slot < 0: nothing to do (ragged batches) */
  VIEKF_MEAS_SUCCESS = 0,
  VIEKF_MEAS_GATED = 1,
  VIEKF_MEAS_NAN = 2,
  VIEKF_MEAS_INVALID = 3,
  VIEKF_MEAS_NEW_FEATURE = 4
} viekf_meas_result;

/* reference include/vi_ekf.h:113-124 (same numeric values) */
typedef enum viekf_meas_type {
  VIEKF_ACC = 0, VIEKF_ALT, VIEKF_ATT, VIEKF_POS, VIEKF_VEL, VIEKF_QZETA, VIEKF_FEAT, VIEKF_PIXEL_VEL,
  VIEKF_DEPTH, VIEKF_INV_DEPTH, VIEKF_TOTAL_MEAS
} viekf_meas_type;

/* per-filter status bits, mirroring the predicates of reference src/vi_ekf/vi_ekf_error.cpp:6-38 */
#define VIEKF_FLAG_NAN 1u            /* NaNsInTheHouse */
#define VIEKF_FLAG_BLOWING_UP 2u     /* BlowingUp (> 1e6) */
#define VIEKF_FLAG_NEGATIVE_DEPTH 4u /* NegativeDepth (seen before fix_depth repaired it) */
#define VIEKF_FLAG_INTERNAL 8u       /* the fused kernel's in-workgroup hand-over timed out: this filter's step is not to be trusted */

/* The keys VIEKF::load reads (reference src/vi_ekf/vi_ekf.cpp:114-131, params/ekf.yaml). */
typedef struct viekf_params {
  double x0[17];
  double P0[16];
  double Qx[16];
  double lambda[16];
  double Qu[6];          /* diagonal (get_yaml_diag) */
  double P0_feat[3];     /* diagonal */
  double Qx_feat[3];     /* diagonal */
  double lambda_feat[3];
  double cam_center[2];
  double focal_len[2];
  double q_b_c[4];
  double p_b_c[3];
  double q_b_u[4];
  double min_depth;
  double keyframe_overlap_threshold;
  int32_t use_drag_term;
  int32_t use_partial_update;
  int32_t use_keyframe_reset;
  char name[64];
} viekf_params;

typedef struct viekf_batch viekf_batch;

/* library */
int viekf_abi_version(void);
const char *viekf_last_error(void);       /* thread-local message of the last failing call */
int viekf_device_count(int32_t *count);   /* VIEKF_OK with *count==0 when no GPU is present */

/* parameters.  Replaces VIEKF::load's YAML reading, src/vi_ekf/vi_ekf.cpp:101-131 */
int viekf_params_default(viekf_params *p); /* zeros, identity quaternions, flags as params/ekf.yaml */
int viekf_params_load_yaml(const char *path, viekf_params *p);

/* construction.  Replaces VIEKF::VIEKF / init() / init(...) / the populate part of load(),
 * src/vi_ekf/vi_ekf.cpp:7-99,134-150: every filter starts at x0, diag(P0 | P0_feat...), no features. */
int viekf_batch_create(int32_t batch, int32_t num_features, const viekf_params *p, int32_t device,
                       viekf_batch **out);
int viekf_batch_destroy(viekf_batch *b);
int viekf_batch_reset(viekf_batch *b);    /* back to the state viekf_batch_create left */
int viekf_batch_dims(const viekf_batch *b, int32_t *batch, int32_t *num_features, int32_t *nx, int32_t *n);
int viekf_batch_get_params(const viekf_batch *b, viekf_params *out);   /* the parameter set the batch was created with */
/* run all later calls of this batch on a caller-owned hipStream_t (e.g. torch's current stream);
 * NULL = HIP's default (null) stream.  A new batch starts on a private non-blocking stream. */
int viekf_batch_set_stream(viekf_batch *b, void *hip_stream);
int viekf_batch_sync(viekf_batch *b);
/* Host-pointer calls normally return when their work is done.  With async_host != 0 the INPUT-ONLY host-pointer calls
 * (viekf_batch_propagate / _propagate_to, and _update_feat / _step / _step_n when `result` is NULL) copy their arguments into
 * pinned staging at call time and return as soon as the launch is queued -- the caller's arrays are free to change at once; every
 * call that hands results back to the host, and viekf_batch_sync, still wait.  The host sequencer (viekf_seq_*) drives its
 * batch this way: a replay of k propagates is k queued launches, not k round trips. */
int viekf_batch_set_async(viekf_batch *b, int32_t async_host);
/* kernel family: 0 = auto, 1 = streaming (P in HBM/L2, any num_features), 2 = resident (P on chip) */
int viekf_batch_set_kernel(viekf_batch *b, int32_t family);
/* Kernel selection knobs for tests and experiments (no reference counterpart; the defaults are what a caller gets and nothing
 * here changes results beyond rounding):
 *   VIEKF_TUNE_RES_INSTANCE  index into the resident family's instance table, -1 = automatic (VIEKF_ERR_UNSUPPORTED if that
 *                            instance does not hold num_features; the automatic choice is then restored)
 *   VIEKF_TUNE_UNIT_LAMBDA   0 = never the instances specialised for lambda_feat = [1, 1, x]
 *   VIEKF_TUNE_BLOCK_GROUP   measurements per pass of the grouped wide-P update: 0 = automatic, 16, 24, 32 (used where it fits the LDS)
 *   VIEKF_TUNE_STREAM_MFMA   0 = the streaming kernels without matrix-core passes (one pass over P per measurement), 1 = with
 *                            them (default), 2 = with them but the propagate in r02's form (operands staged in global scratch)
 *   VIEKF_TUNE_PANEL_SERVICE 0 = the grouped wide-P update with the measurement chain on every thread (r01-r03), 1 (default) = with a
 *                            service wave running it ahead of the others (k_update_feat_panelsvc)
 *   VIEKF_TUNE_TILES         the tile family of the fused step (P as fp64-MFMA accumulator tiles, N = 46..50): 0 / 1 = not used
 *                            (the default: the resident family measures faster on the MI355X at every batch size), 2 = its
 *                            one-filter-per-workgroup form, 3 = its paired form (two filters per workgroup) */
typedef enum viekf_tuning {
  VIEKF_TUNE_RES_INSTANCE = 1, VIEKF_TUNE_UNIT_LAMBDA = 2, VIEKF_TUNE_BLOCK_GROUP = 3, VIEKF_TUNE_STREAM_MFMA = 4, VIEKF_TUNE_TILES = 5,
  VIEKF_TUNE_PANEL_SERVICE = 6
} viekf_tuning;
int viekf_batch_set_tuning(viekf_batch *b, int32_t key, int32_t value);
/* which kernels a feature-update step of this batch launches, as text (for logs and benchmark records; no reference
 * counterpart): e.g. "k_step_resident<7,3> ZU: 3 worker waves x 7 blocks + 1 service wave, 2 workgroups per CU".
 * Writes at most cap bytes including the terminating 0. */
int viekf_batch_describe(const viekf_batch *b, char *out, int32_t cap);

/* state access.  Replaces get_state()/get_covariance()/get_len_features(), include/vi_ekf.h:271-286,
 * and set_x0 / set_imu_bias, src/vi_ekf/vi_ekf.cpp:157-184.  Any pointer may be NULL (skipped).  A covariance handed to
 * set_state is symmetrised, (P + P^T) / 2: the library keeps P exactly symmetric (the reference's Joseph-form update,
 * vi_ekf_meas.cpp:256-257, keeps it symmetric to rounding; the rank-2 form used here is equal to it for symmetric P only). */
int viekf_batch_get_state(viekf_batch *b, double *x, double *P, int32_t *len_features, viekf_mem where);
int viekf_batch_set_state(viekf_batch *b, const double *x, const double *P, const int32_t *len_features,
                          viekf_mem where);
int viekf_batch_get_status(viekf_batch *b, uint32_t *flags /*[batch]*/, viekf_mem where);

/* numeric core of VIEKF::propagate_state, src/vi_ekf/vi_ekf.cpp:262-318
 * (dynamics vi_ekf_dyn.cpp:14-135, boxplus vi_ekf_helper.cpp:88-98, fix_depth :128-156). */
int viekf_batch_propagate(viekf_batch *b, const double *u, const double *dt, viekf_mem where);

/* VIEKF::init_feature, src/vi_ekf/vi_ekf_feat.cpp:6-47, for the filters with mask[i] != 0
 * (mask NULL = all).  depth NaN -> 2*min_depth.  ok[i] = 1 if a slot was taken (may be NULL). */
int viekf_batch_init_feature(viekf_batch *b, const double *pix /*[batch][2]*/, const double *depth /*[batch]*/,
                             const uint8_t *mask, int32_t *ok, viekf_mem where);

/* M sequential ACTIVE FEAT updates per filter: VIEKF::update + h_feat,
 * src/vi_ekf/vi_ekf_meas.cpp:196-278,354-367, applied in array order m = 0..M-1
 * (the caller supplies the reference's reverse in-frame order, vi_ekf_meas.cpp:150-176).
 * R: 2x2 column-major; r_mode 0 = one R for all, 1 = R[batch][4], 2 = R[batch][M][4]. */
int viekf_batch_update_feat(viekf_batch *b, const double *z, const int32_t *slot, int32_t M, const double *R,
                            int32_t r_mode, int32_t *result, viekf_mem where);

/* VIEKF::keep_only_features / clear_feature, src/vi_ekf/vi_ekf_feat.cpp:50-73,81-117: every feature f < len with
 * keep[b][f] == 0 is removed, survivors (state and covariance rows/columns) move up in order, the rest is zeroed.
 * keep [batch][num_features]; new_len [batch] (may be NULL).  The keyframe-overlap test of keep_only_features
 * (:119-139) is host bookkeeping and stays with the caller. */
int viekf_batch_keep_features(viekf_batch *b, const uint8_t *keep, int32_t *new_len, viekf_mem where);

/* Keyframe reset (VIEKF::keyframe_reset, src/vi_ekf/vi_ekf_kfr.cpp:56-157): position <- 0, yaw <- 0, P <- N P N^T.
 * mask [batch] (NULL = every filter): which filters reset (the overlap test of keep_only_features, vi_ekf_feat.cpp:99-104,
 * is the caller's).  edge [batch][17] (may be NULL): the relative pose that the reference folds into its global node frame,
 * {t(3), q_yaw(4; w,x,y,z), cov_pos(9, column-major 3x3), cov_yaw}; the composition itself (:147-149) is host algebra on 7 + 36
 * numbers per filter: viekf_seq_* keeps the node pose and covariance (viekf_seq_get_global_pose / _cov below). */
int viekf_batch_keyframe_reset(viekf_batch *b, const uint8_t *mask, double *edge, viekf_mem where);

/* Read-only evaluations at the current state -- what the reference's log writer records (src/vi_ekf/vi_ekf_log.cpp:6-67):
 *   eval_xdot:    dx_ of VIEKF::dynamics for the input u [batch][6] (src/vi_ekf/vi_ekf_dyn.cpp:6-134; the batch rotates u by
 *                 q_b_u, vi_ekf.cpp:265-267), xdot [batch][n], zero past the active features;
 *   eval_h:       zhat = h(x) of measurement model `type` (src/vi_ekf/vi_ekf_meas.cpp:281-386), zhat [batch][4] (unused entries
 *                 and filters whose slot is not an active feature: NaN); slot [batch] for QZETA / FEAT / DEPTH / INV_DEPTH;
 *   get_cov_diag: diag(P) [batch][n] (what log_state writes as the covariance record).
 * None of them changes the filter. */
int viekf_batch_eval_xdot(viekf_batch *b, const double *u, double *xdot, viekf_mem where);
int viekf_batch_eval_h(viekf_batch *b, int32_t type, const int32_t *slot, double *zhat, viekf_mem where);
int viekf_batch_get_cov_diag(viekf_batch *b, double *diag, viekf_mem where);
/* VIEKF::set_drag_term / get_drag_term (include/vi_ekf.h:290-291): the drag term of the dynamics and of the accelerometer model
 * (src/vi_ekf/vi_ekf_dyn.cpp:44-49,57-67,74-77; vi_ekf_meas.cpp:281-306) is a RUN-TIME switch -- vi_ekf_ros starts with it off and turns
 * it on after take-off (src/vi_ekf_ros.cpp:86,428-429).  Takes effect for every launch after the call. */
int viekf_batch_set_drag_term(viekf_batch *b, int32_t use_drag_term);
int viekf_batch_get_drag_term(const viekf_batch *b, int32_t *use_drag_term);

/* The reference's public TEST HOOKS, evaluated on the device with the device functions the hot kernels use; dense outputs in the
 * reference's (Eigen, column-major) layouts.  None changes the filter.  x / xm / x1 / x2 [batch][nx] are explicit states (for
 * eval_jacobians and eval_H: NULL = the batch's current state); the feature loops run to the batch's current len_features.
 *   eval_jacobians   dynamics(x, u, xdot, dfdx, dfdu), src/vi_ekf/vi_ekf_dyn.cpp:5-134: u [batch][6] is the BODY-frame input as that
 *                    overload takes it (not rotated by q_b_u; propagate_state does that first, vi_ekf.cpp:265-267);
 *                    xdot [batch][n], A [batch][n][n] (n x n column-major), G [batch][6][n] (n x 6 column-major); any of the three NULL
 *   eval_H           h_<type>(x, h, H, id), src/vi_ekf/vi_ekf_meas.cpp:281-386: zhat [batch][4], H [batch][n][3] (the 3 x n hMatrix,
 *                    column-major); slot [batch] for QZETA / FEAT / DEPTH / INV_DEPTH (a slot that is not an active feature: NaN, H = 0)
 *   boxplus          boxplus(x, dx, out), src/vi_ekf/vi_ekf_helper.cpp:88-98: dx [batch][n], out [batch][nx]
 *   boxminus         boxminus(x1, x2, out), :100-111: out [batch][n]
 *   eval_reset_jacobian   keyframe_reset(xm, xp, N), src/vi_ekf/vi_ekf_kfr.cpp:6-12: xp [batch][nx], N [batch][n][n]; either NULL */
int viekf_batch_eval_jacobians(viekf_batch *b, const double *x, const double *u, double *xdot, double *A, double *G, viekf_mem where);
int viekf_batch_eval_h_jacobian(viekf_batch *b, const double *x, int32_t type, const int32_t *slot, double *zhat, double *H, viekf_mem where);
int viekf_batch_boxplus(viekf_batch *b, const double *x, const double *dx, double *out, viekf_mem where);
int viekf_batch_boxminus(viekf_batch *b, const double *x1, const double *x2, double *out, viekf_mem where);
int viekf_batch_eval_reset_jacobian(viekf_batch *b, const double *xm, double *xp, double *N, viekf_mem where);
/* a rectangular block of P of every filter, out [batch][ncols][nrows] (column-major per filter): what get_global_cov reads
 * of P_ (P_.block<3,3>(xPOS|xATT, xPOS|xATT), src/vi_ekf/vi_ekf_kfr.cpp:28-31) without moving the whole covariance */
int viekf_batch_get_cov_block(viekf_batch *b, int32_t row0, int32_t col0, int32_t nrows, int32_t ncols, double *out,
                              viekf_mem where);

/* Bounded device-side history for delayed measurements (the reference rewinds its 250-deep ring of (x,P,t),
 * include/vi_ekf.h:50,156-160, src/vi_ekf/vi_ekf_meas.cpp:45-63).  viekf_batch_history_resize allocates `depth`
 * snapshot slots of the whole batch (depth * batch * (8 n ld + 8 nx) bytes, ld = n rounded up to even for
 * num_features <= 77 and to a multiple of 16 above: choose it, the reference's 250 would be
 * 55 MB per filter at N=50); snapshot copies the live (x, P, len) into a slot, restore copies it back. */
int viekf_batch_history_resize(viekf_batch *b, int32_t depth);
int viekf_batch_snapshot(viekf_batch *b, int32_t slot);
int viekf_batch_restore(viekf_batch *b, int32_t slot);
/* Zero-copy use of the ring, as the reference uses its own (x_[i_], P_[i_] ARE the live state, include/vi_ekf.h:156-160):
 * select makes a slot the live state without copying (slot -1: back to the batch's own buffers); every later call works in
 * that slot.  propagate_to is viekf_batch_propagate whose result lands in slot dst (src/vi_ekf/vi_ekf.cpp:298-306: x_[ip],
 * P_[ip] written from x_[i_], P_[i_]) and which then selects dst: the fused kernel reads P from the old slot and stores
 * it into the new one, so keeping the history costs no extra pass over P.  The feature counts are not part of a slot. */
int viekf_batch_select(viekf_batch *b, int32_t slot);
/* Filters on INDEPENDENT clocks in one batch (one filter per rosbag / trajectory): a participation mask and per-filter ring slots.
 * set_active: mask [batch] (NULL = everybody again): filters with mask[b] == 0 take no part in the following viekf_batch_propagate /
 * _update_feat / _step / _step_n launches -- their state is not touched (the reference's filters are separate objects,
 * include/vi_ekf.h:82: one propagates while another does not).  snapshot_filters / restore_filters: the ring copies of
 * viekf_batch_snapshot / _restore with a slot PER FILTER, slot [batch], < 0 = that filter is skipped: every filter rewinds to and
 * records at its own ring position (src/vi_ekf/vi_ekf_meas.cpp:45-63 per filter).  Both need the live state in the batch's own
 * buffers (no viekf_batch_select). */
int viekf_batch_set_active(viekf_batch *b, const uint8_t *mask, viekf_mem where);
int viekf_batch_snapshot_filters(viekf_batch *b, const int32_t *slot, viekf_mem where);
int viekf_batch_restore_filters(viekf_batch *b, const int32_t *slot, viekf_mem where);
/* (propagate_to advances the WHOLE batch into dst_slot: under a participation mask it returns VIEKF_ERR_INVALID.  Ring slots
 *  handed to snapshot_filters / restore_filters in DEVICE memory cannot be checked by the host: an out-of-range one is skipped
 *  and that filter's VIEKF_FLAG_INTERNAL is raised.) */
int viekf_batch_propagate_to(viekf_batch *b, const double *u, const double *dt, int32_t dst_slot, viekf_mem where);
/* The zero-copy ring for filters on independent clocks: x_[i_], P_[i_] of EVERY filter are a slot of its own
 * (include/vi_ekf.h:156-160, one reference object per filter).  select_filters: slot [batch] in HOST memory, filter b's live state
 * becomes ring slot slot[b] without a copy (< 0: unchanged; the first call names a slot for every filter and needs the live state
 * in the batch's own buffers -- viekf_batch_snapshot_filters puts it into the ring first); every later call (updates, getters,
 * feature changes ...) works on each filter's own slot.  propagate_filters_to: the filters with dst_slot[b] >= 0 step from their
 * live slot INTO dst_slot[b], which becomes their live slot (src/vi_ekf/vi_ekf.cpp:298-306 per filter: the fused kernel loads
 * filter b from slot i_b and stores it into slot i_b + 1, no pass over P besides the step's own); the others are not touched.
 * dst_slot is HOST memory, u / dt as `where` says.  viekf_batch_history_resize(b, 0) brings every live state home again. */
int viekf_batch_select_filters(viekf_batch *b, const int32_t *slot);
int viekf_batch_propagate_filters_to(viekf_batch *b, const double *u, const double *dt, const int32_t *dst_slot, viekf_mem where);
/* K propagates in a row, step k into ring slot dst_slots[k] (all different, none of them the live slot); the last one becomes the
 * live state.  u [K][batch][6], dt [K][batch], 1 <= K <= 64.  This is the replay after a rewind (src/vi_ekf/vi_ekf_meas.cpp:106-118:
 * propagate_state(u, t, false) for every stored input).  Where the fused kernel applies it is ONE launch and only the LAST slot is
 * written -- *intermediates_written = 0: slots dst_slots[0 .. K-2] keep their old contents and the caller has to treat them as
 * unknown (viekf_seq_* re-creates such a slot from the nearest written one if a later measurement rewinds to it); otherwise the
 * steps run one by one, every slot is written and *intermediates_written = 1.  The final state is the same either way. */
int viekf_batch_propagate_n_to(viekf_batch *b, int32_t K, const double *u, const double *dt, const int32_t *dst_slots,
                               int32_t *intermediates_written, viekf_mem where);

/* ONE measurement of any model of the reference's table per filter: VIEKF::update with
 * h_acc/h_alt/h_att/h_pos/h_vel/h_qzeta/h_feat/h_depth/h_inv_depth, src/vi_ekf/vi_ekf_meas.cpp:196-386.
 * type: viekf_meas_type.  z [batch][zdim] (ATT/QZETA: quaternion, zdim 4).  R: rdim x rdim column-major,
 * r_mode 0 = shared, 1 = R[batch][rdim*rdim].  slot [batch]: local feature index for QZETA/FEAT/DEPTH/INV_DEPTH
 * (NULL otherwise).  active [batch] (NULL = all active): 0 = an inactive measurement, which only runs fix_depth as in the
 * reference (:230); 2 = the filter takes no part in this call (result VIEKF_MEAS_SKIPPED).  result [batch]: viekf_meas_result. */
int viekf_batch_update(viekf_batch *b, int32_t type, const double *z, int32_t zdim, const double *R, int32_t rdim,
                       int32_t r_mode, const int32_t *slot, const uint8_t *active, int32_t *result, viekf_mem where);

/* one hot-path step = propagate + M feature updates, fused where the kernel family allows */
int viekf_batch_step(viekf_batch *b, const double *u, const double *dt, const double *z, const int32_t *slot,
                     int32_t M, const double *R, int32_t r_mode, int32_t *result, viekf_mem where);
/* The same with K IMU samples in front of the frame's updates (250 Hz IMU, 30 Hz camera: K = 8 or 9): K x propagate_state,
 * then the M updates, in ONE launch of the fused kernel -- P stays on chip in between, bit for bit what K - 1 calls of
 * viekf_batch_propagate followed by viekf_batch_step give.  u [K][batch][6], dt [K][batch]; 1 <= K <= 64; M may be 0. */
int viekf_batch_step_n(viekf_batch *b, int32_t K, const double *u, const double *dt, const double *z, const int32_t *slot,
                       int32_t M, const double *R, int32_t r_mode, int32_t *result, viekf_mem where);

/* ---------------------------------------------------------------------------------------------------------------------
 * Host sequencer on top of a batch whose filters share ONE clock (same IMU / measurement timestamps, different data):
 * the queueing, rewind and replay logic of the reference class, batched.  All pointers are HOST memory.
 *   viekf_seq_propagate            VIEKF::propagate_state(u, t, save_input = true)   src/vi_ekf/vi_ekf.cpp:262-318
 *   viekf_seq_add_measurement      VIEKF::add_measurement                            src/vi_ekf/vi_ekf_meas.cpp:130-194
 *   viekf_seq_handle_measurements  VIEKF::handle_measurements                        src/vi_ekf/vi_ekf_meas.cpp:6-127
 *   viekf_seq_keep_only_features   VIEKF::keep_only_features (+ keyframe trigger)    src/vi_ekf/vi_ekf_feat.cpp:81-142
 *   viekf_seq_tracked_features     VIEKF::tracked_features                           src/vi_ekf/vi_ekf_feat.cpp:75-78
 *   viekf_seq_init_feature         VIEKF::init_feature                               src/vi_ekf/vi_ekf_feat.cpp:6-47
 * The state history (x, P, t) ring (include/vi_ekf.h:50,156-160; 250 deep there) is the batch's device snapshot ring with
 * `state_hist` slots; a rewind restores x and P but not the feature count, like the reference's ring.  Quirks kept: the
 * input queue stores the input already rotated by q_b_u and the replay rotates it again (vi_ekf.cpp:265-271); a feature
 * measurement with an unknown id initialises the feature at the CURRENT state and is not queued (:140-147); features are
 * numbered by the filter itself (vi_ekf_feat.cpp:29-30).
 * Lockstep: the control flow (which measurement is handled, rewind target, replay) is decided once for the whole batch from
 * the shared times.  A filter that did not queue an entry (NaN measurement, new feature) skips that update but takes part
 * in the rewind / replay of the others -- identical to an independent reference filter whenever q_b_u is the identity or
 * all filters queue the same entries. */
typedef struct viekf_seq viekf_seq;
int viekf_seq_create(viekf_batch *core, int32_t state_hist, int32_t meas_hist, viekf_seq **out);
/* INDEPENDENT CLOCKS: the filters of the batch are fed from different sources (one rosbag / trajectory each: different time
 * stamps, different camera delays) and still share the batch.  Every filter keeps its own time ring, input queue and measurement
 * queue and makes its own handle_measurements decisions (deferral, rewind target, replay length: src/vi_ekf/vi_ekf_meas.cpp:6-127
 * per filter); the device steps of a call are batched over the filters that take the same kind of step (viekf_batch_set_active,
 * viekf_batch_select_filters / _propagate_filters_to: every filter's ring is zero-copy, a rewind is an index).  Each filter's results are those of a batch of one fed the same inputs.  The
 * state history is a snapshot ring here (one copy of (x, P) per propagate, not the zero-copy ring of the shared clock).
 *   viekf_seq_propagate_t / _add_measurement_t take t [batch] and an optional mask [batch] (0 = this filter has no sample in
 *   this call); viekf_seq_propagate / _add_measurement with one t still work (same stamp for everybody); everything else
 *   (handle_measurements, keep_only_features, init_feature, global pose) is shared.  The log writer is lock-step only. */
int viekf_seq_create_independent(viekf_batch *core, int32_t state_hist, int32_t meas_hist, viekf_seq **out);
int viekf_seq_propagate_t(viekf_seq *s, const double *u /* [batch][6] */, const double *t /* [batch] */, const uint8_t *mask);
int viekf_seq_add_measurement_t(viekf_seq *s, const double *t /* [batch] */, int32_t type, const double *z, int32_t zdim,
                                const double *R, int32_t rdim, int32_t active, const int32_t *id, const double *depth,
                                const uint8_t *mask, int32_t *result);
int viekf_seq_destroy(viekf_seq *s);
int viekf_seq_propagate(viekf_seq *s, const double *u /* [batch][6] */, double t);
/* z [batch][zdim]; R rdim x rdim column-major, shared; id [batch] global feature id (NULL = -1); depth [batch] (NULL = NaN);
 * result [batch] (may be NULL): viekf_meas_result per filter */
int viekf_seq_add_measurement(viekf_seq *s, double t, int32_t type, const double *z, int32_t zdim, const double *R,
                              int32_t rdim, int32_t active, const int32_t *id, const double *depth, int32_t *result);
/* A whole camera frame in one call: the `count` FEAT entries that `count` calls of add_measurement(t, z_k, FEAT, R, active, id_k,
 * depth_k), k = 0 .. count - 1, would queue (the loops of src/vi_ekf_ros.cpp:288-306 and test/vi_ekf_test.cpp:30-31).
 * z [batch][count][2]; id [batch][count]; depth [batch][count] or NULL (NaN); R 2x2 column-major; result [batch][count] or NULL.
 * t_per_filter [batch] (independent clocks only; NULL = the one stamp t) and mask [batch] as in viekf_seq_add_measurement_t. */
int viekf_seq_add_frame(viekf_seq *s, double t, const double *t_per_filter, int32_t count, const double *z, const double *R,
                        int32_t active, const int32_t *id, const double *depth, const uint8_t *mask, int32_t *result);
/* gated_ids [batch][cap] / gated_count [batch] (both may be NULL): global ids of the FEAT measurements gated in this call */
int viekf_seq_handle_measurements(viekf_seq *s, int32_t *gated_ids, int32_t cap, int32_t *gated_count);
/* ids [batch][count] global ids to keep (pad with -1); did_reset [batch], edges [batch][17] (may be NULL): keyframe resets
 * triggered by the overlap test and their edges (viekf_batch_keyframe_reset) */
int viekf_seq_keep_only_features(viekf_seq *s, const int32_t *ids, int32_t count, uint8_t *did_reset, double *edges);
int viekf_seq_tracked_features(viekf_seq *s, int32_t *ids /* [batch][num_features] */, int32_t *len /* [batch] */);
/* VIEKF::clear_feature (src/vi_ekf/vi_ekf_feat.cpp:50-73) for every tracked feature NOT in keep_ids [batch][count] (pad with -1): the
 * removal part of keep_only_features without its keyframe-overlap test */
int viekf_seq_drop_features(viekf_seq *s, const int32_t *keep_ids, int32_t count);
/* pix [batch][2]; depth [batch] (NULL = NaN -> 2 min_depth); mask [batch] (NULL = all); ok [batch] (may be NULL): 1 if a slot was taken */
int viekf_seq_init_feature(viekf_seq *s, const double *pix, const double *depth, const uint8_t *mask, int32_t *ok);
int viekf_seq_status(viekf_seq *s, double *t_now, int32_t *ring_index, int32_t *queued, int32_t *inputs);
/* The remaining members of the reference class that its callers use (include/vi_ekf.h:271-292,302,324):
 *   viekf_seq_propagate_state   propagate_state(u, t, save_input)  (save_input = 0: the input is not recorded in the input queue
 *                               and the step is not logged, src/vi_ekf/vi_ekf.cpp:269-272,316-317; shared clock only)
 *   viekf_seq_set_x0            set_x0: x_[i_].topRows(17) = x0 [batch][17]                      vi_ekf.cpp:157-160
 *   viekf_seq_set_imu_bias      set_imu_bias(b_g, b_a), [batch][3] each, into the LIVE state    vi_ekf.cpp:179-183
 *   viekf_seq_keyframe_reset    keyframe_reset(): explicit reset of the filters with mask[b] != 0 (NULL = all) including the node
 *                               frame's move (vi_ekf_kfr.cpp:56-157; vi_ekf_ros calls it right after set_x0, src/vi_ekf_ros.cpp:401-402);
 *                               edges [batch][17] as viekf_batch_keyframe_reset, may be NULL.  The callback of
 *                               register_keyframe_reset_callback is the binding's (include/viekf_shim.hpp).
 *   viekf_seq_get_features      get_depths / get_zetas / get_qzetas / get_zeta                  vi_ekf.cpp:210-246
 *                               depths [batch][N], zetas [batch][N][3], qzetas [batch][N][4]; NaN past a filter's features; any NULL
 *   viekf_seq_get_feat          get_feat(id) / get_depth(id) by global feature id [batch]       vi_ekf.cpp:248-260
 *                               pix [batch][2], depth [batch]; NaN for an id the filter does not track; either may be NULL */
int viekf_seq_propagate_state(viekf_seq *s, const double *u, double t, int32_t save_input);
int viekf_seq_set_x0(viekf_seq *s, const double *x0);
int viekf_seq_set_imu_bias(viekf_seq *s, const double *b_g, const double *b_a);
int viekf_seq_keyframe_reset(viekf_seq *s, const uint8_t *mask, double *edges);
int viekf_seq_get_features(viekf_seq *s, double *depths, double *zetas, double *qzetas);
int viekf_seq_get_feat(viekf_seq *s, const int32_t *id, double *pix, double *depth);

/* Global pose and covariance (relative navigation: the filter state is relative to the last keyframe node).
 *   viekf_seq_get_global_pose   VIEKF::get_global_pose, get_current_node_global_pose   src/vi_ekf/vi_ekf_kfr.cpp:14-21, vi_ekf.cpp:192-195
 *   viekf_seq_get_global_cov    VIEKF::get_global_cov, propagate_global_covariance      src/vi_ekf/vi_ekf_kfr.cpp:23-53
 * and the node update at the end of keyframe_reset (:147-150) happens inside viekf_seq_keep_only_features whenever it resets.
 * SE(3) convention (the reference's Xformd lives in its absent `geometry` submodule; this is the one used here and restated in
 * oracle/seq_oracle.py): a transform is {t(3), q(4; w,x,y,z)} with q the PASSIVE rotation of src/quat.cpp (rota(v) = R(q)^T v);
 *   composition  T1 * T2 = { t1 + q1.rota(t2),  q1 (x) q2 };     Adj(T) = [ R  [t]x R ; 0  R ],  R = q.R()   (6x6, [pos; att])
 *   get_global_pose = node * {x[POS], x[ATT]};     edge covariance = diag-blocks {P[POS,POS], 0 ..., P(ATT+2, ATT+2)} (:59-63,126)
 *   node covariance += Adj(node)^T cov_edge Adj(node);  node = node * edge    (in that order, :149-150)
 *   get_global_cov = node covariance + Adj(node)^T C Adj(node),  C = the [POS,ATT] x [POS,ATT] blocks of P.
 * pose [batch][7] = {t, q}; node [batch][7] (may be NULL): the current node's global pose; cov [batch][36] column-major 6x6. */
int viekf_seq_get_global_pose(viekf_seq *s, double *pose, double *node);
int viekf_seq_get_global_cov(viekf_seq *s, double *cov);

/* VIEKF::init_logger / disable_logger, src/vi_ekf/vi_ekf_log.cpp:69-117: opens <root><name>_{ACC,...,INV_DEPTH,state,cov,feat_id,
 * input,xdot,kf,global_pose}.log and _config.txt / _debug.txt and records filter `filter` of the batch from then on -- the binary
 * record layouts of log_state (:6-35) and log_measurement (:52-67), i.e. what matlab/plot_ekf.m reads.  The global_pose record
 * is get_global_pose (node * relative pose), as :33-34 writes it. */
int viekf_seq_init_logger(viekf_seq *s, const char *root_filename, const char *ekf_name, int32_t filter);
int viekf_seq_disable_logger(viekf_seq *s);
#ifdef __cplusplus
}
#endif
#endif /* VIEKF_H */
