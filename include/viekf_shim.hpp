// viekf_shim.hpp -- the reference's C++ surface over the C ABI (include/viekf.h), for callers written against
// vi_ekf::VIEKF (reference include/vi_ekf.h:82-338): same method names, argument order, defaults and result codes.
// ONE filter = a batch of one behind the host sequencer (viekf_seq_*): measurements are queued and whole frames are
// forwarded by handle_measurements(), the state is read back only when the caller asks for it -- never once per update.
//
// Argument types.  The reference's callers pass Eigen objects (uVector, VectorXd, Vector2d, MatrixXd ...: src/vi_ekf_ros.cpp:174-198,
// 282-309,453-470, test/vi_ekf_test.cpp:24-33).  Every vector / matrix argument here is a TEMPLATE on "anything with .data() and
// .size()" (and .rows() for a noise matrix; column-major storage, Eigen's default) -- Eigen's fixed and dynamic types, std::vector,
// std::array and the 20-line mock vector of tests/cpp/shim_callsites.cpp all qualify, and no Eigen header is needed to compile
// this file (Eigen is absent from the build image).  Raw `const double*` overloads remain for C-style callers.
// Return types.  get_state() / get_covariance() / get_depths() ... return shim::Vec / shim::Mat: contiguous column-major storage
// with .data() .size() .rows() .cols() operator() operator[] topRows() segment() block() diagonal() -- what the call sites above
// apply to the reference's `const xVector&` / `const dxMatrix&`; an Eigen caller maps them (Eigen::Map<const VectorXd>(v.data(),
// v.size())) where it needs Eigen expressions beyond these.
//
//   reference member (include/vi_ekf.h)                      line      here
//   VIEKF(), VIEKF(param_file), load(param_file)             243-249   same (+ num_features, device: run-time here)
//   init() / init(x0, P0, ... 17 arguments)                  250-255   same: the 17 values fill a viekf_params (q_b_u = identity, as
//                                                                      vi_ekf.cpp:64-99 leaves it) and create the filter -- the fixture
//                                                                      of test/jac_test.cpp:118-170
//   now()                                                    257-261   same
//   measurement_functions (public table of h_* pointers)     263       same: measurement_function_ptr (:68) over the typedef names
//                                                                      xVector / zVector / hMatrix, filled in the order of
//                                                                      vi_ekf.cpp:47-61 (h_pixel_vel: the reference's empty TODO)
//   NaNsInTheHouse / BlowingUp / NegativeDepth               266-268   from the per-filter status word
//   global_to_local_feature_id, tracked_features             271-272   same
//   get_depths / get_zetas / get_qzetas / get_zeta           275-278   same (Vec / Mat)
//   get_feat(id) / get_depth(id)                             279,284   same
//   get_current_node_global_pose / get_state / get_covariance / get_covariance_diagonal / get_len_features   280-285  same
//   set_x0 / set_imu_bias / set_drag_term / get_drag_term / get_keyframe_reset   288-292  same (set_drag_term is a run-time switch)
//   init_feature / clear_feature / keep_only_features        294-296   same (clear_feature(id) = keep the others)
//   boxplus / boxminus                                       299-300   same, evaluated on the device
//   step(u, t)                                               301       not offered: declared but never defined in the reference
//   propagate_state(u, t, save_input = true)                 302       same
//   dynamics(x, u, xdot, dfdx, dfdu)                         303       same, evaluated on the device (dense outputs)
//   dynamics(x, u, state, jac)                               304       not offered: fills the private A_, G_, dx_ members
//   handle_measurements / add_measurement                    307-308   same
//   update(measurement_t&)                                   309       not offered: its argument type is private to the class;
//                                                                      measurements enter through add_measurement
//   h_acc ... h_inv_depth (x, h, H, id)                      310-318   same names, evaluated on the device; the feature models map
//                                                                      the GLOBAL id to the slot like vi_ekf_meas.cpp:346,356,371,381;
//                                                                      h_pixel_vel (:319) does nothing, like the reference's TODO
//   propagate_global_covariance                              322       inside get_global_cov / keyframe_reset (viekf_seq.cpp)
//   keyframe_reset(xm, xp, N) / keyframe_reset()             323-324   same
//   register_keyframe_reset_callback                         325       same
//   get_global_pose / get_global_cov                         326-327   {t(3), q(4)} / 6x6 (Xformd lives in the absent geometry submodule)
//   log_state / log_measurement                              330-331   not offered as calls: the sequencer writes the same records
//                                                                      itself (viekf_seq_init_logger)
//   init_logger / disable_logger                             332-333   same files (vi_ekf_log.cpp:79-117)
//   log_global_position                                      334       accepted and ignored: it writes to LOG_GLOBAL, a stream the
//                                                                      reference never opens (vi_ekf_log.cpp:85-97)
//   fix_depth()                                              337       not offered: runs inside every propagate / update on the device
// The typedef names of include/vi_ekf.h:53-61 (xVector, dxVector, dxMatrix, dxuMatrix, uVector, zVector, hMatrix) are shim::Vec /
// shim::Mat here; a caller that has its own (Eigen's fixed-size matrices, the mock of tests/cpp/shim_jactest_callsites.cpp) defines
// them in namespace vi_ekf and #defines VIEKF_SHIM_TYPES before including this file.  Every OUTPUT argument is a template as
// well: it is resized where it has resize(rows, cols) and must otherwise already have the right number of coefficients;
// get_state() / get_covariance() return `const xVector&` / `const dxMatrix&` of whichever types those names stand for.
// The class is movable, not copyable (test/jac_test.cpp:118,169 returns its filter by value: a move).
// No exceptions cross this class, like the reference: a failing call prints to std::cerr and ok() turns false.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <functional>
#include <iostream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "viekf.h"

#ifndef NUM_FEATURES
#define NUM_FEATURES 12   // reference include/vi_ekf.h:39-45 (release default)
#endif

namespace vi_ekf {

namespace shim {
// contiguous column-major storage with the handful of accessors the reference's call sites use on xVector / dxMatrix / VectorXd
struct Mat {
  std::vector<double> d;
  int r = 0, c = 0;
  Mat() {}
  Mat(int rows, int cols) : d((size_t)rows * cols, 0.0), r(rows), c(cols) {}
  void resize(int rows, int cols) { d.assign((size_t)rows * cols, 0.0); r = rows; c = cols; }
  const double* data() const { return d.data(); }
  double* data() { return d.data(); }
  long size() const { return (long)d.size(); }
  int rows() const { return r; }
  int cols() const { return c; }
  double operator()(int i, int j) const { return d[(size_t)i + (size_t)j * r]; }
  double& operator()(int i, int j) { return d[(size_t)i + (size_t)j * r]; }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat o(nr, nc);
    for (int j = 0; j < nc; j++)
      for (int i = 0; i < nr; i++) o(i, j) = (*this)(i0 + i, j0 + j);
    return o;
  }
  Mat diagonal() const {
    Mat o(r < c ? r : c, 1);
    for (int i = 0; i < o.r; i++) o.d[(size_t)i] = (*this)(i, i);
    return o;
  }
};
struct Vec : Mat {
  Vec() {}
  explicit Vec(int n) : Mat(n, 1) {}
  double operator()(int i) const { return d[(size_t)i]; }
  double& operator()(int i) { return d[(size_t)i]; }
  double operator[](int i) const { return d[(size_t)i]; }
  double& operator[](int i) { return d[(size_t)i]; }
  Vec segment(int i0, int n) const {
    Vec o(n);
    for (int i = 0; i < n; i++) o.d[(size_t)i] = d[(size_t)(i0 + i)];
    return o;
  }
  Vec topRows(int n) const { return segment(0, n); }
};
// an output argument is brought to rows x cols where its type can be resized; fixed-size types are left alone
template <class T> auto size_to(T& o, int r, int c, int) -> decltype(o.resize(r, c), void()) { o.resize(r, c); }
template <class T> void size_to(T&, int, int, long) {}
}  // namespace shim

#ifndef VIEKF_SHIM_TYPES   // reference include/vi_ekf.h:53-61
typedef shim::Vec xVector;
typedef shim::Vec dxVector;
typedef shim::Mat dxMatrix;
typedef shim::Mat dxuMatrix;
typedef shim::Vec uVector;
typedef shim::Vec zVector;
typedef shim::Mat hMatrix;
#endif
class VIEKF;
typedef void (VIEKF::*measurement_function_ptr)(const xVector& x, zVector& h, hMatrix& H, const int id) const;   // :68

class VIEKF {
 public:
  // reference include/vi_ekf.h:113-138 (same values as viekf_meas_type / viekf_meas_result)
  typedef enum { ACC, ALT, ATT, POS, VEL, QZETA, FEAT, PIXEL_VEL, DEPTH, INV_DEPTH, TOTAL_MEAS } measurement_type_t;
  typedef enum { MEAS_SUCCESS, MEAS_GATED, MEAS_NAN, MEAS_INVALID, MEAS_NEW_FEATURE } meas_result_t;
  enum : int { xPOS = 0, xVEL = 3, xATT = 6, xB_A = 10, xB_G = 13, xMU = 16, xZ = 17 };          // :87-95
  enum : int { dxPOS = 0, dxVEL = 3, dxATT = 6, dxB_A = 9, dxB_G = 12, dxMU = 15, dxZ = 16 };    // :103-111
  enum : int { uA = 0, uG = 3, uTOTAL = 6 };                                                     // :97-101
  typedef shim::Vec Vec;
  typedef shim::Mat Mat;

  VIEKF() { fill_table(); }
  explicit VIEKF(const std::string& param_file, int num_features = NUM_FEATURES, int device = 0)
      : num_features_(num_features), device_(device) { fill_table(); load(param_file); }
  VIEKF(const VIEKF&) = delete;
  VIEKF& operator=(const VIEKF&) = delete;
  VIEKF(VIEKF&& o) noexcept { fill_table(); take(o); }
  VIEKF& operator=(VIEKF&& o) noexcept { if (this != &o) { release(); take(o); } return *this; }
  ~VIEKF() { release(); }

  std::vector<measurement_function_ptr> measurement_functions;             // include/vi_ekf.h:263, filled as vi_ekf.cpp:47-61

  void init() {}                                                          // :250 (the constructor's reset: nothing is held before load / init)
  // init(x0, P0, ...), vi_ekf.cpp:64-99: the filter from 17 values instead of a parameter file (test/jac_test.cpp:154)
  template <class X0, class PV, class QV, class LV, class QU, class F1, class F2, class F3, class C2, class FL, class QC, class PC>
  void init(const X0& x0, const PV& P0, const QV& Qx, const LV& lambda, const QU& Qu, const F1& P0_feat, const F2& Qx_feat,
            const F3& lambda_feat, const C2& cam_center, const FL& focal_len, const QC& q_b_c, const PC& p_b_c, double min_depth,
            bool use_drag_term, bool use_partial_update, bool use_keyframe_reset, double keyframe_overlap_threshold) {
    release();
    viekf_params p;
    viekf_params_default(&p);                                             // (q_b_u stays the identity: init() never sets q_b_u_)
    auto put = [](double* dst, const double* src, int k) { for (int i = 0; i < k; i++) dst[i] = src[i]; };
    put(p.x0, x0.data(), 17); put(p.P0, P0.data(), 16); put(p.Qx, Qx.data(), 16); put(p.lambda, lambda.data(), 16);
    put(p.Qu, Qu.data(), 6); put(p.P0_feat, P0_feat.data(), 3); put(p.Qx_feat, Qx_feat.data(), 3); put(p.lambda_feat, lambda_feat.data(), 3);
    put(p.cam_center, cam_center.data(), 2); put(p.focal_len, focal_len.data(), 2); put(p.q_b_c, q_b_c.data(), 4); put(p.p_b_c, p_b_c.data(), 3);
    p.min_depth = min_depth;
    p.use_drag_term = use_drag_term ? 1 : 0;
    p.use_partial_update = use_partial_update ? 1 : 0;
    p.use_keyframe_reset = use_keyframe_reset ? 1 : 0;
    p.keyframe_overlap_threshold = keyframe_overlap_threshold;
    start(p);
  }

  void load(const std::string& param_file) {                              // vi_ekf.cpp:101-155
    release();
    viekf_params p;
    if (!check(viekf_params_load_yaml(param_file.c_str(), &p), "load")) return;
    start(p);
  }
  bool ok() const { return ok_; }
  int max_x() const { return nx_; }
  int max_dx() const { return n_; }
  inline double now() const {                                             // :257-261
    return (double)std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::high_resolution_clock::now().time_since_epoch()).count() * 1e-6;
  }

  // ---- state propagation -------------------------------------------------------------------------------------------------
  void propagate_state(const double* u, const double t, bool save_input = true) {   // vi_ekf.cpp:262-318
    if (ok_) check(viekf_seq_propagate_state(seq_, u, t, save_input ? 1 : 0), "propagate_state");
  }
  template <class U, class = decltype(std::declval<const U&>().data())>
  void propagate_state(const U& u, const double t, bool save_input = true) { propagate_state(u.data(), t, save_input); }

  // ---- measurements ------------------------------------------------------------------------------------------------------
  meas_result_t add_measurement(const double t, const double* z, int zdim, const measurement_type_t& meas_type, const double* R,
                                int rdim, bool active = false, const int id = -1, const double depth = NAN) {   // vi_ekf_meas.cpp:130-194
    if (!ok_) return MEAS_INVALID;
    int32_t res = MEAS_INVALID, gid = id;
    check(viekf_seq_add_measurement(seq_, t, (int32_t)meas_type, z, zdim, R, rdim, active ? 1 : 0, &gid, &depth, &res), "add_measurement");
    return (meas_result_t)res;
  }
  // z: anything with data() / size(); R: anything with data() and rows() (Eigen: `const VectorXd& z, ..., const MatrixXd& R`, :308).
  // As in the reference the measurement's dimension is the noise matrix's (meas.rdim = R.rows(), vi_ekf_meas.cpp:160-166).
  template <class Z, class RM, class = decltype(std::declval<const Z&>().data()), class = decltype(std::declval<const RM&>().rows())>
  meas_result_t add_measurement(const double t, const Z& z, const measurement_type_t& meas_type, const RM& R, bool active = false,
                                const int id = -1, const double depth = NAN) {
    return add_measurement(t, z.data(), (int)z.size(), meas_type, R.data(), (int)R.rows(), active, id, depth);
  }
  // (a std::vector noise matrix -- no rows(): square, column-major)
  meas_result_t add_measurement(const double t, const std::vector<double>& z, const measurement_type_t& meas_type,
                                const std::vector<double>& R, bool active = false, const int id = -1, const double depth = NAN) {
    int rdim = 1;
    while (rdim * rdim < (int)R.size()) rdim++;
    return add_measurement(t, z.data(), (int)z.size(), meas_type, R.data(), rdim, active, id, depth);
  }

  void handle_measurements(std::vector<int>* gated_feature_ids = nullptr) {   // vi_ekf_meas.cpp:6-127
    if (!ok_) return;
    std::vector<int32_t> ids((size_t)num_features_ + 8, -1);
    int32_t cnt = 0;
    if (!check(viekf_seq_handle_measurements(seq_, ids.data(), (int32_t)ids.size(), &cnt), "handle_measurements")) return;
    if (gated_feature_ids)
      for (int k = 0; k < cnt && k < (int)ids.size(); k++) gated_feature_ids->push_back(ids[k]);
  }

  // ---- features ----------------------------------------------------------------------------------------------------------
  bool init_feature(const double* l, const int id, const double depth = -1.0) {   // vi_ekf_feat.cpp:6-47
    (void)id;   // the reference pushes its own counter (:29-30)
    if (!ok_) return false;
    int32_t okf = 0;
    check(viekf_seq_init_feature(seq_, l, &depth, nullptr, &okf), "init_feature");
    return okf != 0;
  }
  template <class L, class = decltype(std::declval<const L&>().data())>
  bool init_feature(const L& l, const int id, const double depth = -1.0) { return init_feature(l.data(), id, depth); }

  void keep_only_features(const std::vector<int> features) {              // vi_ekf_feat.cpp:81-142
    if (!ok_) return;
    std::vector<int32_t> ids(features.begin(), features.end());
    if (ids.empty()) ids.push_back(-1);
    uint8_t did = 0;
    if (!check(viekf_seq_keep_only_features(seq_, ids.data(), (int32_t)ids.size(), &did, nullptr), "keep_only_features")) return;
    if (did && keyframe_reset_callback_) keyframe_reset_callback_();      // vi_ekf_kfr.cpp:154-156
  }
  void clear_feature(const int id) {                                      // vi_ekf_feat.cpp:50-73: every other tracked feature stays
    std::vector<int> keep;
    for (int gid : tracked_features())
      if (gid != id) keep.push_back(gid);
    if (!ok_) return;
    // (clear_feature does not run the keyframe-overlap test of keep_only_features, :119-139: the batch call is used directly)
    std::vector<int32_t> ids((size_t)num_features_, -1);
    int32_t len = 0;
    if (!check(viekf_seq_tracked_features(seq_, ids.data(), &len), "clear_feature")) return;
    std::vector<int32_t> k32(keep.begin(), keep.end());
    if (k32.empty()) k32.push_back(-1);
    check(viekf_seq_drop_features(seq_, k32.data(), (int32_t)k32.size()), "clear_feature");
  }
  int global_to_local_feature_id(const int global_id) const {             // vi_ekf_helper.cpp:114-125
    const std::vector<int>& tr = tracked_features();
    for (size_t i = 0; i < tr.size(); i++)
      if (tr[i] == global_id) return (int)i;
    return -1;
  }
  const std::vector<int>& tracked_features() const {                      // :272
    std::vector<int32_t> ids((size_t)num_features_, -1);
    int32_t len = 0;
    tracked_.clear();
    if (ok_ && check(viekf_seq_tracked_features(seq_, ids.data(), &len), "tracked_features"))
      tracked_.assign(ids.begin(), ids.begin() + len);
    return tracked_;
  }

  // ---- keyframe reset ----------------------------------------------------------------------------------------------------
  void keyframe_reset() {                                                 // vi_ekf_kfr.cpp:56-157
    if (!ok_) return;
    if (!check(viekf_seq_keyframe_reset(seq_, nullptr, nullptr), "keyframe_reset")) return;
    if (keyframe_reset_callback_) keyframe_reset_callback_();             // :154-156
  }
  template <class X, class OX, class ON, class = decltype(std::declval<const X&>().data())>
  void keyframe_reset(const X& xm, OX& xp, ON& N) {                       // :6-12 (test hook: the filter itself is left alone)
    if (fits(xp, nx_, 1, "keyframe_reset") && fits(N, n_, n_, "keyframe_reset"))
      check(viekf_batch_eval_reset_jacobian(core_, xm.data(), xp.data(), N.data(), VIEKF_HOST), "keyframe_reset(xm, xp, N)");
  }
  void register_keyframe_reset_callback(std::function<void(void)> cb) { keyframe_reset_callback_ = cb; }   // :325

  // ---- getters and setters -----------------------------------------------------------------------------------------------
  const xVector& get_state() {                                            // :281 (the caller's xVector under VIEKF_SHIM_TYPES)
    if (fits(x_, nx_, 1, "get_state")) check(viekf_batch_get_state(core_, x_.data(), nullptr, nullptr, VIEKF_HOST), "get_state");
    return x_;
  }
  const dxMatrix& get_covariance() {                                      // :282 (MAX_DX x MAX_DX)
    if (fits(P_, n_, n_, "get_covariance")) check(viekf_batch_get_state(core_, nullptr, P_.data(), nullptr, VIEKF_HOST), "get_covariance");
    return P_;
  }
  Vec get_covariance_diagonal() {                                         // :283
    Vec d(n_);
    if (ok_) check(viekf_batch_get_cov_diag(core_, d.data(), VIEKF_HOST), "get_covariance_diagonal");
    return d;
  }
  int get_len_features() {                                                // :285
    int32_t len = 0;
    if (ok_) check(viekf_batch_get_state(core_, nullptr, nullptr, &len, VIEKF_HOST), "get_len_features");
    return len;
  }
  Vec get_depths() {                                                      // :275, vi_ekf.cpp:210-218 (len_features entries)
    const int len = get_len_features();
    Vec all(num_features_), o(len);
    if (ok_) check(viekf_seq_get_features(seq_, all.data(), nullptr, nullptr), "get_depths");
    for (int i = 0; i < len; i++) o[i] = all[i];
    return o;
  }
  Mat get_zetas() { return feat_matrix(3); }                              // :276, vi_ekf.cpp:220-229 (3 x len_features)
  Mat get_qzetas() { return feat_matrix(4); }                             // :277, vi_ekf.cpp:231-239 (4 x len_features)
  Vec get_zeta(const int i) {                                             // :278, vi_ekf.cpp:241-245
    Mat z = feat_matrix(3, true);
    Vec o(3);
    for (int k = 0; k < 3; k++) o[k] = (i >= 0 && i < z.cols()) ? z(k, i) : NAN;
    return o;
  }
  Vec get_feat(const int id) {                                            // :279, vi_ekf.cpp:253-260
    Vec o(2);
    int32_t gid = id;
    if (ok_) check(viekf_seq_get_feat(seq_, &gid, o.data(), nullptr), "get_feat");
    return o;
  }
  double get_depth(const int id) {                                        // :284, vi_ekf.cpp:247-251
    double d = NAN;
    int32_t gid = id;
    if (ok_) check(viekf_seq_get_feat(seq_, &gid, nullptr, &d), "get_depth");
    return d;
  }
  template <class X, class = decltype(std::declval<const X&>().data())>
  void set_x0(const X& x0) { if (ok_) check(viekf_seq_set_x0(seq_, x0.data()), "set_x0"); }             // :288, vi_ekf.cpp:157-160
  template <class G, class A, class = decltype(std::declval<const G&>().data()), class = decltype(std::declval<const A&>().data())>
  void set_imu_bias(const G& b_g, const A& b_a) {                         // :289, vi_ekf.cpp:179-183
    if (ok_) check(viekf_seq_set_imu_bias(seq_, b_g.data(), b_a.data()), "set_imu_bias");
  }
  void set_drag_term(const bool use_drag_term) {                          // :290 (a run-time switch: src/vi_ekf_ros.cpp:86,428-429)
    if (ok_ && check(viekf_batch_set_drag_term(core_, use_drag_term ? 1 : 0), "set_drag_term")) params_.use_drag_term = use_drag_term;
  }
  bool get_drag_term() const { return params_.use_drag_term != 0; }       // :291
  bool get_keyframe_reset() const { return params_.use_keyframe_reset != 0; }   // :292

  void get_global_pose(double t[3], double q[4]) { pose(t, q, nullptr, nullptr); }                   // :326
  void get_current_node_global_pose(double t[3], double q[4]) { pose(nullptr, nullptr, t, q); }      // :280
  void get_global_cov(double cov[36]) {                                   // :327 (column-major 6 x 6)
    for (int i = 0; i < 36; i++) cov[i] = 0.0;
    if (ok_) check(viekf_seq_get_global_cov(seq_, cov), "get_global_cov");
  }
  Mat get_global_cov() { Mat c(6, 6); get_global_cov(c.data()); return c; }

  bool NaNsInTheHouse() { return (status() & VIEKF_FLAG_NAN) != 0; }     // vi_ekf_error.cpp:6-38
  bool BlowingUp() { return (status() & VIEKF_FLAG_BLOWING_UP) != 0; }
  bool NegativeDepth() { return (status() & VIEKF_FLAG_NEGATIVE_DEPTH) != 0; }

  // ---- the reference's public test hooks, evaluated on the device (outputs: any type with data(), see the header) ----------
  template <class X, class D, class O, class = decltype(std::declval<const X&>().data()), class = decltype(std::declval<const D&>().data())>
  void boxplus(const X& x, const D& dx, O& out) const {                   // :299, vi_ekf_helper.cpp:88-98
    if (fits(out, nx_, 1, "boxplus")) check(viekf_batch_boxplus(core_, x.data(), dx.data(), out.data(), VIEKF_HOST), "boxplus");
  }
  template <class X1, class X2, class O, class = decltype(std::declval<const X1&>().data()), class = decltype(std::declval<const X2&>().data())>
  void boxminus(const X1& x1, const X2& x2, O& out) const {               // :300, vi_ekf_helper.cpp:100-111
    if (fits(out, n_, 1, "boxminus")) check(viekf_batch_boxminus(core_, x1.data(), x2.data(), out.data(), VIEKF_HOST), "boxminus");
  }
  template <class X, class U, class OX, class OA, class OG, class = decltype(std::declval<const X&>().data()),
            class = decltype(std::declval<const U&>().data())>
  void dynamics(const X& x, const U& u, OX& xdot, OA& dfdx, OG& dfdu) {   // :303, vi_ekf_dyn.cpp:5-11
    if (fits(xdot, n_, 1, "dynamics") && fits(dfdx, n_, n_, "dynamics") && fits(dfdu, n_, 6, "dynamics"))
      check(viekf_batch_eval_jacobians(core_, x.data(), u.data(), xdot.data(), dfdx.data(), dfdu.data(), VIEKF_HOST), "dynamics");
  }
  // h_<type>(x, h, H, id): h is the 4-entry zVector, H the 3 x MAX_DX hMatrix (:310-318, vi_ekf_meas.cpp:281-386).  The feature
  // models take the GLOBAL feature id and look its slot up (vi_ekf_meas.cpp:346,356,371,381); an id that is not tracked evaluates
  // to NaN (the reference indexes with -1 there).
  template <class X, class ZV, class HM, class = decltype(std::declval<const X&>().data())>
  void h(measurement_type_t type, const X& x, ZV& hv, HM& H, const int id) const {
    if (!fits(hv, 4, 1, "h") || !fits(H, 3, n_, "h")) return;
    int32_t slot = id;
    if (type == QZETA || type == FEAT || type == DEPTH || type == INV_DEPTH) {
      slot = global_to_local_feature_id(id);
      if (slot < 0) {
        for (int i = 0; i < 4; i++) hv.data()[i] = NAN;
        for (long i = 0; i < 3L * n_; i++) H.data()[i] = NAN;
        return;
      }
    }
    check(viekf_batch_eval_h_jacobian(core_, x.data(), (int32_t)type, &slot, hv.data(), H.data(), VIEKF_HOST), "h");
  }
  template <class X, class ZV, class HM> void h_acc(const X& x, ZV& hv, HM& H, const int id) const { h(ACC, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_alt(const X& x, ZV& hv, HM& H, const int id) const { h(ALT, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_att(const X& x, ZV& hv, HM& H, const int id) const { h(ATT, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_pos(const X& x, ZV& hv, HM& H, const int id) const { h(POS, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_vel(const X& x, ZV& hv, HM& H, const int id) const { h(VEL, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_qzeta(const X& x, ZV& hv, HM& H, const int id) const { h(QZETA, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_feat(const X& x, ZV& hv, HM& H, const int id) const { h(FEAT, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_depth(const X& x, ZV& hv, HM& H, const int id) const { h(DEPTH, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_inv_depth(const X& x, ZV& hv, HM& H, const int id) const { h(INV_DEPTH, x, hv, H, id); }
  template <class X, class ZV, class HM> void h_pixel_vel(const X&, ZV&, HM&, const int) const {}   // :319, vi_ekf_meas.cpp:388-395: "TODO"

  // ---- logger ------------------------------------------------------------------------------------------------------------
  void init_logger(std::string root_filename, std::string prefix = "") {  // vi_ekf_log.cpp:79-117
    if (ok_) check(viekf_seq_init_logger(seq_, root_filename.c_str(), prefix.c_str(), 0), "init_logger");
  }
  void disable_logger() { if (ok_) check(viekf_seq_disable_logger(seq_), "disable_logger"); }
  template <class T> void log_global_position(const T&) {}                // :334: LOG_GLOBAL is never opened (vi_ekf_log.cpp:85-97)

 private:
  void start(const viekf_params& p) {                                     // the filter + its sequencer from a parameter set
    params_ = p;
    if (!check(viekf_batch_create(1, num_features_, &p, device_, &core_), "create")) return;
    if (!check(viekf_seq_create(core_, 250, 200, &seq_), "seq_create")) return;   // LEN_STATE_HIST / LEN_MEAS_HIST, :50-51
    nx_ = 17 + 5 * num_features_;
    n_ = 16 + 3 * num_features_;
    ok_ = true;
  }
  void fill_table() {                                                     // vi_ekf.cpp:47-61
    measurement_functions.assign(TOTAL_MEAS, nullptr);
    measurement_functions[ACC] = &VIEKF::h_acc;
    measurement_functions[ALT] = &VIEKF::h_alt;
    measurement_functions[ATT] = &VIEKF::h_att;
    measurement_functions[POS] = &VIEKF::h_pos;
    measurement_functions[VEL] = &VIEKF::h_vel;
    measurement_functions[QZETA] = &VIEKF::h_qzeta;
    measurement_functions[FEAT] = &VIEKF::h_feat;
    measurement_functions[DEPTH] = &VIEKF::h_depth;
    measurement_functions[INV_DEPTH] = &VIEKF::h_inv_depth;
    measurement_functions[PIXEL_VEL] = &VIEKF::h_pixel_vel;
  }
  void take(VIEKF& o) {
    num_features_ = o.num_features_; device_ = o.device_; nx_ = o.nx_; n_ = o.n_; ok_ = o.ok_; params_ = o.params_;
    core_ = o.core_; seq_ = o.seq_; keyframe_reset_callback_ = std::move(o.keyframe_reset_callback_);
    o.core_ = nullptr; o.seq_ = nullptr; o.ok_ = false;
  }
  template <class O> bool fits(O& out, int rows, int cols, const char* what) const {
    if (!ok_) return false;
    shim::size_to(out, rows, cols, 0);
    if ((long)out.size() == (long)rows * cols) return true;
    std::cerr << "VIEKF::" << what << ": an output argument holds " << out.size() << " coefficients, " << (long)rows * cols << " needed\n";
    ok_ = false;
    return false;
  }
  bool check(int rc, const char* what) const {
    if (rc == VIEKF_OK) return true;
    std::cerr << "VIEKF::" << what << ": " << viekf_last_error() << " (" << rc << ")\n";   // diagnostics to cerr, as the reference
    ok_ = false;
    return false;
  }
  uint32_t status() {
    uint32_t f = 0;
    if (ok_) check(viekf_batch_get_status(core_, &f, VIEKF_HOST), "status");
    return f;
  }
  Mat feat_matrix(int rows, bool all = false) {
    const int len = all ? num_features_ : get_len_features();
    std::vector<double> buf((size_t)num_features_ * rows, NAN);
    if (ok_) check(viekf_seq_get_features(seq_, nullptr, rows == 3 ? buf.data() : nullptr, rows == 4 ? buf.data() : nullptr), "get_zetas");
    Mat o(rows, len);
    for (int i = 0; i < len; i++)
      for (int k = 0; k < rows; k++) o(k, i) = buf[(size_t)i * rows + k];
    return o;
  }
  void pose(double* t, double* q, double* nt, double* nq) {
    double p[7] = {0, 0, 0, 1, 0, 0, 0}, nd[7] = {0, 0, 0, 1, 0, 0, 0};
    if (ok_) check(viekf_seq_get_global_pose(seq_, p, nd), "get_global_pose");
    for (int i = 0; i < 3; i++) { if (t) t[i] = p[i]; if (nt) nt[i] = nd[i]; }
    for (int i = 0; i < 4; i++) { if (q) q[i] = p[3 + i]; if (nq) nq[i] = nd[3 + i]; }
  }
  void release() {
    if (seq_) viekf_seq_destroy(seq_);
    if (core_) viekf_batch_destroy(core_);
    seq_ = nullptr; core_ = nullptr; ok_ = false;
  }

  int num_features_ = NUM_FEATURES, device_ = 0, nx_ = 0, n_ = 0;
  mutable bool ok_ = false;
  viekf_params params_ = viekf_params();
  viekf_batch* core_ = nullptr;
  viekf_seq* seq_ = nullptr;
  xVector x_;
  dxMatrix P_;
  mutable std::vector<int> tracked_;
  std::function<void(void)> keyframe_reset_callback_;
};

}  // namespace vi_ekf
