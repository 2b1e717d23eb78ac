// viekf_shim.hpp -- the reference's C++ surface over the C ABI (include/viekf.h), for callers written against
// vi_ekf::VIEKF (reference include/vi_ekf.h:82-338): same method names, argument order and result codes, plain
// arrays / std::vector where the reference has Eigen types (Eigen is absent on the build box: an Eigen-typed
// overload is one Eigen::Map per argument on top of these).  ONE filter = a batch of one behind the host sequencer
// (viekf_seq_*): measurements are queued and whole frames are forwarded by handle_measurements(), the state is
// read back only when the caller asks for it -- never once per update.
//
//   reference member (include/vi_ekf.h)                                  here
//   VIEKF(const string& param_file), load()            :243,249      VIEKF(param_file[, num_features, device]), load()
//   propagate_state(u, t, save_input = true)           :302          propagate_state(const double u[6], t)
//   add_measurement(t, z, type, R, active, id, depth)  :308          same order; z / R as std::vector (R column-major)
//   handle_measurements(vector<int>* gated = nullptr)  :307          same
//   init_feature(l, id, depth = -1.0)                  :294          same (id ignored, as vi_ekf_feat.cpp:29-30 does)
//   keep_only_features(vector<int>)                    :296          same (+ the keyframe-reset callback, :325)
//   get_state() / get_covariance() / _diagonal()       :277-279      std::vector<double> (x: MAX_X; P: MAX_DX^2 column-major)
//   get_len_features() / tracked_features()            :266,281      same
//   get_global_pose() / get_global_cov()               :326-327      {t(3), q(4)} / 6x6 column-major
//   get_current_node_global_pose()                     :276          {t(3), q(4)}
//   NaNsInTheHouse() / BlowingUp() / NegativeDepth()   :261-263      from the per-filter status word
//   init_logger(root, prefix) / disable_logger()       :332-333      same files (vi_ekf_log.cpp:79-117)
// Not offered: the test hooks (boxplus / dynamics / h_* with Eigen outputs, measurement_functions) -- the parity tests
// reach those through the C ABI's evaluation calls; clear_feature(id) alone (use keep_only_features).
// No exceptions cross this class, like the reference: a failing call prints to std::cerr and ok() turns false.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "viekf.h"

#ifndef NUM_FEATURES
#define NUM_FEATURES 12   // reference include/vi_ekf.h:39-45 (release default)
#endif

namespace vi_ekf {

class VIEKF {
 public:
  // reference include/vi_ekf.h:113-138 (same values as viekf_meas_type / viekf_meas_result)
  typedef enum { ACC, ALT, ATT, POS, VEL, QZETA, FEAT, PIXEL_VEL, DEPTH, INV_DEPTH, TOTAL_MEAS } measurement_type_t;
  typedef enum { MEAS_SUCCESS, MEAS_GATED, MEAS_NAN, MEAS_INVALID, MEAS_NEW_FEATURE } meas_result_t;
  enum : int { xPOS = 0, xVEL = 3, xATT = 6, xB_A = 10, xB_G = 13, xMU = 16, xZ = 17 };          // :87-95
  enum : int { dxPOS = 0, dxVEL = 3, dxATT = 6, dxB_A = 9, dxB_G = 12, dxMU = 15, dxZ = 16 };    // :103-111

  VIEKF() {}
  explicit VIEKF(const std::string& param_file, int num_features = NUM_FEATURES, int device = 0)
      : num_features_(num_features), device_(device) { load(param_file); }
  VIEKF(const VIEKF&) = delete;
  VIEKF& operator=(const VIEKF&) = delete;
  ~VIEKF() { release(); }

  void load(const std::string& param_file) {                              // vi_ekf.cpp:101-155
    release();
    viekf_params p;
    if (!check(viekf_params_load_yaml(param_file.c_str(), &p), "load")) return;
    params_ = p;
    if (!check(viekf_batch_create(1, num_features_, &p, device_, &core_), "create")) return;
    if (!check(viekf_seq_create(core_, 250, 200, &seq_), "seq_create")) return;   // LEN_STATE_HIST / LEN_MEAS_HIST, :50-51
    nx_ = 17 + 5 * num_features_;
    n_ = 16 + 3 * num_features_;
    ok_ = true;
  }
  bool ok() const { return ok_; }
  int max_x() const { return nx_; }
  int max_dx() const { return n_; }

  void propagate_state(const double u[6], const double t) {               // vi_ekf.cpp:262-318
    if (ok_) check(viekf_seq_propagate(seq_, u, t), "propagate_state");
  }

  meas_result_t add_measurement(const double t, const std::vector<double>& z, const measurement_type_t& meas_type,
                                const std::vector<double>& R, bool active = false, const int id = -1,
                                const double depth = NAN) {               // vi_ekf_meas.cpp:130-194
    if (!ok_) return MEAS_INVALID;
    int rdim = 1;
    while (rdim * rdim < (int)R.size()) rdim++;
    int32_t res = MEAS_INVALID, gid = id;
    check(viekf_seq_add_measurement(seq_, t, (int32_t)meas_type, z.data(), (int32_t)z.size(), R.data(), rdim, active ? 1 : 0,
                                    &gid, &depth, &res), "add_measurement");
    return (meas_result_t)res;
  }

  void handle_measurements(std::vector<int>* gated_feature_ids = nullptr) {   // vi_ekf_meas.cpp:6-127
    if (!ok_) return;
    std::vector<int32_t> ids((size_t)num_features_ + 8, -1);
    int32_t cnt = 0;
    if (!check(viekf_seq_handle_measurements(seq_, ids.data(), (int32_t)ids.size(), &cnt), "handle_measurements")) return;
    if (gated_feature_ids)
      for (int k = 0; k < cnt && k < (int)ids.size(); k++) gated_feature_ids->push_back(ids[k]);
  }

  bool init_feature(const double l[2], const int id, const double depth = -1.0) {   // vi_ekf_feat.cpp:6-47
    (void)id;   // the reference pushes its own counter (:29-30)
    if (!ok_) return false;
    int32_t okf = 0;
    check(viekf_seq_init_feature(seq_, l, &depth, nullptr, &okf), "init_feature");
    return okf != 0;
  }

  void keep_only_features(const std::vector<int> features) {              // vi_ekf_feat.cpp:81-142
    if (!ok_) return;
    std::vector<int32_t> ids(features.begin(), features.end());
    if (ids.empty()) ids.push_back(-1);
    uint8_t did = 0;
    if (!check(viekf_seq_keep_only_features(seq_, ids.data(), (int32_t)ids.size(), &did, nullptr), "keep_only_features")) return;
    if (did && keyframe_reset_callback_) keyframe_reset_callback_();      // vi_ekf_kfr.cpp:154-156
  }
  void register_keyframe_reset_callback(std::function<void(void)> cb) { keyframe_reset_callback_ = cb; }   // :325

  const std::vector<double>& get_state() {                                // :277
    x_.assign((size_t)nx_, 0.0);
    if (ok_) check(viekf_batch_get_state(core_, x_.data(), nullptr, nullptr, VIEKF_HOST), "get_state");
    return x_;
  }
  const std::vector<double>& get_covariance() {                           // :278 (column-major MAX_DX x MAX_DX)
    P_.assign((size_t)n_ * n_, 0.0);
    if (ok_) check(viekf_batch_get_state(core_, nullptr, P_.data(), nullptr, VIEKF_HOST), "get_covariance");
    return P_;
  }
  std::vector<double> get_covariance_diagonal() {                         // :279
    std::vector<double> d((size_t)n_, 0.0);
    if (ok_) check(viekf_batch_get_cov_diag(core_, d.data(), VIEKF_HOST), "get_covariance_diagonal");
    return d;
  }
  int get_len_features() {                                                // :281
    int32_t len = 0;
    if (ok_) check(viekf_batch_get_state(core_, nullptr, nullptr, &len, VIEKF_HOST), "get_len_features");
    return len;
  }
  const std::vector<int>& tracked_features() {                            // :266
    std::vector<int32_t> ids((size_t)num_features_, -1);
    int32_t len = 0;
    tracked_.clear();
    if (ok_ && check(viekf_seq_tracked_features(seq_, ids.data(), &len), "tracked_features"))
      tracked_.assign(ids.begin(), ids.begin() + len);
    return tracked_;
  }
  void get_global_pose(double t[3], double q[4]) { pose(t, q, nullptr, nullptr); }                   // :326
  void get_current_node_global_pose(double t[3], double q[4]) { pose(nullptr, nullptr, t, q); }      // :276
  void get_global_cov(double cov[36]) {                                   // :327 (column-major 6 x 6)
    for (int i = 0; i < 36; i++) cov[i] = 0.0;
    if (ok_) check(viekf_seq_get_global_cov(seq_, cov), "get_global_cov");
  }

  bool NaNsInTheHouse() { return (status() & VIEKF_FLAG_NAN) != 0; }     // vi_ekf_error.cpp:6-38
  bool BlowingUp() { return (status() & VIEKF_FLAG_BLOWING_UP) != 0; }
  bool NegativeDepth() { return (status() & VIEKF_FLAG_NEGATIVE_DEPTH) != 0; }

  void init_logger(std::string root_filename, std::string prefix = "") {  // vi_ekf_log.cpp:79-117
    if (ok_) check(viekf_seq_init_logger(seq_, root_filename.c_str(), prefix.c_str(), 0), "init_logger");
  }
  void disable_logger() { if (ok_) check(viekf_seq_disable_logger(seq_), "disable_logger"); }

 private:
  bool check(int rc, const char* what) {
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
  bool ok_ = false;
  viekf_params params_;
  viekf_batch* core_ = nullptr;
  viekf_seq* seq_ = nullptr;
  std::vector<double> x_, P_;
  std::vector<int> tracked_;
  std::function<void(void)> keyframe_reset_callback_;
};

}  // namespace vi_ekf
