// shim_callsites.cpp -- the reference's own CALL EXPRESSIONS against include/viekf_shim.hpp, with Eigen replaced by a small mock
// vector / matrix type (Eigen is absent from the build image; the shim's arguments are templates on "anything with data() /
// size() / rows()", so the same expressions compile against Eigen types unchanged).  The expressions are those of
//   src/vi_ekf_ros.cpp:82-86 (load, register_keyframe_reset_callback, set_drag_term), :174 (propagate_state(imu_, t)),
//   :180-191 (ACC by get_drag_term), :195-198 (ATT), :282 (keep_only_features), :302-304 (FEAT / DEPTH), :308-309
//   (handle_measurements(&gated_ids)), :398-402 (x0 = get_state().topRows(xZ); set_x0; keyframe_reset), :428-429 (drag term on),
//   :453-470 (POS / ATT / VEL / ALT),   test/vi_ekf_test.cpp:24-33 (propagate_state(z, t, true), add_measurement(..., NAN))
// driven by a scripted flight (tests/test_abi_consumers.py); what the getters return goes to out.bin for comparison with the
// restated plumbing.  Event records: 32 doubles each, [code, t, payload...]:
//   1 imu callback        imu[6] at 2..7, use_acc at 8, is_flying at 9, q_att[4] at 10..13, use_imu_att at 14
//   2 camera frame        count at 2, then (id, x, y, depth) x count from 3 (at most 7 features per record), use_depth at 31
//   3 truth callback      z_pos[3] at 2..4, z_att[4] at 5..8, truth_active at 9, is_flying at 10, z_alt at 11
//   4 keep_only_features  count at 2, ids at 3..
//   5 first-truth init    z_pos[3] at 2..4, z_att[4] at 5..8     (set_x0 + keyframe_reset)
//   6 take-off            (drag term on if it was off)
//   7 sim callbacks       imu[6] at 2..7 (vi_ekf_test: propagate_state(z, t, true))
//   8 set_imu_bias        b_g[3] at 2..4, b_a[3] at 5..7
//   9 clear_feature       id at 2
// usage: shim_callsites params.yaml num_features events.bin out.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "viekf_shim.hpp"

// ---- the mock: fixed-size column-major matrix with the members the call sites use -------------------------------------------
template <int R, int C = 1>
struct M {
  double d[R * C] = {};
  const double* data() const { return d; }
  double* data() { return d; }
  long size() const { return R * C; }
  int rows() const { return R; }
  int cols() const { return C; }
  double& operator()(int i, int j = 0) { return d[i + j * R]; }
  double operator()(int i, int j = 0) const { return d[i + j * R]; }
  M() {}
  template <class V, class = decltype(std::declval<const V&>().data())>
  M(const V& v) { for (int i = 0; i < R * C; i++) d[i] = v.data()[i]; }   // (Matrix<double, xZ, 1> x0 = ekf_.get_state().topRows(xZ))
  template <int RR, int CC> M<RR, CC> block(int i0, int j0) const { M<RR, CC> o; for (int j = 0; j < CC; j++) for (int i = 0; i < RR; i++) o(i, j) = (*this)(i0 + i, j0 + j); return o; }
  template <int RR> void set_block(int i0, const M<RR, 1>& v) { for (int i = 0; i < RR; i++) d[i0 + i] = v(i); }
  double norm() const { double s = 0; for (double v : d) s += v * v; return std::sqrt(s); }
  static M Identity() { M o; for (int i = 0; i < (R < C ? R : C); i++) o(i, i) = 1.0; return o; }
  M operator*(double k) const { M o; for (int i = 0; i < R * C; i++) o.d[i] = d[i] * k; return o; }
};
typedef M<6> Vector6d; typedef M<4> Vector4d; typedef M<3> Vector3d; typedef M<2> Vector2d; typedef M<1> Vector1d;
typedef M<2, 2> Matrix2d; typedef M<3, 3> Matrix3d; typedef M<1, 1> Matrix1d;

struct VIEKF_ROS {   // the members of the reference's adapter that the mirrored call sites touch (include/vi_ekf_ros.h)
  vi_ekf::VIEKF ekf_;
  Vector6d imu_;
  Vector2d z_acc_drag_, z_feat_; Vector3d z_acc_grav_; Vector4d z_att_; Vector1d z_alt_, z_depth_;
  Matrix2d acc_R_drag_, feat_R_; Matrix3d acc_R_grav_, att_R_, pos_R_; Matrix1d alt_R_, depth_R_;
  bool use_acc_ = true, is_flying_ = false, use_truth_ = false, use_imu_att_ = false, use_features_ = true, use_depth_ = false,
       use_drag_term_ = true, got_depth_ = true;
  int resets = 0;
  std::vector<double> results;
  std::vector<int> gated_ids;
  void keyframe_reset_callback() { resets++; }
};

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  VIEKF_ROS r;
  r.ekf_.~VIEKF();
  new (&r.ekf_) vi_ekf::VIEKF(argv[1], std::atoi(argv[2]));   // (num_features is a run-time value here)
  vi_ekf::VIEKF& ekf_ = r.ekf_;
  if (!ekf_.ok()) return 3;
  ekf_.register_keyframe_reset_callback(std::bind(&VIEKF_ROS::keyframe_reset_callback, &r));   // vi_ekf_ros.cpp:83
  ekf_.set_drag_term(false);                                                                     // :86  Start out not using the drag term
  r.acc_R_drag_ = Matrix2d::Identity() * 0.5; r.acc_R_grav_ = Matrix3d::Identity() * 0.5; r.att_R_ = Matrix3d::Identity() * 0.01;
  r.pos_R_ = Matrix3d::Identity() * 0.01; r.feat_R_ = Matrix2d::Identity() * 10.0; r.alt_R_ = Matrix1d::Identity() * 0.05;
  r.depth_R_ = Matrix1d::Identity() * 0.1;
  FILE* f = std::fopen(argv[3], "rb");
  if (!f) return 4;
  double e[32];
  while (std::fread(e, sizeof(double), 32, f) == 32) {
    const int code = (int)e[0];
    const double t = e[1];
    if (code == 1) {                                          // ---- VIEKF_ROS::imu_callback, vi_ekf_ros.cpp:147-199
      for (int i = 0; i < 6; i++) r.imu_(i) = e[2 + i];
      r.use_acc_ = e[8] != 0.0; r.is_flying_ = e[9] != 0.0; r.use_imu_att_ = e[14] != 0.0;
      ekf_.propagate_state(r.imu_, t);                                                           // :174
      if (ekf_.get_drag_term() == true) {                                                        // :180
        r.z_acc_drag_ = r.imu_.block<2, 1>(0, 0);
        r.results.push_back(ekf_.add_measurement(t, r.z_acc_drag_, vi_ekf::VIEKF::ACC, r.acc_R_drag_, r.use_acc_ && r.is_flying_));   // :183
      } else {
        r.z_acc_grav_ = r.imu_.block<3, 1>(0, 0);
        double norm = r.z_acc_grav_.norm();
        if (norm < 9.80665 * 1.15 && norm > 9.80665 * 0.85)
          r.results.push_back(ekf_.add_measurement(t, r.z_acc_grav_, vi_ekf::VIEKF::ACC, r.acc_R_grav_, r.use_acc_));                // :190
      }
      for (int i = 0; i < 4; i++) r.z_att_(i) = e[10 + i];
      if (r.use_imu_att_)
        r.results.push_back(ekf_.add_measurement(t, r.z_att_, vi_ekf::VIEKF::ATT, r.att_R_, (r.use_truth_) ? true : r.use_imu_att_));   // :198
    } else if (code == 2) {                                   // ---- VIEKF_ROS::color_image_callback, :254-314 (the tracker's output is scripted)
      const int cnt = (int)e[2];
      r.use_depth_ = e[31] != 0.0;
      std::vector<int> ids_;
      for (int i = 0; i < cnt; i++) ids_.push_back((int)e[3 + 4 * i]);
      for (int i = 0; i < cnt; i++) {
        float depth = (float)e[3 + 4 * i + 3];
        r.z_feat_(0) = e[3 + 4 * i + 1]; r.z_feat_(1) = e[3 + 4 * i + 2];
        r.z_depth_(0) = depth;
        int result = ekf_.add_measurement(t, r.z_feat_, vi_ekf::VIEKF::FEAT, r.feat_R_, r.use_features_, ids_[i], (r.use_depth_) ? depth : NAN);   // :302
        r.results.push_back(result);
        if (result == vi_ekf::VIEKF::MEAS_SUCCESS && r.got_depth_ && !(depth != depth))
          r.results.push_back(ekf_.add_measurement(t, r.z_depth_, vi_ekf::VIEKF::DEPTH, r.depth_R_, r.use_depth_, ids_[i]));          // :304
      }
      std::vector<int> gated_ids;
      ekf_.handle_measurements(&gated_ids);                                                      // :309
      for (auto it = gated_ids.begin(); it != gated_ids.end(); it++) r.gated_ids.push_back(*it);
    } else if (code == 3) {                                   // ---- VIEKF_ROS::truth_callback, :438-471
      Vector3d z_pos; Vector4d z_att;
      for (int i = 0; i < 3; i++) z_pos(i) = e[2 + i];
      for (int i = 0; i < 4; i++) z_att(i) = e[5 + i];
      bool truth_active = e[9] != 0.0;
      r.is_flying_ = e[10] != 0.0;
      r.z_alt_(0) = e[11];
      r.results.push_back(ekf_.add_measurement(t, z_pos, vi_ekf::VIEKF::POS, r.pos_R_, truth_active));       // :453
      r.results.push_back(ekf_.add_measurement(t, z_att, vi_ekf::VIEKF::ATT, r.att_R_, truth_active));       // :454
      ekf_.handle_measurements();                                                                            // :455
      if (!r.is_flying_) {
        Vector3d meas;                                       // Vector3d::Zero()
        Matrix3d R = Matrix3d::Identity() * 1e-8;
        r.results.push_back(ekf_.add_measurement(t, meas, vi_ekf::VIEKF::VEL, R, true));                      // :464
      }
      r.results.push_back(ekf_.add_measurement(t, r.z_alt_, vi_ekf::VIEKF::ALT, r.alt_R_, !truth_active));   // :470
    } else if (code == 4) {
      std::vector<int> ids_;
      for (int k = 0; k < (int)e[2]; k++) ids_.push_back((int)e[3 + k]);
      ekf_.keep_only_features(ids_);                                                             // :282
    } else if (code == 5) {                                   // ---- first truth message, :396-403
      Vector3d z_pos; Vector4d z_att;
      for (int i = 0; i < 3; i++) z_pos(i) = e[2 + i];
      for (int i = 0; i < 4; i++) z_att(i) = e[5 + i];
      M<vi_ekf::VIEKF::xZ, 1> x0 = ekf_.get_state().topRows(vi_ekf::VIEKF::xZ);                 // :398
      x0.set_block<3>((int)vi_ekf::VIEKF::xPOS, z_pos);                                          // :399  x0.block<3,1>(xPOS,0) = z_pos
      x0.set_block<4>((int)vi_ekf::VIEKF::xATT, z_att);                                          // :400
      ekf_.set_x0(x0);                                                                           // :401
      ekf_.keyframe_reset();                                                                     // :402
    } else if (code == 6) {
      if (r.use_drag_term_ == true && ekf_.get_drag_term() == false)                             // :428
        ekf_.set_drag_term(true);
    } else if (code == 7) {                                   // ---- test/vi_ekf_test.cpp:24-27
      Vector6d z;
      for (int i = 0; i < 6; i++) z(i) = e[2 + i];
      ekf_.propagate_state(z, t, true);
    } else if (code == 8) {
      Vector3d b_g, b_a;
      for (int i = 0; i < 3; i++) { b_g(i) = e[2 + i]; b_a(i) = e[5 + i]; }
      ekf_.set_imu_bias(b_g, b_a);
    } else if (code == 9) {
      ekf_.clear_feature((int)e[2]);
    }
    if (!ekf_.ok()) return 5;
  }
  std::fclose(f);
  FILE* o = std::fopen(argv[4], "wb");
  if (!o) return 6;
  auto put = [&](const double* p, size_t n) { std::fwrite(p, sizeof(double), n, o); };
  const vi_ekf::VIEKF::Vec& x = ekf_.get_state();
  const vi_ekf::VIEKF::Mat& P = ekf_.get_covariance();
  const std::vector<int>& tr = ekf_.tracked_features();
  const int len = ekf_.get_len_features();
  double hdr[6] = {(double)x.size(), (double)ekf_.max_dx(), (double)len, (double)tr.size(), (double)r.gated_ids.size(), (double)r.results.size()};
  put(hdr, 6);
  put(x.data(), (size_t)x.size());
  put(P.data(), (size_t)P.size());
  for (int v : tr) { double d = v; put(&d, 1); }
  for (int v : r.gated_ids) { double d = v; put(&d, 1); }
  put(r.results.data(), r.results.size());
  double t3[3], q[4], nt[3], nq[4];
  ekf_.get_global_pose(t3, q);
  ekf_.get_current_node_global_pose(nt, nq);
  vi_ekf::VIEKF::Mat cov = ekf_.get_global_cov();
  put(t3, 3); put(q, 4); put(nt, 3); put(nq, 4); put(cov.data(), 36);
  // the feature getters (include/vi_ekf.h:275-279,284)
  vi_ekf::VIEKF::Vec depths = ekf_.get_depths();
  vi_ekf::VIEKF::Mat zetas = ekf_.get_zetas(), qzetas = ekf_.get_qzetas();
  put(depths.data(), (size_t)depths.size()); put(zetas.data(), (size_t)zetas.size()); put(qzetas.data(), (size_t)qzetas.size());
  for (int v : tr) {
    vi_ekf::VIEKF::Vec px = ekf_.get_feat(v);
    double dd = ekf_.get_depth(v);
    put(px.data(), 2); put(&dd, 1);
    vi_ekf::VIEKF::Vec zi = ekf_.get_zeta(ekf_.global_to_local_feature_id(v));
    put(zi.data(), 3);
  }
  vi_ekf::VIEKF::Vec dg = ekf_.get_covariance_diagonal();
  put(dg.data(), (size_t)dg.size());
  double tail[4] = {(double)r.resets, (double)((ekf_.NaNsInTheHouse() ? 1 : 0) | (ekf_.BlowingUp() ? 2 : 0)), ekf_.get_drag_term() ? 1.0 : 0.0,
                    ekf_.get_keyframe_reset() ? 1.0 : 0.0};
  put(tail, 4);
  std::fclose(o);
  return 0;
}
