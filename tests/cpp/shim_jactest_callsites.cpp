// shim_jactest_callsites.cpp -- the call expressions of the reference's only ASSERTING test, test/jac_test.cpp, against
// include/viekf_shim.hpp on the GPU: the 17-argument init(...) fixture (:118-170), every h_* through a member-function pointer
// (CALL_MEMBER_FN(ekf, fn)(x0, z0, H, id), :172-226, the pointers &VIEKF::h_acc ... and the public measurement_functions table),
// f_tilde (:283-304) inside the dfdx / dfdu tests (:306-428), the manifold identities (:246-280) and the keyframe-reset test
// (:446-487), with the reference's tolerances.  Eigen is absent from the build image: xVector ... hMatrix are a small
// fixed-size mock here, handed to the shim through VIEKF_SHIM_TYPES exactly as Eigen's typedefs would be; the quaternion algebra
// the reference takes from its (absent) geometry submodule is a few lines below (conventions of src/quat.cpp).  Seeds are fixed
// (the reference seeds from the wall clock, :490).  Prints one line per property; exit code 0 = every tolerance held.
// usage: shim_jactest_callsites [iterations]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#ifndef NUM_FEATURES
#define NUM_FEATURES 12
#endif
#define MAX_X (17 + NUM_FEATURES * 5)
#define MAX_DX (16 + NUM_FEATURES * 3)

// ---- the mock of the Eigen types: column-major, run-time shape in Mx, compile-time shape in F<R, C> ---------------------------
static std::mt19937_64 g_rng(20240917);
static double urand() { return std::uniform_real_distribution<double>(-1.0, 1.0)(g_rng); }   // Eigen's ::Random() is U[-1, 1]

struct Mx {
  int r = 0, c = 0;
  std::vector<double> d;
  Mx() {}
  Mx(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
  const double* data() const { return d.data(); }
  double* data() { return d.data(); }
  long size() const { return (long)d.size(); }
  int rows() const { return r; }
  int cols() const { return c; }
  double& operator()(int i, int j = 0) { return d[(size_t)i + (size_t)j * r]; }
  double operator()(int i, int j = 0) const { return d[(size_t)i + (size_t)j * r]; }
  void setZero() { for (double& v : d) v = 0.0; }
  void setZero(int rows, int cols) { *this = Mx(rows, cols); }
  void setOnes() { for (double& v : d) v = 1.0; }
  void setRandom() { for (double& v : d) v = urand(); }
  double norm() const { double s = 0; for (double v : d) s += v * v; return std::sqrt(s); }
  Mx block(int i0, int j0, int nr, int nc) const {
    Mx o(nr, nc);
    for (int j = 0; j < nc; j++) for (int i = 0; i < nr; i++) o(i, j) = (*this)(i0 + i, j0 + j);
    return o;
  }
  void set_block(int i0, int j0, const Mx& v) { for (int j = 0; j < v.c; j++) for (int i = 0; i < v.r; i++) (*this)(i0 + i, j0 + j) = v(i, j); }
  Mx col(int j) const { return block(0, j, r, 1); }
  Mx topRows(int k) const { return block(0, 0, k, c); }
  double maxabs() const { double m = 0; for (double v : d) m = std::fabs(v) > m ? std::fabs(v) : m; return m; }
};
static Mx operator+(const Mx& a, const Mx& b) { Mx o = a; for (size_t i = 0; i < o.d.size(); i++) o.d[i] += b.d[i]; return o; }
static Mx operator-(const Mx& a, const Mx& b) { Mx o = a; for (size_t i = 0; i < o.d.size(); i++) o.d[i] -= b.d[i]; return o; }
static Mx operator-(const Mx& a) { Mx o = a; for (double& v : o.d) v = -v; return o; }
static Mx operator*(const Mx& a, double k) { Mx o = a; for (double& v : o.d) v *= k; return o; }
static Mx operator/(const Mx& a, double k) { Mx o = a; for (double& v : o.d) v /= k; return o; }

template <int R, int C = 1>
struct F : Mx {
  F() : Mx(R, C) {}
  F(const Mx& m) : Mx(R, C) { for (size_t i = 0; i < d.size() && i < m.d.size(); i++) d[i] = m.d[i]; }
  F& operator=(const Mx& m) { for (size_t i = 0; i < d.size() && i < m.d.size(); i++) d[i] = m.d[i]; return *this; }
  static F Random() { F o; o.setRandom(); return o; }
  static F Identity() { F o; for (int i = 0; i < (R < C ? R : C); i++) o(i, i) = 1.0; return o; }
};
typedef F<2> Vector2d; typedef F<3> Vector3d; typedef F<4> Vector4d;
typedef Mx MatrixXd;

namespace vi_ekf {                       // reference include/vi_ekf.h:53-61
typedef F<MAX_X, 1> xVector;
typedef F<MAX_DX, 1> dxVector;
typedef F<MAX_DX, MAX_DX> dxMatrix;
typedef F<MAX_DX, 6> dxuMatrix;
typedef F<6, 1> uVector;
typedef F<4, 1> zVector;
typedef F<3, MAX_DX> hMatrix;
}
#define VIEKF_SHIM_TYPES
#include "viekf_shim.hpp"
using namespace vi_ekf;

// ---- what the reference takes from geometry/quat.h (src/quat.cpp: otimes :304-312, exp :64-80, roll / pitch / yaw :211-224) ---
static Vector4d q_otimes(const Mx& a, const Mx& b) {
  Vector4d o;
  o(0) = a(0) * b(0) - a(1) * b(1) - a(2) * b(2) - a(3) * b(3);
  o(1) = a(0) * b(1) + a(1) * b(0) + a(2) * b(3) - a(3) * b(2);
  o(2) = a(0) * b(2) - a(1) * b(3) + a(2) * b(0) + a(3) * b(1);
  o(3) = a(0) * b(3) + a(1) * b(2) - a(2) * b(1) + a(3) * b(0);
  return o;
}
static Vector4d q_plus(const Mx& q, const Mx& v) {      // Quatd + Vector3d = q (x) exp(v)
  const double th = v.norm();
  Vector4d e;
  e(0) = std::cos(th / 2.0);
  const double s = th > 1e-4 ? std::sin(th / 2.0) / th : 0.5;
  for (int i = 0; i < 3; i++) e(1 + i) = s * v(i);
  return q_otimes(q, e);
}
static double q_roll(const Mx& q) { return std::atan2(2.0 * (q(0) * q(1) + q(2) * q(3)), 1.0 - 2.0 * (q(1) * q(1) + q(2) * q(2))); }
static double q_pitch(const Mx& q) { return std::asin(2.0 * (q(0) * q(2) - q(3) * q(1))); }
static double q_yaw(const Mx& q) { return std::atan2(2.0 * (q(0) * q(3) + q(1) * q(2)), 1.0 - 2.0 * (q(2) * q(2) + q(3) * q(3))); }

#define CALL_MEMBER_FN(objectptr, ptrToMember) ((objectptr).*(ptrToMember))

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { g_fail++; std::printf("  FAILED %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static std::map<std::string, std::vector<int>> make_indexes() {           // test/jac_test.cpp:62-78
  std::map<std::string, std::vector<int>> tmp;
  tmp["dxPOS"] = {0, 3}; tmp["dxVEL"] = {3, 3}; tmp["dxATT"] = {6, 3}; tmp["dxB_A"] = {9, 3}; tmp["dxB_G"] = {12, 3}; tmp["dxMU"] = {15, 1};
  tmp["uA"] = {0, 3}; tmp["uG"] = {3, 3};
  for (int i = 0; i < 50; i++) { tmp["dxZETA_" + std::to_string(i)] = {16 + 3 * i, 2}; tmp["dxRHO_" + std::to_string(i)] = {16 + 3 * i + 2, 1}; }
  return tmp;
}
static std::map<std::string, std::vector<int>> indexes = make_indexes();

static bool check_block(const std::string& row_id, const std::string& col_id, const Mx& analytical, const Mx& fd, double tolerance = 1e-3) {
  const std::vector<int> row = indexes[row_id], col = indexes[col_id];
  const double e = (analytical - fd).block(row[0], col[0], row[1], col[1]).maxabs();
  if (e > tolerance) { std::printf("  Error in Jacobian %s, %s: %.3e > %.1e\n", row_id.c_str(), col_id.c_str(), e, tolerance); return true; }
  return false;
}

// ---- test/jac_test.cpp:118-170 -------------------------------------------------------------------------------------------------
VIEKF init_jacobians_test(xVector& x0, uVector& u0) {
  x0.setZero();
  x0(VIEKF::xATT) = 1.0;
  x0(VIEKF::xMU) = 0.2;
  x0.set_block((int)VIEKF::xPOS, 0, x0.block((int)VIEKF::xPOS, 0, 3, 1) + Vector3d::Random() * 100.0);
  x0.set_block((int)VIEKF::xVEL, 0, x0.block((int)VIEKF::xVEL, 0, 3, 1) + Vector3d::Random() * 10.0);
  x0.set_block((int)VIEKF::xATT, 0, q_plus(x0.block((int)VIEKF::xATT, 0, 4, 1), Vector3d::Random()));
  x0.set_block((int)VIEKF::xB_A, 0, x0.block((int)VIEKF::xB_A, 0, 3, 1) + Vector3d::Random() * 1.0);
  x0.set_block((int)VIEKF::xB_G, 0, x0.block((int)VIEKF::xB_G, 0, 3, 1) + Vector3d::Random() * 0.5);
  x0((int)VIEKF::xMU, 0) += (static_cast<double>(rand()) / (static_cast<double>(RAND_MAX))) * 0.05;

  VIEKF ekf;
  F<VIEKF::dxZ, 1> P0, Qx, gamma;
  P0.setOnes();
  Qx.setOnes();
  gamma.setOnes();
  uVector Qu;
  Qu.setOnes();
  Vector3d P0feat, Qxfeat, gammafeat;
  P0feat.setOnes();
  Qxfeat.setOnes();
  gammafeat.setOnes();
  Vector2d cam_center = Vector2d::Random();
  cam_center(0) = 320 - 25 + std::rand() % 50; cam_center(1) = 240 - 25 + std::rand() % 50;
  Vector2d focal_len;
  focal_len(0) = 250 + double(rand()) / RAND_MAX * 50; focal_len(1) = 250 + double(rand()) / RAND_MAX * 50;
  Vector4d qr = Vector4d::Random();
  qr = qr / qr.norm();                                  // Quatd::Random().elements()
  Vector4d half; half.setOnes(); half = half * 0.5;
  Vector4d q_b_c = half + qr;                           // NOT re-normalised, as in the reference (:146)
  Vector3d p_b_c = Vector3d::Random() * 0.5;
  F<17, 1> state_0;
  state_0 = x0.block(0, 0, 17, 1);
  ekf.init(state_0, P0, Qx, gamma, Qu, P0feat, Qxfeat, gammafeat, cam_center, focal_len, q_b_c, p_b_c, 2.0, true, true, true, 0.0);

  for (int i = 0; i < NUM_FEATURES; i++) {
    Vector2d l;
    l(0) = std::rand() % 640; l(1) = std::rand() % 480;
    double depth = 1.0 + double(rand()) / double(RAND_MAX) * 20.0;
    ekf.init_feature(l, i, depth);
  }
  x0 = ekf.get_state();

  u0.setZero();
  u0.set_block((int)VIEKF::uA, 0, u0.block((int)VIEKF::uA, 0, 3, 1) + Vector3d::Random() * 1.0);
  u0.set_block((int)VIEKF::uG, 0, u0.block((int)VIEKF::uG, 0, 3, 1) + Vector3d::Random() * 1.0);
  return ekf;
}

// ---- test/jac_test.cpp:172-226 -------------------------------------------------------------------------------------------------
int htest(measurement_function_ptr fn, VIEKF& ekf, const VIEKF::measurement_type_t type, const int id, const int dim, double tol = 1e-3) {
  int num_errors = 0;
  xVector x0 = ekf.get_state();
  zVector z0;
  MatrixXd a_dhdx;
  a_dhdx.setZero(dim, MAX_DX);

  hMatrix H;
  CALL_MEMBER_FN(ekf, fn)(x0, z0, H, id);
  a_dhdx = H.topRows(dim);

  MatrixXd d_dhdx;
  d_dhdx.setZero(dim, MAX_DX);

  F<MAX_DX, MAX_DX> I = F<MAX_DX, MAX_DX>::Identity();
  double epsilon = 1e-6;

  zVector z_prime;
  hMatrix dummy_H;
  xVector x_prime;
  for (int i = 0; i < a_dhdx.cols(); i++) {
    ekf.boxplus(ekf.get_state(), (I.col(i) * epsilon), x_prime);

    CALL_MEMBER_FN(ekf, fn)(x_prime, z_prime, dummy_H, id);

    Mx dcol;
    if (type == VIEKF::QZETA || type == VIEKF::ATT) {
      // q_feat_boxminus(Quatd(z_prime), Quatd(z0)) / (Quatd(z_prime) - Quatd(z0)): the geometry library is absent, so the two
      // quaternions are put into a state each and the filter's own boxminus returns the same difference (vi_ekf_helper.cpp:103,108)
      xVector xa = x0, xb = x0;
      const int at = type == VIEKF::ATT ? (int)VIEKF::xATT : (int)VIEKF::xZ + 5 * id;
      xa.set_block(at, 0, z_prime.block(0, 0, 4, 1));
      xb.set_block(at, 0, z0.block(0, 0, 4, 1));
      dxVector dd;
      ekf.boxminus(xa, xb, dd);
      dcol = type == VIEKF::ATT ? dd.block((int)VIEKF::dxATT, 0, 3, 1) : dd.block((int)VIEKF::dxZ + 3 * id, 0, 2, 1);
      dcol = dcol / epsilon;
    } else {
      dcol = (z_prime.topRows(dim) - z0.topRows(dim)) / epsilon;
    }
    d_dhdx.set_block(0, i, dcol);
  }

  MatrixXd error = a_dhdx - d_dhdx;
  double err_threshold = std::max(tol * a_dhdx.norm(), tol);

  for (std::map<std::string, std::vector<int>>::iterator it = indexes.begin(); it != indexes.end(); ++it) {
    if (it->second[0] + it->second[1] > error.cols()) continue;
    MatrixXd block_error = error.block(0, it->second[0], error.rows(), it->second[1]);
    if (block_error.maxabs() > err_threshold) {
      num_errors += 1;
      std::printf("  Error in Measurement %d_%d, %s: %.3e (thresh = %.3e)\n", (int)type, id, it->first.c_str(), block_error.maxabs(), err_threshold);
    }
  }
  return num_errors;
}

static void XVECTOR_EQUAL(VIEKF& ekf, xVector& x1, xVector& x2) {      // :228-244 (quaternions compared through the manifold difference)
  dxVector d;
  ekf.boxminus(x1, x2, d);
  EXPECT(d.maxabs() <= 1e-8, "x1 [-] x2 = %.3e", d.maxabs());
}

// ---- :246-280 ----------------------------------------------------------------------------------------------------------------------
void VIEKF_manifold(int iters) {
  xVector x, x2, x3;
  uVector u;
  dxVector dx, dx1, dx2;
  for (int j = 0; j < iters; j++) {
    vi_ekf::VIEKF ekf = init_jacobians_test(x, u);
    vi_ekf::VIEKF dummyekf = init_jacobians_test(x2, u);
    dx.setZero();

    ekf.boxplus(x, dx, x3);                                               // (x [+] 0) == x
    EXPECT((x3 - x).maxabs() <= 1e-8, "x [+] 0 != x: %.3e", (x3 - x).maxabs());

    ekf.boxminus(x2, x, dx2);                                             // (x [+] (x2 [-] x)) = x2
    ekf.boxplus(x, dx2, x3);
    XVECTOR_EQUAL(ekf, x3, x2);

    dx.setRandom();                                                       // ((x [+] dx) [-] x) == dx
    ekf.boxplus(x, dx, x3);
    ekf.boxminus(x3, x, dx2);
    EXPECT((dx2 - dx).maxabs() <= 1e-8, "(x [+] dx) [-] x != dx: %.3e", (dx2 - dx).maxabs());

    dx1.setRandom();                                                      // ||(x [+] dx1) [-] (x [+] dx2)|| < || dx1 - dx2 ||
    dx2.setRandom();
    ekf.boxplus(x, dx1, x2);
    ekf.boxplus(x, dx2, x3);
    ekf.boxminus(x2, x3, dx);
    EXPECT(dx.norm() <= (dx - dx2).norm(), "contraction: %.6f > %.6f", dx.norm(), (dx - dx2).norm());
    EXPECT(ekf.ok() && dummyekf.ok(), "a shim call failed");
  }
}

// ---- :283-304 ----------------------------------------------------------------------------------------------------------------------
void f_tilde(const dxVector& x_tilde, const xVector& x_hat, const uVector& u, const double& dt, vi_ekf::VIEKF& ekf, dxVector& dx_tilde) {
  xVector x, x_plus, x_minus, x_hat_plus, x_hat_minus;
  dxMatrix dummydfdx;
  dxuMatrix dummydfdu;
  dxVector dx, dx_hat, x_tilde_plus, x_tilde_minus;
  ekf.boxplus(x_hat, x_tilde, x);

  ekf.dynamics(x, u, dx, dummydfdx, dummydfdu);
  ekf.dynamics(x_hat, u, dx_hat, dummydfdx, dummydfdu);

  ekf.boxplus(x, dx * dt, x_plus);
  ekf.boxplus(x, -dx * dt, x_minus);
  ekf.boxplus(x_hat, dx_hat * dt, x_hat_plus);
  ekf.boxplus(x_hat, -dx_hat * dt, x_hat_minus);

  ekf.boxminus(x_plus, x_hat_plus, x_tilde_plus);
  ekf.boxminus(x_minus, x_hat_minus, x_tilde_minus);

  dx_tilde = (x_tilde_plus - x_tilde_minus) / (2 * dt);
}

// ---- :306-372 ----------------------------------------------------------------------------------------------------------------------
void VIEKF_dfdx_test(int iters) {
  xVector x_hat;
  uVector u;
  dxVector x_tilde, x_tilde_plus, x_tilde_minus, dx_tilde, dx_tilde_plus, dx_tilde_minus;
  dxMatrix Idx = dxMatrix::Identity();
  double epsilon = 1e-5;
  double dt = 1e-3;
  dxMatrix d_dfdx;
  dxVector dummy_dx;
  dxuMatrix a_dfdu;
  dxMatrix a_dfdx;
  for (int j = 0; j < iters; j++) {
    vi_ekf::VIEKF ekf = init_jacobians_test(x_hat, u);
    ekf.dynamics(x_hat, u, dummy_dx, a_dfdx, a_dfdu);
    d_dfdx.setZero();
    x_tilde = dxVector::Random() * epsilon;
    f_tilde(x_tilde, x_hat, u, dt, ekf, dx_tilde);
    for (int i = 0; i < d_dfdx.cols(); i++) {
      x_tilde_plus = dx_tilde + Idx.col(i) * epsilon;
      x_tilde_minus = dx_tilde - Idx.col(i) * epsilon;
      f_tilde(x_tilde_plus, x_hat, u, dt, ekf, dx_tilde_plus);
      f_tilde(x_tilde_minus, x_hat, u, dt, ekf, dx_tilde_minus);
      d_dfdx.set_block(0, i, (dx_tilde_plus - dx_tilde_minus) / (2 * epsilon));
    }
    int bad = 0;
    bad += check_block("dxPOS", "dxVEL", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxPOS", "dxATT", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxVEL", "dxVEL", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxVEL", "dxATT", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxVEL", "dxB_A", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxVEL", "dxB_G", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxVEL", "dxMU", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxATT", "dxATT", a_dfdx, d_dfdx, 1e-2);
    bad += check_block("dxATT", "dxB_G", a_dfdx, d_dfdx, 1e-2);
    for (int i = 0; i < ekf.get_len_features(); i++) {
      std::string zeta_key = "dxZETA_" + std::to_string(i);
      std::string rho_key = "dxRHO_" + std::to_string(i);
      bad += check_block(zeta_key, "dxVEL", a_dfdx, d_dfdx, 5e-1);
      bad += check_block(zeta_key, "dxB_G", a_dfdx, d_dfdx, 5e-1);
      bad += check_block(zeta_key, zeta_key, a_dfdx, d_dfdx, 5e-1);
      bad += check_block(zeta_key, rho_key, a_dfdx, d_dfdx, 5e-1);
      bad += check_block(rho_key, "dxVEL", a_dfdx, d_dfdx, 5e-1);
      bad += check_block(rho_key, "dxB_G", a_dfdx, d_dfdx, 5e-1);
      bad += check_block(rho_key, zeta_key, a_dfdx, d_dfdx, 5e-1);
      bad += check_block(rho_key, rho_key, a_dfdx, d_dfdx, 5e-1);
    }
    EXPECT(bad == 0, "dfdx: %d blocks out of tolerance", bad);
    EXPECT(ekf.ok() && ekf.get_len_features() == NUM_FEATURES, "shim state");
  }
}

// ---- :374-428 ----------------------------------------------------------------------------------------------------------------------
void VIEKF_dfdu_test(int iters) {
  xVector x_hat;
  uVector u;
  dxVector x_tilde, x_tilde_plus, x_tilde_minus, dx_tilde, dx_tilde_plus, dx_tilde_minus;
  F<MAX_DX, 6> Iu;
  Iu.setZero();
  for (int k = 0; k < 6; k++) Iu((int)VIEKF::dxB_A + k, k) = 1.0;         // Iu.block<6,6>(VIEKF::dxB_A,0).setIdentity()
  double epsilon = 1e-5;
  double dt = 1e-3;
  dxuMatrix d_dfdu;
  dxVector dummy_dx;
  dxMatrix a_dfdx;
  dxuMatrix a_dfdu;
  for (int j = 0; j < iters; j++) {
    vi_ekf::VIEKF ekf = init_jacobians_test(x_hat, u);
    ekf.dynamics(x_hat, u, dummy_dx, a_dfdx, a_dfdu);
    d_dfdu.setZero();
    x_tilde = dxVector::Random() * epsilon;
    f_tilde(x_tilde, x_hat, u, dt, ekf, dx_tilde);
    for (int i = 0; i < d_dfdu.cols(); i++) {
      x_tilde_plus = dx_tilde + Iu.col(i) * epsilon;
      x_tilde_minus = dx_tilde - Iu.col(i) * epsilon;
      f_tilde(x_tilde_plus, x_hat, u, dt, ekf, dx_tilde_plus);
      f_tilde(x_tilde_minus, x_hat, u, dt, ekf, dx_tilde_minus);
      d_dfdu.set_block(0, i, (dx_tilde_plus - dx_tilde_minus) / (2 * epsilon));
    }
    int bad = 0;
    bad += check_block("dxVEL", "uA", a_dfdu, d_dfdu, 1e-2);
    bad += check_block("dxVEL", "uG", a_dfdu, d_dfdu, 1e-2);
    bad += check_block("dxATT", "uG", a_dfdu, d_dfdu, 1e-2);
    for (int i = 0; i < ekf.get_len_features(); i++) {
      std::string zeta_key = "dxZETA_" + std::to_string(i);
      std::string rho_key = "dxRHO_" + std::to_string(i);
      bad += check_block(zeta_key, "uG", a_dfdu, d_dfdu, 5e-1);
      bad += check_block(rho_key, "uG", a_dfdu, d_dfdu, 5e-1);
    }
    EXPECT(bad == 0, "dfdu: %d blocks out of tolerance", bad);
    EXPECT(ekf.ok(), "shim state");
  }
}

// ---- :430-444 ----------------------------------------------------------------------------------------------------------------------
void VI_EKF_h_test(int iters) {
  xVector x0;
  uVector u0;
  for (int j = 0; j < iters; j++) {
    vi_ekf::VIEKF ekf = init_jacobians_test(x0, u0);
    int bad = 0;
    bad += htest(&VIEKF::h_acc, ekf, VIEKF::ACC, 0, 2);
    bad += htest(&VIEKF::h_pos, ekf, VIEKF::POS, 0, 3);
    bad += htest(&VIEKF::h_vel, ekf, VIEKF::VEL, 0, 3);
    bad += htest(&VIEKF::h_alt, ekf, VIEKF::ALT, 0, 1);
    ekf.set_drag_term(true);
    bad += htest(&VIEKF::h_att, ekf, VIEKF::ATT, 0, 3);
    ekf.set_drag_term(false);
    bad += htest(&VIEKF::h_att, ekf, VIEKF::ATT, 0, 3);
    for (int i = 0; i < ekf.get_len_features(); i++) {
      bad += htest(&VIEKF::h_feat, ekf, VIEKF::FEAT, i, 2, 1e-1);
      bad += htest(&VIEKF::h_qzeta, ekf, VIEKF::QZETA, i, 2);
      bad += htest(&VIEKF::h_depth, ekf, VIEKF::DEPTH, i, 1);
      bad += htest(&VIEKF::h_inv_depth, ekf, VIEKF::INV_DEPTH, i, 1);
    }
    // the public table (include/vi_ekf.h:263, filled as vi_ekf.cpp:47-61) holds the same functions, in the enum's order
    EXPECT((int)ekf.measurement_functions.size() == (int)VIEKF::TOTAL_MEAS, "measurement_functions has %d entries", (int)ekf.measurement_functions.size());
    bad += htest(ekf.measurement_functions[VIEKF::ACC], ekf, VIEKF::ACC, 0, 2);
    bad += htest(ekf.measurement_functions[VIEKF::FEAT], ekf, VIEKF::FEAT, NUM_FEATURES - 1, 2, 1e-1);
    bad += htest(ekf.measurement_functions[VIEKF::INV_DEPTH], ekf, VIEKF::INV_DEPTH, 1, 1);
    zVector zz; hMatrix HH; zz.setOnes(); HH.setOnes();
    CALL_MEMBER_FN(ekf, ekf.measurement_functions[VIEKF::PIXEL_VEL])(x0, zz, HH, 0);   // the reference's empty TODO: touches nothing
    EXPECT(zz(0) == 1.0 && HH(0, 0) == 1.0, "h_pixel_vel wrote something");
    EXPECT(bad == 0, "h_test: %d blocks out of tolerance", bad);
    EXPECT(ekf.ok(), "shim state");
  }
}

// ---- :446-487 ----------------------------------------------------------------------------------------------------------------------
void VIEKF_KF_reset_test(int iters) {
  uVector u0;
  dxMatrix d_dxpdxm;
  dxMatrix a_dxpdxm;
  xVector xm;
  xVector xp;
  dxMatrix dummy;
  dxMatrix I_dx = dxMatrix::Identity();
  xVector xm_prime;
  xVector xp_prime;
  dxVector d_xp;
  for (int j = 0; j < iters; j++) {
    vi_ekf::VIEKF ekf = init_jacobians_test(xm, u0);
    ekf.keyframe_reset(xm, xp, a_dxpdxm);
    Mx qm = xm.block((int)VIEKF::xATT, 0, 4, 1);
    Mx qp = xp.block((int)VIEKF::xATT, 0, 4, 1);
    EXPECT(std::fabs(q_roll(qm) - q_roll(qp)) <= 1e-8, "roll changed");
    EXPECT(std::fabs(q_pitch(qm) - q_pitch(qp)) <= 1e-8, "pitch changed");
    EXPECT(std::fabs(q_yaw(qp)) <= 1e-8, "yaw not reset");

    d_dxpdxm.setZero();
    double epsilon = 1e-6;
    for (int i = 0; i < d_dxpdxm.cols(); i++) {
      ekf.boxplus(xm, (I_dx.col(i) * epsilon), xm_prime);
      ekf.keyframe_reset(xm_prime, xp_prime, dummy);
      ekf.boxminus(xp_prime, xp, d_xp);
      d_dxpdxm.set_block(0, i, d_xp / epsilon);
    }
    EXPECT(!check_block("dxPOS", "dxPOS", a_dxpdxm, d_dxpdxm), "KF reset dxPOS");
    EXPECT(!check_block("dxATT", "dxATT", a_dxpdxm, d_dxpdxm, 1e-1), "KF reset dxATT");
    EXPECT(ekf.ok(), "shim state");
  }
  EXPECT((a_dxpdxm - d_dxpdxm).maxabs() <= 1e-1, "KF reset: whole matrix %.3e", (a_dxpdxm - d_dxpdxm).maxabs());   // check_all, :486
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 3;
  srand(12345);                                                            // (:490 seeds from the clock)
  struct { const char* name; void (*fn)(int); } tests[] = {{"manifold", VIEKF_manifold}, {"dfdx_test", VIEKF_dfdx_test},
      {"dfdu_test", VIEKF_dfdu_test}, {"h_test", VI_EKF_h_test}, {"KF_reset_test", VIEKF_KF_reset_test}};
  for (auto& t : tests) {
    const int before = g_fail;
    t.fn(iters);
    std::printf("VI_EKF.%s: %s (%d iterations)\n", t.name, g_fail == before ? "OK" : "FAILED", iters);
  }
  return g_fail ? 1 : 0;
}
