// shim_driver.cpp -- drives include/viekf_shim.hpp (the reference-shaped C++ class over the C ABI) through a scripted
// flight written by tests/test_abi_consumers.py and dumps what the reference's getters return, for comparison with the
// restated plumbing (oracle/seq_oracle.py).  Event records: 32 doubles each, [code, t, payload...]:
//   1 propagate_state  u[6] at 2..7
//   2 add_measurement  type, zdim, z[4], rdim, R[9], active, id, depth  at 2..19
//   3 handle_measurements
//   4 keep_only_features  count at 2, ids at 3..
//   5 init_feature  l[2] at 2..3, id at 4, depth at 5
// usage: shim_driver params.yaml num_features events.bin out.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "viekf_shim.hpp"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  vi_ekf::VIEKF ekf(argv[1], std::atoi(argv[2]));
  if (!ekf.ok()) return 3;
  int resets = 0;
  ekf.register_keyframe_reset_callback([&resets]() { resets++; });
  FILE* f = std::fopen(argv[3], "rb");
  if (!f) return 4;
  double e[32];
  std::vector<int> gated;
  std::vector<double> results;
  while (std::fread(e, sizeof(double), 32, f) == 32) {
    const int code = (int)e[0];
    const double t = e[1];
    if (code == 1) {
      ekf.propagate_state(e + 2, t);
    } else if (code == 2) {
      const int zdim = (int)e[3], rdim = (int)e[8];
      std::vector<double> z(e + 4, e + 4 + zdim), R(e + 9, e + 9 + rdim * rdim);
      results.push_back((double)ekf.add_measurement(t, z, (vi_ekf::VIEKF::measurement_type_t)(int)e[2], R, e[18] != 0.0, (int)e[19],
                                                    e[20]));
    } else if (code == 3) {
      ekf.handle_measurements(&gated);
    } else if (code == 4) {
      std::vector<int> ids;
      for (int k = 0; k < (int)e[2]; k++) ids.push_back((int)e[3 + k]);
      ekf.keep_only_features(ids);
    } else if (code == 5) {
      results.push_back(ekf.init_feature(e + 2, (int)e[4], e[5]) ? 1.0 : 0.0);
    }
    if (!ekf.ok()) return 5;
  }
  std::fclose(f);
  FILE* o = std::fopen(argv[4], "wb");
  if (!o) return 6;
  auto put = [&](const double* p, size_t n) { std::fwrite(p, sizeof(double), n, o); };
  const vi_ekf::VIEKF::Vec& x = ekf.get_state();
  const vi_ekf::VIEKF::Mat& P = ekf.get_covariance();
  const std::vector<int>& tr = ekf.tracked_features();
  double hdr[6] = {(double)x.size(), (double)ekf.max_dx(), (double)ekf.get_len_features(), (double)tr.size(), (double)gated.size(),
                   (double)results.size()};
  put(hdr, 6);
  put(x.data(), (size_t)x.size());
  put(P.data(), (size_t)P.size());
  for (int v : tr) { double d = v; put(&d, 1); }
  for (int v : gated) { double d = v; put(&d, 1); }
  put(results.data(), results.size());
  double t[3], q[4], cov[36], nt[3], nq[4];
  ekf.get_global_pose(t, q);
  ekf.get_current_node_global_pose(nt, nq);
  ekf.get_global_cov(cov);
  put(t, 3); put(q, 4); put(nt, 3); put(nq, 4); put(cov, 36);
  double tail[2] = {(double)resets, (double)((ekf.NaNsInTheHouse() ? 1 : 0) | (ekf.BlowingUp() ? 2 : 0))};
  put(tail, 2);
  std::fclose(o);
  return 0;
}
