// seq_sanitized_driver.cpp -- drives the host sequencer (vi_ekf_amd/csrc/viekf_seq.cpp) over tests/cpp/seq_host_stub.cpp through
// the scenarios of tests/test_gpu_sequencer.py -- shared clock with camera delays 0 / 10.5 / 30 ms (rewind + replay every frame),
// ring wrap-around, queue trims, measurements from the future and from before the history, keep_only_features with keyframe resets,
// the log writer, independent clocks with different periods / origins / delays -- in a -fsanitize=address,undefined build.
// Checks what the stub makes checkable: every filter's propagated time equals its clock's span, replays included exactly once per
// rewound interval (x[0] = sum of dt along the final time line).  Exit code 0 = all scenarios ran clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/viekf.h"

#define CHECK(e) do { int rc_ = (e); if (rc_ != 0) { std::fprintf(stderr, "%s failed: %d (line %d)\n", #e, rc_, __LINE__); std::exit(10); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }

static void shared_clock(const char* yaml, double delay, int state_hist, int meas_hist, bool logger, const char* tmp) {
  viekf_params p;
  CHECK(viekf_params_load_yaml(yaml, &p));
  const int B = 5, N = 7, nx = 17 + 5 * N;
  viekf_batch* core; viekf_seq* s;
  CHECK(viekf_batch_create(B, N, &p, 0, &core));
  CHECK(viekf_seq_create(core, state_hist, meas_hist, &s));
  if (logger) CHECK(viekf_seq_init_logger(s, tmp, "san", 2));
  std::vector<double> u((size_t)B * 6, 0.0), z((size_t)B * N * 2, 100.0), R = {10.0, 0.0, 0.0, 10.0};
  std::vector<int32_t> ids((size_t)B * N), res((size_t)B * N);
  for (int b = 0; b < B; b++) for (int i = 0; i < N; i++) ids[(size_t)b * N + i] = i;
  const double dt = 0.004;
  double t_end = 0.0;
  for (int k = 0; k < 400; k++) {
    const double t = dt * k;
    for (auto& v : u) v = urand();
    CHECK(viekf_seq_propagate(s, u.data(), t));
    t_end = t;
    if (k % 8 == 3) {
      double tz = t - delay;
      if (k == 43) tz = t + 1.0;            // from the future: deferred (vi_ekf_meas.cpp:24-28) -- handled much later
      if (k == 83) tz = t - 10.0;           // older than every input: refused / dropped (:39-43,59-64)
      if (k % 16 == 3) z[3] = NAN; else z[3] = 100.0;   // a NaN entry for filter 0 (MEAS_NAN: that filter skips the entry)
      CHECK(viekf_seq_add_frame(s, tz, nullptr, N, z.data(), R.data(), 1, ids.data(), nullptr, nullptr, res.data()));
      std::vector<int32_t> gid((size_t)B * 8), gc(B);
      CHECK(viekf_seq_handle_measurements(s, gid.data(), 8, gc.data()));
    }
    if (k % 50 == 17) {                     // other measurement models in the queue between the frames
      std::vector<double> za((size_t)B * 3, 0.1), Ra = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      CHECK(viekf_seq_add_measurement(s, t - 0.002, VIEKF_POS, za.data(), 3, Ra.data(), 3, 1, nullptr, nullptr, nullptr));
      CHECK(viekf_seq_handle_measurements(s, nullptr, 0, nullptr));
    }
    if (k == 150 || k == 300) {             // drop features: compaction, overlap test, keyframe reset, node frame
      std::vector<int32_t> keep((size_t)B * 3);
      for (int b = 0; b < B; b++) { keep[(size_t)b * 3] = 0; keep[(size_t)b * 3 + 1] = (k == 150) ? 2 : -1; keep[(size_t)b * 3 + 2] = 5; }
      std::vector<uint8_t> did(B); std::vector<double> edges((size_t)B * 17);
      CHECK(viekf_seq_keep_only_features(s, keep.data(), 3, did.data(), edges.data()));
      std::vector<double> pose((size_t)B * 7), node((size_t)B * 7), cov((size_t)B * 36);
      CHECK(viekf_seq_get_global_pose(s, pose.data(), node.data()));
      CHECK(viekf_seq_get_global_cov(s, cov.data()));
    }
    if (k == 200) { std::vector<double> x0((size_t)B * 17, 0.0); for (int b = 0; b < B; b++) x0[(size_t)b * 17 + 6] = 1.0; CHECK(viekf_seq_set_x0(s, x0.data())); CHECK(viekf_seq_keyframe_reset(s, nullptr, nullptr)); }
  }
  std::vector<double> x((size_t)B * nx);
  CHECK(viekf_batch_get_state(core, x.data(), nullptr, nullptr, VIEKF_HOST));
  // set_x0 at k = 200 writes x_[i_] only (vi_ekf.cpp:157-160): the stub's time counter restarts from 0 there -- unless a later
  // frame is stamped BEFORE that moment (camera delay 30 ms = 7.5 IMU periods): its rewind goes back to a ring slot from before
  // the edit and the replay carries on from there, in the reference as here: the edit is lost and the counter spans the whole run
  const double want = (delay > 0.02) ? t_end : t_end - dt * 200;
  for (int b = 0; b < B; b++)
    if (std::fabs(x[(size_t)b * nx] - want) > 1e-9) { std::fprintf(stderr, "filter %d: propagated %.9f s, its time line spans %.9f s (delay %.4f)\n", b, x[(size_t)b * nx], want, delay); std::exit(11); }
  std::vector<int32_t> tr((size_t)B * N), ln(B);
  CHECK(viekf_seq_tracked_features(s, tr.data(), ln.data()));
  double tn; int32_t ri, q, in;
  CHECK(viekf_seq_status(s, &tn, &ri, &q, &in));
  // (the queues are trimmed at the end of a handle_measurements that ran to its end, vi_ekf_meas.cpp:119-126: between two of them
  //  the input queue holds up to one frame interval of samples more, the measurement queue up to one frame more)
  if (q > meas_hist + N + 1 || in > state_hist + 16) { std::fprintf(stderr, "queues not trimmed: %d %d\n", q, in); std::exit(12); }
  if (logger) CHECK(viekf_seq_disable_logger(s));
  CHECK(viekf_seq_destroy(s));
  CHECK(viekf_batch_destroy(core));
}

// A ring that wraps past the start of a fused replay while steps of that replay are still unmaterialised, then a second, slower
// sensor stamped INSIDE the replayed span: the reference's ring still holds x_ / P_ of that step (vi_ekf_meas.cpp:46-57).
// Frames every 8 steps stamped 30 ms back (a fused replay of 8 steps per frame), ring of `state_hist` slots, an altimeter reading
// `late_steps` steps after each frame stamped `back` steps + 2 ms before its arrival.  The stub poisons what a fused replay skips,
// so a rewind that lands on a skipped slot, or on a newer state, shows up in the propagated time.
extern "C" int stub_write_every_slot;   // tests/cpp/seq_host_stub.cpp
static std::vector<double> late_second_sensor_run(const char* yaml, int state_hist, int late_steps, int back, int period = 8, bool with_alt = false);
// ... and the same flight with a core that writes every ring slot (nothing to re-create, no orphan): the stub's state depends on
// the order of the steps and on the input each step got, so the two final states are equal bit for bit only if every re-created
// state was re-created from the right start with the right inputs.
static void late_second_sensor(const char* yaml, int state_hist, int late_steps, int back, int period = 8, bool with_alt = false) {
  const unsigned long long seed = rng_state;
  stub_write_every_slot = 0;
  const std::vector<double> fused = late_second_sensor_run(yaml, state_hist, late_steps, back, period, with_alt);
  rng_state = seed;
  stub_write_every_slot = 1;
  const std::vector<double> plain = late_second_sensor_run(yaml, state_hist, late_steps, back, period, with_alt);
  stub_write_every_slot = 0;
  for (size_t i = 0; i < fused.size(); i++)
    if (!(fused[i] == plain[i])) { std::fprintf(stderr, "late sensor (H %d, +%d, -%d): state entry %zu differs between the fused and the every-slot core: %.17g vs %.17g\n", state_hist, late_steps, back, i, fused[i], plain[i]); std::exit(16); }
}
static std::vector<double> late_second_sensor_run(const char* yaml, int state_hist, int late_steps, int back, int period, bool with_alt) {
  viekf_params p;
  CHECK(viekf_params_load_yaml(yaml, &p));
  const int B = 3, N = 4, nx = 17 + 5 * N;
  viekf_batch* core; viekf_seq* s;
  CHECK(viekf_batch_create(B, N, &p, 0, &core));
  CHECK(viekf_seq_create(core, state_hist, 100000, &s));   // (no queue trim here: an entry that disappears was erased)
  std::vector<double> u((size_t)B * 6, 0.0), z((size_t)B * N * 2, 100.0), R = {10.0, 0.0, 0.0, 10.0}, za((size_t)B, 2.0), Ra = {0.01};
  std::vector<int32_t> ids((size_t)B * N);
  for (int b = 0; b < B; b++) for (int i = 0; i < N; i++) ids[(size_t)b * N + i] = i;
  const double dt = 0.004;
  double t_end = 0.0;
  int handled_late = 0;
  for (int k = 0; k < 200; k++) {
    const double t = dt * k;
    for (auto& v : u) v = urand();
    CHECK(viekf_seq_propagate(s, u.data(), t));
    t_end = t;
    if (k % period == 3) {
      CHECK(viekf_seq_add_frame(s, t - 0.03, nullptr, N, z.data(), R.data(), 1, ids.data(), nullptr, nullptr, nullptr));
      if (with_alt)    // (the frame's own altimeter reading 1 ms later: the fused replay then starts from ITS slot)
        CHECK(viekf_seq_add_measurement(s, t - 0.03 + 0.001, VIEKF_ALT, za.data(), 1, Ra.data(), 1, 1, nullptr, nullptr, nullptr));
      CHECK(viekf_seq_handle_measurements(s, nullptr, 0, nullptr));
    }
    if (k > 16 && k % period == (3 + late_steps) % period) {
      int32_t q0, q1;
      CHECK(viekf_seq_add_measurement(s, t - dt * back + 0.002, VIEKF_ALT, za.data(), 1, Ra.data(), 1, 1, nullptr, nullptr, nullptr));
      CHECK(viekf_seq_status(s, nullptr, nullptr, &q0, nullptr));
      CHECK(viekf_seq_handle_measurements(s, nullptr, 0, nullptr));
      CHECK(viekf_seq_status(s, nullptr, nullptr, &q1, nullptr));
      handled_late += (q1 == q0);          // (still queued as handled, not erased as "older than the state history")
    }
  }
  std::vector<double> x((size_t)B * nx);
  CHECK(viekf_batch_get_state(core, x.data(), nullptr, nullptr, VIEKF_HOST));
  for (int b = 0; b < B; b++)
    if (!(std::fabs(x[(size_t)b * nx] - t_end) <= 1e-9)) { std::fprintf(stderr, "late sensor (H %d, +%d, -%d): filter %d propagated %.9f s, its time line spans %.9f s\n", state_hist, late_steps, back, b, x[(size_t)b * nx], t_end); std::exit(14); }
  if (handled_late < 20) { std::fprintf(stderr, "late sensor (H %d): only %d of the late readings were fused\n", state_hist, handled_late); std::exit(15); }
  CHECK(viekf_seq_destroy(s));
  CHECK(viekf_batch_destroy(core));
  return x;
}

static void independent_clocks(const char* yaml, const int B) {   // (B >= 256: the queues are planned on several host threads)
  viekf_params p;
  CHECK(viekf_params_load_yaml(yaml, &p));
  const int N = 5, nx = 17 + 5 * N;
  viekf_batch* core; viekf_seq* s;
  CHECK(viekf_batch_create(B, N, &p, 0, &core));
  CHECK(viekf_seq_create_independent(core, 32, 40, &s));
  const double origin[6] = {0.0, 1000.0, -50.0, 3.0, 0.0, 7.5}, period[6] = {0.004, 0.005, 0.004, 0.01, 0.0025, 0.004},
               delay[6] = {0.0, 0.03, 0.0105, 0.02, 0.0, 0.05};
  std::vector<double> u((size_t)B * 6), tt(B), z((size_t)B * N * 2, 50.0), R = {10.0, 0.0, 0.0, 10.0};
  std::vector<int32_t> ids((size_t)B * N);
  for (int b = 0; b < B; b++) for (int i = 0; i < N; i++) ids[(size_t)b * N + i] = i;
  std::vector<int> steps(B, 0);
  std::vector<double> last(B, 0.0), first(B, NAN);
  for (int round = 0; round < 500; round++) {
    std::vector<uint8_t> mask(B, 0);
    for (int b = 0; b < B; b++) {
      if (urand() < 0.7) { mask[b] = 1; tt[b] = origin[b % 6] + period[b % 6] * steps[b]; steps[b]++; last[b] = tt[b]; if (std::isnan(first[b])) first[b] = tt[b]; }
      for (int c = 0; c < 6; c++) u[(size_t)b * 6 + c] = urand();
    }
    CHECK(viekf_seq_propagate_t(s, u.data(), tt.data(), mask.data()));
    if (round % 7 == 4) {
      std::vector<double> tz(B);
      std::vector<uint8_t> m2(B, 0);
      for (int b = 0; b < B; b++) { tz[b] = last[b] - delay[b % 6]; m2[b] = (steps[b] > 2 && urand() < 0.8) ? 1 : 0; }
      CHECK(viekf_seq_add_frame(s, 0.0, tz.data(), N, z.data(), R.data(), 1, ids.data(), nullptr, m2.data(), nullptr));
      CHECK(viekf_seq_handle_measurements(s, nullptr, 0, nullptr));
    }
    if (round == 250) {
      std::vector<int32_t> keep((size_t)B * 2);
      for (int b = 0; b < B; b++) { keep[(size_t)b * 2] = 1; keep[(size_t)b * 2 + 1] = 3; }
      CHECK(viekf_seq_keep_only_features(s, keep.data(), 2, nullptr, nullptr));
    }
  }
  std::vector<double> x((size_t)B * nx);
  CHECK(viekf_batch_get_state(core, x.data(), nullptr, nullptr, VIEKF_HOST));
  for (int b = 0; b < B; b++) {
    const double want = last[b] - first[b];
    if (std::fabs(x[(size_t)b * nx] - want) > 1e-6) { std::fprintf(stderr, "independent filter %d: propagated %.9f s, its clock spans %.9f s\n", b, x[(size_t)b * nx], want); std::exit(13); }
  }
  CHECK(viekf_seq_destroy(s));
  CHECK(viekf_batch_destroy(core));
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const char* yaml = argv[1];
  const std::string tmp = std::string(argv[2]) + "/";
  shared_clock(yaml, 0.0, 250, 200, false, tmp.c_str());
  shared_clock(yaml, 0.0105, 250, 200, true, tmp.c_str());
  shared_clock(yaml, 0.03, 250, 200, false, tmp.c_str());
  shared_clock(yaml, 0.03, 12, 9, false, tmp.c_str());      // a ring barely longer than the rewind, a queue shorter than a frame + extras
  late_second_sensor(yaml, 12, 6, 10);     // the replay's start slot overwritten four steps earlier: served from the orphan buffer
  late_second_sensor(yaml, 12, 6, 9);      //   (an even / odd number of steps from the orphan: either buffer of the ping-pong ends it)
  late_second_sensor(yaml, 12, 5, 10);
  late_second_sensor(yaml, 13, 6, 11);
  late_second_sensor(yaml, 24, 2, 6);      // the start slot still in the ring: re-created from it
  // (back is in steps + 2 ms: 9 steps back - 0.5 = 0.034 s, the flights of tests/test_gpu_sequencer.py)
  late_second_sensor(yaml, 12, 6, 10, 7, true);
  late_second_sensor(yaml, 12, 6, 9, 7, true);
  late_second_sensor(yaml, 12, 5, 10, 7, true);
  late_second_sensor(yaml, 13, 6, 11, 7, true);
  late_second_sensor(yaml, 11, 6, 8, 7, true);
  independent_clocks(yaml, 6);
  independent_clocks(yaml, 300);
  std::printf("sanitized sequencer scenarios: ok\n");
  return 0;
}
