// viekf_seq.cpp -- host sequencer over a viekf_batch (include/viekf.h, "Host sequencer"): the reference's input deque,
// measurement queue, state-history ring with rewind / replay, global feature ids and the keyframe trigger, batched for
// filters that share one clock.  Uses only the public C ABI of the batch; every numeric operation runs on the device.
//
// Reference lines followed (byu-magicc/VI-EKF): src/vi_ekf/vi_ekf.cpp:262-318 (propagate_state bookkeeping),
// src/vi_ekf/vi_ekf_meas.cpp:6-127 (handle_measurements), :130-194 (add_measurement), src/vi_ekf/vi_ekf_feat.cpp:29-30,
// 81-142 (feature numbering, keep_only_features), src/vi_ekf/vi_ekf_helper.cpp:114-125 (global_to_local_feature_id).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/viekf.h"

#ifdef VIEKF_SEQ_TRACE
#include <cstdio>
#define TR(...) std::fprintf(stderr, __VA_ARGS__)
#else
#define TR(...) do {} while (0)
#endif

namespace {

struct SeqMeas {                      // measurement_t, include/vi_ekf.h:167-179 (per-filter payload for the whole batch)
  double t;
  int type, zdim, rdim;
  std::vector<double> z;              // [B][zdim]
  std::vector<double> R;              // rdim x rdim
  bool active;
  std::vector<int32_t> id;            // [B] global feature id
  std::vector<uint8_t> present;       // [B] this filter queued the entry
  bool handled;
  // count > 1: a whole camera frame queued by viekf_seq_add_frame as ONE entry -- `count` FEAT measurements with the same stamp
  // and R, z [B][count][2], id / present [B][count], stored in the order handle_measurements consumes same-stamp entries (the
  // reverse of their insertion, vi_ekf_meas.cpp:150-176 with :16-18,96-98).  A later entry with the same stamp is inserted behind
  // every earlier one (:150-156), never between two of them, so the block is what its `count` separate entries would be.
  int count = 1;
};

void rota(const double* q, const double* v, double* o) {   // src/quat.cpp:279-283
  const double* b = q + 1;
  double t[3] = {2.0 * (b[1] * v[2] - b[2] * v[1]), 2.0 * (b[2] * v[0] - b[0] * v[2]), 2.0 * (b[0] * v[1] - b[1] * v[0])};
  double c[3] = {b[1] * t[2] - b[2] * t[1], b[2] * t[0] - b[0] * t[2], b[0] * t[1] - b[1] * t[0]};
  for (int i = 0; i < 3; i++) o[i] = v[i] + q[0] * t[i] + c[i];
}

void otimes(const double* a, const double* b, double* o) {   // src/quat.cpp:304-312
  const double r0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double r1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double r2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double r3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

// T1 * T2 = { t1 + q1.rota(t2), q1 (x) q2 }  (the SE(3) convention stated in include/viekf.h; transforms are {t(3), q(4)})
void xform_compose(const double* T1, const double* T2, double* out) {
  double r[3], q[4];
  rota(T1 + 3, T2, r);
  otimes(T1 + 3, T2 + 3, q);
  for (int i = 0; i < 3; i++) out[i] = T1[i] + r[i];
  for (int i = 0; i < 4; i++) out[3 + i] = q[i];
}

// out (6x6, column-major) += Adj(T)^T C Adj(T),  Adj(T) = [ R  [t]x R ; 0  R ],  R = q.R() (passive, src/quat.cpp:226-242)
// -- VIEKF::propagate_global_covariance, src/vi_ekf/vi_ekf_kfr.cpp:47-53
void add_adj_cov(const double* T, const double* C, double* out) {
  const double w = T[3], x = T[4], y = T[5], z = T[6];
  const double R[3][3] = {{1 - 2 * y * y - 2 * z * z, 2 * x * y + 2 * w * z, 2 * x * z - 2 * w * y},
                          {2 * x * y - 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z + 2 * w * x},
                          {2 * x * z + 2 * w * y, 2 * y * z - 2 * w * x, 1 - 2 * x * x - 2 * y * y}};
  const double S[3][3] = {{0, -T[2], T[1]}, {T[2], 0, -T[0]}, {-T[1], T[0], 0}};
  double A[6][6] = {};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i][j] = R[i][j];
      A[3 + i][3 + j] = R[i][j];
      double sr = 0.0;
      for (int k = 0; k < 3; k++) sr += S[i][k] * R[k][j];
      A[i][3 + j] = sr;
    }
  double CA[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double v = 0.0;
      for (int k = 0; k < 6; k++) v += C[i + 6 * k] * A[k][j];
      CA[i][j] = v;
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double v = 0.0;
      for (int k = 0; k < 6; k++) v += A[k][i] * CA[k][j];
      out[i + 6 * j] += v;
    }
}

}  // namespace

// Independent clocks (viekf_seq_create_independent): every filter keeps its own time ring, input deque and measurement queue
// -- the reference class's members, one set per filter -- and the device work of a call is batched over the filters that
// take the same kind of step.
// A camera frame handed over by viekf_seq_add_frame, shared by the queues of all filters that took entries from it: the pixels
// are stored once, every filter's queue holds ONE entry that points here (r04: 50 queue entries of 136 bytes per filter and frame,
// copied again into the step list, were most of the host time of this mode).  Rows are in PROCESSING order (the reverse of the
// insertion order, vi_ekf_meas.cpp:150-176 with :16-18,96-98): member j of row b is feature count - 1 - j of the call.
struct FrameData {
  int count = 0;
  std::vector<double> z;              // [B][count][2]
  std::vector<int32_t> id;            // [B][count]
  std::vector<uint8_t> present;       // [B][count]: this filter queued the member (not NaN, a tracked feature, not before its start)
  double R[4] = {};
};
struct FMeas {                        // measurement_t, include/vi_ekf.h:167-179 (one filter)
  double t;
  int type, zdim, rdim;
  double z[4], R[9];
  bool active;
  int32_t id;
  bool handled;
  std::shared_ptr<const FrameData> frame;   // set: a frame block -- members [first, frame->count) of this filter's row
  int first = 0;
  int weight = 1;                     // how many measurements of the reference's queue this entry stands for (:121-122)
};
struct FInput { double t; double u[6]; };   // (t, rotated u), vi_ekf.cpp:269-272
struct FilterSeq {
  std::vector<double> t;                                           // t_ ring
  int i = 0;                                                       // i_
  double start_t = NAN;
  std::deque<FInput> u;                                            // newest first
  std::deque<FMeas> zbuf;                                          // newest first
  long queued = 0;                                                 // sum of the entries' weights = zbuf_.size() of the reference
};
struct SeqOp {                        // one device step of one filter's handle_measurements, in the order it must run
  enum Kind { REWIND, PROP, FEAT_FRAME, GENERIC } kind;
  int slot = -1;                      // REWIND: ring slot to restore;  PROP: ring slot to record into
  double u[6] = {}, dt = 0.0;         // PROP
  std::vector<FMeas> meas;            // FEAT_FRAME from single entries: in processing order;  GENERIC: one entry
  std::shared_ptr<const FrameData> frame;   // FEAT_FRAME from a frame block: members [first, frame->count) of the filter's row
  int first = 0;
  int count() const { return frame ? frame->count - first : (int)meas.size(); }
};

struct viekf_seq {
  viekf_batch* core = nullptr;
  bool indep = false;                                              // independent clocks
  std::vector<FilterSeq> fs;
  int B = 0, N = 0, H = 0, MH = 0;
  viekf_params prm;
  std::vector<double> t;                                           // t_ ring
  std::vector<uint8_t> mat;                                        // ring slot holds its state (0: a step inside a fused replay --
                                                                   // its time is known, its state is re-created on demand)
  // Shared clock: ring slot -> buffer of the core's history ring.  The core holds H + 1 buffers; the one outside the ring (`extra`)
  // is free or holds the ORPHAN: the state a fused replay started from, taken out of the ring (a swap of two indices, no copy) at
  // the moment its slot was due to be overwritten while steps of that replay were still unmaterialised -- the reference's ring
  // still has x_ / P_ for those steps (vi_ekf_meas.cpp:46-57), so a late measurement may rewind to one of them.
  std::vector<int> phys;
  int extra = 0;
  bool orph = false;
  double orph_t = NAN;
  struct Step { int slot; double t; std::vector<double> u; };
  std::vector<Step> orph_steps;                                    // the orphan's unmaterialised steps, oldest first
  std::vector<std::vector<double>> slot_u;                         // per ring slot: the (rotated) input [B][6] its step was made with
  int i = 0;                                                       // i_
  double start_t = NAN;
  std::deque<std::pair<double, std::vector<double>>> u;            // (t, rotated u [B][6]), newest first
  std::deque<SeqMeas> zbuf;                                        // newest first
  std::vector<std::vector<int32_t>> ids;                           // current_feature_ids_ per filter
  std::vector<std::vector<int32_t>> slot_of;                       // per filter: global id -> local slot (-1: not tracked); the ids
                                                                   // are the filter's own counter (vi_ekf_feat.cpp:29-30): dense
  std::vector<int32_t> next_id;                                    // next_feature_id_ per filter
  std::vector<std::vector<int32_t>> kf_feats;                      // keyframe_features_ per filter
  std::vector<double> node;                                        // current_node_global_pose_ per filter: {t(3), q(4)}
  std::vector<double> node_cov;                                    // global_pose_cov_ per filter: 6x6 column-major
  std::string err;
  // binary logs of ONE filter of the batch, file for file what VIEKF::init_logger opens (src/vi_ekf/vi_ekf_log.cpp:79-117)
  std::vector<std::ofstream> log;                                  // empty = logging off
  int log_filter = 0;
};

namespace {

// vi_ekf_helper.cpp:114-125 (a linear std::find there; a frame of 50 features on 1024 filters asks 100,000 times per frame: table)
int local_id(const viekf_seq* s, int b, int gid) {
  const auto& m = s->slot_of[b];
  return (gid >= 0 && gid < (int)m.size()) ? m[(size_t)gid] : -1;
}
void rebuild_slots(viekf_seq* s, int b) {
  auto& m = s->slot_of[b];
  std::fill(m.begin(), m.end(), -1);
  const auto& v = s->ids[b];
  for (size_t l = 0; l < v.size(); l++) {
    if (v[l] < 0) continue;
    if ((size_t)v[l] >= m.size()) m.resize((size_t)v[l] + 1, -1);
    m[(size_t)v[l]] = (int32_t)l;
  }
}
void push_feature(viekf_seq* s, int b) {                           // vi_ekf_feat.cpp:29-30: the filter numbers features itself
  const int32_t gid = s->next_id[b];
  s->ids[b].push_back(gid);
  s->next_id[b] += 1;
  auto& m = s->slot_of[b];
  if ((size_t)gid >= m.size()) m.resize((size_t)gid + 1, -1);
  m[(size_t)gid] = (int32_t)s->ids[b].size() - 1;
}

// ---- log writer (src/vi_ekf/vi_ekf_log.cpp) ------------------------------------------------------------------------------
enum { kTotalMeas = 10, LOG_STATE = kTotalMeas, LOG_COV, LOG_FEATURE_IDS, LOG_INPUT, LOG_XDOT, LOG_GLOBAL, LOG_CONF, LOG_KF,
       LOG_DEBUG, LOG_GLOBAL_POSE, TOTAL_LOGS };                   // include/vi_ekf.h:141-153
const char* const kMeasNames[kTotalMeas] = {"ACC", "ALT", "ATT", "POS", "VEL", "QZETA", "FEAT", "PIXEL_VEL", "DEPTH",
                                            "INV_DEPTH"};           // include/vi_ekf.h:340-354

void wr(std::ofstream& f, const double* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(double) * n); }

// log_state, vi_ekf_log.cpp:6-35: (t, x), (t, diag P), (t, u), (t, dx), (t, ids), (t, pose) of the logged filter.
// x / Pd / xdot are whole-batch arrays [B][nx] / [B][n] / [B][n]; ub is the logged filter's ROTATED input (or NULL = zeros).
void log_state(viekf_seq* s, double t, const std::vector<double>& x, const std::vector<double>& Pd, const double* ub,
               const double* xdot) {
  const int f = s->log_filter, nx = 17 + 5 * s->N, n = 16 + 3 * s->N;
  const double zeros6[6] = {0, 0, 0, 0, 0, 0};
  wr(s->log[LOG_STATE], &t, 1); wr(s->log[LOG_STATE], x.data() + (size_t)f * nx, nx);
  wr(s->log[LOG_COV], &t, 1); wr(s->log[LOG_COV], Pd.data() + (size_t)f * n, n);
  wr(s->log[LOG_INPUT], &t, 1); wr(s->log[LOG_INPUT], ub ? ub : zeros6, 6);
  wr(s->log[LOG_XDOT], &t, 1);
  if (xdot) wr(s->log[LOG_XDOT], xdot + (size_t)f * n, n);
  else { std::vector<double> z((size_t)n, 0.0); wr(s->log[LOG_XDOT], z.data(), n); }
  wr(s->log[LOG_FEATURE_IDS], &t, 1);
  for (int i = 0; i < s->N; i++) {
    const double idd = i < (int)s->ids[f].size() ? (double)s->ids[f][i] : -1.0;
    wr(s->log[LOG_FEATURE_IDS], &idd, 1);
  }
  // get_global_pose() = current_node_global_pose_ * (p, q)   (vi_ekf_log.cpp:33-34, vi_ekf_kfr.cpp:14-21)
  const double* xf = x.data() + (size_t)f * nx;
  const double rel[7] = {xf[0], xf[1], xf[2], xf[6], xf[7], xf[8], xf[9]};
  double gp[7];
  xform_compose(s->node.data() + 7 * (size_t)f, rel, gp);
  wr(s->log[LOG_GLOBAL_POSE], &t, 1); wr(s->log[LOG_GLOBAL_POSE], gp, 3); wr(s->log[LOG_GLOBAL_POSE], gp + 3, 4);
}

int fetch_state_and_diag(viekf_seq* s, std::vector<double>& x, std::vector<double>& Pd) {
  x.resize((size_t)s->B * (17 + 5 * s->N));
  Pd.resize((size_t)s->B * (16 + 3 * s->N));
  if (int rc = viekf_batch_get_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  return viekf_batch_get_cov_diag(s->core, Pd.data(), VIEKF_HOST);
}

// Eigen's default stream format of a row vector: coefficients right-aligned to the widest one, separated by one space
std::string eigen_row(const double* v, int n) {
  std::vector<std::string> t(n);
  size_t w = 0;
  for (int i = 0; i < n; i++) { std::ostringstream o; o << v[i]; t[i] = o.str(); w = std::max(w, t[i].size()); }
  std::string r;
  for (int i = 0; i < n; i++) { if (i) r += " "; r += std::string(w - t[i].size(), ' ') + t[i]; }
  return r;
}

// Ring slot `ip` (the oldest state of the ring) is about to be overwritten.  If it is the slot a fused replay started from and the
// replay's next step is still unmaterialised, its buffer leaves the ring as the orphan (with the list of those steps and their
// inputs) and the slot gets the buffer that was outside: a later rewind into the span can still be served (rewind_to).
void before_overwrite(viekf_seq* s, int ip) {
  const int nxt = (ip + 1) % s->H;
  if (!s->mat[ip] || s->mat[nxt]) return;
  s->orph_steps.clear();
  for (int q = nxt; !s->mat[q] && q != ip; q = (q + 1) % s->H) s->orph_steps.push_back({q, s->t[q], s->slot_u[q]});
  s->orph = true;
  s->orph_t = s->t[ip];
  std::swap(s->extra, s->phys[ip]);
  TR("  orphan: slot %d (t %.4f) leaves the ring, %zu steps kept, its buffer %d, slot gets %d\n", ip, s->orph_t, s->orph_steps.size(), s->extra, s->phys[ip]);
}

// numeric core of propagate_state (vi_ekf.cpp:291-311) with ring bookkeeping; `u` is what the caller hands to
// propagate_state (the batch rotates it by q_b_u itself, :265-267)
int propagate_core(viekf_seq* s, const double* u, double t, bool save_input) {
  const int B = s->B;
  if (save_input) {
    std::vector<double> ub((size_t)B * 6);
    for (int b = 0; b < B; b++) {
      rota(s->prm.q_b_u, u + 6 * b, ub.data() + 6 * b);
      rota(s->prm.q_b_u, u + 6 * b + 3, ub.data() + 6 * b + 3);
    }
    s->u.emplace_front(t, std::move(ub));                          // :269-272 (the ROTATED input is stored)
  }
  if (std::isnan(s->start_t)) {                                    // :274-279
    s->start_t = t;
    s->t[s->i] = t;
    return VIEKF_OK;
  }
  const double dt = t - s->t[s->i];
  if (std::fabs(dt) < 1e-6) return VIEKF_OK;                       // :281-283
  if (dt < 0) return VIEKF_OK;                                     // :285-289 ("I won't let you")
  std::vector<double> dts((size_t)B, dt);
  const bool logging = save_input && !s->log.empty();              // :316-317 (replays are not logged)
  std::vector<double> xdot;
  if (logging) {                                                   // dx_ belongs to the state BEFORE the step (:293)
    xdot.resize((size_t)B * (16 + 3 * s->N));
    if (int rc = viekf_batch_eval_xdot(s->core, u, xdot.data(), VIEKF_HOST)) return rc;
  }
  const int ip = (s->i + 1) % s->H;                                // :298: x_[ip], P_[ip] are written from x_[i_], P_[i_] --
  before_overwrite(s, ip);
  TR("  prop: slot %d (buf %d, t %.4f) -> slot %d (buf %d, t %.4f)\n", s->i, s->phys[s->i], s->t[s->i], ip, s->phys[ip], t);
  if (int rc = viekf_batch_propagate_to(s->core, u, dts.data(), s->phys[ip], VIEKF_HOST)) return rc;   // the old slot stays as history
  s->i = ip;                                                       // :306
  s->t[s->i] = t;
  s->mat[s->i] = 1;
  if (logging) {
    std::vector<double> x, Pd;
    if (int rc = fetch_state_and_diag(s, x, Pd)) return rc;
    log_state(s, t, x, Pd, s->u.front().second.data() + 6 * (size_t)s->log_filter, xdot.data());
  }
  return VIEKF_OK;
}

// The replay that closes handle_measurements (vi_ekf_meas.cpp:106-118): propagate_state(u, t, false) for the stored inputs
// u[from], ..., u[0], each to its own time stamp -- as ONE call where the core fuses it (P stays on chip through the whole replay and
// only the last ring slot is written).  The slots in between keep their time stamps and are marked as not holding their state;
// materialize() re-creates one from the nearest slot that does if a later, late measurement rewinds into the span.
int replay_inputs(viekf_seq* s, size_t from) {
  const int B = s->B;
  std::vector<size_t> idx;
  std::vector<double> dtv;
  double tprev = s->t[s->i];
  for (size_t k = from + 1; k-- > 0;) {
    const double dt = s->u[k].first - tprev;
    if (std::fabs(dt) < 1e-6 || dt < 0) continue;                  // propagate_core makes no step then (vi_ekf.cpp:281-289)
    idx.push_back(k); dtv.push_back(dt); tprev = s->u[k].first;
  }
  size_t done = 0;
  while (done < idx.size()) {
    const int K = (int)std::min<size_t>(std::min<size_t>(64, idx.size() - done), (size_t)std::max(1, s->H - 1));
    if (K == 1) {
      if (int rc = propagate_core(s, s->u[idx[done]].second.data(), s->u[idx[done]].first, false)) return rc;
      done++;
      continue;
    }
    std::vector<double> U((size_t)K * B * 6), DT((size_t)K * B);
    std::vector<int32_t> slots(K);
    std::vector<int32_t> pslots(K);
    for (int k = 0; k < K; k++) {
      std::memcpy(U.data() + (size_t)k * B * 6, s->u[idx[done + k]].second.data(), sizeof(double) * 6 * (size_t)B);
      std::fill(DT.begin() + (size_t)k * B, DT.begin() + (size_t)(k + 1) * B, dtv[done + k]);
      slots[k] = (s->i + 1 + k) % s->H;
      before_overwrite(s, slots[k]);
      pslots[k] = s->phys[slots[k]];
    }
    int32_t written = 1;
    TR("  replay: %d steps from slot %d (buf %d, t %.4f) into slots %d.. (last buf %d)\n", K, s->i, s->phys[s->i], s->t[s->i], slots[0], pslots[K - 1]);
    if (int rc = viekf_batch_propagate_n_to(s->core, K, U.data(), DT.data(), pslots.data(), &written, VIEKF_HOST)) return rc;
    for (int k = 0; k < K; k++) {
      s->t[slots[k]] = s->u[idx[done + k]].first;
      s->mat[slots[k]] = (written || k == K - 1) ? 1 : 0;
      if (!s->mat[slots[k]]) s->slot_u[slots[k]] = s->u[idx[done + k]].second;   // (what re-creating this step later needs)
    }
    s->i = slots[K - 1];
    done += (size_t)K;
  }
  return VIEKF_OK;
}

// makes ring slot `target` the live state (the rewind, vi_ekf_meas.cpp:50-52).  A slot inside a fused replay is first re-created:
// from the nearest older slot that holds its state, forward step by step with the inputs those steps were made with -- or, when that
// slot has been overwritten since (the ring wrapped past it), from the orphan buffer that left the ring then (before_overwrite).
// *found = false: the state cannot be re-created (not reached with the bookkeeping above; the caller then treats the measurement
// as older than the state history, vi_ekf_meas.cpp:59-64, instead of failing on it for ever).
int rewind_to(viekf_seq* s, int target, bool* found) {
  const int H = s->H, B = s->B;
  *found = true;
  TR("rewind to slot %d (t %.4f, mat %d) from slot %d\n", target, s->t[target], (int)s->mat[target], s->i);
  if (s->mat[target]) {
    if (target != s->i) {
      if (int rc = viekf_batch_select(s->core, s->phys[target])) return rc;
      s->i = target;
    }
    return VIEKF_OK;
  }
  int jv = target, back = 0;
  do { jv = (jv + H - 1) % H; back++; } while (!s->mat[jv] && back < H);
  if (s->mat[jv] && s->t[jv] < s->t[target]) {                     // (an overwritten start shows up as a NEWER state here)
    if (jv != s->i) {
      if (int rc = viekf_batch_select(s->core, s->phys[jv])) return rc;
      s->i = jv;
    }
    for (int q = 1; q <= back; q++) {
      const int slot = (jv + q) % H;
      if (s->slot_u[slot].size() != (size_t)B * 6) return VIEKF_ERR_INVALID;
      const std::vector<double> uq = s->slot_u[slot];              // (a copy: propagate_core may not touch slot_u, but stay safe)
      if (int rc = propagate_core(s, uq.data(), s->t[slot], false)) return rc;   // writes slot (i + 1) % H = `slot`, marks it
      if (s->i != slot) return VIEKF_ERR_INVALID;
    }
    return VIEKF_OK;
  }
  // the orphan: steps [0, idx] of its list lead to the target
  int idx = -1;
  if (s->orph)
    for (size_t k = 0; k < s->orph_steps.size(); k++)
      if (s->orph_steps[k].slot == target && s->orph_steps[k].t == s->t[target]) { idx = (int)k; break; }
  if (idx < 0) { *found = false; return VIEKF_OK; }
  // ping-pong between the target's buffer and a second free one: another still unmaterialised step's, or the orphan buffer itself
  // once it has been read (nothing else can need it then: no other step of its replay is left in the ring)
  int owner = -1;
  for (size_t k = 0; k < s->orph_steps.size() && owner < 0; k++) {
    const auto& st = s->orph_steps[k];
    if ((int)k != idx && !s->mat[st.slot] && s->t[st.slot] == st.t) owner = st.slot;
  }
  const int bufA = s->phys[target], bufB = owner >= 0 ? s->phys[owner] : s->extra;
  if (int rc = viekf_batch_select(s->core, s->extra)) return rc;
  double tprev = s->orph_t;
  std::vector<double> dts((size_t)B);
  int last = bufA;
  for (int k = 0; k <= idx; k++) {
    const auto& st = s->orph_steps[(size_t)k];
    if (st.u.size() != (size_t)B * 6) return VIEKF_ERR_INVALID;
    std::fill(dts.begin(), dts.end(), st.t - tprev);
    last = (k % 2 == 0) ? bufA : bufB;
    if (int rc = viekf_batch_propagate_to(s->core, st.u.data(), dts.data(), last, VIEKF_HOST)) return rc;
    tprev = st.t;
  }
  if (owner < 0 && idx >= 1) s->orph = false;                      // (the orphan buffer was written by step 1)
  if (last != bufA) {                                              // the state ended in the other buffer: the two trade places
    if (owner >= 0) std::swap(s->phys[target], s->phys[owner]);
    else std::swap(s->phys[target], s->extra);
  }
  s->mat[target] = 1;
  s->i = target;
  return VIEKF_OK;
}

int update_entry(viekf_seq* s, SeqMeas& m, std::vector<int32_t>& res) {   // VIEKF::update, vi_ekf_meas.cpp:196-278
  const int B = s->B;
  m.handled = true;                                                // :198
  res.assign(B, VIEKF_MEAS_SKIPPED);
  std::vector<int32_t> slot(B, -1);
  const bool needs_slot = m.type == VIEKF_QZETA || m.type == VIEKF_FEAT || m.type == VIEKF_DEPTH || m.type == VIEKF_INV_DEPTH;
  std::vector<uint8_t> act(B);
  for (int b = 0; b < B; b++) {
    if (needs_slot) slot[b] = m.present[b] ? local_id(s, b, m.id[b]) : -1;
    act[b] = m.present[b] ? (m.active ? 1 : 0) : 2;                // 2 = this filter skips the entry altogether
  }
  const int lf = s->log_filter;
  const bool logging = !s->log.empty() && m.present[lf] && m.type >= 0 && m.type < kTotalMeas;
  std::vector<double> zhat;
  if (logging) {                                                   // zhat_ = h(x) at the state the update starts from (:201-207)
    zhat.assign((size_t)B * 4, 0.0);
    if (int rc = viekf_batch_eval_h(s->core, m.type, needs_slot ? slot.data() : nullptr, zhat.data(), VIEKF_HOST)) return rc;
  }
  int rc;
  if (m.type == VIEKF_FEAT && m.active) {
    rc = viekf_batch_update_feat(s->core, m.z.data(), slot.data(), 1, m.R.data(), 0, res.data(), VIEKF_HOST);
  } else {
    rc = viekf_batch_update(s->core, m.type, m.z.data(), m.zdim, m.R.data(), m.rdim, 0, needs_slot ? slot.data() : nullptr,
                            act.data(), res.data(), VIEKF_HOST);
  }
  // log_measurement, vi_ekf_log.cpp:52-67, called at the END of update() (vi_ekf_meas.cpp:276): a gated update returned before
  if (rc == VIEKF_OK && logging && res[lf] != VIEKF_MEAS_GATED && res[lf] != VIEKF_MEAS_INVALID && res[lf] != VIEKF_MEAS_NAN) {
    std::ofstream& f = s->log[m.type];
    const double tt = s->t[s->i] - s->start_t;
    wr(f, &tt, 1);
    wr(f, m.z.data() + (size_t)lf * m.zdim, m.zdim);
    wr(f, zhat.data() + (size_t)lf * 4, m.zdim);
    const double ac = m.active ? 1.0 : 0.0;
    wr(f, &ac, 1);
    if (needs_slot || m.type == 7) { const double idd = (double)m.id[lf]; wr(f, &idd, 1); }
  }
  return rc;
}

// entries of one camera frame: unhandled active FEAT measurements with the same time stamp and the same R
bool frame_mate(const SeqMeas& a, const SeqMeas& z) {
  return a.count == 1 && !a.handled && a.type == VIEKF_FEAT && a.active && a.t == z.t && a.zdim == z.zdim && a.rdim == z.rdim && a.R == z.R;
}

// the k entries zbuf[zi], zbuf[zi-1], ..., zbuf[zi-k+1] (the order handle_measurements visits them) as one launch;
// res [B][k]
int update_frame(viekf_seq* s, long zi, int k, std::vector<int32_t>& res) {
  const int B = s->B;
  std::vector<double> z((size_t)B * k * 2);
  std::vector<int32_t> slot((size_t)B * k, -1);
  for (int j = 0; j < k; j++) {
    SeqMeas& m = s->zbuf[zi - j];
    m.handled = true;                                              // :198
    for (int b = 0; b < B; b++) {
      z[((size_t)b * k + j) * 2] = m.z[(size_t)b * 2];
      z[((size_t)b * k + j) * 2 + 1] = m.z[(size_t)b * 2 + 1];
      slot[(size_t)b * k + j] = m.present[b] ? local_id(s, b, m.id[b]) : -1;
    }
  }
  res.assign((size_t)B * k, VIEKF_MEAS_SKIPPED);
  return viekf_batch_update_feat(s->core, z.data(), slot.data(), k, s->zbuf[zi].R.data(), 0, res.data(), VIEKF_HOST);
}

// a frame block (SeqMeas::count > 1) as one launch; res [B][count] (NULL: nobody asked for the result codes -- the launch is then
// queued without waiting for it)
int update_block(viekf_seq* s, SeqMeas& m, std::vector<int32_t>* res) {
  const int B = s->B, k = m.count;
  m.handled = true;                                                // :198
  std::vector<int32_t> slot((size_t)B * k, -1);
  for (int b = 0; b < B; b++)
    for (int j = 0; j < k; j++) {
      const size_t e = (size_t)b * k + j;
      slot[e] = m.present[e] ? local_id(s, b, m.id[e]) : -1;
    }
  if (res) res->assign((size_t)B * k, VIEKF_MEAS_SKIPPED);
  return viekf_batch_update_feat(s->core, m.z.data(), slot.data(), k, m.R.data(), 0, res ? res->data() : nullptr, VIEKF_HOST);
}

// ---- independent clocks --------------------------------------------------------------------------------------------------
// propagate_state's bookkeeping for ONE filter (vi_ekf.cpp:262-318); returns true and the interval if the numeric core is to
// run -- the caller batches those -- and advances the filter's ring position then.
bool plan_propagate(viekf_seq* s, int b, const double* u_in, double t, bool save_input, double* dt_out, int* dst_slot) {
  FilterSeq& f = s->fs[b];
  if (save_input) {
    FInput in;
    in.t = t;
    rota(s->prm.q_b_u, u_in, in.u);
    rota(s->prm.q_b_u, u_in + 3, in.u + 3);
    f.u.push_front(in);                                            // :269-272 (the ROTATED input is stored)
  }
  if (std::isnan(f.start_t)) { f.start_t = t; f.t[f.i] = t; return false; }   // :274-279
  const double dt = t - f.t[f.i];
  if (std::fabs(dt) < 1e-6 || dt < 0) return false;                // :281-289
  const int ip = (f.i + 1) % s->H;                                 // :298
  f.i = ip;                                                        // :306
  f.t[ip] = t;
  *dt_out = dt;
  *dst_slot = ip;
  return true;
}

// handle_measurements of ONE filter (vi_ekf_meas.cpp:6-127) as a list of device steps; the host bookkeeping (ring position,
// times, handled flags, queue trims) is done here, the steps run afterwards, batched over the filters
void plan_handle(viekf_seq* s, int b, std::vector<SeqOp>& ops) {
  FilterSeq& f = s->fs[b];
  auto prop = [&](const double* ur, double t) {                     // propagate_state(u, t, false): replays are not stored again
    SeqOp op;
    op.kind = SeqOp::PROP;
    if (plan_propagate(s, b, ur, t, false, &op.dt, &op.slot)) {
      std::memcpy(op.u, ur, sizeof op.u);
      ops.push_back(std::move(op));
    }
  };
  if (f.zbuf.empty() || f.u.empty()) return;                       // :12-13
  long zi = (long)f.zbuf.size() - 1;                               // :16-18 oldest unhandled
  while (f.zbuf[zi].handled && zi != 0) zi--;
  if (zi == 0 && f.zbuf[zi].handled) return;                       // :21-22
  if (f.zbuf[zi].t > f.u[0].t) return;                             // :24-28 from the future
  size_t ui = 0;                                                   // :32-38 input just before the measurement
  while (ui != f.u.size()) {
    if (f.zbuf[zi].t > f.u[ui].t) break;
    ui++;
  }
  if (ui == f.u.size() || f.zbuf[zi].t <= f.u[ui].t) return;       // :39-43 not enough input history
  int k = s->H, target = -1;                                       // :46-57 rewind
  while (k > 0) {
    const int j = (f.i + k) % s->H;
    if (f.t[j] <= f.u[ui].t) { target = j; break; }
    k--;
  }
  if (k == 0) { f.queued -= f.zbuf[zi].weight; f.zbuf.erase(f.zbuf.begin() + zi); return; }   // :59-64 not enough state history
  if (target != f.i) {
    SeqOp op;
    op.kind = SeqOp::REWIND;
    op.slot = target;
    ops.push_back(std::move(op));
    f.i = target;
  }
  auto mate = [](const FMeas& a, const FMeas& z) {                 // single entries of one camera frame (see update_frame)
    return !a.frame && !a.handled && a.type == VIEKF_FEAT && a.active && a.t == z.t && a.zdim == z.zdim && a.rdim == z.rdim &&
           std::memcmp(a.R, z.R, sizeof(double) * 4) == 0;
  };
  ui--;                                                            // :74
  while (ui != 0) {                                                // :75
    bool left_inner_by_break = false;
    while (f.zbuf[zi].t <= f.u[ui].t) {                            // :78
      FMeas& z = f.zbuf[zi];
      if (f.t[f.i] < z.t) prop(f.u[ui].u, z.t);                    // :81-82
      if (!z.handled) {                                            // :87-95
        SeqOp op;
        op.slot = f.i;   // (an update changes x_[i_], P_[i_] in place, vi_ekf_meas.cpp:254-271: that slot IS the filter's live state)
        if (z.frame) {   // a frame block: its members in one launch, nothing copied
          op.kind = SeqOp::FEAT_FRAME;
          op.frame = z.frame;
          op.first = z.first;
          z.handled = true;                                        // :198
        } else {
          long zl = zi;
          if (z.type == VIEKF_FEAT && z.active)
            while (zl > 0 && mate(f.zbuf[zl - 1], z)) zl--;
          op.kind = (z.type == VIEKF_FEAT && z.active) ? SeqOp::FEAT_FRAME : SeqOp::GENERIC;
          for (long q = zi; q >= zl; q--) { f.zbuf[q].handled = true; op.meas.push_back(f.zbuf[q]); }   // :198
          zi = zl;
        }
        ops.push_back(std::move(op));
      }
      if (zi != 0) {                                               // :97-105
        zi--;
        while (f.u[ui].t < f.zbuf[zi].t && ui != 0) { prop(f.u[ui].u, f.u[ui].t); ui--; }
      } else {                                                     // :106-115
        while (ui != 0) { prop(f.u[ui].u, f.u[ui].t); ui--; }
        left_inner_by_break = true;
        break;
      }
    }
    if (!left_inner_by_break) break;
  }
  prop(f.u[ui].u, f.u[ui].t);                                      // :118
  while (f.queued > s->MH && !f.zbuf.empty()) {                    // :121-122: single measurements leave from the old end; a frame
    FMeas& e = f.zbuf.back();                                      // block gives up the members it would consume first
    const long drop = f.queued - s->MH;
    if (drop >= e.weight) { f.queued -= e.weight; f.zbuf.pop_back(); continue; }
    const uint8_t* pr = e.frame->present.data() + (size_t)b * e.frame->count;
    long left = drop;
    while (left > 0 && e.first < e.frame->count) { if (pr[e.first]) left--; e.first++; }
    e.weight -= (int)drop;
    f.queued -= drop;
  }
  while ((int)f.u.size() > s->H) f.u.pop_back();                   // :125-126
}

// VIEKF::keyframe_reset(), src/vi_ekf/vi_ekf_kfr.cpp:56-157, for the filters with reset[b] != 0: the device resets state and
// covariance and hands back the edge; the node frame moves here (:147-150).  edges [B][17] may be NULL.
int reset_and_move_node(viekf_seq* s, const std::vector<uint8_t>& reset, double* edges) {
  const int B = s->B;
  std::vector<double> eb((size_t)B * 17, 0.0);
  if (int rc = viekf_batch_keyframe_reset(s->core, reset.data(), eb.data(), VIEKF_HOST)) return rc;
  for (int b = 0; b < B; b++) {
    if (!reset[b]) continue;
    const double* e = eb.data() + 17 * (size_t)b;                // {t(3), q_yaw(4), cov_pos(9), cov_yaw}
    double C[36] = {};                                           // edge.cov: position block and the yaw variance (:59-63,126)
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) C[r + 6 * c] = e[7 + r + 3 * c];
    C[5 + 6 * 5] = e[16];
    double* nd = s->node.data() + 7 * (size_t)b;
    add_adj_cov(nd, C, s->node_cov.data() + 36 * (size_t)b);      // :149 (with the node pose BEFORE it moves)
    double nn[7];
    xform_compose(nd, e, nn);                                    // :150
    std::memcpy(nd, nn, sizeof nn);
  }
  if (edges) std::memcpy(edges, eb.data(), sizeof(double) * 17 * (size_t)B);
  return VIEKF_OK;
}

// independent clocks: the reference changes x_[i_], P_[i_] in place (updates, init_feature, clear_feature, keyframe reset), and
// that slot is what a later rewind finds
int refresh_slots(viekf_seq*, const std::vector<uint8_t>&) {
  // (r04: every filter's live state IS its ring slot -- viekf_batch_select_filters -- so an in-place change is already there)
  return VIEKF_OK;
}

// runs the planned steps: per round the next step of every filter, one masked launch per kind of step
int run_ops(viekf_seq* s, std::vector<std::vector<SeqOp>>& ops, std::vector<std::vector<int32_t>>& gated, bool want_gated) {
  const int B = s->B;
  std::vector<size_t> head(B, 0);
  std::vector<uint8_t> mask(B);
  std::vector<int32_t> slot(B);
  for (;;) {
    bool any = false;
    for (int b = 0; b < B; b++) any |= head[b] < ops[b].size();
    if (!any) break;
    auto group = [&](SeqOp::Kind k, int gtype) {                   // filters whose next step is of this kind (and type)
      bool got = false;
      for (int b = 0; b < B; b++) {
        const bool in = head[b] < ops[b].size() && ops[b][head[b]].kind == k &&
                        (k != SeqOp::GENERIC || ops[b][head[b]].meas[0].type == gtype);
        mask[b] = in ? 1 : 0;
        got |= in;
      }
      return got;
    };
    std::vector<uint8_t> done(B, 0);
    if (group(SeqOp::REWIND, 0)) {
      for (int b = 0; b < B; b++) slot[b] = mask[b] ? ops[b][head[b]].slot : -1;
      if (int rc = viekf_batch_select_filters(s->core, slot.data())) return rc;   // the rewind (:50-52): an index per filter
      for (int b = 0; b < B; b++) done[b] |= mask[b];
    }
    if (group(SeqOp::PROP, 0)) {
      std::vector<double> u((size_t)B * 6, 0.0), dt(B, 0.0);
      for (int b = 0; b < B; b++) {
        if (!mask[b] || done[b]) { mask[b] = 0; continue; }
        std::memcpy(u.data() + 6 * (size_t)b, ops[b][head[b]].u, sizeof(double) * 6);
        dt[b] = ops[b][head[b]].dt;
        slot[b] = ops[b][head[b]].slot;
      }
      for (int b = 0; b < B; b++) if (!mask[b]) slot[b] = -1;
      // x_[ip], P_[ip] written from x_[i_], P_[i_] (vi_ekf.cpp:298-306) for every filter that steps: slot i_b -> slot i_b + 1
      if (int rc = viekf_batch_propagate_filters_to(s->core, u.data(), dt.data(), slot.data(), VIEKF_HOST)) return rc;
      for (int b = 0; b < B; b++) done[b] |= mask[b];
    }
    if (group(SeqOp::FEAT_FRAME, 0)) {
      int M = 0;
      for (int b = 0; b < B; b++) {
        if (done[b]) mask[b] = 0;
        if (mask[b]) M = std::max(M, ops[b][head[b]].count());
      }
      if (M > 0) {
        // R may differ between filters: r_mode 1 (one R per filter)
        std::vector<double> z((size_t)B * M * 2, 0.0), R((size_t)B * 4, 0.0);
        std::vector<int32_t> sl((size_t)B * M, -1), res((size_t)B * M, VIEKF_MEAS_SKIPPED);
        for (int b = 0; b < B; b++) {
          if (!mask[b]) continue;
          const SeqOp& op = ops[b][head[b]];
          if (op.frame) {
            const FrameData& fd = *op.frame;
            const int cnt = fd.count - op.first;
            const size_t row = (size_t)b * fd.count + op.first;
            std::memcpy(R.data() + 4 * (size_t)b, fd.R, sizeof(double) * 4);
            std::memcpy(z.data() + (size_t)b * M * 2, fd.z.data() + row * 2, sizeof(double) * 2 * (size_t)cnt);
            for (int j = 0; j < cnt; j++) sl[(size_t)b * M + j] = fd.present[row + j] ? local_id(s, b, fd.id[row + j]) : -1;
            continue;
          }
          std::memcpy(R.data() + 4 * (size_t)b, op.meas[0].R, sizeof(double) * 4);
          for (size_t j = 0; j < op.meas.size(); j++) {
            z[((size_t)b * M + j) * 2] = op.meas[j].z[0];
            z[((size_t)b * M + j) * 2 + 1] = op.meas[j].z[1];
            sl[(size_t)b * M + j] = local_id(s, b, op.meas[j].id);
          }
        }
        if (int rc = viekf_batch_set_active(s->core, mask.data(), VIEKF_HOST)) return rc;
        // (nobody asked for the gated ids: no result codes, so the launch is queued and not waited for)
        int rc = viekf_batch_update_feat(s->core, z.data(), sl.data(), M, R.data(), 1, want_gated ? res.data() : nullptr, VIEKF_HOST);
        (void)viekf_batch_set_active(s->core, nullptr, VIEKF_HOST);
        if (rc) return rc;
        for (int b = 0; b < B; b++) {
          if (!mask[b]) continue;
          const SeqOp& op = ops[b][head[b]];
          for (int j = 0; want_gated && j < op.count(); j++)
            if (res[(size_t)b * M + j] == VIEKF_MEAS_GATED)
              gated[b].push_back(op.frame ? op.frame->id[(size_t)b * op.frame->count + op.first + j] : op.meas[(size_t)j].id);
          done[b] = 1;
        }
      }
    }
    for (int type = 0; type < VIEKF_TOTAL_MEAS; type++) {
      if (!group(SeqOp::GENERIC, type)) continue;
      int zdim = 0, rdim = 0;
      for (int b = 0; b < B; b++) {
        if (done[b]) mask[b] = 0;
        if (mask[b]) { zdim = ops[b][head[b]].meas[0].zdim; rdim = ops[b][head[b]].meas[0].rdim; }
      }
      if (zdim == 0) continue;
      std::vector<double> z((size_t)B * zdim, 0.0), R((size_t)B * rdim * rdim, 0.0);
      std::vector<int32_t> sl(B, -1), res(B, VIEKF_MEAS_SKIPPED);
      std::vector<uint8_t> act(B, 2);                              // 2 = this filter takes no part
      const bool needs_slot = type == VIEKF_QZETA || type == VIEKF_FEAT || type == VIEKF_DEPTH || type == VIEKF_INV_DEPTH;
      for (int b = 0; b < B; b++) {
        if (!mask[b]) continue;
        const FMeas& m = ops[b][head[b]].meas[0];
        if (m.zdim != zdim || m.rdim != rdim) { mask[b] = 0; continue; }   // (a later round takes it)
        std::memcpy(z.data() + (size_t)b * zdim, m.z, sizeof(double) * zdim);
        std::memcpy(R.data() + (size_t)b * rdim * rdim, m.R, sizeof(double) * rdim * rdim);
        if (needs_slot) sl[b] = local_id(s, b, m.id);
        act[b] = m.active ? 1 : 0;
      }
      if (int rc = viekf_batch_update(s->core, type, z.data(), zdim, R.data(), rdim, 1, needs_slot ? sl.data() : nullptr, act.data(),
                                      res.data(), VIEKF_HOST))
        return rc;
      for (int b = 0; b < B; b++) {
        if (!mask[b]) continue;
        if (type == VIEKF_FEAT && res[b] == VIEKF_MEAS_GATED) gated[b].push_back(ops[b][head[b]].meas[0].id);
        done[b] = 1;
      }
    }
    for (int b = 0; b < B; b++) head[b] += done[b];
  }
  return VIEKF_OK;
}

}  // namespace

extern "C" {

int viekf_seq_create(viekf_batch* core, int32_t state_hist, int32_t meas_hist, viekf_seq** out) {
  if (!core || !out || state_hist < 2 || meas_hist < 1) return VIEKF_ERR_INVALID;
  viekf_seq* s = new viekf_seq;
  s->core = core;
  int32_t B, N, nx, n;
  if (int rc = viekf_batch_dims(core, &B, &N, &nx, &n)) { delete s; return rc; }
  if (int rc = viekf_batch_get_params(core, &s->prm)) { delete s; return rc; }
  if (int rc = viekf_batch_history_resize(core, state_hist + 1)) { delete s; return rc; }   // (+ 1: the buffer outside the ring)
  if (int rc = viekf_batch_snapshot(core, 0)) { delete s; return rc; }     // the live state moves into ring slot i_ = 0
  if (int rc = viekf_batch_select(core, 0)) { delete s; return rc; }
  s->B = B; s->N = N; s->H = state_hist; s->MH = meas_hist;
  s->t.assign(state_hist, NAN);                                    // vi_ekf.cpp:22-27
  s->mat.assign(state_hist, 1);
  s->phys.resize(state_hist);
  for (int k = 0; k < state_hist; k++) s->phys[k] = k;
  s->extra = state_hist;
  s->slot_u.assign(state_hist, {});
  s->ids.assign(B, {});
  s->slot_of.assign(B, {});
  s->next_id.assign(B, 0);
  s->kf_feats.assign(B, {});
  s->node.assign((size_t)B * 7, 0.0);                              // Xformd::Identity(), vi_ekf.cpp:38
  for (int b = 0; b < B; b++) s->node[7 * (size_t)b + 3] = 1.0;
  s->node_cov.assign((size_t)B * 36, 0.0);                         // vi_ekf.cpp:39
  // input-only device steps (propagates, updates whose result codes nobody asked for) are queued, not waited for: a replay of k
  // propagates is k launches back to back; every call that returns data to the host still synchronises
  if (int rc = viekf_batch_set_async(core, 1)) { delete s; return rc; }
  *out = s;
  return VIEKF_OK;
}

int viekf_seq_destroy(viekf_seq* s) {
  if (s && s->core) {   // hand the live state back to the batch's own buffers
    (void)viekf_batch_set_async(s->core, 0);
    (void)viekf_batch_history_resize(s->core, 0);
  }
  delete s;
  return VIEKF_OK;
}

int viekf_seq_create_independent(viekf_batch* core, int32_t state_hist, int32_t meas_hist, viekf_seq** out) {
  if (!core || !out || state_hist < 2 || meas_hist < 1) return VIEKF_ERR_INVALID;
  viekf_seq* s = new viekf_seq;
  s->core = core;
  s->indep = true;
  int32_t B, N, nx, n;
  if (int rc = viekf_batch_dims(core, &B, &N, &nx, &n)) { delete s; return rc; }
  if (int rc = viekf_batch_get_params(core, &s->prm)) { delete s; return rc; }
  if (int rc = viekf_batch_history_resize(core, state_hist)) { delete s; return rc; }
  s->B = B; s->N = N; s->H = state_hist; s->MH = meas_hist;
  s->fs.resize(B);
  for (auto& f : s->fs) f.t.assign(state_hist, NAN);               // vi_ekf.cpp:22-27
  std::vector<int32_t> zero(B, 0);
  if (int rc = viekf_batch_snapshot_filters(core, zero.data(), VIEKF_HOST)) { delete s; return rc; }   // x_[0], P_[0]
  if (int rc = viekf_batch_select_filters(core, zero.data())) { delete s; return rc; }                 // ... ARE the live state
  s->ids.assign(B, {});
  s->slot_of.assign(B, {});
  s->next_id.assign(B, 0);
  s->kf_feats.assign(B, {});
  s->node.assign((size_t)B * 7, 0.0);
  for (int b = 0; b < B; b++) s->node[7 * (size_t)b + 3] = 1.0;
  s->node_cov.assign((size_t)B * 36, 0.0);
  if (int rc = viekf_batch_set_async(core, 1)) { delete s; return rc; }
  *out = s;
  return VIEKF_OK;
}

int viekf_seq_propagate_t(viekf_seq* s, const double* u, const double* t, const uint8_t* mask) {
  if (!s || !u || !t || !s->indep) return VIEKF_ERR_INVALID;
  const int B = s->B;
  std::vector<uint8_t> act(B, 0);
  std::vector<double> dt(B, 0.0);
  std::vector<int32_t> slot(B, -1);
  bool any = false;
  for (int b = 0; b < B; b++) {
    if (mask && !mask[b]) continue;
    if (plan_propagate(s, b, u + 6 * (size_t)b, t[b], true, &dt[b], &slot[b])) { act[b] = 1; any = true; }
  }
  if (!any) return VIEKF_OK;
  return viekf_batch_propagate_filters_to(s->core, u, dt.data(), slot.data(), VIEKF_HOST);   // x_[ip], P_[ip] of every filter that steps
}

int viekf_seq_add_measurement_t(viekf_seq* s, const double* t, int32_t type, const double* z, int32_t zdim, const double* R,
                                int32_t rdim, int32_t active, const int32_t* id, const double* depth, const uint8_t* mask,
                                int32_t* result) {
  if (!s || !t || !z || !R || !s->indep || zdim < 1 || zdim > 4 || rdim < 1 || rdim > 3) return VIEKF_ERR_INVALID;
  const int B = s->B;
  std::vector<int32_t> res(B, VIEKF_MEAS_SKIPPED);
  std::vector<uint8_t> newf(B, 0);
  bool any_new = false;
  for (int b = 0; b < B; b++) {
    if (mask && !mask[b]) continue;
    FilterSeq& f = s->fs[b];
    res[b] = VIEKF_MEAS_SUCCESS;
    if (t[b] < f.start_t) { res[b] = VIEKF_MEAS_INVALID; continue; }                 // :133-134
    bool isnan_ = false;
    for (int k = 0; k < zdim; k++) isnan_ |= std::isnan(z[(size_t)b * zdim + k]);
    if (isnan_) { res[b] = VIEKF_MEAS_NAN; continue; }                                // :136-137
    const int gid = id ? id[b] : -1;
    if (type == VIEKF_FEAT && gid >= 0 && local_id(s, b, gid) < 0) {                  // :140-147
      res[b] = VIEKF_MEAS_NEW_FEATURE;
      if ((int)s->ids[b].size() < s->N) { newf[b] = 1; any_new = true; }             // vi_ekf_feat.cpp:9-10
      continue;
    }
    FMeas m;
    m.t = t[b]; m.type = type; m.zdim = zdim; m.rdim = rdim; m.active = active != 0; m.handled = false; m.id = gid;
    std::memset(m.z, 0, sizeof m.z);
    std::memset(m.R, 0, sizeof m.R);
    std::memcpy(m.z, z + (size_t)b * zdim, sizeof(double) * zdim);
    std::memcpy(m.R, R, sizeof(double) * rdim * rdim);
    size_t k = 0;                                                                    // :150-156
    while (k < f.zbuf.size() && !(f.zbuf[k].t < m.t)) k++;
    f.zbuf.insert(f.zbuf.begin() + (long)k, m);                                      // :169-175
    f.queued += 1;
  }
  if (any_new) {   // init_feature at the CURRENT state (vi_ekf_feat.cpp:6-47); numbered by the filter itself (:29-30)
    std::vector<double> dep(B, NAN);
    if (depth) dep.assign(depth, depth + B);
    std::vector<int32_t> ok(B, 0);
    if (int rc = viekf_batch_init_feature(s->core, z, dep.data(), newf.data(), ok.data(), VIEKF_HOST)) return rc;
    for (int b = 0; b < B; b++)
      if (newf[b] && ok[b]) push_feature(s, b);
    if (int rc = refresh_slots(s, newf)) return rc;
  }
  if (result) std::memcpy(result, res.data(), sizeof(int32_t) * B);
  return VIEKF_OK;
}

int viekf_seq_propagate(viekf_seq* s, const double* u, double t) {
  if (!s || !u) return VIEKF_ERR_INVALID;
  if (s->indep) { std::vector<double> tt(s->B, t); return viekf_seq_propagate_t(s, u, tt.data(), nullptr); }
  return propagate_core(s, u, t, true);
}

int viekf_seq_add_measurement(viekf_seq* s, double t, int32_t type, const double* z, int32_t zdim, const double* R,
                              int32_t rdim, int32_t active, const int32_t* id, const double* depth, int32_t* result) {
  if (!s || !z || !R || zdim < 1 || zdim > 4 || rdim < 1 || rdim > 3) return VIEKF_ERR_INVALID;
  if (s->indep) {
    std::vector<double> tt(s->B, t);
    return viekf_seq_add_measurement_t(s, tt.data(), type, z, zdim, R, rdim, active, id, depth, nullptr, result);
  }
  const int B = s->B;
  std::vector<int32_t> res(B, VIEKF_MEAS_SUCCESS);
  SeqMeas m;
  m.t = t; m.type = type; m.zdim = zdim; m.rdim = rdim; m.active = active != 0; m.handled = false;
  m.z.assign(z, z + (size_t)B * zdim);
  m.R.assign(R, R + (size_t)rdim * rdim);
  m.id.assign(B, -1);
  m.present.assign(B, 1);
  std::vector<uint8_t> newf(B, 0);
  bool any_new = false, any_present = false;
  for (int b = 0; b < B; b++) {
    if (id) m.id[b] = id[b];
    if (t < s->start_t) { res[b] = VIEKF_MEAS_INVALID; m.present[b] = 0; continue; }   // :133-134
    bool isnan_ = false;
    for (int k = 0; k < zdim; k++) isnan_ |= std::isnan(z[(size_t)b * zdim + k]);
    if (isnan_) { res[b] = VIEKF_MEAS_NAN; m.present[b] = 0; continue; }               // :136-137
    if (type == VIEKF_FEAT && m.id[b] >= 0 && local_id(s, b, m.id[b]) < 0) {           // :140-147
      res[b] = VIEKF_MEAS_NEW_FEATURE;
      m.present[b] = 0;
      if ((int)s->ids[b].size() < s->N) { newf[b] = 1; any_new = true; }              // vi_ekf_feat.cpp:9-10
      continue;
    }
    any_present = true;
  }
  if (any_new) {   // init_feature at the CURRENT state (vi_ekf_feat.cpp:6-47); numbered by the filter itself (:29-30)
    std::vector<double> dep(B, NAN);
    if (depth) dep.assign(depth, depth + B);
    std::vector<int32_t> ok(B, 0);
    if (int rc = viekf_batch_init_feature(s->core, z, dep.data(), newf.data(), ok.data(), VIEKF_HOST)) return rc;
    for (int b = 0; b < B; b++)
      if (newf[b] && ok[b]) push_feature(s, b);
    }
  if (any_present) {
    size_t k = 0;                                                                      // :150-156
    while (k < s->zbuf.size() && !(s->zbuf[k].t < t)) k++;
    s->zbuf.insert(s->zbuf.begin() + (long)k, std::move(m));                           // :169-175
  }
  if (result) std::memcpy(result, res.data(), sizeof(int32_t) * B);
  return VIEKF_OK;
}

// A camera frame in ONE call: the `count` FEAT entries of a frame, exactly what `count` calls of add_measurement(t, z_k, FEAT, R,
// active, id_k, depth_k) in the order k = 0 .. count - 1 do (the loop of src/vi_ekf_ros.cpp:288-306 / test/vi_ekf_test.cpp:30-31) --
// without the per-call crossing of the language boundary.  z [B][count][2], id [B][count], depth [B][count] or NULL (NaN),
// t: one stamp, or with t_per_filter [B] (independent clocks) and an optional mask [B]; result [B][count] or NULL.
int viekf_seq_add_frame(viekf_seq* s, double t, const double* t_per_filter, int32_t count, const double* z, const double* R,
                        int32_t active, const int32_t* id, const double* depth, const uint8_t* mask, int32_t* result) {
  if (!s || !z || !R || !id || count < 0) return VIEKF_ERR_INVALID;
  if (t_per_filter && !s->indep) return VIEKF_ERR_INVALID;
  const int B = s->B;
  if (!s->indep && s->log.empty() && active && count > 1) {
    // shared clock, no log writer: ONE queue entry for the frame (SeqMeas::count).  Per feature k = 0 .. count - 1, in order, the
    // tests of add_measurement (vi_ekf_meas.cpp:133-147): before the start, NaN, an unknown id starts a feature at the CURRENT state
    SeqMeas m;
    m.t = t; m.type = VIEKF_FEAT; m.zdim = 2; m.rdim = 2; m.active = true; m.handled = false; m.count = count;
    m.R.assign(R, R + 4);
    m.z.assign((size_t)B * count * 2, 0.0);
    m.id.assign((size_t)B * count, -1);
    m.present.assign((size_t)B * count, 0);
    bool any_present = false;
    std::vector<uint8_t> newf(B);
    std::vector<double> zk((size_t)B * 2), dk((size_t)B, NAN);
    std::vector<int32_t> ok(B);
    for (int k = 0; k < count; k++) {
      const int j = count - 1 - k;                                 // position in processing order (last added is consumed first)
      bool any_new = false;
      for (int b = 0; b < B; b++) {
        const size_t e = (size_t)b * count + k, o = (size_t)b * count + j;
        int32_t r = VIEKF_MEAS_SUCCESS;
        newf[b] = 0;
        if (t < s->start_t) r = VIEKF_MEAS_INVALID;                                   // :133-134
        else if (std::isnan(z[2 * e]) || std::isnan(z[2 * e + 1])) r = VIEKF_MEAS_NAN;   // :136-137
        else if (id[e] >= 0 && local_id(s, b, id[e]) < 0) {                            // :140-147
          r = VIEKF_MEAS_NEW_FEATURE;
          if ((int)s->ids[b].size() < s->N) { newf[b] = 1; any_new = true; }          // vi_ekf_feat.cpp:9-10
        } else {
          m.present[o] = 1;
          any_present = true;
        }
        m.z[2 * o] = z[2 * e]; m.z[2 * o + 1] = z[2 * e + 1];
        m.id[o] = id[e];
        if (result) result[e] = r;
      }
      if (any_new) {   // init_feature at the CURRENT state (vi_ekf_feat.cpp:6-47); numbered by the filter itself (:29-30)
        for (int b = 0; b < B; b++) {
          const size_t e = (size_t)b * count + k;
          zk[2 * (size_t)b] = z[2 * e]; zk[2 * (size_t)b + 1] = z[2 * e + 1];
          dk[b] = depth ? depth[e] : NAN;
        }
        if (int rc = viekf_batch_init_feature(s->core, zk.data(), dk.data(), newf.data(), ok.data(), VIEKF_HOST)) return rc;
        for (int b = 0; b < B; b++)
          if (newf[b] && ok[b]) push_feature(s, b);
      }
    }
    if (any_present) {
      size_t k = 0;                                                                    // :150-156
      while (k < s->zbuf.size() && !(s->zbuf[k].t < t)) k++;
      s->zbuf.insert(s->zbuf.begin() + (long)k, std::move(m));                         // :169-175
    }
    return VIEKF_OK;
  }
  if (s->indep && count > 1 && active) {
    // independent clocks: the same per-feature tests in the same order (viekf_seq_add_measurement_t); the frame's pixels are stored
    // ONCE (FrameData) and every filter that accepted members queues one entry pointing at its row
    auto fd = std::make_shared<FrameData>();
    fd->count = count;
    fd->z.assign((size_t)B * count * 2, 0.0);
    fd->id.assign((size_t)B * count, -1);
    fd->present.assign((size_t)B * count, 0);
    std::memcpy(fd->R, R, sizeof(double) * 4);
    std::vector<int> npres(B, 0);
    std::vector<uint8_t> newf(B);
    std::vector<double> zk((size_t)B * 2), dk((size_t)B, NAN);
    std::vector<int32_t> ok(B);
    for (int k = 0; k < count; k++) {
      const int j = count - 1 - k;                                 // position in processing order (last added is consumed first)
      bool any_new = false;
      for (int b = 0; b < B; b++) {
        const size_t e = (size_t)b * count + k, o = (size_t)b * count + j;
        newf[b] = 0;
        if (result) result[e] = VIEKF_MEAS_SKIPPED;
        if (mask && !mask[b]) continue;
        const FilterSeq& f = s->fs[b];
        const double tb = t_per_filter ? t_per_filter[b] : t;
        int32_t r = VIEKF_MEAS_SUCCESS;
        if (tb < f.start_t) r = VIEKF_MEAS_INVALID;                                   // :133-134
        else if (std::isnan(z[2 * e]) || std::isnan(z[2 * e + 1])) r = VIEKF_MEAS_NAN;   // :136-137
        else if (id[e] >= 0 && local_id(s, b, id[e]) < 0) {                            // :140-147
          r = VIEKF_MEAS_NEW_FEATURE;
          if ((int)s->ids[b].size() < s->N) { newf[b] = 1; any_new = true; }          // vi_ekf_feat.cpp:9-10
        } else {
          fd->present[o] = 1;
          npres[b]++;
        }
        fd->z[2 * o] = z[2 * e]; fd->z[2 * o + 1] = z[2 * e + 1];
        fd->id[o] = id[e];
        if (result) result[e] = r;
      }
      if (any_new) {   // init_feature at the CURRENT state (vi_ekf_feat.cpp:6-47); numbered by the filter itself (:29-30)
        for (int b = 0; b < B; b++) {
          const size_t e = (size_t)b * count + k;
          zk[2 * (size_t)b] = z[2 * e]; zk[2 * (size_t)b + 1] = z[2 * e + 1];
          dk[b] = depth ? depth[e] : NAN;
        }
        if (int rc = viekf_batch_init_feature(s->core, zk.data(), dk.data(), newf.data(), ok.data(), VIEKF_HOST)) return rc;
        for (int b = 0; b < B; b++)
          if (newf[b] && ok[b]) push_feature(s, b);
      }
    }
    for (int b = 0; b < B; b++) {
      if (!npres[b]) continue;
      FilterSeq& f = s->fs[b];
      FMeas m;
      m.t = t_per_filter ? t_per_filter[b] : t; m.type = VIEKF_FEAT; m.zdim = 2; m.rdim = 2; m.active = true; m.handled = false; m.id = -1;
      std::memset(m.z, 0, sizeof m.z);
      std::memset(m.R, 0, sizeof m.R);
      std::memcpy(m.R, R, sizeof(double) * 4);
      m.frame = fd; m.first = 0; m.weight = npres[b];
      size_t k = 0;                                                                    // :150-156
      while (k < f.zbuf.size() && !(f.zbuf[k].t < m.t)) k++;
      f.zbuf.insert(f.zbuf.begin() + (long)k, std::move(m));                           // :169-175
      f.queued += npres[b];
    }
    return VIEKF_OK;
  }
  std::vector<double> zk((size_t)B * 2), dk((size_t)B, NAN);
  std::vector<int32_t> ik(B), rk(B);
  for (int k = 0; k < count; k++) {
    for (int b = 0; b < B; b++) {
      const size_t e = (size_t)b * count + k;
      zk[2 * (size_t)b] = z[2 * e]; zk[2 * (size_t)b + 1] = z[2 * e + 1];
      ik[b] = id[e];
      if (depth) dk[b] = depth[e];
    }
    int rc;
    if (t_per_filter) rc = viekf_seq_add_measurement_t(s, t_per_filter, VIEKF_FEAT, zk.data(), 2, R, 2, active, ik.data(), dk.data(), mask, rk.data());
    else rc = viekf_seq_add_measurement(s, t, VIEKF_FEAT, zk.data(), 2, R, 2, active, ik.data(), dk.data(), rk.data());
    if (rc) return rc;
    if (result)
      for (int b = 0; b < B; b++) result[(size_t)b * count + k] = rk[b];
  }
  return VIEKF_OK;
}

// VIEKF::init_feature(l, id, depth), src/vi_ekf/vi_ekf_feat.cpp:6-47, with the sequencer's feature bookkeeping: the caller's id
// is ignored like there (:29-30: the filter pushes its own counter)
int viekf_seq_init_feature(viekf_seq* s, const double* pix, const double* depth, const uint8_t* mask, int32_t* ok) {
  if (!s || !pix) return VIEKF_ERR_INVALID;
  const int B = s->B;
  std::vector<uint8_t> m(B, 1);
  if (mask) m.assign(mask, mask + B);
  for (int b = 0; b < B; b++)
    if ((int)s->ids[b].size() >= s->N) m[b] = 0;                   // :9-10 (full: refused)
  std::vector<double> dep(B, NAN);
  if (depth) dep.assign(depth, depth + B);
  std::vector<int32_t> okv(B, 0);
  if (int rc = viekf_batch_init_feature(s->core, pix, dep.data(), m.data(), okv.data(), VIEKF_HOST)) return rc;
  for (int b = 0; b < B; b++)
    if (m[b] && okv[b]) push_feature(s, b);
  if (s->indep)
    if (int rc = refresh_slots(s, m)) return rc;
  if (ok) std::memcpy(ok, okv.data(), sizeof(int32_t) * B);
  return VIEKF_OK;
}

int viekf_seq_handle_measurements(viekf_seq* s, int32_t* gated_ids, int32_t cap, int32_t* gated_count) {
  if (!s) return VIEKF_ERR_INVALID;
  const int B = s->B;
  std::vector<std::vector<int32_t>> gated(B);
  auto finish = [&]() {
    if (gated_count) for (int b = 0; b < B; b++) gated_count[b] = (int32_t)gated[b].size();
    if (gated_ids && cap > 0)
      for (int b = 0; b < B; b++)
        for (int k = 0; k < cap; k++) gated_ids[(size_t)b * cap + k] = k < (int)gated[b].size() ? gated[b][k] : -1;
    return VIEKF_OK;
  };
  if (s->indep) {   // every filter plans its own rewind / replay; the device steps run batched by kind
    std::vector<std::vector<SeqOp>> ops(B);
    // (the filters' queues are separate objects: planned side by side on a few host threads when there are many of them)
    const int nth = B >= 256 ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nth > 1) {
      std::vector<std::thread> pool;
      for (int w = 0; w < nth; w++)
        pool.emplace_back([&, w]() { for (int b = B * w / nth; b < B * (w + 1) / nth; b++) { ops[b].reserve(16); plan_handle(s, b, ops[b]); } });
      for (auto& th : pool) th.join();
    } else {
      for (int b = 0; b < B; b++) plan_handle(s, b, ops[b]);
    }
    if (int rc = run_ops(s, ops, gated, gated_ids != nullptr || gated_count != nullptr)) return rc;
    return finish();
  }
  const bool want_gated = gated_ids != nullptr || gated_count != nullptr;   // (nobody asked: the frame's launch is not waited for)
  if (s->zbuf.empty() || s->u.empty()) return finish();            // :12-13
  long zi = (long)s->zbuf.size() - 1;                              // :16-18 oldest unhandled
  while (s->zbuf[zi].handled && zi != 0) zi--;
  if (zi == 0 && s->zbuf[zi].handled) return finish();             // :21-22
  if (s->zbuf[zi].t > s->u[0].first) return finish();              // :24-28 from the future
  size_t ui = 0;                                                   // :32-38 input just before the measurement
  while (ui != s->u.size()) {
    if (s->zbuf[zi].t > s->u[ui].first) break;
    ui++;
  }
  if (ui == s->u.size() || s->zbuf[zi].t <= s->u[ui].first) return finish();   // :39-43 not enough input history
  int k = s->H, target = -1;                                       // :46-57 rewind
  while (k > 0) {
    const int j = (s->i + k) % s->H;
    if (s->t[j] <= s->u[ui].first) { target = j; break; }
    k--;
  }
  if (k == 0) {                                                    // :59-64 not enough state history
    s->zbuf.erase(s->zbuf.begin() + zi);
    return finish();
  }
  if (target != s->i) {   // rewind = the ring slot becomes the live state (:50-52); feature counts are not part of the ring
    bool found = true;
    if (int rc = rewind_to(s, target, &found)) return rc;
    if (!found) {                                                  // as :59-64: no state to go back to -- the measurement is dropped
      s->zbuf.erase(s->zbuf.begin() + zi);
      return finish();
    }
  }
  std::vector<int32_t> res;
  size_t tail = 0;
  ui--;                                                            // :74
  while (ui != 0) {                                                // :75
    bool left_inner_by_break = false;
    while (s->zbuf[zi].t <= s->u[ui].first) {                      // :78
      SeqMeas& z = s->zbuf[zi];
      if (s->t[s->i] < z.t)                                        // :81-82
        if (int rc = propagate_core(s, s->u[ui].second.data(), z.t, false)) return rc;
      if (!z.handled) {                                            // :87-95
        // A camera frame queues one FEAT entry per feature, all with the same time stamp: the reference applies them one
        // after the other with nothing in between (no propagate: :81 is false, :100 does not fire), so they go to the device
        // as ONE launch of M sequential updates -- the same arithmetic, P crosses HBM once per frame instead of once per
        // feature.  (Not while logging: the log wants zhat before every single update.)
        long zl = zi;
        if (s->log.empty() && z.type == VIEKF_FEAT && z.active && z.count == 1)
          while (zl > 0 && frame_mate(s->zbuf[zl - 1], z)) zl--;
        if (z.count > 1) {
          std::vector<int32_t> resk;
          if (int rc = update_block(s, z, want_gated ? &resk : nullptr)) return rc;
          if (want_gated)
            for (int j = 0; j < z.count; j++)
              for (int b = 0; b < B; b++)
                if (resk[(size_t)b * z.count + j] == VIEKF_MEAS_GATED) gated[b].push_back(z.id[(size_t)b * z.count + j]);
        } else if (zl < zi) {
          const int k = (int)(zi - zl + 1);
          std::vector<int32_t> resk;
          if (int rc = update_frame(s, zi, k, resk)) return rc;
          for (int j = 0; j < k; j++)
            for (int b = 0; b < B; b++)
              if (resk[(size_t)b * k + j] == VIEKF_MEAS_GATED) gated[b].push_back(s->zbuf[zi - j].id[b]);
          zi = zl;                                                 // (the bookkeeping below continues from the frame's last entry)
        } else {
          if (int rc = update_entry(s, z, res)) return rc;
          if (z.type == VIEKF_FEAT)
            for (int b = 0; b < B; b++)
              if (res[b] == VIEKF_MEAS_GATED) gated[b].push_back(z.id[b]);
        }
      }
      if (zi != 0) {                                               // :97-105
        zi--;
        while (s->u[ui].first < s->zbuf[zi].t && ui != 0) {
          if (int rc = propagate_core(s, s->u[ui].second.data(), s->u[ui].first, false)) return rc;
          ui--;
        }
      } else {                                                     // :106-115: every remaining input, down to u[1] ...
        tail = ui;
        ui = 0;
        left_inner_by_break = true;
        break;
      }
    }
    if (!left_inner_by_break) break;   // (the inner condition can only fail with ui == 0)
  }
  if (int rc = replay_inputs(s, tail)) return rc;                  // ... and u[0] (:118), fused into one launch where possible
  {   // :121-122: single measurements leave from the old end until LEN_MEAS_HIST are left; a frame block weighs its `count` and
      // gives up only as many of its members as the reference would pop -- the ones it consumes first sit at the very end
    long total = 0;
    for (const auto& e : s->zbuf) total += e.count;
    while (total > s->MH && !s->zbuf.empty()) {
      SeqMeas& e = s->zbuf.back();
      const long drop = total - s->MH;
      if (drop >= e.count) { total -= e.count; s->zbuf.pop_back(); continue; }
      const int keep = e.count - (int)drop;
      std::vector<double> z((size_t)B * keep * 2);
      std::vector<int32_t> id((size_t)B * keep);
      std::vector<uint8_t> pr((size_t)B * keep);
      for (int b = 0; b < B; b++)
        for (int j = 0; j < keep; j++) {
          const size_t o = (size_t)b * keep + j, f = (size_t)b * e.count + drop + j;
          z[2 * o] = e.z[2 * f]; z[2 * o + 1] = e.z[2 * f + 1]; id[o] = e.id[f]; pr[o] = e.present[f];
        }
      e.z.swap(z); e.id.swap(id); e.present.swap(pr);
      e.count = keep;
      total -= drop;
    }
  }
  while ((int)s->u.size() > s->H) s->u.pop_back();                 // :125-126
  return finish();
}

static int keep_features_impl(viekf_seq* s, const int32_t* ids, int32_t count, bool kf_test, uint8_t* did_reset, double* edges) {
  if (!s || (count > 0 && !ids) || count < 0) return VIEKF_ERR_INVALID;
  const int B = s->B, N = s->N;
  std::vector<uint8_t> keep((size_t)B * N, 0), reset(B, 0);
  bool any_drop = false, any_reset = false;
  const bool use_kf = kf_test && s->prm.use_keyframe_reset != 0;
  for (int b = 0; b < B; b++) {
    std::vector<int32_t> want;
    for (int k = 0; k < count; k++)
      if (ids[(size_t)b * count + k] >= 0) want.push_back(ids[(size_t)b * count + k]);
    int overlap = 0;
    std::vector<int32_t> kept;
    for (size_t l = 0; l < s->ids[b].size(); l++) {                // vi_ekf_feat.cpp:85-113
      const int gid = s->ids[b][l];
      if (std::find(want.begin(), want.end(), gid) != want.end()) {
        keep[(size_t)b * N + l] = 1;
        kept.push_back(gid);
        if (use_kf && std::find(s->kf_feats[b].begin(), s->kf_feats[b].end(), gid) != s->kf_feats[b].end()) overlap++;
      } else {
        any_drop = true;
      }
    }
    s->ids[b] = kept;
    rebuild_slots(s, b);
    if (use_kf && !s->kf_feats[b].empty() &&
        (double)overlap / (double)s->kf_feats[b].size() < s->prm.keyframe_overlap_threshold) {   // :119-130
      reset[b] = 1;
      any_reset = true;
      s->kf_feats[b] = want;
    } else if (use_kf && s->kf_feats[b].empty()) {                 // :131-139
      s->kf_feats[b] = want;
    }
  }
  if (any_drop) {
    if (int rc = viekf_batch_keep_features(s->core, keep.data(), nullptr, VIEKF_HOST)) return rc;
    }
  if (edges) std::memset(edges, 0, sizeof(double) * 17 * (size_t)B);
  if (any_reset)
    if (int rc = reset_and_move_node(s, reset, edges)) return rc;
  if (s->indep && (any_drop || any_reset)) {
    std::vector<uint8_t> all(B, 1);
    if (int rc = refresh_slots(s, all)) return rc;
  }
  if (did_reset) std::memcpy(did_reset, reset.data(), B);
  return VIEKF_OK;
}

int viekf_seq_keep_only_features(viekf_seq* s, const int32_t* ids, int32_t count, uint8_t* did_reset, double* edges) {
  return keep_features_impl(s, ids, count, true, did_reset, edges);
}
// VIEKF::clear_feature (src/vi_ekf/vi_ekf_feat.cpp:50-73) for every tracked feature NOT listed: the removal of keep_only_features
// without its keyframe-overlap test (:119-139 belongs to keep_only_features alone)
int viekf_seq_drop_features(viekf_seq* s, const int32_t* keep_ids, int32_t count) {
  return keep_features_impl(s, keep_ids, count, false, nullptr, nullptr);
}

// VIEKF::propagate_state(u, t, save_input), src/vi_ekf/vi_ekf.cpp:262-318, with the flag the reference's own replay uses
int viekf_seq_propagate_state(viekf_seq* s, const double* u, double t, int32_t save_input) {
  if (!s || !u) return VIEKF_ERR_INVALID;
  if (save_input) return viekf_seq_propagate(s, u, t);
  if (s->indep) return VIEKF_ERR_UNSUPPORTED;   // (independent clocks: inputs are always recorded, per filter)
  return propagate_core(s, u, t, false);
}

namespace {
// x_[i_] of every filter, edited in place on the host (set_x0 / set_imu_bias are start-up calls): one read, one write
int edit_live_state(viekf_seq* s, const std::function<void(int, double*)>& f) {
  const int nx = 17 + 5 * s->N;
  std::vector<double> x((size_t)s->B * nx);
  if (int rc = viekf_batch_get_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  for (int b = 0; b < s->B; b++) f(b, x.data() + (size_t)b * nx);
  if (int rc = viekf_batch_set_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  if (s->indep) { std::vector<uint8_t> all(s->B, 1); return refresh_slots(s, all); }
  return VIEKF_OK;
}
}  // namespace

int viekf_seq_set_x0(viekf_seq* s, const double* x0) {            // VIEKF::set_x0, vi_ekf.cpp:157-160: x_[i_].topRows(xZ) = x0
  if (!s || !x0) return VIEKF_ERR_INVALID;
  return edit_live_state(s, [&](int b, double* x) { std::memcpy(x, x0 + 17 * (size_t)b, sizeof(double) * 17); });
}

int viekf_seq_set_imu_bias(viekf_seq* s, const double* b_g, const double* b_a) {   // VIEKF::set_imu_bias, vi_ekf.cpp:179-183
  if (!s || !b_g || !b_a) return VIEKF_ERR_INVALID;
  return edit_live_state(s, [&](int b, double* x) {
    for (int i = 0; i < 3; i++) { x[13 + i] = b_g[3 * (size_t)b + i]; x[10 + i] = b_a[3 * (size_t)b + i]; }
  });
}

int viekf_seq_keyframe_reset(viekf_seq* s, const uint8_t* mask, double* edges) {   // VIEKF::keyframe_reset(), vi_ekf_kfr.cpp:56-157
  if (!s) return VIEKF_ERR_INVALID;
  std::vector<uint8_t> reset(s->B, 1);
  if (mask) reset.assign(mask, mask + s->B);
  if (edges) std::memset(edges, 0, sizeof(double) * 17 * (size_t)s->B);
  if (int rc = reset_and_move_node(s, reset, edges)) return rc;
  if (s->indep) return refresh_slots(s, reset);
  return VIEKF_OK;
}

// get_depths / get_zetas / get_qzetas / get_zeta (vi_ekf.cpp:210-246): per feature slot 1 / rho, zeta = q_zeta.rota(e_z), q_zeta;
// slots past a filter's len_features: NaN.  Any output may be NULL.
int viekf_seq_get_features(viekf_seq* s, double* depths, double* zetas, double* qzetas) {
  if (!s) return VIEKF_ERR_INVALID;
  const int nx = 17 + 5 * s->N, N = s->N;
  std::vector<double> x((size_t)s->B * nx);
  if (int rc = viekf_batch_get_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  const double ez[3] = {0.0, 0.0, 1.0};
  for (int b = 0; b < s->B; b++)
    for (int i = 0; i < N; i++) {
      const double* f = x.data() + (size_t)b * nx + 17 + 5 * i;
      const bool on = i < (int)s->ids[b].size();
      if (depths) depths[(size_t)b * N + i] = on ? 1.0 / f[4] : NAN;
      if (zetas) {
        double z[3] = {NAN, NAN, NAN};
        if (on) rota(f, ez, z);
        std::memcpy(zetas + ((size_t)b * N + i) * 3, z, sizeof z);
      }
      if (qzetas)
        for (int k = 0; k < 4; k++) qzetas[((size_t)b * N + i) * 4 + k] = on ? f[k] : NAN;
    }
  return VIEKF_OK;
}

// get_feat(id) / get_depth(id) (vi_ekf.cpp:248-260) by GLOBAL feature id per filter: pix [B][2] = cam_F zeta / zeta_z + cam_center,
// depth [B] = 1 / rho; an id the filter does not track: NaN (the reference indexes with -1 there).  Either output may be NULL.
int viekf_seq_get_feat(viekf_seq* s, const int32_t* id, double* pix, double* depth) {
  if (!s || !id) return VIEKF_ERR_INVALID;
  const int nx = 17 + 5 * s->N;
  std::vector<double> x((size_t)s->B * nx);
  if (int rc = viekf_batch_get_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  const double ez[3] = {0.0, 0.0, 1.0};
  for (int b = 0; b < s->B; b++) {
    const int i = local_id(s, b, id[b]);
    double px[2] = {NAN, NAN}, d = NAN;
    if (i >= 0) {
      const double* f = x.data() + (size_t)b * nx + 17 + 5 * i;
      double z[3];
      rota(f, ez, z);
      px[0] = s->prm.focal_len[0] * z[0] / z[2] + s->prm.cam_center[0];
      px[1] = s->prm.focal_len[1] * z[1] / z[2] + s->prm.cam_center[1];
      d = 1.0 / f[4];
    }
    if (pix) { pix[2 * (size_t)b] = px[0]; pix[2 * (size_t)b + 1] = px[1]; }
    if (depth) depth[b] = d;
  }
  return VIEKF_OK;
}

int viekf_seq_get_global_pose(viekf_seq* s, double* pose, double* node) {   // vi_ekf_kfr.cpp:14-21, vi_ekf.cpp:192-195
  if (!s || !pose) return VIEKF_ERR_INVALID;
  const int nx = 17 + 5 * s->N;
  std::vector<double> x((size_t)s->B * nx);
  if (int rc = viekf_batch_get_state(s->core, x.data(), nullptr, nullptr, VIEKF_HOST)) return rc;
  for (int b = 0; b < s->B; b++) {
    const double* xf = x.data() + (size_t)b * nx;
    const double rel[7] = {xf[0], xf[1], xf[2], xf[6], xf[7], xf[8], xf[9]};
    xform_compose(s->node.data() + 7 * (size_t)b, rel, pose + 7 * (size_t)b);
  }
  if (node) std::memcpy(node, s->node.data(), sizeof(double) * 7 * (size_t)s->B);
  return VIEKF_OK;
}

int viekf_seq_get_global_cov(viekf_seq* s, double* cov) {   // vi_ekf_kfr.cpp:23-35
  if (!s || !cov) return VIEKF_ERR_INVALID;
  std::vector<double> blk((size_t)s->B * 81);                       // P[0:9, 0:9]: holds the POS (0..2) and ATT (6..8) blocks
  if (int rc = viekf_batch_get_cov_block(s->core, 0, 0, 9, 9, blk.data(), VIEKF_HOST)) return rc;
  for (int b = 0; b < s->B; b++) {
    const double* Pb = blk.data() + 81 * (size_t)b;
    double C[36];
    for (int c = 0; c < 6; c++)
      for (int r = 0; r < 6; r++) C[r + 6 * c] = Pb[(r < 3 ? r : r + 3) + 9 * (c < 3 ? c : c + 3)];
    double* o = cov + 36 * (size_t)b;
    std::memcpy(o, s->node_cov.data() + 36 * (size_t)b, sizeof(double) * 36);
    add_adj_cov(s->node.data() + 7 * (size_t)b, C, o);
  }
  return VIEKF_OK;
}

int viekf_seq_init_logger(viekf_seq* s, const char* root_filename, const char* ekf_name, int32_t filter) {   // vi_ekf_log.cpp:79-117
  if (!s || !root_filename || !ekf_name || filter < 0 || filter >= s->B) return VIEKF_ERR_INVALID;
  if (s->indep) return VIEKF_ERR_UNSUPPORTED;   // (the log writer records the lock-step flow: one shared clock)
  const std::string base = std::string(root_filename) + ekf_name;
  s->log.clear();
  s->log.resize(TOTAL_LOGS);
  auto open = [&](int i, const std::string& suffix) { s->log[i].open(base + suffix, std::ofstream::out | std::ofstream::trunc); };
  for (int i = 0; i < kTotalMeas; i++) open(i, std::string("_") + kMeasNames[i] + ".log");
  open(LOG_STATE, "_state.log"); open(LOG_COV, "_cov.log"); open(LOG_FEATURE_IDS, "_feat_id.log"); open(LOG_CONF, "_config.txt");
  open(LOG_INPUT, "_input.log"); open(LOG_XDOT, "_xdot.log"); open(LOG_KF, "_kf.log"); open(LOG_DEBUG, "_debug.txt");
  open(LOG_GLOBAL_POSE, "_global_pose.log");
  for (int i = 0; i < TOTAL_LOGS; i++)
    if (i != LOG_GLOBAL && !s->log[i].is_open()) { s->log.clear(); return VIEKF_ERR_INVALID; }   // (LOG_GLOBAL is never opened, :85-97)
  s->log_filter = filter;
  std::vector<double> x, Pd;
  if (int rc = fetch_state_and_diag(s, x, Pd)) { s->log.clear(); return rc; }
  const int nx = 17 + 5 * s->N, n = 16 + 3 * s->N;
  const viekf_params& p = s->prm;
  std::ofstream& c = s->log[LOG_CONF];                              // :100-116, line for line
  c << "Test Num: " << root_filename << "\n";
  c << "x0" << eigen_row(x.data() + (size_t)filter * nx, 17) << "\n";
  c << "P0: " << eigen_row(Pd.data() + (size_t)filter * n, std::min(17, n)) << "\n";   // (the reference prints xZ = 17 entries)
  c << "P0_feat: " << eigen_row(p.P0_feat, 3) << "\n";
  c << "Qx: " << eigen_row(p.Qx, 16) << "\n";
  c << "Qx_feat: " << eigen_row(p.Qx_feat, 3) << "\n";
  c << "Qu: " << eigen_row(p.Qu, 6) << "\n";
  c << "q_b_c: " << eigen_row(p.q_b_c, 4) << "\n";
  c << "p_b_c: " << eigen_row(p.p_b_c, 3) << "\n";
  c << "lambda: " << eigen_row(p.lambda, 16) << "\n";
  c << "lambda_feat: " << eigen_row(p.lambda_feat, 3) << "\n";
  c << "using partial_update: " << (p.use_partial_update != 0) << "\n";
  c << "using keyframe reset: " << (p.use_keyframe_reset != 0) << "\n";
  c << "using drag Term: " << (p.use_drag_term != 0) << "\n";
  c << "keyframe overlap: " << p.keyframe_overlap_threshold << "\n";
  c << "num features: " << s->N << "\n";
  c << "min_depth: " << p.min_depth << std::endl;
  log_state(s, 0.0, x, Pd, nullptr, nullptr);                      // the constructor's first record, vi_ekf.cpp:154
  return VIEKF_OK;
}

int viekf_seq_disable_logger(viekf_seq* s) {                       // vi_ekf_log.cpp:69-77
  if (!s) return VIEKF_ERR_INVALID;
  for (auto& f : s->log) if (f.is_open()) f.close();
  s->log.clear();
  return VIEKF_OK;
}

int viekf_seq_tracked_features(viekf_seq* s, int32_t* ids, int32_t* len) {
  if (!s) return VIEKF_ERR_INVALID;
  for (int b = 0; b < s->B; b++) {
    if (len) len[b] = (int32_t)s->ids[b].size();
    if (ids)
      for (int k = 0; k < s->N; k++) ids[(size_t)b * s->N + k] = k < (int)s->ids[b].size() ? s->ids[b][k] : -1;
  }
  return VIEKF_OK;
}

int viekf_seq_status(viekf_seq* s, double* t_now, int32_t* ring_index, int32_t* queued, int32_t* inputs) {
  if (!s) return VIEKF_ERR_INVALID;
  if (s->indep) {   // (independent clocks: the first filter's)
    const FilterSeq& f = s->fs[0];
    if (t_now) *t_now = f.t[f.i];
    if (ring_index) *ring_index = f.i;
    if (queued) *queued = (int32_t)f.zbuf.size();
    if (inputs) *inputs = (int32_t)f.u.size();
    return VIEKF_OK;
  }
  if (t_now) *t_now = s->t[s->i];
  if (ring_index) *ring_index = s->i;
  if (queued) *queued = (int32_t)s->zbuf.size();
  if (inputs) *inputs = (int32_t)s->u.size();
  return VIEKF_OK;
}

}  // extern "C"
