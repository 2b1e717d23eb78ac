// seq_host_stub.cpp -- TEST ONLY (never shipped, never linked into libviekf_hip.so): a host stand-in for the viekf_batch_* entry
// points that viekf_seq.cpp calls, so that the sequencer's queue / ring / rewind code (977 lines of deque and index arithmetic
// that otherwise only ever run against the GPU library) can be built with -fsanitize=address,undefined and driven on the CPU
// (tests/test_seq_sanitized.py).  It is NOT a filter: a propagate adds dt to a per-filter counter in x[0] and counts in x[1], an
// update counts in x[2], the ring copies whole states -- enough for the driver to check that every filter ends where its own
// time line says it should, and for the sanitizers to see every index the sequencer forms.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/viekf.h"

struct viekf_batch {
  int B, N, nx, n;
  viekf_params p;
  std::vector<double> x, P;            // live [B][nx], [B][n*n]
  std::vector<int32_t> len;
  std::vector<std::vector<double>> rx, rP;   // ring slots
  int live = -1;
  std::vector<uint8_t> active;
  bool active_on = false;
  long calls = 0;
  double* xs(int slot) { return slot < 0 ? x.data() : rx[(size_t)slot].data(); }
  double* Ps(int slot) { return slot < 0 ? P.data() : rP[(size_t)slot].data(); }
  bool on(int b) const { return !active_on || active[(size_t)b]; }
  // per-filter live ring slots (viekf_batch_select_filters): filter i's live state is slot lf[i]
  bool per_filter = false;
  std::vector<int> lf;
  double* xf(int i) { return xs(per_filter ? lf[(size_t)i] : live) + (size_t)i * nx; }
  double* Pf(int i) { return Ps(per_filter ? lf[(size_t)i] : live) + (size_t)i * n * n; }
};

extern "C" {
const char* viekf_last_error(void) { return "stub"; }
int viekf_batch_create(int32_t batch, int32_t num_features, const viekf_params* p, int32_t, viekf_batch** out) {
  viekf_batch* b = new viekf_batch;
  b->B = batch; b->N = num_features; b->nx = 17 + 5 * num_features; b->n = 16 + 3 * num_features; b->p = *p;
  b->x.assign((size_t)batch * b->nx, 0.0); b->P.assign((size_t)batch * b->n * b->n, 0.0); b->len.assign(batch, 0);
  for (int i = 0; i < batch; i++) b->x[(size_t)i * b->nx + 6] = 1.0;
  *out = b;
  return VIEKF_OK;
}
int viekf_batch_destroy(viekf_batch* b) { delete b; return VIEKF_OK; }
int viekf_batch_dims(const viekf_batch* b, int32_t* batch, int32_t* nf, int32_t* nx, int32_t* n) {
  if (batch) *batch = b->B; if (nf) *nf = b->N; if (nx) *nx = b->nx; if (n) *n = b->n;
  return VIEKF_OK;
}
int viekf_batch_get_params(const viekf_batch* b, viekf_params* out) { *out = b->p; return VIEKF_OK; }
int viekf_batch_set_async(viekf_batch*, int32_t) { return VIEKF_OK; }
int viekf_batch_history_resize(viekf_batch* b, int32_t depth) {
  if (b->per_filter) {
    std::vector<double> x(b->x.size()), P(b->P.size());
    for (int i = 0; i < b->B; i++) { std::memcpy(x.data() + (size_t)i * b->nx, b->xf(i), sizeof(double) * b->nx); std::memcpy(P.data() + (size_t)i * b->n * b->n, b->Pf(i), sizeof(double) * b->n * b->n); }
    b->x = x; b->P = P; b->per_filter = false;
  }
  if (b->live >= 0) { b->x = b->rx[(size_t)b->live]; b->P = b->rP[(size_t)b->live]; b->live = -1; }
  b->rx.assign((size_t)depth, std::vector<double>(b->x.size(), NAN));
  b->rP.assign((size_t)depth, std::vector<double>(b->P.size(), NAN));
  return VIEKF_OK;
}
int viekf_batch_snapshot(viekf_batch* b, int32_t slot) {
  if (slot < 0 || slot >= (int)b->rx.size()) return VIEKF_ERR_INVALID;
  if (slot != b->live) { b->rx[(size_t)slot].assign(b->xs(b->live), b->xs(b->live) + b->x.size()); b->rP[(size_t)slot].assign(b->Ps(b->live), b->Ps(b->live) + b->P.size()); }
  return VIEKF_OK;
}
int viekf_batch_select(viekf_batch* b, int32_t slot) {
  if (slot < -1 || slot >= (int)b->rx.size()) return VIEKF_ERR_INVALID;
  b->live = slot;
  return VIEKF_OK;
}
static void prop(viekf_batch* b, int src, int dst, const double* u, const double* dt) {
  for (int i = 0; i < b->B; i++) {
    double* xo = b->xs(dst) + (size_t)i * b->nx;
    const double* xi = b->xs(src) + (size_t)i * b->nx;
    if (xo != xi) std::memcpy(xo, xi, sizeof(double) * b->nx);
    if (!b->on(i)) continue;
    xo[0] += dt[i]; xo[1] += 1.0; xo[3] += u[6 * (size_t)i];
    xo[7] = xo[7] * 1.0000001 + u[6 * (size_t)i] * dt[i] + 0.37;   // depends on the ORDER of the steps and on which input each one got
  }
  if (src != dst) std::memcpy(b->Ps(dst), b->Ps(src), sizeof(double) * b->P.size());
}
int viekf_batch_propagate(viekf_batch* b, const double* u, const double* dt, viekf_mem) {
  b->calls++;
  if (b->per_filter) {
    for (int i = 0; i < b->B; i++) { if (!b->on(i)) continue; double* x = b->xf(i); x[0] += dt[i]; x[1] += 1.0; x[3] += u[6 * (size_t)i]; x[7] = x[7] * 1.0000001 + u[6 * (size_t)i] * dt[i] + 0.37; }
    return VIEKF_OK;
  }
  prop(b, b->live, b->live, u, dt);
  return VIEKF_OK;
}
int viekf_batch_select_filters(viekf_batch* b, const int32_t* slot) {
  if (b->live >= 0 || b->rx.empty()) return VIEKF_ERR_INVALID;
  for (int i = 0; i < b->B; i++) if (slot[i] >= (int)b->rx.size() || (slot[i] < 0 && !b->per_filter)) return VIEKF_ERR_INVALID;
  if (!b->per_filter) { b->lf.assign((size_t)b->B, 0); b->per_filter = true; }
  for (int i = 0; i < b->B; i++) if (slot[i] >= 0) b->lf[(size_t)i] = slot[i];
  return VIEKF_OK;
}
int viekf_batch_propagate_filters_to(viekf_batch* b, const double* u, const double* dt, const int32_t* dst, viekf_mem) {
  if (!b->per_filter) return VIEKF_ERR_INVALID;
  for (int i = 0; i < b->B; i++) if (dst[i] >= (int)b->rx.size() || (dst[i] >= 0 && dst[i] == b->lf[(size_t)i])) return VIEKF_ERR_INVALID;
  b->calls++;
  for (int i = 0; i < b->B; i++) {
    if (dst[i] < 0) continue;
    double* xi = b->xf(i); double* Pi = b->Pf(i);
    double* xo = b->rx[(size_t)dst[i]].data() + (size_t)i * b->nx;
    double* Po = b->rP[(size_t)dst[i]].data() + (size_t)i * b->n * b->n;
    std::memcpy(xo, xi, sizeof(double) * b->nx); std::memcpy(Po, Pi, sizeof(double) * b->n * b->n);
    xo[0] += dt[i]; xo[1] += 1.0; xo[3] += u[6 * (size_t)i];
    xo[7] = xo[7] * 1.0000001 + u[6 * (size_t)i] * dt[i] + 0.37;
    b->lf[(size_t)i] = dst[i];
  }
  return VIEKF_OK;
}
int viekf_batch_propagate_to(viekf_batch* b, const double* u, const double* dt, int32_t dst, viekf_mem) {
  if (dst < 0 || dst >= (int)b->rx.size() || b->active_on) return VIEKF_ERR_INVALID;
  b->calls++;
  prop(b, b->live, dst, u, dt);
  b->live = dst;
  return VIEKF_OK;
}
// the fused form: only the LAST slot is written (the stub poisons the ones in between, as stale device memory would be wrong);
// stub_write_every_slot != 0: the HBM-path family's behaviour, every slot written -- the run the fused one is compared with
int stub_write_every_slot = 0;
int viekf_batch_propagate_n_to(viekf_batch* b, int32_t K, const double* u, const double* dt, const int32_t* dst, int32_t* written, viekf_mem) {
  if (K < 1 || K > 64 || b->active_on) return VIEKF_ERR_INVALID;
  for (int k = 0; k < K; k++)
    if (dst[k] < 0 || dst[k] >= (int)b->rx.size() || dst[k] == b->live) return VIEKF_ERR_INVALID;
  b->calls++;
  if (stub_write_every_slot) {
    for (int k = 0; k < K; k++) { prop(b, b->live, dst[k], u + (size_t)6 * b->B * k, dt + (size_t)b->B * k); b->live = dst[k]; }
    if (written) *written = 1;
    return VIEKF_OK;
  }
  int cur = b->live;
  for (int k = 0; k < K; k++) {
    prop(b, cur, dst[K - 1], u + (size_t)6 * b->B * k, dt + (size_t)b->B * k);
    cur = dst[K - 1];
  }
  for (int k = 0; k + 1 < K; k++)
    for (size_t e = 0; e < b->rx[(size_t)dst[k]].size(); e++) b->rx[(size_t)dst[k]][e] = NAN;
  b->live = dst[K - 1];
  if (written) *written = K == 1 ? 1 : 0;
  return VIEKF_OK;
}
int viekf_batch_update_feat(viekf_batch* b, const double* z, const int32_t* slot, int32_t M, const double* R, int32_t, int32_t* result, viekf_mem) {
  b->calls++;
  for (int i = 0; i < b->B; i++)
    for (int m = 0; m < M; m++) {
      const int sl = slot[(size_t)i * M + m];
      int code = 0;
      if (sl < 0) code = -1;
      else if (sl >= b->len[(size_t)i]) code = 3;
      else if (std::isnan(z[((size_t)i * M + m) * 2])) code = 2;
      else if (b->on(i)) { b->xf(i)[2] += 1.0; b->xf(i)[7] = b->xf(i)[7] * 0.9999 + 0.01 * z[((size_t)i * M + m) * 2]; }
      if (result) result[(size_t)i * M + m] = b->on(i) ? code : -1;
    }
  (void)R;
  return VIEKF_OK;
}
int viekf_batch_update(viekf_batch* b, int32_t type, const double* z, int32_t zdim, const double*, int32_t, int32_t, const int32_t* slot,
                       const uint8_t* active, int32_t* result, viekf_mem) {
  b->calls++;
  for (int i = 0; i < b->B; i++) {
    int code = 0;
    if (active && active[i] == 2) code = -1;
    else if (slot && (type == 5 || type == 6 || type == 8 || type == 9) && (slot[i] < 0 || slot[i] >= b->len[(size_t)i])) code = slot[i] < 0 ? -1 : 3;
    else if (std::isnan(z[(size_t)i * zdim])) code = 2;
    else { b->xf(i)[4] += 1.0; b->xf(i)[7] = b->xf(i)[7] * 0.999 + z[(size_t)i * zdim]; }
    if (result) result[i] = code;
  }
  return VIEKF_OK;
}
int viekf_batch_init_feature(viekf_batch* b, const double* pix, const double*, const uint8_t* mask, int32_t* ok, viekf_mem) {
  for (int i = 0; i < b->B; i++) {
    int took = 0;
    if ((!mask || mask[i]) && b->len[(size_t)i] < b->N) {
      b->xf(i)[17 + 5 * b->len[(size_t)i]] = pix[2 * (size_t)i];
      b->len[(size_t)i]++;
      took = 1;
    }
    if (ok) ok[i] = took;
  }
  return VIEKF_OK;
}
int viekf_batch_keep_features(viekf_batch* b, const uint8_t* keep, int32_t* new_len, viekf_mem) {
  for (int i = 0; i < b->B; i++) {
    int k = 0;
    double* x = b->xf(i);
    for (int f = 0; f < b->len[(size_t)i]; f++)
      if (keep[(size_t)i * b->N + f]) { std::memmove(x + 17 + 5 * k, x + 17 + 5 * f, sizeof(double) * 5); k++; }
    for (int f = k; f < b->N; f++) std::memset(x + 17 + 5 * f, 0, sizeof(double) * 5);
    b->len[(size_t)i] = k;
    if (new_len) new_len[i] = k;
  }
  return VIEKF_OK;
}
int viekf_batch_keyframe_reset(viekf_batch* b, const uint8_t* mask, double* edge, viekf_mem) {
  for (int i = 0; i < b->B; i++) {
    if (mask && !mask[i]) continue;
    double* x = b->xf(i);
    if (edge) { double* e = edge + 17 * (size_t)i; std::memset(e, 0, sizeof(double) * 17); e[0] = x[0]; e[3] = 1.0; }
    x[5] += 1.0;
  }
  return VIEKF_OK;
}
int viekf_batch_get_state(viekf_batch* b, double* x, double* P, int32_t* len, viekf_mem) {
  for (int i = 0; i < b->B; i++) {
    if (x) std::memcpy(x + (size_t)i * b->nx, b->xf(i), sizeof(double) * b->nx);
    if (P) std::memcpy(P + (size_t)i * b->n * b->n, b->Pf(i), sizeof(double) * b->n * b->n);
  }
  if (len) std::memcpy(len, b->len.data(), sizeof(int32_t) * b->B);
  return VIEKF_OK;
}
int viekf_batch_set_state(viekf_batch* b, const double* x, const double* P, const int32_t* len, viekf_mem) {
  for (int i = 0; i < b->B; i++) {
    if (x) std::memcpy(b->xf(i), x + (size_t)i * b->nx, sizeof(double) * b->nx);
    if (P) std::memcpy(b->Pf(i), P + (size_t)i * b->n * b->n, sizeof(double) * b->n * b->n);
  }
  if (len) std::memcpy(b->len.data(), len, sizeof(int32_t) * b->B);
  return VIEKF_OK;
}
int viekf_batch_get_cov_diag(viekf_batch* b, double* d, viekf_mem) { std::memset(d, 0, sizeof(double) * (size_t)b->B * b->n); return VIEKF_OK; }
int viekf_batch_get_cov_block(viekf_batch* b, int32_t, int32_t, int32_t nr, int32_t nc, double* out, viekf_mem) {
  std::memset(out, 0, sizeof(double) * (size_t)b->B * nr * nc);
  return VIEKF_OK;
}
int viekf_batch_eval_xdot(viekf_batch* b, const double*, double* xdot, viekf_mem) { std::memset(xdot, 0, sizeof(double) * (size_t)b->B * b->n); return VIEKF_OK; }
int viekf_batch_eval_h(viekf_batch* b, int32_t, const int32_t*, double* zhat, viekf_mem) { std::memset(zhat, 0, sizeof(double) * 4 * (size_t)b->B); return VIEKF_OK; }
int viekf_batch_set_active(viekf_batch* b, const uint8_t* mask, viekf_mem) {
  b->active_on = mask != nullptr;
  if (mask) b->active.assign(mask, mask + b->B);
  return VIEKF_OK;
}
static int ring_filters(viekf_batch* b, const int32_t* slot, int to_ring) {
  if (b->live >= 0) return VIEKF_ERR_INVALID;
  for (int i = 0; i < b->B; i++) {
    const int sl = slot[i];
    if (sl < 0) continue;
    if (sl >= (int)b->rx.size()) return VIEKF_ERR_INVALID;
    double* rx = b->rx[(size_t)sl].data() + (size_t)i * b->nx;
    double* lx = b->per_filter ? b->xf(i) : b->x.data() + (size_t)i * b->nx;
    double* rP = b->rP[(size_t)sl].data() + (size_t)i * b->n * b->n;
    double* lP = b->per_filter ? b->Pf(i) : b->P.data() + (size_t)i * b->n * b->n;
    if (to_ring) { std::memcpy(rx, lx, sizeof(double) * b->nx); std::memcpy(rP, lP, sizeof(double) * b->n * b->n); }
    else { std::memcpy(lx, rx, sizeof(double) * b->nx); std::memcpy(lP, rP, sizeof(double) * b->n * b->n); }
  }
  return VIEKF_OK;
}
int viekf_batch_snapshot_filters(viekf_batch* b, const int32_t* slot, viekf_mem) { return ring_filters(b, slot, 1); }
int viekf_batch_restore_filters(viekf_batch* b, const int32_t* slot, viekf_mem) { return ring_filters(b, slot, 0); }
}  // extern "C"
