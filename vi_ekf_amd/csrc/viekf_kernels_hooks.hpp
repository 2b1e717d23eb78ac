// viekf_kernels_hooks.hpp -- read-only evaluation kernels behind the reference's public TEST HOOKS: the five-argument
// dynamics(x, u, xdot, dfdx, dfdu) (src/vi_ekf/vi_ekf_dyn.cpp:5-11), h_*(x, h, H, id) (vi_ekf_meas.cpp:281-386), boxplus / boxminus
// (vi_ekf_helper.cpp:88-111) and keyframe_reset(xm, xp, N) (vi_ekf_kfr.cpp:6-12).  They hand out, as DENSE matrices in the
// reference's (Eigen, column-major) layout, what the hot kernels only ever hold in structured form -- so that the reference's
// jac_test properties (test/jac_test.cpp:245-487) and the symbolic Jacobians can be checked against DEVICE output.  They use the
// same device functions as the hot kernels (body_dynamics, feature_dynamics, meas_model, body_boxplus, q_feat_boxplus ...).
// None of them changes a filter; the state they evaluate at is an explicit argument (NULL = the batch's current state).
#pragma once
#include "viekf_kernels_stream.hpp"

namespace viekf {

#ifndef VIEKF_INSTANCES_ONLY
// dynamics(x, u, xdot, dfdx, dfdu): u is the body-frame input, as that overload takes it (NOT rotated by q_b_u: propagate_state
// does that before it calls dynamics, vi_ekf.cpp:265-267,295).  xdot [B][n], A [B][n][n], G [B][6][n] (column-major n x n / n x 6),
// everything the reference leaves zero is zero.  One workgroup per filter; x_in [B][nx] or NULL.
__global__ __launch_bounds__(256) void k_eval_jacobians(StreamArgs a, const double* __restrict__ x_in, const double* __restrict__ u_all,
                                                        double* __restrict__ xdot_out, double* __restrict__ A_out,
                                                        double* __restrict__ G_out) {
  __shared__ double Abb[256], Gb[96], xdb[16];
  __shared__ BodyCtx ctx;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, len = a.len[b];
  const double* xs = x_in ? x_in + (long)b * a.nx : a.x + a.si(b) * a.nxs;
  double* A = A_out ? A_out + (long)b * n * n : nullptr;
  double* G = G_out ? G_out + (long)b * n * 6 : nullptr;
  double* xd = xdot_out ? xdot_out + (long)b * n : nullptr;
  if (A) for (long e = tid; e < (long)n * n; e += 256) A[e] = 0.0;
  if (G) for (int e = tid; e < n * 6; e += 256) G[e] = 0.0;
  if (xd) for (int e = tid; e < n; e += 256) xd[e] = 0.0;
  if (tid == 0) {
    body_ctx(xs, u_all + (long)b * 6, *a.dp, ctx);
    body_dynamics(ctx, *a.dp, xdb, Abb, Gb);                              // vi_ekf_dyn.cpp:42-80
  }
  __syncthreads();
  for (int e = tid; e < 256; e += 256) { const int r = e >> 4, c = e & 15; if (A) A[r + (long)c * n] = Abb[e]; }
  if (tid < 96) { const int r = tid / 6, c = tid % 6; if (G) G[r + (long)c * n] = Gb[tid]; }
  if (tid < 16 && xd) xd[tid] = xdb[tid];
  for (int f = tid; f < len; f += 256) {                                  // :96-134
    double xd3[3], Afv[9], Afg[9], Aff[9];
    feature_dynamics(xs + xZ + 5 * f, xs[xZ + 5 * f + 4], ctx, xd3, Afv, Afg, Aff);
    const int r0 = dxZ + 3 * f;
    for (int r = 0; r < 3; r++) {
      if (xd) xd[r0 + r] = xd3[r];
      for (int c = 0; c < 3; c++) {
        if (A) {
          A[(r0 + r) + (long)(dxVEL + c) * n] = Afv[r * 3 + c];
          A[(r0 + r) + (long)(dxB_G + c) * n] = Afg[r * 3 + c];
          A[(r0 + r) + (long)(r0 + c) * n] = Aff[r * 3 + c];
        }
        if (G) G[(r0 + r) + (long)(3 + c) * n] = Afg[r * 3 + c];           // :131-132 (the uG block equals the B_G block)
      }
    }
  }
}

// h_type(x, h, H, id): zhat [B][4] (unused entries 0), H [B][n][3] = the 3 x n hMatrix column-major (rows past the model's
// dimension 0).  One lane per filter; filters whose slot is not an active feature get NaN in zhat and a zero H.
__global__ void k_eval_H(StreamArgs a, const double* __restrict__ x_in, int type, const int* __restrict__ slot_all,
                         double* __restrict__ zhat_out, double* __restrict__ H_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n;
  const double* xs = x_in ? x_in + (long)b * a.nx : a.x + a.si(b) * a.nxs;
  double* H = H_out + (long)b * n * 3;
  for (int e = 0; e < 3 * n; e++) H[e] = 0.0;
  const bool needs_slot = type == MT_QZETA || type == MT_FEAT || type == MT_DEPTH || type == MT_INV_DEPTH;
  const int slot = (needs_slot && slot_all) ? slot_all[b] : 0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  double zhat[4] = {0.0, 0.0, 0.0, 0.0};
  if (needs_slot && (slot < 0 || slot >= a.len[b])) {
    for (int i = 0; i < 4; i++) zhat_out[(long)b * 4 + i] = nan;
    return;
  }
  int cols[6], nc = 0;
  double Hc[18];
  meas_model(type, xs, slot, *a.dp, zhat, cols, Hc, nc);
  for (int c = 0; c < nc; c++)
    for (int r = 0; r < 3; r++) H[r + 3L * cols[c]] = Hc[r * 6 + c];
  for (int i = 0; i < 4; i++) zhat_out[(long)b * 4 + i] = zhat[i];
}

// boxplus(x, dx, out) (minus = 0) / boxminus(x1, x2, out) (minus = 1) over the filter's ACTIVE features (len of the batch's
// filter b, as the reference loops to len_features_): x, x2 [B][nx]; dx / out_dx [B][n]; out_x [B][nx] (entries past the active
// features copied from x).  One workgroup of 64 per filter: lane 0 the body part, a lane per feature.
__global__ __launch_bounds__(64) void k_boxops(StreamArgs a, int minus, const double* __restrict__ x1_all, const double* __restrict__ v_all,
                                               double* __restrict__ out_all) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, nx = a.nx, len = a.len[b];
  const double* x1 = x1_all + (long)b * nx;
  if (!minus) {
    const double* dx = v_all + (long)b * n;
    double* o = out_all + (long)b * nx;
    for (int i = xZ + 5 * len + tid; i < nx; i += 64) o[i] = x1[i];
    if (tid == 0) {
      double xo[17];
      body_boxplus(x1, dx, xo);                                           // vi_ekf_helper.cpp:90-92
      for (int i = 0; i < 17; i++) o[i] = xo[i];
    }
    for (int f = tid; f < len; f += 64) {                                 // :93-97
      double qn[4];
      q_feat_boxplus(x1 + xZ + 5 * f, dx[dxZ + 3 * f], dx[dxZ + 3 * f + 1], qn);
      for (int i = 0; i < 4; i++) o[xZ + 5 * f + i] = qn[i];
      o[xZ + 5 * f + 4] = x1[xZ + 5 * f + 4] + dx[dxZ + 3 * f + 2];
    }
  } else {
    const double* x2 = v_all + (long)b * nx;
    double* o = out_all + (long)b * n;
    for (int i = dxZ + 3 * len + tid; i < n; i += 64) o[i] = 0.0;
    if (tid == 0) {                                                       // :102-104
      for (int i = 0; i < 6; i++) o[dxPOS + i] = x1[xPOS + i] - x2[xPOS + i];
      double d3[3];
      q_boxminus_dev(x1 + xATT, x2 + xATT, d3);
      for (int i = 0; i < 3; i++) o[dxATT + i] = d3[i];
      for (int i = 0; i < 7; i++) o[dxB_A + i] = x1[xB_A + i] - x2[xB_A + i];
    }
    for (int f = tid; f < len; f += 64) {                                 // :106-110
      double d2[2];
      q_feat_boxminus_dev(x1 + xZ + 5 * f, x2 + xZ + 5 * f, d2);
      o[dxZ + 3 * f] = d2[0]; o[dxZ + 3 * f + 1] = d2[1];
      o[dxZ + 3 * f + 2] = x1[xZ + 5 * f + 4] - x2[xZ + 5 * f + 4];
    }
  }
}

// keyframe_reset(xm, xp, N) (vi_ekf_kfr.cpp:6-12 over :56-145): the reset map of the state -- position <- 0, yaw <- 0 -- and its
// Jacobian N = I with a zero position block and the attitude block of the RMEKF paper (:134-142); xm, xp [B][nx], N [B][n][n]
// column-major.  The same expressions as k_keyframe_reset, which applies them to the filter.
__global__ __launch_bounds__(256) void k_eval_reset(StreamArgs a, const double* __restrict__ xm_all, double* __restrict__ xp_all,
                                                    double* __restrict__ N_all) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, nx = a.nx;
  const double* xm = xm_all + (long)b * nx;
  const double* q = xm + xATT;
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const double roll = atan2(2.0 * (qw * qx + qy * qz), 1.0 - 2.0 * (qx * qx + qy * qy));    // src/quat.cpp:211-214
  const double pitch = asin(2.0 * (qw * qy - qz * qx));                                     // :216-219
  const double cp = cos(roll), sp = sin(roll), tt = tan(pitch);                             // vi_ekf_kfr.cpp:134-136
  const double Na[9] = {1.0, sp * tt, cp * tt, 0.0, cp * cp, -cp * sp, 0.0, -cp * sp, sp * sp};   // row-major (:139-142)
  if (xp_all) {
    double* xp = xp_all + (long)b * nx;
    for (int i = tid; i < nx; i += 256) xp[i] = xm[i];
    __syncthreads();
    if (tid == 0) {
      const double cr = cos(roll / 2.0), ct = cos(pitch / 2.0), sr = sin(roll / 2.0), st = sin(pitch / 2.0);
      xp[xPOS] = 0.0; xp[xPOS + 1] = 0.0; xp[xPOS + 2] = 0.0;                                 // :65
      xp[xATT] = cr * ct; xp[xATT + 1] = sr * ct; xp[xATT + 2] = cr * st; xp[xATT + 3] = -sr * st;   // from_euler(roll, pitch, 0)
    }
  }
  if (N_all) {
    double* Nm = N_all + (long)b * n * n;
    for (long e = tid; e < (long)n * n; e += 256) {
      const int r = (int)(e % n), c = (int)(e / n);
      double v = (r == c) ? 1.0 : 0.0;
      if (r < 3 && c < 3) v = 0.0;
      if (r >= dxATT && r < dxATT + 3 && c >= dxATT && c < dxATT + 3) v = Na[(r - dxATT) * 3 + (c - dxATT)];
      Nm[e] = v;
    }
  }
}
#endif

}  // namespace viekf
