/* abi_smoke.c -- a C11 consumer of include/viekf.h (built by tests/test_abi_consumers.py with
 * gcc -std=c11 -Wall -Wextra -Werror -pedantic): proves the header is plain C and that a C program links and
 * drives the library.  Without a GPU it exercises the host-only entry points and checks that the compute entry
 * points refuse to run (no CPU fallback); with one (argv[1] = "gpu") it runs a propagate + feature updates through
 * the reference-shaped call sequence and prints the state for the Python side to compare with the oracle. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "viekf.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != VIEKF_OK) {                                                          \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, viekf_last_error());          \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main(int argc, char **argv) {
  viekf_params p;
  int32_t ndev = -1;
  if (viekf_abi_version() != VIEKF_ABI_VERSION) return 2;
  CHECK(viekf_params_default(&p));
  if (p.q_b_c[0] != 1.0 || p.use_partial_update != 1) return 3;
  if (argc > 2) CHECK(viekf_params_load_yaml(argv[2], &p));
  CHECK(viekf_device_count(&ndev));
  if (viekf_params_default(NULL) != VIEKF_ERR_INVALID) return 4;
  if (argc < 2 || strcmp(argv[1], "gpu") != 0) {
    viekf_batch *b = NULL;
    if (ndev == 0 && viekf_batch_create(2, 3, &p, 0, &b) != VIEKF_ERR_NO_DEVICE) return 5;   /* no CPU fallback */
    if (b) viekf_batch_destroy(b);
    printf("abi ok: version %d, %d device(s)\n", viekf_abi_version(), (int)ndev);
    return 0;
  }
  {
    enum { B = 2, N = 4, NX = 17 + 5 * N };
    viekf_batch *b = NULL;
    double u[B][6], dt[B], pix[B][2], z[B][N][2], R[4] = {10.0, 0.0, 0.0, 10.0}, x[B][NX];
    int32_t slot[B][N], res[B][N], ok[B];
    int i, f, k;
    CHECK(viekf_batch_create(B, N, &p, 0, &b));
    for (f = 0; f < N; f++) {
      for (i = 0; i < B; i++) { pix[i][0] = 200.0 + 60.0 * f + 5.0 * i; pix[i][1] = 150.0 + 40.0 * f; }
      CHECK(viekf_batch_init_feature(b, &pix[0][0], NULL, NULL, ok, VIEKF_HOST));
      if (!ok[0] || !ok[1]) return 6;
    }
    for (k = 0; k < 3; k++) {
      for (i = 0; i < B; i++) {
        const double uu[6] = {0.1, -0.05, -9.80665, 0.01, 0.02, -0.01};
        memcpy(u[i], uu, sizeof uu);
        dt[i] = 0.004;
        for (f = 0; f < N; f++) {
          slot[i][f] = N - 1 - f;   /* the reference's reverse in-frame order */
          z[i][f][0] = 200.0 + 60.0 * (N - 1 - f) + 5.0 * i + 0.3 * k;
          z[i][f][1] = 150.0 + 40.0 * (N - 1 - f) - 0.2 * k;
        }
      }
      CHECK(viekf_batch_step(b, &u[0][0], dt, &z[0][0][0], &slot[0][0], N, R, 0, &res[0][0], VIEKF_HOST));
      for (i = 0; i < B; i++)
        for (f = 0; f < N; f++)
          if (res[i][f] != VIEKF_MEAS_SUCCESS) return 7;
    }
    CHECK(viekf_batch_get_state(b, &x[0][0], NULL, NULL, VIEKF_HOST));
    for (i = 0; i < B; i++) {
      for (k = 0; k < NX; k++) printf("%.17g ", x[i][k]);
      printf("\n");
    }
    CHECK(viekf_batch_destroy(b));
  }
  return 0;
}
