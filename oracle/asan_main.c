/* oracle/asan_main.c -- TEST INFRASTRUCTURE: drives the oracle under AddressSanitizer / UBSan (`make -C oracle asan`).
 *   oracle_asan COUNTS B BINS [AGE_GENERATIONS [COAL_FILE]]
 * COUNTS is a count-table file in the .colate_mat layout (grid line, then per replicate a line of shared and a line
 * of not-shared counts; what `Colate --counts_out` writes).  Runs the epoch builders, a capped EM (60 iterations) and
 * one E-step per replicate, the age grid, and the mt19937 / block-weight restatement; prints a checksum. */
#include <stdio.h>
#include <stdlib.h>

#include "colate_oracle.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  int B = atoi(argv[2]);
  double age = argc > 4 ? atof(argv[4]) : 0.0;
  double grid[256], grid_file[256], epochs[512], init[512];
  int A = oracle_age_grid(grid, 256);
  FILE* f = fopen(argv[1], "r");
  if (!f || A < 1 || B < 1) return 2;
  double* sh = (double*)calloc((size_t)B * A, sizeof(double));
  double* ns = (double*)calloc((size_t)B * A, sizeof(double));
  for (int b = 0; b < A; b++)
    if (fscanf(f, "%lf", &grid_file[b]) != 1) return 2;
  for (int i = 0; i < B; i++) {
    for (int b = 0; b < A; b++)
      if (fscanf(f, "%lf", &sh[(size_t)i * A + b]) != 1) return 2;
    for (int b = 0; b < A; b++)
      if (fscanf(f, "%lf", &ns[(size_t)i * A + b]) != 1) return 2;
  }
  fclose(f);
  int ep_null = 0, E;
  if (argc > 5) {
    E = oracle_epochs_from_coal(argv[5], age, epochs, init, 512);
  } else {
    E = oracle_epochs_from_bins(argv[3], age, 28.0, epochs, 512, &ep_null);
    for (int e = 0; e < E; e++) init[e] = 1.0 / 20000.0;
  }
  if (E < 2) return 3;
  double* rates = (double*)calloc((size_t)B * E, sizeof(double));
  double* ll = (double*)calloc((size_t)B, sizeof(double));
  int* iters = (int*)calloc((size_t)B, sizeof(int));
  int* flags = (int*)calloc((size_t)B, sizeof(int));
  oracle_em_batch(B, E, A, grid_file, sh, ns, epochs, init, 60, 20, 1e-7, 5e-9, rates, iters, ll, flags);
  double sum = 0.0;
  double* N = (double*)calloc((size_t)E, sizeof(double));
  double* D = (double*)calloc((size_t)E, sizeof(double));
  for (int i = 0; i < B; i++) {
    int fl = 0;
    sum += oracle_estep(E, A, epochs, rates + (size_t)i * E, grid_file, sh + (size_t)i * A, ns + (size_t)i * A, N, D, &fl);
    for (int e = 0; e < E; e++) sum += rates[(size_t)i * E + e];
  }
  oracle_mt19937 g;
  oracle_mt_seed(&g, 12345u);
  double w[64];
  oracle_block_weights(&g, 64, 3, w);
  for (int j = 0; j < 64; j++) sum += w[j];
  sum += oracle_uniform_real01(&g) + oracle_uniform_int(&g, 17);
  printf("oracle_asan ok: E=%d iters[0]=%d checksum=%.17g\n", E, iters[0], sum);
  free(sh), free(ns), free(rates), free(ll), free(iters), free(flags), free(N), free(D);
  return 0;
}
