/* oracle/colate_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's `Colate --mode mut` EM path
 * (leospeidel/Colate, include/coal/coal_EM.cpp and include/coal/coal.cpp).
 * It is the *checker* for the HIP product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under colate_amd/ may include, link or call it.
 *
 * Parity pin: this restatement is checked against the reference itself
 * (compiled from /root/reference into oracle/_ref/ by oracle/Makefile) and
 * against the golden vectors committed under tests/golden/ that were generated
 * from that reference build (tests/golden/make_golden.py).
 *
 * All arithmetic is IEEE double, no FMA contraction (-ffp-contract=off), libm
 * exp/log/log1p -- the same operations, in the same order, as the reference.
 */
#ifndef COLATE_ORACLE_H
#define COLATE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* flags returned per replicate (the reference would abort on these asserts,
 * coal.cpp:3711-3714, 3724-3727) */
#define ORACLE_FLAG_NAN 1      /* a num/denom entry was NaN          */
#define ORACLE_FLAG_NEG 2      /* a num/denom entry was negative     */
#define ORACLE_FLAG_MAXITER 4  /* iteration cap reached (no `break`) */

/* coal_EM.cpp:5-31 / 33-58 */
double oracle_logsumexp(double loga, double logb);
double oracle_logminusexp(double loga, double logb);

/* coal_EM.cpp:97-151 as called from the constructor coal_EM.hpp:38-50
 * (t_int = epochs, ep_index = identity). */
void oracle_get_AB(int E, const double* epochs, const double* rates, double* A_ep, double* B_ep);

/* coal_EM.cpp:153-295 and 297-468, age_begin == age_end == age only (the only
 * way coal.cpp:3708/3721 calls them).  num/denom are [E]; returns the
 * log-normaliser (0.0 and zeroed outputs on the failure path). */
double oracle_em_shared(int E, const double* epochs, const double* rates, const double* A_ep,
                        const double* B_ep, double age, double* num, double* denom);
double oracle_em_notshared(int E, const double* epochs, const double* rates, const double* A_ep,
                           const double* B_ep, double age, double* num, double* denom);

/* coal.cpp:3698-3733: one E-step over the age grid.  Returns log-likelihood;
 * num_acc/den_acc [E] are overwritten with the accumulated sums; *flags |= NAN/NEG. */
double oracle_estep(int E, int A, const double* epochs, const double* rates, const double* age_grid,
                    const double* cnt_shared, const double* cnt_notshared, double* num_acc,
                    double* den_acc, int* flags);

/* coal.cpp:3771-3815 (EM branch only): rates updated in place from num_acc/den_acc. */
void oracle_mstep(int E, const double* num_acc, const double* den_acc, double rate_floor,
                  double* rates);

/* coal.cpp:3675-3827 for one replicate: EM to the reference's stop rule
 * (`ll/prev_ll > 1 - rel_tol` and `iter > min_iter`, cap max_iter).
 * out_iters = the `iter` at which the loop broke (what the reference prints as
 * "Total iterations"), or max_iter if the cap was reached. */
void oracle_em_run(int E, int A, const double* age_grid, const double* epochs,
                   const double* init_rates, const double* cnt_shared, const double* cnt_notshared,
                   int max_iter, int min_iter, double rel_tol, double rate_floor, double* out_rates,
                   int* out_iters, double* out_loglik, int* out_flags);

/* B replicates, rows of cnt_* are [B][A]; out_rates [B][E]. */
void oracle_em_batch(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                     const double* cnt_notshared, const double* epochs, const double* init_rates,
                     int max_iter, int min_iter, double rel_tol, double rate_floor,
                     double* out_rates, int* out_iters, double* out_loglik, int* out_flags);

/* coal.cpp:3126-3137: age_bin[0]=0, age_bin[k]=exp((k-1)/10)/10; returns A (=185). */
int oracle_age_grid(double* age_grid, int cap);

/* coal.cpp:3551-3632 (--bins x,y,step; std::stof semantics -> strtof) with the
 * ancient-sample insertion rule.  age in generations (max(target,ref)/years_per_gen).
 * Returns E (or <0 on format error); *ep_null as coal.cpp:3622. */
int oracle_epochs_from_bins(const char* bins, double age, double years_per_gen, double* epochs,
                            int cap, int* ep_null);

/* coal.cpp:3508-3549 + 3638-3646: epochs (float-parsed) and starting rates from a `--coal` file. */
int oracle_epochs_from_coal(const char* path, double age, double* epochs, double* init_rates, int cap);

/* coal.cpp:3350-3451 for one replicate, given the block weights:
 * weighted block sums (3358-3390; only emp row 0 is live) and the F
 * redistribution of emp row 0 into the shared counts (3392-3441).
 * blocks_* are [nb][A] (emp = row 0 of the reference's A*A tables). */
void oracle_bootstrap_counts(int nb, int A, const double* age_grid, double age,
                             const double* weights, const double* sh_block,
                             const double* ns_block, const double* sh_emp_block,
                             const double* ns_emp_block, double* cnt_shared,
                             double* cnt_notshared);

/* std::mt19937 (32-bit Mersenne twister) + libstdc++-11 distributions, restated:
 * the reference draws its block-bootstrap weights (coal.cpp:3350-3357) and its
 * age samples (coal.cpp:2262, 2282) from one such generator. */
typedef struct {
  unsigned int mt[624];
  int idx;
} oracle_mt19937;
void oracle_mt_seed(oracle_mt19937* g, unsigned int seed);
unsigned int oracle_mt_next(oracle_mt19937* g);
/* std::uniform_int_distribution<int>(0, n-1)(rng), GCC 11 (Lemire) algorithm */
int oracle_uniform_int(oracle_mt19937* g, int n);
/* std::uniform_real_distribution<double>(0,1)(rng) = generate_canonical<double,53> */
double oracle_uniform_real01(oracle_mt19937* g);
/* coal.cpp:3350-3357: weights[nb]; B==1 -> all ones, else multinomial */
void oracle_block_weights(oracle_mt19937* g, int nb, int num_bootstrap, double* weights);

/* Checker-side experiment, OFF (seed 0) by default: with a non-zero seed every exp/log/log1p result of the EM
 * path moves to a neighbouring double at random -- "the same source on another < 1 ulp libm".  Not thread-safe. */
void oracle_set_libm_noise(unsigned long long seed);

#ifdef __cplusplus
}
#endif
#endif
