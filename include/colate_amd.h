/* include/colate_amd.h -- C ABI of libcolate_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of leospeidel/Colate: `Colate --mode mut` on
 * precomputed .colate.in inputs, i.e. the block-bootstrap driver and the EM loop
 * that turns age-binned shared / not-shared mutation counts into pairwise
 * coalescence rates.  The reference has no FFI; its de-facto boundary is the C++
 * class `coal_EM` (include/coal/coal_EM.hpp:14-63) used at exactly one site,
 * include/coal/coal.cpp:3698-3721, inside the per-replicate EM loop
 * include/coal/coal.cpp:3675-3827.  Calling a GPU once per age bin would be
 * meaningless, so the replacement boundary is one coarse call per batch of
 * bootstrap replicates (colate_em_batch), plus the single E-step
 * (colate_em_estep) that corresponds to one pass of coal.cpp:3698-3733.
 * INTEGRATION.md shows the patch a maintainer would apply to coal.cpp.
 *
 * Conventions: plain pointers and sizes, row-major, IEEE double; caller owns
 * every buffer; the library keeps no state between calls besides the selected
 * device.  Every function returns 0 on success or a negative COLATE_E* code and
 * never aborts; colate_last_error() gives a message.  "_device" variants take
 * pointers to device (HBM) memory and enqueue on a HIP stream without
 * synchronising; the plain variants take host pointers and are synchronous.
 * There is NO CPU fallback: without a usable HIP device the compute entry
 * points fail with COLATE_ENODEVICE.
 */
#ifndef COLATE_AMD_H
#define COLATE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define COLATE_OK 0
#define COLATE_EINVAL (-1)    /* bad argument (sizes, NULL, unsorted grids, ...) */
#define COLATE_ENODEVICE (-2) /* no usable HIP device                            */
#define COLATE_EHIP (-3)      /* a HIP runtime call failed                       */
#define COLATE_ELIMIT (-4)    /* E above 1024 or A above 256 (compiled limits)   */
#define COLATE_EIO (-5)       /* file could not be read / written                */

/* per-replicate flags in out_flags[]: conditions on which the reference aborts
 * through assert() (coal.cpp:3711-3714, coal_EM.cpp:128-129, 351) or that it
 * cannot report (iteration cap) */
#define COLATE_FLAG_NAN 1
#define COLATE_FLAG_NEG 2
#define COLATE_FLAG_MAXITER 4
/* Not an error: the last COLATE_UNRESOLVED_EPOCHS(flags) epochs of this replicate are below the resolution of the
 * reference's own arithmetic -- the survival probability there is so small that the `integ` term of its denominators
 * (coal_EM.cpp:270-274, 445-449) is rounding residue, and the reference's printed rate changes by more than 1e-8
 * (relative; by orders of magnitude a few epochs further on) when its libm's exp()/log() return a neighbouring double.
 * Rates of such epochs are returned (deep in that regime they are the floor, as in the reference) but are outside the
 * 1e-6 parity claim; all other epochs are inside it.  out_flags == 0 therefore still means: clean and fully
 * reproducible.  With --bins 3,7,0.2 (23 epochs) no epoch is ever unresolved on whole-genome tables; with
 * --bins 2,7.95,0.05 (122 epochs) the last ~17 are (profiles/parity/).  The verdict is about the
 * fixed point: it is made for runs that end by the stop rule at the reference's tolerance (COLATE_DEFAULT_REL_TOL); a run cut
 * by max_iter (COLATE_FLAG_MAXITER) or stopped at a looser tolerance still carries its path. */
#define COLATE_FLAG_UNRESOLVED 8
/* the error-like bits (NAN | NEG | MAXITER): COLATE_STATUS_FLAGS(f) != 0 means "the reference would have aborted, or the
 * iteration cap ended the run"; COLATE_FLAG_UNRESOLVED is deliberately not part of it */
#define COLATE_STATUS_FLAGS(flags) ((flags) & 0x07)
#define COLATE_UNRESOLVED_EPOCHS(flags) ((int)((unsigned)(flags) >> 8))

/* compiled limits of the EM kernel: up to 16 epochs per lane of a 64-lane wave (the reference builds any number of epochs,
 * coal.cpp:3551-3632: `--bins 3,7,0.01` gives 404; beyond 256 a slower instantiation runs), one age bin per thread */
#define COLATE_MAX_EPOCHS 1024
#define COLATE_MAX_AGE_BINS 256

/* reference defaults (coal.cpp:3656, 3822, 3798-3803, 3636) */
#define COLATE_DEFAULT_MAX_ITER 100000
#define COLATE_DEFAULT_MIN_ITER 1000
#define COLATE_DEFAULT_REL_TOL 1e-7
#define COLATE_DEFAULT_RATE_FLOOR 5e-9
#define COLATE_DEFAULT_INIT_RATE (1.0 / 20000.0)

const char* colate_version(void);
const char* colate_last_error(void);
int colate_device_count(void);       /* >= 0, or COLATE_ENODEVICE */
int colate_set_device(int ordinal);  /* device used by the calling thread's later calls */
/* Creates the device's HIP context now (from any thread) instead of inside the first compute call: lets a host overlap
 * the few hundred ms a fresh process pays for it with its own input parsing.  No reference counterpart (CPU code). */
int colate_warm_up(int ordinal);
/* 1 once this process has talked to the HIP runtime through this library (any compute or device entry point), else 0.
 * Such a process must not fork() children that use the GPU: colate_mut_main refuses `--ranks N` (which forks one
 * process per GPU) when this is set -- start `Colate --ranks N` as a fresh process instead. */
int colate_device_touched(void);
/* Diagnostic: which build of the EM kernel a batch of this shape runs on the current device
 * (0 latency/max-ilp, 1 latency/default scheduler, 2 throughput), or a negative code. */
int colate_em_kernel_variant(int B, int E);
/* Diagnostic: force that choice for E <= 128 (0, 1 or 2; any other value = automatic again).  Process-wide.  The
 * environment variable COLATE_EM_VARIANT=latency-ilp|latency|throughput sets the initial value (read once). */
int colate_em_force_variant(int variant);

/* ---- the EM hot path ------------------------------------------------------
 * Replaces coal.cpp:3675-3827 (bootstrap EM driver: coal_EM construction,
 * EM_shared/EM_notshared per age bin, accumulation, M-step, floor, stop rule)
 * for B replicates at once.
 *   age_grid[A]            ascending age-bin representatives (coal.cpp:3126-3137; A = 185)
 *   cnt_shared[B][A], cnt_notshared[B][A]   bootstrap count tables (coal.cpp:3344-3451)
 *   epochs[E]              epoch starts in generations, epochs[0] <= age_grid[0], non-decreasing
 *   init_rates[E]          starting rates (coal.cpp:3636-3646)
 *   max_iter, min_iter, rel_tol, rate_floor   100000, 1000, 1e-7, 5e-9 in the reference
 *   out_rates[B][E], out_iters[B] (the reference's "Total iterations"), out_loglik[B], out_flags[B]
 */
int colate_em_batch(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                    const double* cnt_notshared, const double* epochs, const double* init_rates,
                    int max_iter, int min_iter, double rel_tol, double rate_floor,
                    double* out_rates, int* out_iters, double* out_loglik, int* out_flags);

/* Same, all pointers in device memory, asynchronous on `hip_stream` (a
 * hipStream_t, NULL = default stream).  epochs_per_replicate / rates_per_replicate
 * != 0 select [B][E] layouts for epochs / init_rates (batched all-pairs, where
 * every (target, reference) pair brings its own epochs); 0 = one shared [E] row.
 * Host-side argument checks that need the data (sortedness) are skipped. */
int colate_em_batch_device(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                           const double* cnt_notshared, const double* epochs,
                           int epochs_per_replicate, const double* init_rates,
                           int rates_per_replicate, int max_iter, int min_iter, double rel_tol,
                           double rate_floor, double* out_rates, int* out_iters,
                           double* out_loglik, int* out_flags, void* hip_stream);

/* Host-pointer variant with epochs[B][E] and init_rates[B][E] per replicate: one launch for many
 * (target, reference) pairs whose epochs differ (an ancient sample inserts its age as an epoch,
 * coal.cpp:3597-3624); all rows must have the same E.  Used by `Colate --pairs` (batched all-pairs). */
int colate_em_batch_rows(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                         const double* cnt_notshared, const double* epochs, const double* init_rates,
                         int max_iter, int min_iter, double rel_tol, double rate_floor,
                         double* out_rates, int* out_iters, double* out_loglik, int* out_flags);

/* Host-pointer variant that shards the B replicates over several GPUs of the node from ONE
 * process: contiguous, balanced ranges (replicate i of device d = global lo_d + i), one stream per
 * device, all launches in flight together, results gathered in replicate order.  `devices` lists
 * `num_devices` HIP ordinals (an ordinal may repeat: its shards then share that GPU).  The
 * one-process-per-GPU form of the same sharding (torch.distributed + RCCL all-gather) is
 * colate_amd/distributed.py; there is no exchange between shards inside the EM. */
int colate_em_batch_sharded(int num_devices, const int* devices, int B, int E, int A,
                            const double* age_grid, const double* cnt_shared,
                            const double* cnt_notshared, const double* epochs,
                            const double* init_rates, int max_iter, int min_iter, double rel_tol,
                            double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                            int* out_flags);

/* The same sharding for the per-row form (epochs[B][E], init_rates[B][E], as colate_em_batch_rows):
 * the batched all-pairs run over several GPUs. */
int colate_em_batch_rows_sharded(int num_devices, const int* devices, int B, int E, int A,
                                 const double* age_grid, const double* cnt_shared,
                                 const double* cnt_notshared, const double* epochs,
                                 const double* init_rates, int max_iter, int min_iter, double rel_tol,
                                 double rate_floor, double* out_rates, int* out_iters,
                                 double* out_loglik, int* out_flags);

/* One E-step = one pass of coal.cpp:3698-3733 for each of B replicates with the
 * rates given per replicate: rates[B][E] -> num_acc[B][E], den_acc[B][E]
 * (coal_rates_num / coal_rates_denom), loglik[B], flags[B]. */
int colate_em_estep(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                    const double* cnt_notshared, const double* epochs, const double* rates,
                    double* num_acc, double* den_acc, double* loglik, int* flags);
int colate_em_estep_device(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                           const double* cnt_notshared, const double* epochs, const double* rates,
                           double* num_acc, double* den_acc, double* loglik, int* flags,
                           void* hip_stream);

/* ---- host-side pieces of mut() around the hot path (CPU, no device needed) ----
 * coal.cpp:3126-3137: the 185-point age grid.  Returns A or COLATE_EINVAL if cap < A. */
int colate_age_grid(double* age_grid, int cap);

/* coal.cpp:3551-3632: epochs from `--bins x,y,step` (std::stof semantics), with the
 * ancient-sample rule; age = max(target_age, reference_age)/years_per_gen in
 * generations.  Returns E (>0) or a negative code; *ep_null as coal.cpp:3622. */
int colate_epochs_from_bins(const char* bins, double age, double years_per_gen, double* epochs,
                            int cap, int* ep_null);

/* coal.cpp:3508-3549 + 3638-3646: epochs and initial rates from a `--coal` file.
 * Returns E or a negative code. */
int colate_epochs_from_coal(const char* path, double age, double* epochs, double* init_rates,
                            int cap);

/* coal.cpp:3344-3451: block bootstrap.  Draws the multinomial block weights for
 * num_bootstrap replicates from std::mt19937 state `rng_state` (opaque, from
 * colate_rng_*), forms the weighted block sums and applies the F
 * redistribution of the age_begin<=0 mutations.  Tables are [nb][A]; the emp
 * tables hold row 0 of the reference's A*A tables (the only row it reads). */
void* colate_rng_create(unsigned int seed); /* std::mt19937, coal.cpp:3157-3162 */
void colate_rng_destroy(void* rng_state);
int colate_bootstrap_counts(void* rng_state, int num_bootstrap, int nb, int A,
                            const double* age_grid, double age, const double* sh_block,
                            const double* ns_block, const double* sh_emp_block,
                            const double* ns_emp_block, double* cnt_shared,
                            double* cnt_notshared);

/* coal.cpp:3350-3357 alone: the multinomial block weights w[num_bootstrap][nb] (all ones when
 * num_bootstrap == 1), drawn from the same std::mt19937 -- the host half of the GPU bootstrap below. */
int colate_bootstrap_weights(void* rng_state, int num_bootstrap, int nb, double* weights);

/* coal.cpp:3358-3451 alone, on the host, from weights drawn before (colate_bootstrap_weights): the weighted block sums
 * and the F redistribution.  colate_bootstrap_counts = colate_bootstrap_weights + this.  The host twin of the GPU
 * bootstrap below (bit-identical); what `--counts_only` runs, which needs no device. */
int colate_bootstrap_counts_from_weights(int num_bootstrap, int nb, int A, const double* age_grid, double age,
                                         const double* weights, const double* sh_block, const double* ns_block,
                                         const double* sh_emp_block, const double* ns_emp_block, double* cnt_shared,
                                         double* cnt_notshared);

/* coal.cpp:3358-3441 on the GPU: weighted block sums + F redistribution for B replicates, all
 * pointers in device memory (tables [nb][A], weights [B][nb], cnt_* [B][A]), asynchronous on
 * hip_stream; results are bit-identical to colate_bootstrap_counts.  `status` (device int, may be
 * NULL) receives 1 if the sample age lies outside the age grid. */
int colate_bootstrap_counts_device(int B, int nb, int A, const double* age_grid, double age,
                                   const double* weights, const double* sh_block, const double* ns_block,
                                   const double* sh_emp_block, const double* ns_emp_block,
                                   double* cnt_shared, double* cnt_notshared, int* status,
                                   void* hip_stream);

/* Block tables -> rates in one call, host pointers: uploads the [nb][A] tables and the weights
 * w[B][nb], runs the bootstrap kernel and the EM kernel back to back on the device (the count
 * tables never visit the host) and returns the EM outputs; out_cnt_shared / out_cnt_notshared
 * (each [B][A], may be NULL) additionally receive the count tables.  This is what `Colate --mode
 * mut` of colate_amd runs after reading the inputs (coal.cpp:3344-3451 + 3675-3827). */
int colate_bootstrap_em_batch(int B, int nb, int E, int A, const double* age_grid, double age,
                              const double* weights, const double* sh_block, const double* ns_block,
                              const double* sh_emp_block, const double* ns_emp_block,
                              const double* epochs, const double* init_rates, int max_iter,
                              int min_iter, double rel_tol, double rate_floor, double* out_rates,
                              int* out_iters, double* out_loglik, int* out_flags,
                              double* out_cnt_shared, double* out_cnt_notshared);

/* ---- batched all-pairs (BASELINE configs[4]) --------------------------------------------
 * The reference is run once per (target, reference) pair (coal.cpp:2071-2321 + 3071-3863 each time).  Here G pairs
 * ("groups") go through ONE bootstrap launch and ONE EM launch: group g has its own genome-block tables (group_nb[g]
 * blocks; the four tables of all groups are concatenated in group order, [sum_g nb_g][A]), its own sample age, its own
 * epochs[g][E] / init_rates[g][E] (an ancient sample inserts its age as an epoch, coal.cpp:3597-3624; all groups of a
 * call have the same E) and B bootstrap replicates with weights [B][nb_g], concatenated in group order as well.
 * Row r = g * B + i of every output is replicate i of group g; results are bit-identical to G separate
 * colate_bootstrap_em_batch calls.  out_cnt_* ([G*B][A], may be NULL) receive the count tables. */
int colate_bootstrap_em_batch_groups(int G, int B, int E, int A, const double* age_grid, const int* group_nb,
                                     const double* group_age, const double* weights, const double* sh_block,
                                     const double* ns_block, const double* sh_emp_block,
                                     const double* ns_emp_block, const double* epochs, const double* init_rates,
                                     int max_iter, int min_iter, double rel_tol, double rate_floor,
                                     double* out_rates, int* out_iters, double* out_loglik, int* out_flags,
                                     double* out_cnt_shared, double* out_cnt_notshared);
/* The bootstrap of rows [row_lo, row_hi) of such a batch, all pointers in device memory, asynchronous on hip_stream.
 * The group arrays describe groups [group_first, group_first + G) (every row must belong to one of them):
 * group_block_off[g] = number of genome blocks in front of group g in the tables given, group_weight_off[g] = number
 * of weights in front of its [B][nb_g] weights; cnt_*[row_hi - row_lo][A]; `status` (device int, required) receives 1
 * if a sample age lies outside the age grid. */
int colate_bootstrap_counts_groups_device(int G, int B, int group_first, int row_lo, int row_hi, int A,
                                          const double* age_grid, const int* group_nb,
                                          const long long* group_block_off, const long long* group_weight_off,
                                          const double* group_age, const double* weights, const double* sh_block,
                                          const double* ns_block, const double* sh_emp_block,
                                          const double* ns_emp_block, double* cnt_shared, double* cnt_notshared,
                                          int* status, void* hip_stream);

/* The host-pointer entry points above keep one device buffer, one pinned staging buffer and one stream per calling
 * thread between calls (grown on demand); this frees them. */
int colate_release_workspace(void);

/* ---- one process per GPU: replicate shards + ONE RCCL all-gather over xGMI -----------
 * The reference runs its replicates one after the other (coal.cpp:3675-3846); they are independent, so rank r of
 * `nranks` processes runs the contiguous range colate_shard_bounds(B, nranks, r) on ITS current device and a
 * single ncclAllGather of the packed results (rates, log-likelihood, iterations, flags: (8E + 16) bytes per
 * replicate) gives every rank all B results in replicate order -- bit-identical to the one-process run.  No other
 * exchange exists on the path.  Rank 0 obtains the 128-byte id (colate_comm_unique_id) and hands it to the others
 * by any means (a pipe, a file, MPI, torch.distributed's store); every rank then calls colate_comm_create after
 * selecting its device (colate_set_device).  RCCL (librccl.so.1) is loaded on first use.  `Colate --ranks N` is
 * the command-line form (it forks N such processes itself); colate_amd/distributed.py + bench.py are the
 * torch.distributed form of the same sharding. */
#define COLATE_COMM_ID_BYTES 128
int colate_shard_bounds(int B, int nranks, int rank, int* lo, int* hi);
int colate_comm_unique_id(void* id /* [COLATE_COMM_ID_BYTES] */);
int colate_comm_create(const void* id, int nranks, int rank, void** comm);
int colate_comm_destroy(void* comm);
/* colate_em_batch over the communicator: every rank passes the FULL host arrays (cnt_*[B][A]) and receives the
 * full outputs; it computes rows [lo, hi) only. */
int colate_em_batch_allgather(void* comm, int B, int E, int A, const double* age_grid, const double* cnt_shared,
                              const double* cnt_notshared, const double* epochs, const double* init_rates,
                              int max_iter, int min_iter, double rel_tol, double rate_floor, double* out_rates,
                              int* out_iters, double* out_loglik, int* out_flags);
/* colate_bootstrap_em_batch over the communicator (weights[B][nb] in full on every rank: they come from the
 * run's one std::mt19937 stream, which every rank replays identically from --seed). */
int colate_bootstrap_em_batch_allgather(void* comm, int B, int nb, int E, int A, const double* age_grid, double age,
                                        const double* weights, const double* sh_block, const double* ns_block,
                                        const double* sh_emp_block, const double* ns_emp_block, const double* epochs,
                                        const double* init_rates, int max_iter, int min_iter, double rel_tol,
                                        double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                                        int* out_flags);

/* colate_bootstrap_em_batch_groups over the communicator: the G * B rows are sharded like replicates
 * (colate_shard_bounds(G * B, nranks, rank)); a rank passes the arrays of the groups its rows belong to ONLY --
 * groups [group_first, group_first + group_count) with group_first = lo / B, group_count = (hi - 1) / B - lo / B + 1 --
 * so that it needs to read and fill the tables of those pairs alone; a rank without rows passes group_count = 0.
 * Every rank receives all G * B results.  `Colate --pairs FILE --ranks N` is the command-line form. */
int colate_bootstrap_em_batch_groups_allgather(void* comm, int G, int B, int group_first, int group_count, int E, int A,
                                               const double* age_grid, const int* group_nb, const double* group_age,
                                               const double* weights, const double* sh_block, const double* ns_block,
                                               const double* sh_emp_block, const double* ns_emp_block,
                                               const double* epochs, const double* init_rates, int max_iter,
                                               int min_iter, double rel_tol, double rate_floor, double* out_rates,
                                               int* out_iters, double* out_loglik, int* out_flags);

/* coal.cpp:3660-3672, 3830-3847: the .coal text (6 significant digits, trailing blank). */
int colate_write_coal(const char* path, int B, int E, const double* epochs, const double* rates,
                      int is_ancient, int ep_null);

/* The whole `Colate --mode mut` command line for the .colate.in / .colate_mat
 * inputs (Colate.cpp:6-116 -> coal.cpp:3071-3863): same option names, same
 * stderr progress lines, same .coal output.  Returns the process exit code. */
int colate_mut_main(int argc, char** argv);

#ifdef __cplusplus
}
#endif
#endif /* COLATE_AMD_H */
