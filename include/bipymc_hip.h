/*
 * bipymc_hip.h -- C ABI of the MI355X-native DE-MC / DREAM population sampler.
 *
 * This is the drop-in boundary for the per-generation hot path of
 * wgurecky/bipymc's parallel samplers (reference paths relative to
 * /root/reference):
 *
 *   DeMcMpi._mcmc_run            bipymc/demc.py:63-151   generation driver, a/b pools, 2x Allgather
 *   DeMcMpi._update_chain_pool   bipymc/demc.py:153-196  DE-MC proposal + Metropolis
 *   DreamMpi._update_chain_pool  bipymc/dream.py:32-107  DREAM proposal (CR mask, multi-pair jump, jitter)
 *   DreamMpi._update_cr_ratios   bipymc/dream.py:119-140 crossover-probability adaptation
 *   DeMc._mut_prop_ratio / metropolis_accept  bipymc/samplers.py:328-336
 *   var_ball / var_box           bipymc/util.py:5-28
 *   McmcChain history            bipymc/chain.py:13-29,51-54,122-124
 *   targets                      bipymc/utils/{d100_gauss,dblgauss_rv,banana_rv}.py
 *   param_est / _super_chain     bipymc/demc.py:235-270
 *
 * The reference is pure Python and has no FFI of its own; the binding a
 * maintainer would add is the ctypes stub in INTEGRATION.md (the host-side
 * classes in bipymc_amd/{demc,dream}.py are that stub, written out).
 *
 * Conventions: opaque handle; every call returns 0 on success, non-zero on
 * error, and bpm_last_error() returns the message of the calling thread's last
 * failure.  No exceptions cross the boundary.  All buffers are caller-owned
 * host memory, row-major float64 unless stated; the library copies.  One host
 * thread per handle.  With world_size > 1 every rank must make the same
 * sequence of collective calls (create, init/set_state, begin_run, step,
 * propose/commit), exactly as all MPI ranks of the reference enter
 * comm.Allgather / comm.Barrier in lock-step.
 */
#ifndef BIPYMC_HIP_H
#define BIPYMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: bpm_step_timed gained n_launches (round 2) and the entry points below it were added; a caller built against version 1 is
 * refused by bpm_create / by the binding's bpm_abi_version check instead of passing a short argument list. */
#define BPM_ABI_VERSION 2

/* algo */
#define BPM_ALGO_DEMC 0  /* bipymc/demc.py:153-196 */
#define BPM_ALGO_DREAM 1 /* bipymc/dream.py:32-107 */
#define BPM_ALGO_DEMC_SYNC 2 /* bipymc/samplers.py:237-308 serial DeMc with delayed_accept=True: all chains
                              propose against the generation's starting states, updates banked; no a/b pools */

/* target ids: which ln_like_fn is evaluated on the device */
#define BPM_TARGET_HOST_CALLBACK 0  /* arbitrary Python ln_like_fn (samplers.py:36-43): propose/commit */
#define BPM_TARGET_GAUSS_EQUICORR 1 /* utils/d100_gauss.py:14-35; params [rho,c0,a,b,1/sigma_0..d-1] */
#define BPM_TARGET_MIXTURE_PAIRS 2  /* utils/dblgauss_rv.py:11-32 (d=2) and its pairwise d-dim extension */
#define BPM_TARGET_BANANA_2D 3      /* utils/banana_rv.py:11-37 */

#define BPM_MAX_CR 8
#define BPM_UID_BYTES 128
#define BPM_PUSH_BLOB_BYTES 256

typedef struct bpm_sampler* bpm_handle_t;

/* Constructor arguments: DeMcMpi.__init__ (demc.py:14-32) / DreamMpi.__init__ (dream.py:17-30). */
typedef struct bpm_config {
    int32_t abi_version; /* BPM_ABI_VERSION */
    int32_t algo;        /* BPM_ALGO_* */
    int32_t n_chains;    /* global number of chains, >= 4 (samplers.py:249), divisible by world_size */
    int32_t dim;         /* len(theta_0) or kwargs["dim"] (demc.py:20-23); 1 ... 131055 and n_chains * (dim + 2) < 2^31 (the reference has no limit;
                          * rows wider than 512 coordinates run on the looped kernel, bipymc_amd/csrc/kernels_wide.h) */
    int32_t target_id;   /* BPM_TARGET_* */
    int32_t n_target_params;
    const double* target_params; /* copied */
    uint64_t seed;               /* np.random.seed(...) analogue: key of every Philox stream */
    int32_t device;              /* HIP device ordinal */
    int32_t rank;                /* comm.rank; owns chains array_split(range(N), size)[rank] (demc.py:39) */
    int32_t world_size;          /* comm.size */
    const char* nccl_uid;        /* BPM_UID_BYTES from bpm_get_unique_id on rank 0; NULL if world_size==1 */
    /* DREAM kwargs (dream.py:20-27) */
    double gamma_scale; /* 1.0 */
    int32_t del_pairs;  /* 3; 1 ... 10 */
    int32_t burnin_gen; /* 300 */
    int32_t n_cr_gen;   /* 50 */
    int32_t n_cr;       /* 3, <= BPM_MAX_CR */
    /* build-defined extensions (absent from the reference) */
    double p_snooker;      /* DE-MC only: probability of a snooker update (0 = off) */
    int32_t outlier_every; /* DREAM burn-in: IQR outlier-chain reset every this many generations (0 = off) */
    int32_t keep_history;  /* 1: append every generation (chain.py:51-54); 0: keep current state only */
    int32_t running_moments; /* 1: keep, per generation, the population sums sum_i x_ij and sum_i x_ij^2 (2 dim doubles): bpm_reduce_moments
                              * then answers for any burn-in of WHOLE generations without a resident history -- param_est (demc.py:235-248)
                              * of 10^5 generations of config 2 would otherwise need 660 GB.  Two small dispatches per generation. */
    int32_t _pad2;
} bpm_config_t;

/* run_mcmc(**kwargs) (demc.py:73-75,161-162; dream.py:40-41). */
typedef struct bpm_run_opts {
    double flip;      /* 0.5; clipped to [0,1] */
    int32_t shuffle;  /* 1 */
    int32_t _pad;
    double epsilon;   /* < 0: default (DE-MC 1e-15, DREAM 1e-12); std of the normal jitter */
    double u_epsilon; /* < 0: default 1e-2 (DREAM uniform jitter half-width) */
    double gamma;     /* <= 0: default 2.38/sqrt(2 dim) (DE-MC only) */
} bpm_run_opts_t;

typedef struct bpm_stats {
    int64_t local_n_accepted; /* demc.py:67,190 (this run, this rank) */
    int64_t local_n_rejected; /* demc.py:68,193: starts at 1 */
    int64_t n_nan_alpha;      /* updates whose Metropolis ratio was NaN (reference raises ValueError) */
    int64_t k_gen;            /* generations done in the current run (demc.py:78,134) */
    int64_t t_abs;            /* generations done since creation */
    int64_t history_rows;     /* generations stored + 1 (row 0 = initial state) */
    int64_t n_outlier_resets;
    int32_t n_cr;
    int32_t _pad;
    double p_cr[BPM_MAX_CR];         /* dream.py:114 */
    double delta_m[BPM_MAX_CR];      /* dream.py:117 */
    double n_cr_updates[BPM_MAX_CR]; /* dream.py:115 */
} bpm_stats_t;

const char* bpm_last_error(void);
int bpm_abi_version(void);
/* Provenance: the first 16 hex digits of the SHA-256 over the sources this binary was built from (bipymc_amd/csrc/Makefile: ID_SRCS), baked in at
 * compile time.  The binding recomputes it from the tree and refuses a stale library; profiles/traffic_*.json carry the id the counters were taken
 * on.  (No counterpart in the reference: pure Python has no build.) */
const char* bpm_build_id(void);
/* GPUs visible to this process (hipGetDeviceCount): what one rank per GPU needs to pick its device where the reference
 * picks nothing (mpi4py ranks share the host's cores, demc.py:15). */
int bpm_device_count(int32_t* out);

/* rank 0 calls this and ships the bytes to the other ranks (mpi4py bcast / torch.distributed). */
int bpm_get_unique_id(char out[BPM_UID_BYTES]);

int bpm_create(const bpm_config_t* cfg, bpm_handle_t* out);
/* Non-zero when the library's own AQL queue had failed (a wait ran into its limit) and could not be quiesced: the handle is gone, but
 * its device buffers were deliberately leaked instead of freed under kernels that may still run (bpm_last_error says so). */
int bpm_destroy(bpm_handle_t h);

/* McmcChain.__init__ for every chain (chain.py:25-27): state0 = theta_0 + N(0, diag(varepsilon)),
 * varepsilon = VARIANCES, length dim (all > 0, else no jitter: util.py:12).  History := [state0]. */
int bpm_init_chains(bpm_handle_t h, const double* theta_0, const double* varepsilon);
/* Set the (N, dim) state matrix in global-id order (warm start, demc.py:46-51).  History := [X]. */
int bpm_set_state(bpm_handle_t h, const double* X);
int bpm_get_state(bpm_handle_t h, double* X);
/* Warm start with full histories (demc.py:46-51,217-233; chain.py:82-93): hist_local is
 * (rows, n_local, dim) for this rank's chains, X the (N, dim) current state of all chains. */
int bpm_set_history(bpm_handle_t h, int64_t rows, const double* hist_local, const double* X);
/* cached ln_like of the current state.  bpm_set_loglike: HOST-CALLBACK targets only (they must set the n_local local values after every
 * (re)initialisation; a sampler with a device target evaluates them itself and refuses the call). */
int bpm_set_loglike(bpm_handle_t h, const double* ll_local); /* n_local values */
int bpm_get_loglike(bpm_handle_t h, double* ll_local);

/* _mcmc_run prologue (demc.py:67-78): counters reset, k_gen = 0, kwargs latched. */
int bpm_begin_run(bpm_handle_t h, const bpm_run_opts_t* opts);
/* n_gens iterations of the while loop of demc.py:79-140, entirely on the device. Asynchronous. */
int bpm_step(bpm_handle_t h, int64_t n_gens);
/* same, synchronous, timed on the device with two HIP events BOUND TO UPDATE-KERNEL DISPATCHES of the sampler's stream (no
 * marker packets: an event-record pair alone costs ~17 us here, 7 % of a 20-generation window): elapsed_ms = end of update
 * launch number n/4 (the first one when n_gens < 8: binding an event costs the host 10-30 us, which a call starting from an
 * idle GPU can only absorb once the host runs ahead) -> end of the last one, n_launches = the launches that interval covers.
 * Average launch period = elapsed_ms / n_launches. */
int bpm_step_timed(bpm_handle_t h, int64_t n_gens, float* elapsed_ms, int64_t* n_launches);
/* the figures of the last bpm_step_timed call (which may be given NULL, NULL: reading the events costs tens of microseconds of
 * host time that a caller timing the call with its own clock does not want inside) */
int bpm_get_step_time(bpm_handle_t h, float* elapsed_ms, int64_t* n_launches);
int bpm_synchronize(bpm_handle_t h);
/* Exchange policy for world_size > 1: what the Allgather of demc.py:93-94,116-117 moves.  Outside DREAM's CR
 * adaptation (where delta / cr_idx of every chain travel with the dense block) a half generation only changes the rows
 * that were ACCEPTED, and everything a proposal is made of -- the replicated state matrix, counter-addressed draws,
 * update records -- is on every rank already.  mode 2 (default), "replay": owners all-gather ONE BYTE per local chain
 * (accepted or not) and every rank recomputes the other ranks' accepted proposals into its replica with the update
 * kernel's own proposal code (bit-identical by construction).  mode 1, "rows": ranks all-gather fixed-capacity packed
 * blocks (up to 4 sub-blocks [count | ids | rows] per rank) and scatter them; generations run in chunks of 64 under a
 * device-side checkpoint, a chunk in which a sub-block overflowed is rolled back and repeated with mode 0; cap > 0
 * sets the capacity (rows per sub-block per half generation) of the next chunk, afterwards it follows the largest
 * count seen.  mode 0: the dense all-gather of whole blocks.  The same values on every rank. */
int bpm_set_exchange(bpm_handle_t h, int32_t mode, int32_t cap);
/* out[8] = {mode, current capacity of mode 1, chunks run with mode 1, chunks of mode 1 repeated dense,
 *           generations exchanged by replay, generations exchanged by push, bits 0-1: 1 if the push exchange is connected (2: with the
 *           control block in fine-grained memory of its own) | bit 2 / bit 3: the arena probe of bpm_push_selftest passed with system- /
 *           agent-scope packet fences | bit 4: it ran on the library's own queue,
 *           barriers so far | bit 62 when the push exchange fences at agent scope} */
int bpm_get_exchange_stats(bpm_handle_t h, int64_t* out);
/* mode 3, "push" (the default once connected): where the reference's ranks meet in MPI_Allgather twice per generation
 * (demc.py:93-94,116-117), the OWNER of a chain stores an accepted row -- during CR adaptation also every update's (delta, cr)
 * statistic, dream.py:92 -- straight into every other rank's replica of the state matrix from inside the update kernel: the ranks'
 * exchange buffers are mapped into each other's address space (hipIpcOpenMemHandle; peer memory over xGMI on a multi-GPU node).  A
 * one-wavefront kernel per half generation announces "done" in every peer's control block and waits for the peers' announcements; no
 * collective, no recomputation, and the rank's kernels run on the library's own queue like a single-GPU sampler's.
 * cap (mode 3): 0 = the update kernels' packets acquire and release at SYSTEM scope, what the HSA memory model asks for between agents
 * (default); 1 = at AGENT scope, 6 us less per half generation on MI355X -- the pushed rows themselves are system-scope write-through
 * stores either way; whether the cheaper fences suffice between the GPUs of a node is a property of the platform that a caller
 * verifies by comparing the ranks' replicas (bench.py does, before it times anything).
 *   bpm_push_export   blob[BPM_PUSH_BLOB_BYTES]: what the other ranks need to map this rank's buffer
 *   bpm_push_connect  blobs = the exports of ALL ranks in rank order (world_size x BPM_PUSH_BLOB_BYTES), moved by the caller's own
 *                     communicator (the reference's counterpart is mpi_comm itself, demc.py:15)
 *   bpm_push_selftest collective; handles = this process's ranks (one, or the R ranks of a local test group).  (1) every rank writes a pattern
 *                     into every peer's control block, one barrier, every rank checks; (2) every rank stores probe rows into ITS block of
 *                     every peer's arena (first and last row, the block's last slots, both ends of its outlier block: where its update
 *                     kernels will push) from the queue and under the packet fences the update kernels use, a barrier, every rank verifies
 *                     what arrived and restores what was there -- with system-scope fences (*ok = (1) and this), then with the agent-scope
 *                     form (bpm_get_exchange_stats out[6] bit 3).  A mapping that points elsewhere, or stores that are not visible behind
 *                     the hand-over, end as *ok = 0 (-> an RCCL exchange), not as a fault or a divergence in the first generation.
 *   bpm_destroy       of a connected rank of a world of processes announces "closing" to its peers and waits, bounded
 *                     (BPM_PUSH_CLOSE_TIMEOUT_S, default 3 s), for theirs before it unmaps and frees: the library orders the teardown, the
 *                     caller needs no barrier of its own.  A rank that still exchanges with a closed rank gets an error at its next
 *                     bpm_synchronize ("closed its sampler") instead of waiting for its time limit.
 * A sampler created with a nccl_uid that starts with "BPMPUSH" has no RCCL communicator at all (ranks sharing one GPU, which RCCL
 * refuses; nodes without RCCL): the push exchange is then its only one.  Every rank must enter bpm_step within BPM_PUSH_TIMEOUT_S
 * (default 30) seconds of the others; a rank that waited longer reports it at the next bpm_synchronize. */
int bpm_push_export(bpm_handle_t h, void* blob);
int bpm_push_connect(bpm_handle_t h, const void* blobs);
int bpm_push_selftest(bpm_handle_t* handles, int32_t R, int32_t* ok);
/* How the generation loop's kernels reach the GPU (no counterpart in the reference: its loop is the Python interpreter,
 * demc.py:79-140).  A sampler with a device target dispatches its generation loop through the library's own user-mode AQL queue
 * (packets written by the library, bipymc_amd/csrc/aql_queue.h) -- update kernels, table builds, the CR reduction kernels of burn-in,
 * with the push exchange also the cross-rank barrier kernels of a rank of a world; RCCL exchanges and every other entry point use the
 * HIP stream.  out[6] = {1 if the sampler has such a queue, update-kernel dispatches of the calling thread through it, update-kernel
 * launches of the calling thread through the HIP stream, 1 if work may be in flight on the queue, 0 (was: experimental memory type),
 * packet fences of the update kernels on the queue: 3 acquire + release, 1 acquire only (the default: what later kernels read leaves
 * through write-through stores)}.
 * BPM_DIRECT_QUEUE=0 in the environment disables the queue. */
int bpm_get_launch_stats(bpm_handle_t h, int64_t* out);
/* direct != 0: use the library's own queue where the sampler has one (the default); 0: HIP stream launches only.
 * fence: packet fences of the update kernels on that queue: -1 keep, 3 agent-scope acquire + release on every packet with plain
 * stores (what a HIP stream does), 1 acquire only with write-through stores of what later kernels read (the default where the queue
 * exists); 0 is refused (fence-less packets belong to an experiment build only).  Results do not depend on the
 * launch path (tested). */
int bpm_set_launch_path(bpm_handle_t h, int32_t direct, int32_t fence);
/* Host-callback ln_like_fn (samplers.py:36-43): one half generation = propose + commit.
 * bpm_propose writes the proposals of this rank's chains of the current phase into out_prop
 * (n_out rows of dim) and their global ids into out_ids (capacity n_local each); the caller evaluates
 * ln_like_fn row by row and hands the values to bpm_commit, which does the Metropolis test
 * (samplers.py:328-336), the append (chain.py:51-54) and, after the second phase, k_gen += 1. */
int bpm_propose(bpm_handle_t h, double* out_prop, int32_t* out_ids, int32_t* n_out);
int bpm_commit(bpm_handle_t h, const double* ll_prop);
/* The same half generation with the read-back OVERLAPPED with the caller's evaluation (round 5).  bpm_propose_begin launches the proposal kernel and
 * queues the device-to-host copies of the proposals in n_chunks pieces (1 ... 256) into pinned staging OF THE LIBRARY, then returns at once;
 * bpm_propose_chunk(k) waits for piece k only and returns pointers INTO that staging -- *n_rows rows of *ld doubles (the first dim are the proposal;
 * ld = dim rounded up to even) in work-item order and their global chain ids (-1: an idle work item of a rank of a world; its value is ignored) --
 * valid until bpm_commit_end: while the caller evaluates ln_like_fn on piece k (samplers.py:36-43) the DMA of the following pieces proceeds.
 * bpm_commit_chunk(k, ll) hands in the *n_rows values of piece k (any order of k; the library copies), bpm_commit_end does what bpm_commit does.
 * bpm_propose_chunk -- and only it -- may be called from several host threads at once (a pool evaluating the pieces in parallel). */
int bpm_propose_begin(bpm_handle_t h, int32_t n_chunks);
int bpm_propose_chunk(bpm_handle_t h, int32_t k, const double** rows, const int32_t** ids, int32_t* n_rows, int32_t* ld);
int bpm_commit_chunk(bpm_handle_t h, int32_t k, const double* ll);
int bpm_commit_end(bpm_handle_t h);
/* ... and with the likelihood evaluated ON THE DEVICE by the caller's own framework (a vectorised torch / cupy likelihood): nothing crosses PCIe.
 * bpm_propose_device returns, complete, *rows_dev = DEVICE pointer to the half generation's *n_rows proposal rows of *ld doubles (work-item order)
 * and *ids_dev = device pointer to their global chain ids (-1: idle work item); bpm_commit_device takes a DEVICE pointer to *n_rows float64 values
 * in the same order (memory of this sampler's GPU; the library waits for the whole device first -- the values may come from any stream -- and
 * copies them).  bpm_state_device / bpm_set_loglike_device are the device-resident forms of bpm_get_state (local chains only: n_local rows where
 * they lie, read-only) + bpm_set_loglike, needed once per (re)initialisation. */
int bpm_propose_device(bpm_handle_t h, const double** rows_dev, const int32_t** ids_dev, int32_t* n_rows, int32_t* ld);
int bpm_commit_device(bpm_handle_t h, const double* ll_dev);
int bpm_state_device(bpm_handle_t h, const double** rows_dev, int32_t* n_rows, int32_t* ld);
int bpm_set_loglike_device(bpm_handle_t h, const double* ll_dev);
/* ... and with the likelihood given as HIP SOURCE (round 5; samplers.py:36-43 takes any Python callable -- this is the form of it that runs at device
 * speed with no framework in the process): `hip_source` defines
 *     __device__ double ln_like(const double* x, int d, const double* p)       (x: one parameter vector; p: the caller's parameter block = ln_kwargs)
 * (or, per coordinate: `#define BPM_LN_LIKE_TERMS K` + ln_like_terms(xj, j, d, p, acc) adding coordinate j's contribution to K <= 8 sums +
 * ln_like_finish(acc, d, p): inside the update kernel every lane of a chain then adds its own coordinates' terms; bipymc_amd/csrc/user_likelihood.h)
 * bpm_set_device_likelihood compiles it with hiprtc (loaded on demand) into one kernel -- a thread per proposal row -- that runs between the library's
 * proposal and commit kernels, copies the n_params doubles of `params` to the device and evaluates the current states (what bpm_set_loglike takes from
 * the host).  From then on bpm_step drives this host-callback sampler like one with a shipped target: no host code inside a generation.  f64 arithmetic
 * is compiled unfused (-ffp-contract=off) like the library's own.  A source that does not compile: error, the compiler's log in bpm_last_error.
 * bpm_refresh_device_loglike re-evaluates the current states (after bpm_set_state / bpm_init_chains / a warm start).
 * bpm_check_device_likelihood compiles only (no sampler, no GPU needed; arch NULL = "gfx950"): 0 or -1 with the reason in `log` (and bpm_last_error). */
int bpm_set_device_likelihood(bpm_handle_t h, const char* hip_source, const double* params, int32_t n_params);
int bpm_refresh_device_loglike(bpm_handle_t h);
int bpm_check_device_likelihood(const char* hip_source, const char* arch, char* log, int64_t log_cap);
/* bpm_set_device_likelihood also compiles the UPDATE KERNEL ITSELF around the caller's function (the library carries its kernel source; rows of up to
 * 512 coordinates): bpm_step then needs one launch per half generation instead of three (proposal / likelihood / commit).  Whatever fails on that way
 * leaves the three-kernel form in use: *fused says which runs, `why` (up to why_cap - 1 characters, may be NULL) why not the fused one.
 * The module's kernels are dispatched like the library's own (its AQL queue finds them by name).  Environment: BPM_USER_FUSED=0 does not attempt the
 * fused form, =2 launches it on the HIP stream (A/B). */
int bpm_get_device_likelihood_info(bpm_handle_t h, int32_t* fused, char* why, int64_t why_cap);

/* history of this rank's chains: out[(g - g_lo) * n_local * dim + i * dim + j], g in [g_lo, g_hi) */
int bpm_get_history(bpm_handle_t h, int64_t g_lo, int64_t g_hi, double* out);
int bpm_get_loglike_history(bpm_handle_t h, int64_t g_lo, int64_t g_hi, double* out);
int bpm_reserve_history(bpm_handle_t h, int64_t total_rows);
/* param_est (demc.py:235-248) without moving the history: for this rank's rows of the interleaved
 * super chain (row g*N + i = chain i at generation g) with row index >= n_burn, returns count,
 * sum_j (x - shift_j), sum_j (x - shift_j)^2 and shift_j (length dim each; shift is identical on every
 * rank).  mean_j = shift_j + S1/n, var_j = S2/n - (S1/n)^2 after summing S1, S2, n over ranks.  A sampler created with
 * running_moments and without a resident history answers from its per-generation sums: n_burn must then be a multiple of n_chains. */
int bpm_reduce_moments(bpm_handle_t h, int64_t n_burn, double* sum, double* sumsq, double* shift, int64_t* count);
int bpm_get_stats(bpm_handle_t h, bpm_stats_t* out);
/* checkpoint what the reference forgets: p_cr, delta_m, n_cr_updates, t_abs */
int bpm_set_adapt_state(bpm_handle_t h, const double* p_cr, const double* delta_m, const double* n_cr_updates,
                        int64_t t_abs);

/* ln_like of n points (row-major (n, dim)) with the sampler's device target: what `ln_like_fn(theta)` returns for the shipped analytic
 * targets (utils/d100_gauss.py:14-35, dblgauss_rv.py:11-32, banana_rv.py:11-40) */
int bpm_eval_loglike(bpm_handle_t h, const double* X, int32_t n, double* out);
/* (the test surface -- bpm_debug_*, bpm_selftest_philox, bpm_set_trace / bpm_get_trace, bpm_local_group_step, bpm_step_profiled, the
 * BPM_TEST_PATHS kernel-path switches -- is NOT part of this library: it is compiled only into build_variants/libbipymc_test.so and declared
 * in include/bipymc_hip_test.h; the product's kernel-argument block has no trace fields) */

#ifdef __cplusplus
}
#endif
#endif /* BIPYMC_HIP_H */
