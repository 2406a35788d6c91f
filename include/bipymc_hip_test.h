/*
 * bipymc_hip_test.h -- the TEST SURFACE of the MI355X-native DE-MC / DREAM sampler.
 *
 * Everything declared here is compiled ONLY with -DBPM_TEST_HOOKS into build_variants/libbipymc_test.so (bipymc_amd/csrc/Makefile); the
 * product library bipymc_amd/libbipymc_hip.so exports none of it and does not read BPM_TEST_PATHS.  The test variant is the same
 * source otherwise: it also exports every entry point of bipymc_hip.h.  None of these has a counterpart in the reference (its test
 * surface is its public classes, tests/test_*.py of /root/reference).
 *
 * BPM_TEST_PATHS (environment, read once per process by the test variant): comma-separated alternative kernel paths that the tests pin to
 * the default one bit for bit -- mode1, noplan, noperm, planall, nohot, groupqueues, wt8, ctrlarena, serial, hosttiming, histchain
 * (bipymc_amd/csrc/sampler.hip: test_path).
 */
#ifndef BIPYMC_HIP_TEST_H
#define BIPYMC_HIP_TEST_H

#include "bipymc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The world_size > 1 path on ONE GPU: create R handles with rank = 0..R-1, world_size = R and a nccl_uid that starts with "BPMLOCAL" (no RCCL
 * involved; the product library refuses such a uid); this call advances all of them n_gens generations in lock-step, doing the per-half-generation
 * all-gather (demc.py:93-94,116-117) with device copies -- or, once connected, through the push exchange. */
int bpm_local_group_step(bpm_handle_t* handles, int32_t R, int64_t n_gens);
/* bpm_step with a HIP event pair around every update-kernel launch: summed kernel time and launch count (bench.py's cross-check of the
 * per-launch duration its roofline is priced with, run outside the timed region).  n_gens <= 4096. */
int bpm_step_profiled(bpm_handle_t h, int64_t n_gens, double* kernel_ms_sum, int64_t* n_launches);
/* per-chain integer/float trace of the LAST generation (parity tests; the trace fields exist in the test variant's kernel-argument block only):
 * out_i32[n_local*32] = (cr_idx, d_prime, gamma_jump, accepted, snooker, partner ids[23], ...),
 * out_f64[n_local*4] = (alpha, ll_prop, delta, gamma), out_mask[n_local*dim] = CR mask.  A tracing sampler launches on the HIP stream. */
int bpm_set_trace(bpm_handle_t h, int32_t on);
int bpm_get_trace(bpm_handle_t h, int32_t* out_i32, double* out_f64, uint8_t* out_mask);
/* bpm_destroy's decision about the device buffers as a pure function (1 free, 0 leak), and the injection of a failed queue (the device's
 * queue is unusable for the rest of the process afterwards: child processes only) */
int bpm_debug_destroy_plan(int32_t queue_failed, int32_t quiesced);
int bpm_debug_fail_queue(bpm_handle_t h, int32_t refuse_quiesce);
/* no-op packets on the handle's own AQL queue until its next packet takes position `pos` (0 ... 254) of an epoch of 256 packets;
 * *widx = the queue's write index afterwards (a test then puts a drain's packets at a chosen place of the ring) */
int bpm_debug_queue_pad(bpm_handle_t h, int32_t pos, int64_t* widx);
/* why packets may drop the release fence only together with write-through stores -- 48 dependent dispatches with acquire-only packets and
 * PLAIN stores hand every block of a 2 MB buffer from workgroup to workgroup (XCD to XCD); *wrong = elements that missed an update (> 0 on
 * hipMalloc memory: the probe must FAIL there), -1 if there is no queue.  coherent_alloc != 0 (the memory type of round 2's experiment) is an
 * error outside the experiment build. */
int bpm_debug_coherence_probe(int32_t device, int32_t coherent_alloc, int64_t* wrong);
/* the order statistics, first argmax and cut (Q1 - 2 IQR) the outlier check would select from `omega` (n_chains values) */
int bpm_debug_outlier_select(bpm_handle_t h, const double* omega, double out[6]);
/* the update and the replay kernel of the handle's last half generation re-launched `reps` times and timed (destructive; tools/emulate_ranks.py) */
int bpm_debug_time_kernels(bpm_handle_t h, int32_t reps, float* update_us, float* replay_us);
/* the inline Philox4x32-10 against rocRAND's device engine: n blocks of (seed, subsequence i, block i) from both */
int bpm_selftest_philox(int32_t device, int32_t n, uint64_t seed, uint32_t* out_mine, uint32_t* out_rocrand);
/* per-generation host decisions (flip, shuffle order and its inverse) for generation t */
int bpm_debug_perm(bpm_handle_t h, int64_t t, int32_t shuffle, double flip_prob, int32_t* out_order, int32_t* out_inverse, int32_t* out_flip);

#ifdef __cplusplus
}
#endif
#endif /* BIPYMC_HIP_TEST_H */
