/* test_hooks.h -- C entry points for the TESTS of this repository (tests/, tools/).  NOT part of the drop-in boundary: the
 * public headers (include/gpuntt_c.h, include/gpuntt/...) do not declare them and a GPU-NTT user has no use for them.
 *
 *   gpuntt_test_set_hook(name, value)   everything gpuntt_set_option takes, plus the hooks
 *       path = fast-strict       a call the fast kernels cannot take throws; no generic launch behind any call
 *       path = generic-capped    the generic kernels of an RNS call on the capped grid they use behind a go-flag
 *       no_scratch = 0 | 1       the drop-in calls behave as if their twiddle scratch could not be allocated
 *       rns_force_fallback = 0|1 the preparation kernel's own fall-back serves every drop-in RNS Merge call
 *       u32_e32 = mask           32-bit Merge rings 2^12 .. 2^15 on the 32-coefficients-per-lane kernels (bit n = ring 2^n;
 *                                bit 16: the full-tile contiguous pass of larger rings)
 *       two_sweep_big = 0 | 1    experiment: 64-bit rings 2^23 / 2^24 forward in two sweeps on 16384-coefficient tiles
 *       reset_predictions = 1    the family prediction of the RNS overloads forgets every stack it has seen
 *   gpuntt_test_launch_log_start()      start recording the kernel of every launch the library enqueues (all threads)
 *   gpuntt_test_launch_log_take(buf, n) stop; the kernels since start, space-separated ("prep_twiddles merge_pass_lazy:31 ..."),
 *                                       fast kernels with their lazy range behind a colon; returns the length needed
 *   gpuntt_test_scratch_stats(out[6])   the twiddle scratch of captured calls, which their graph owns (prep.hip): buffers handed to
 *                                       a graph, of those reported dead by their graph, buffers pooled now, buffers re-used from
 *                                       the pool, chains erased, chains alive
 * Options are snapshot once per API call (prep.hip), so a hook set while another thread's call is in flight does not change
 * that call. */
#ifndef GPUNTT_TEST_HOOKS_H
#define GPUNTT_TEST_HOOKS_H
#ifdef __cplusplus
extern "C"
{
#endif
    int gpuntt_test_set_hook(const char* name, const char* value);
    int gpuntt_test_launch_log_start(void);
    int gpuntt_test_launch_log_take(char* buf, int capacity);
    int gpuntt_test_scratch_stats(unsigned long long out[6]);
#ifdef __cplusplus
}
#endif
#endif
