/* radixsort_hip_experiments.h — options of the EXPERIMENTS build of the engine only.
 *
 * The product library (radix-sort_amd/libradixsort_hip.so) does not know these values: rsx_set_option returns
 * RSX_INITIALIZATION_FAILED ("unknown option") for them.  They exist in the build made with
 *   tools/build_variant.sh experiments -DRSX_EXPERIMENTS   ->   tools/_variants/libradixsort_hip_experiments.so
 * which carries the alternatives that were measured and rejected (profiles/r03_tuning_log.md), so that A/B runs and their
 * parity tests (tests/test_gpu_experiments.py) stay possible.  Same results as the product in every mode.
 */
#ifndef RADIXSORT_HIP_EXPERIMENTS_H
#define RADIXSORT_HIP_EXPERIMENTS_H

typedef enum rsx_experimental_option {
    RSX_XOPT_DEBUG_RAISE_SCAN_TIMEOUT = 16, /* tests only: enqueue the store a timed-out fused scan makes (see rsx_check_status) */
    RSX_XOPT_INLINE_SCAN = 17,      /* 1: tables beyond the self-scan and of at most RSX_XOPT_INLINE_SCAN_MAX_GROUPS scan groups get no scan launch:
                                       the first workgroups of every reorder launch scan the pass's table (the fused scan's workgroup body, entries
                                       published write-through with a per-group ready word) before they turn to their tiles.  passes + 1 dependent
                                       launches instead of 2 passes + 1; measured 5-20 % SLOWER than the scan launch it removes.  The co-residency
                                       bound comes from the occupancy of the inline reorder kernels themselves (halved), queried when the option is set. */
    RSX_XOPT_INLINE_SCAN_MAX_GROUPS = 18, /* (default 64 = 2^26 keys) largest table, in scan groups, that takes the inline scan */
    RSX_XOPT_REORDER8_KERNEL = 19,  /* scatter kernel of the 8-bit passes: 1 = the product's (two ranking rounds); 2 = keys and payload make ONE trip
                                       through LDS (better on Range, worse elsewhere); 3 = ranks from one returning LDS atomic per key on per-wave
                                       counters (1.6x slower on random keys).  Kernel 3's STABILITY — the payload order of equal keys — rests on LDS
                                       atomics serving conflicting lanes in ascending lane order: measured behaviour of gfx950, NOT an architectural
                                       guarantee, probed once per engine (a failed probe falls back to kernel 1).  Experimental; keys always sort. */
    RSX_XOPT_REORDER8_STAY = 20     /* kernel 1 of the 8-bit passes as a grid that stays: N > 0 launches N workgroups per CU that walk the tiles of
                                       their XCD's range and prefetch the next tile while they rank the current one (1.5x slower); 0 / -1 = off. */
} rsx_experimental_option;

#endif /* RADIXSORT_HIP_EXPERIMENTS_H */
