// rsx_kernels.hpp — hand-written HIP kernels of the LSD radix sort for gfx950 (MI355X).
//
// Four device steps per 4-bit pass, the counterparts of the reference's OpenCL
// kernels (/root/reference/src/kernels/RadixSort.cl):
//
//   histogram_kernel      <- `histogram`        (RadixSort.cl:16-71)
//   scan_blocks_kernel    <- `scanhistograms` #1 (RadixSort.cl:125-181, RadixSortGPU.cpp:66-113)
//   scan_globsum_kernel   <- `scanhistograms` #2 (RadixSortGPU.cpp:115-152)
//   paste_kernel          <- `pastehistograms`   (RadixSort.cl:185-197)
//   reorder_kernel        <- `reorder`           (RadixSort.cl:74-119)
//
// What is kept from the reference is the algorithmic contract only: the digit of a
// key is `((key + OFFSET) >> (pass*4)) & 15`, the counter table is digit-major, its
// global exclusive prefix sum gives the first output slot of every (digit, producer)
// pair, and producers emit their keys in index order — hence a stable pass.
//
// What is different is everything that touches the machine.  The reference gives each
// of 1024 work-items a contiguous n/1024 sub-list (no access is coalesced, 16
// work-groups).  Here a *tile* of THREADS*KPT consecutive keys is one workgroup's
// unit of work, the table is [digit][tile], and inside a tile the reference's idea is
// replayed at LDS scale: THREADS "virtual processors" each own KPT consecutive keys and
// 16 private 16-bit counters in LDS, a raking DPP scan over the [digit][thread]
// counters yields every key's slot in the tile-local sorted order, keys are staged
// through LDS in that order, and the tile leaves as (up to) 16 runs of consecutive
// addresses.  HBM sees 16-byte loads of whole 64-byte per-lane rows and run-coalesced
// stores.  Inside rsx_sort the reorder of pass p also counts pass p+1's digits per
// output tile (look-ahead), so only the first pass reads the keys for a histogram.
//
// No MFMA: this is an integer permutation bounded by HBM bandwidth.
//
// The kernels live in one header per step; this file is the single include of rsx_capi.hip:
//   rsx_common.hpp     constants, key vectors, digit functions, XCD tile mapping, wave / block scans
//   rsx_histogram.hpp  histogram_kernel
//   rsx_scan.hpp       scan_blocks / scan_globsum / paste / paste_scan / scan_fused / scan_small
//   rsx_reorder.hpp    reorder_kernel (plain, look-ahead, self-scan, ranged)
//   rsx_tile_sort.hpp  tile_sort_kernel
//   rsx_radix8.hpp     histogram8 / scan8_* / reorder8
//   rsx_util.hpp       reference-geometry diagnostics and the small helpers of the multi-GPU partition
//   rsx_msd.hpp        the sharded sort's exchange step on the top B <= 8 bits: wave-major bucket starts, device-side plan, per-wave push
#pragma once

#include "rsx_common.hpp"
#include "rsx_histogram.hpp"
#include "rsx_scan.hpp"
#include "rsx_reorder.hpp"
#include "rsx_tile_sort.hpp"
#include "rsx_radix8.hpp"
#include "rsx_util.hpp"
#include "rsx_msd.hpp"
#ifdef RSX_EXPERIMENTS
#include "rsx_radix8_experiments.hpp"      // rejected alternatives, A/B builds only
#endif
