/* radixsort_hip.h — C ABI of the MI355X (gfx950) LSD radix-sort engine.
 *
 * This is the drop-in boundary for the reference's GPU hot path: the private
 * steps of RadixSortGPU<T> (/root/reference/src/RadixSortGPU.h:95-109) and the
 * device-buffer set of ComputeDeviceData<T> (src/ComputeDeviceData.cpp:42-77),
 * re-expressed as plain C so that C++ (radix-sort_amd/host), ctypes, cgo or JNI
 * hosts bind the same symbols.  No C++ types, no exceptions and no ownership
 * transfer cross this boundary.  Every function returns an rsx_status whose
 * values map 1:1 onto the reference's `enum class OperationStatus`
 * (src/OperationStatus.h:4-17).
 *
 * One engine = one device + one HIP stream + one set of device buffers
 * (inputKeys/outputKeys ping-pong, optional uint32 payload ping-pong, digit
 * table, block sums).  Engines share nothing; an engine is not thread-safe
 * (same as the reference, which swaps buffer names in place,
 * src/RadixSortGPU.cpp:263-266).
 *
 * There is no CPU fallback behind these symbols: without a HIP device every
 * entry point that needs one fails with RSX_INITIALIZATION_FAILED.
 */
#ifndef RADIXSORT_HIP_H
#define RADIXSORT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == OperationStatus (src/OperationStatus.h:4-17), same order, same values == */
typedef enum rsx_status {
    RSX_OK = 0,
    RSX_HOST_BUFFERS_FAILED = 1,
    RSX_INITIALIZATION_FAILED = 2,
    RSX_DATA_UPLOAD_FAILED = 3,
    RSX_CALCULATION_FAILED = 4,
    RSX_DATA_DOWNLOAD_FAILED = 5,
    RSX_CLEANUP_FAILED = 6,
    RSX_RESIZE_FAILED = 7,
    RSX_KERNEL_CREATION_FAILED = 8,   /* unreachable: kernels are AOT-compiled for gfx950 */
    RSX_PROGRAM_CREATION_FAILED = 9,  /* unreachable, kept for value parity */
    RSX_NO_SOURCE_FOUND = 10,         /* unreachable, kept for value parity */
    RSX_LOADING_SOURCE_FAILED = 11    /* unreachable, kept for value parity */
} rsx_status;

typedef struct rsx_engine rsx_engine;   /* opaque */

/* Digit geometry of the sort: 4-bit digits, 16 buckets, bits/4 passes
 * (src/Parameters.h:25,45,47). */
#define RSX_RADIX_BITS 4
#define RSX_RADIX 16

/* Options for rsx_set_option. */
typedef enum rsx_option {
    RSX_OPT_PROFILE = 0,      /* HIP events around launches -> RuntimesGPU: 0 off, 1 every launch, 2 reorder only */
    RSX_OPT_XCD_REMAP = 1,    /* 1 (default): consecutive tiles run on one XCD (L2 merges run seams) */
    RSX_OPT_FIRST_PASS = 2,   /* first pass of rsx_sort (default 0) */
    RSX_OPT_LAST_PASS = 3,    /* one past the last pass of rsx_sort (default bits/4) */
    RSX_OPT_REF_DIAGNOSTICS = 5, /* 1: rsx_download fills hist_out / globsum_out in the REFERENCE's geometry
                                 (16384-word [digit][group][item] table after paste, 512 scanned block sums,
                                 src/RadixSortGPU.cpp:412-428), recomputed from the last pass's input; needs a key
                                 count that is a multiple of 1024.  0 (default): the engine's own [digit][tile] table. */
    RSX_OPT_GRAPH = 6,        /* 1: rsx_sort / rsx_sort_from of at most 2^22 keys are captured once into a hipGraph and
                                 replayed; ignored while profiling and on the null stream.  Default 0: measured on
                                 MI355X the replay is no faster than eager launches (the ~3 us dependent-kernel
                                 boundary, not host launch cost, sets the 0.1 ms floor of a 33-kernel sort). */
    RSX_OPT_SMALL_SCAN = 7,   /* 1 (default): inside rsx_sort, tables of at most 1024 tiles (2^22 keys) are scanned and
                                 pasted by ONE workgroup in one launch instead of three; the step API is unaffected */
    RSX_OPT_TILE_SORT = 8,    /* 1 (default): inside rsx_sort, inputs of at most one tile (4096 keys) are sorted by ONE workgroup in
                                 ONE launch, every pass inside LDS; buffers, table and group sums end up as the pass chain
                                 leaves them.  Not taken while RSX_OPT_PROFILE is 1 (per-launch timings of the steps). */
    RSX_OPT_FUSED_SCAN = 9,   /* 1 (default): inside rsx_sort, tables of up to 512 scan groups (2^29 keys) are scanned and pasted in
                                 ONE launch whose workgroups hand their group sums to each other through tagged 8-byte
                                 granules; 0: scan #1, then scan #2 + paste (two launches).  Same table either way. */
    RSX_OPT_FUSED_SCAN_MAX_GROUPS = 15, /* largest table, in scan groups of 256 tiles (2^20 keys each), that takes the fused scan.  Its workgroups
                                 wait for each other inside the launch, so the whole grid must be resident at once: rsx_create asks
                                 hipOccupancyMaxActiveBlocksPerMultiprocessor x the CU count how many workgroups the device holds and takes
                                 HALF of it as the default (at most 512), which leaves room for a second engine scanning on another
                                 stream of the same device; engines that scan at the same time share that budget — a caller running k > 2
                                 of them concurrently on one device sets resident / k here.  Values are clamped to what is resident; -1 =
                                 default; 0 = never.  Larger tables take scan #1, then scan #2 + paste (two launches), same table. */
    RSX_OPT_RADIX_BITS = 10,  /* digit width of the rsx_sort chain: 4 (default, the reference's _NUM_BITS_PER_RADIX, src/Parameters.h:25) or 8.
                                 With 8 a pass sorts by a whole byte (two stable 4-bit rounds inside LDS, one scatter of up to
                                 256 runs per tile): half the passes over HBM.  Same result.  Pass ranges (RSX_OPT_FIRST_PASS /
                                 LAST_PASS, rsx_sort_from_to) stay in units of 4-bit passes: a range of whole bytes runs 8-bit passes, an odd
                                 range that starts on a byte boundary runs them for its whole bytes and one 4-bit pass for the last
                                 nibble, any other range runs the 4-bit chain; the step API and the diagnostic tables are those of
                                 4-bit passes. */
    RSX_OPT_SELF_SCAN = 11,   /* 1 (default): inside rsx_sort, tables of 2..1024 tiles (up to 2^22 keys) get no scan launch: every reorder
                                 workgroup derives the 16 first slots of its tile from the raw [tile][16] counts itself while
                                 its keys are on their way (passes + 1 dependent launches instead of 2 passes + 2) */
    RSX_OPT_SMALL_TILE_MAX_KEYS = 12, /* (default 2^19) self-scan sorts of at most this many keys (<= 2^20) run on tiles of 1024 keys (4 per thread) instead of
                                 4096: a shorter per-tile dependency chain for latency-bound small sorts.  The engine's own table
                                 read-back is not produced in that geometry (the reference-geometry diagnostics are). */
    RSX_OPT_SELF_SCAN_MAX_TILES = 14, /* (default 1024) largest table, in tiles (<= 1024), that takes the SELF_SCAN path */
    RSX_OPT_XCD_PHASE = 13,   /* (default -1) with XCD_REMAP: XCD x enters its tile range x * value tiles in and wraps round, so that the eight
                                 XCDs do not walk ranges that start n/8 apart in lockstep (same HBM channels); -1 = an eighth of a
                                 range, 0 = lockstep.  Placement only: results are identical. */
    RSX_OPT_LOOKAHEAD = 4     /* 1 (default): inside rsx_sort the reorder of pass p also counts pass p+1's digits per
                                 output tile, so only the first pass runs the histogram kernel; 0: every pass runs
                                 histogram -> scan -> paste -> reorder separately.  Results are identical. */
} rsx_option;

/* Per-phase launch timings in milliseconds, the RuntimesGPU fields
 * (src/RadixSortGPU.h:18-24) with Statistics semantics (src/Statistics.h). */
typedef struct rsx_phase_stat {
    double min_ms, max_ms, avg_ms, sum_ms;
    uint64_t n;
} rsx_phase_stat;

typedef struct rsx_runtimes {
    rsx_phase_stat histogram;  /* timeHisto   */
    rsx_phase_stat scan;       /* timeScan    (two launches per pass, as in the reference) */
    rsx_phase_stat paste;      /* timePaste   */
    rsx_phase_stat reorder;    /* timeReorder */
    rsx_phase_stat total;      /* whole rsx_sort calls (first launch -> last launch done) */
} rsx_runtimes;

typedef struct rsx_geometry {
    uint32_t tile_threads;     /* workgroup size of histogram/reorder (two scatter variants over 64-bit keys rank the same tile with 2x the threads and half the keys per thread) */
    uint32_t keys_per_thread;
    uint32_t tile_keys;        /* keys per tile = table column */
    uint32_t scan_block;       /* tiles per scan group: a group = this many consecutive tiles of ONE digit */
    uint64_t num_keys;         /* active length (rsx_resize / rsx_upload) */
    uint64_t capacity;
    uint64_t num_tiles;        /* ceil(num_keys / tile_keys) */
    uint64_t table_len;        /* RSX_RADIX * num_tiles, layout [digit][tile] */
    uint64_t num_scan_blocks;  /* 16 * ceil(num_tiles / scan_block) = live entries of globsum, layout [digit][group] */
    uint32_t num_passes;       /* key bits / 4 */
    uint32_t key_bytes;
    uint32_t fused_scan_resident;   /* workgroups of the fused table scan the device holds at once (occupancy query x CU count) */
    uint32_t fused_scan_max_groups; /* largest table, in scan groups, that takes the fused scan (RSX_OPT_FUSED_SCAN_MAX_GROUPS) */
} rsx_geometry;

/* ---- device --------------------------------------------------------------- */
/* Replaces ComputeState::init's device discovery (Common/ComputeState.cpp:14-104). */
int rsx_device_count(int* count);
int rsx_device_name(int device, char* buf, size_t buflen);
const char* rsx_last_error(void);   /* thread-local text of the last failure */
const char* rsx_version(void);

/* ---- lifetime ---------------------------------------------------------------
 * rsx_create replaces RadixSortGPU<T>::initialize (src/RadixSortGPU.cpp:452-543)
 * + ComputeDeviceData's constructor: allocates inputKeys/outputKeys
 * (capacity*key_bytes each), inputPermutations/outputPermutations
 * (capacity*4 each, only with has_payload), the digit table and block sums.
 * key_bytes is 4 or 8; is_signed selects the OFFSET treatment of signed keys
 * (src/RadixSortGPU.cpp:436-440, RadixSort.cl:51,114).  The engine creates its
 * own stream; rsx_set_stream substitutes a caller-owned hipStream_t.
 * rsx_destroy replaces release() (src/RadixSortGPU.cpp:445-449). */
int rsx_create(rsx_engine** out, int device, int key_bytes, int is_signed, int has_payload, uint64_t capacity);
int rsx_destroy(rsx_engine* e);
int rsx_set_stream(rsx_engine* e, void* hip_stream);
int rsx_get_stream(const rsx_engine* e, void** hip_stream);   /* the hipStream_t every call of this engine is enqueued on */
int rsx_set_option(rsx_engine* e, int option, int64_t value);
int rsx_get_geometry(const rsx_engine* e, rsx_geometry* out);

/* Sets the active number of keys (any value <= capacity; the reference's
 * 1024-rounding lives in the host class, RadixSortGPU::Resize,
 * src/RadixSortGPU.cpp:288-297). */
int rsx_resize(rsx_engine* e, uint64_t num_keys);

/* ---- transfers ------------------------------------------------------------
 * rsx_upload   = CopyDataToDevice + finish (src/RadixSortGPU.cpp:300-308,366-387):
 *                host keys -> inputKeys, host perm -> inputPermutations (payload
 *                engines only; perm may be NULL otherwise).  Sets num_keys = n.
 * rsx_fill_pad = padGPUData (:270-285): fills inputKeys from byte_offset to the
 *                end of the active length with numeric_limits<T>::max()-1.
 * rsx_download = CopyDataFromDevice + finish (:349-357,390-429): sorted keys,
 *                payload (NULL to skip), the first hist_cap entries of the digit
 *                table and the first globsum_cap block sums of the last pass. */
int rsx_upload(rsx_engine* e, const void* host_keys, const uint32_t* host_perm, uint64_t n);
int rsx_fill_pad(rsx_engine* e, uint64_t byte_offset);
int rsx_download(rsx_engine* e, void* host_keys_out, uint32_t* host_perm_out,
                 uint32_t* hist_out, uint64_t hist_cap, uint32_t* globsum_out, uint64_t globsum_cap);

/* Pinned host memory for the transfers (the reference leaves a "consider CL_MEM_USE_HOST_PTR" note at
 * src/ComputeDeviceData.cpp:26; pageable memcpy is what its avgTotalGPU column pays for).  rsx_pin_host
 * page-locks a caller-owned range so that rsx_upload / rsx_download DMA straight from / to it;
 * rsx_unpin_host undoes it.  Both are optional; unpinned buffers keep working. */
int rsx_pin_host(rsx_engine* e, void* host_ptr, uint64_t bytes);
int rsx_unpin_host(rsx_engine* e, void* host_ptr);

/* End-to-end beyond one upload -> sort -> download at a time (the reference's avgTotalGPU column pays for all
 * three in sequence, src/CRadixSortTask.cpp:357-378; its visualizer sorts straight out of mapped host memory,
 * examples/visualize/visualize.cpp:801-854).
 * rsx_pipeline_submit: asynchronous.  Copies n keys (and permutation) from host memory into a device inbox on an
 *   upload stream, sorts them behind that copy on the engine's stream, and copies the result to host_keys_out on a
 *   download stream.  Two jobs are in flight at once (two inboxes, two outboxes, allocated on first use): the
 *   upload of job i+1 and the download of job i-1 run beside the sort of job i.  Host buffers must stay untouched
 *   until rsx_pipeline_wait and should be pinned (rsx_pin_host), otherwise the copies serialise on the host.
 * rsx_pipeline_wait: waits for every submitted job.
 * rsx_host_device_pointer: the device-side address of pinned (rsx_pin_host) host memory, for the zero-copy form:
 *   rsx_sort_from_to(e, <that address>, ...) reads the keys over PCIe in its first pass (and histogram) and
 *   writes its last pass straight into mapped host memory. */
int rsx_pipeline_submit(rsx_engine* e, const void* host_keys, const uint32_t* host_perm, uint64_t n, void* host_keys_out, uint32_t* host_perm_out);
int rsx_pipeline_wait(rsx_engine* e);
int rsx_host_device_pointer(rsx_engine* e, void* host_ptr, void** device_ptr);

/* ---- the hot path, step by step ---------------------------------------------
 * Asynchronous on the engine's stream; no host synchronisation inside.
 * rsx_histogram = RadixSortGPU::Histogram        (src/RadixSortGPU.cpp:16-61)
 * rsx_scan      = ScanHistogram, scans #1 and #2 (:64-152)
 * rsx_paste     = ScanHistogram, paste part      (:155-195)
 * rsx_reorder   = Reorder incl. the buffer swap  (:199-267)
 * rsx_sort      = calculate's pass loop          (:311-346) */
int rsx_histogram(rsx_engine* e, int pass);
int rsx_scan(rsx_engine* e);
int rsx_paste(rsx_engine* e);
int rsx_reorder(rsx_engine* e, int pass);
int rsx_sort(rsx_engine* e);
int rsx_sync(rsx_engine* e);   /* CommandQueue.finish() */
/* The fused table scan bounds its polls; a workgroup whose poll ran out (its grid was not resident at once — a device
 * shared with long-running kernels of other processes, see RSX_OPT_FUSED_SCAN_MAX_GROUPS) stores to a word of mapped host
 * memory and finishes, leaving that sort's result undefined.  The word is reported ONCE — by rsx_sync, rsx_download,
 * rsx_pipeline_wait (after their synchronisation), and by rsx_check_status / rsx_copy_result without synchronising, i.e. for
 * sorts that have already finished — with RSX_CALCULATION_FAILED / RSX_DATA_DOWNLOAD_FAILED, and is then cleared: the
 * engine stays usable.  Asynchronous callers (rsx_sort_from_to into their own buffers) end a batch with rsx_sync, or
 * call rsx_check_status after synchronising the stream themselves. */
int rsx_check_status(rsx_engine* e);

/* ---- device-resident callers (PyTorch / RCCL plumbing) ----------------------
 * rsx_sort_from: sorts n keys that already live in HBM at d_keys (16-byte
 *   aligned, not modified) with optional payload d_payload; the result stays in
 *   the engine (rsx_result_device / rsx_copy_result / rsx_download).
 * rsx_partition: ONE stable radix pass on the bit field [shift, shift+bits) of
 *   external keys into caller-provided output buffers; bucket_offsets receives
 *   (1<<bits)+1 exclusive offsets.  This is the bucket-grouping step of the
 *   multi-GPU exchange.  Synchronises the stream before returning.
 *   All partition entry points read keys and payload 16 bytes per lane: d_keys and d_payload must be 16-byte
 *   aligned (RSX_HOST_BUFFERS_FAILED otherwise); outputs need the alignment of their element type only,
 *   except where stated. */
/* Aliasing: d_keys may be the start of one of the engine's own two key buffers (the pointer
 * rsx_result_device returns; with a payload engine d_payload must then be the matching payload
 * buffer) — the sort then runs through the internal ping-pong, as rsx_sort does.  Any other overlap of
 * the input with the engine's buffers, of rsx_sort_from_to's output with them, or of input and output
 * with each other is refused with RSX_HOST_BUFFERS_FAILED.  After rsx_sort_from_to the result lives in
 * the caller's buffer only: rsx_download / rsx_copy_result of keys fail and rsx_result_device yields NULL
 * until the next sort or upload. */
int rsx_sort_from(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n);
int rsx_partition(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n,
                  int shift, int bits, void* d_keys_out, uint32_t* d_payload_out, uint64_t* bucket_offsets);
/* Range-adaptive partition for the multi-GPU exchange (keys whose top bits are all equal — small
 * ranges, sorted inputs — would otherwise land on one rank):
 * rsx_key_range: min and max of n device-resident keys in unsigned sort order (key ^ sign bit),
 *   returned as uint64; n == 0 gives lo = UINT64_MAX, hi = 0.  Synchronises.
 * rsx_partition_range: like rsx_partition with x = (key ^ sign) - lo and
 *   bucket = min(mul ? mulhi(x, mul) : x >> shift, 15), mul = floor(16 * 2^keybits / (hi - lo + 1)):
 *   16 equal-width buckets over [lo, hi] (mul == 0 for ranges of at most 16 values);
 *   bucket_offsets receives 17 entries. */
/* The same bit-field partition in two halves, so that the host can look at the bucket sizes (and
 * exchange them between ranks) BEFORE anything is moved: rsx_partition_count runs the histogram
 * and returns the (1<<bits) bucket sizes (synchronises); rsx_partition_scatter must follow on the
 * same keys / n / bit field and does scan + paste + reorder into the caller's buffers (asynchronous). */
int rsx_partition_count(rsx_engine* e, const void* d_keys, uint64_t n, int shift, int bits, uint64_t* bucket_counts);
int rsx_partition_scatter(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, int shift, int bits,
                          void* d_keys_out, uint32_t* d_payload_out);
/* Splitter partition (multi-GPU exchange on arbitrary distributions, heavy ties included):
 * rsx_sample_keys: `count` (<= 4096) keys, one per stratum of n/count consecutive keys, in unsigned
 *   sort order (key ^ sign bit) as uint64.  Synchronises.
 * rsx_partition_count_split: nsplit (1..7) strictly increasing splitters in that same order;
 *   bucket(x) = 2 * #{splitters < x} + [x equals a splitter]: even buckets are the open intervals, odd
 *   buckets hold exactly the keys equal to a splitter (which the caller may cut anywhere, in (rank,
 *   index) order).  Returns the 2*nsplit+1 bucket sizes.  Synchronises.
 * rsx_partition_scatter_split: must follow on the same keys; scan + paste + reorder by those buckets. */
int rsx_sample_keys(rsx_engine* e, const void* d_keys, uint64_t n, uint32_t count, uint64_t* samples);
int rsx_partition_count_split(rsx_engine* e, const void* d_keys, uint64_t n, const uint64_t* splitters, int nsplit, uint64_t* bucket_counts);
int rsx_partition_scatter_split(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, void* d_keys_out, uint32_t* d_payload_out);
/* rsx_sort_from_to: rsx_sort_from over passes [first_pass, last_pass) (4-bit pass units, whatever the digit width of the chain) whose last pass
 * writes to the caller's d_keys_out / d_payload_out (any alignment), e.g. at an offset inside the final array — the local sort of one
 * wave of the sharded sort, whose keys share their top bits. */
int rsx_sort_from_to(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, int first_pass, int last_pass, void* d_keys_out,
                     uint32_t* d_payload_out);
/* Receive buffers other ranks can write to (the peer-store exchange below): rsx_peer_alloc (hipMalloc + an IPC handle to hand to the other
 * PROCESSES, which map it with rsx_peer_open / rsx_peer_close — peer access over xGMI); ranks that are threads of one process use the
 * pointer itself (after rsx_peer_enable, once per other device). */
#define RSX_IPC_HANDLE_BYTES 64
int rsx_peer_alloc(rsx_engine* e, uint64_t bytes, void** d_ptr, void* ipc_handle /* RSX_IPC_HANDLE_BYTES bytes, or NULL */);
int rsx_peer_free(rsx_engine* e, void* d_ptr);
int rsx_peer_open(rsx_engine* e, const void* ipc_handle, void** d_ptr);
int rsx_peer_close(rsx_engine* e, void* d_ptr);
int rsx_peer_enable(rsx_engine* e, int peer_device);
/* ---- exchange step of the sharded sort on the top B <= 8 key bits (nothing in the reference: one device, one in-order
 * queue, Common/ComputeState.cpp:88-101).  world = 1, 2, 4, 8 or 16 ranks own k = 2^bits / world consecutive buckets of the top `bits`
 * bits each; bucket rank * k + w belongs to WAVE w, and every rank receives its k waves one after the other, so that wave w can be sorted —
 * all its keys at a rank share the top `bits` bits: passes [0, ceil((keybits - bits) / 4)) suffice — while wave w + 1 is still on the links.
 * More bits = more, smaller waves = a smaller exposed first wave.  All calls are asynchronous on the engine's stream unless stated.
 *   rsx_msd_count    one read of the shard: d_counts (DEVICE memory, 256 x uint64) receives the keys per bucket in natural bucket order
 *                    (entries past 2^bits are 0) — the row the caller all_gathers; no host synchronisation.
 *   rsx_msd_scatter  must follow on the same keys: groups the shard (stable) into d_staging in wave-major order [wave][destination rank]
 *                    (inside a segment: by the remaining bits of the key's top byte).  Needs only THIS rank's counts, i.e. it may run while
 *                    the all_gather is still in flight.
 *   rsx_msd_plan     from the gathered table d_table[source rank * stride + bucket] (+ every rank's receive and output capacity in keys at
 *                    [.. + cap_at] and [.. + cap_at + 1]) computes ON THE DEVICE, on hip_stream (NULL = the engine's): where each of this
 *                    rank's (wave, destination) segments lands in the destination's receive buffer (sources in rank order; grouping 0: every
 *                    wave starts 16-byte aligned and is sorted by itself; grouping 1, "doubling groups": only waves 0, 1, 2, 4, 8, ... do, the
 *                    waves of a group {0} {1} {2,3} {4..7} ... lie gap-free and are sorted together — radix-sort_amd/host/ShardPlanner.h), what this rank receives per wave, every rank's load, and the capacity verdict — then copies the host's part to
 *                    pinned memory.  rsx_msd_plan_wait blocks the HOST until that copy has landed (the device never waits for the host) and
 *                    returns it: wave_start / wave_count (2^bits / world entries, in keys, THIS rank's receive buffer), loads (world entries),
 *                    verdict (0 = go; bit r = rank r's buffers are too small; bit 32 + r = rank r's status word, [.. + cap_at + 2] of its row, was
 *                    non-zero: its engine reported an error of an earlier step — non-zero on one rank is non-zero on all, and no push writes anything).
 *   rsx_msd_push     wave `wave`: copies this rank's segments of that wave from staging straight into the destinations' receive buffers:
 *                    d_peer_keys / d_peer_payload = DEVICE arrays of `world` base addresses as THIS rank addresses them (rsx_peer_alloc /
 *                    rsx_peer_open / rsx_peer_enable).  `parts` workgroups per destination (<= 0: max(16, 128 / world)): a link-bound copy that leaves the CUs to
 *                    the local sorts.  hip_stream (NULL = the engine's): the stream the copy is enqueued on — a second stream lets wave w + 1
 *                    travel while wave w is sorted on the engine's; the call makes it wait for the plan and for rsx_msd_scatter itself.
 *                    The caller fences the wave across ranks (one tiny all_reduce, or its own flags) before sorting it.
 * Plumbing for hosts that do not link HIP themselves: rsx_copy_to_device / _from_device / _on_device (asynchronous on the engine's
 * stream; pageable host memory serialises, pin it with rsx_pin_host), rsx_wait_for (e's stream waits for everything enqueued on
 * other's stream so far — engines of one process, any devices) and rsx_record_mark / rsx_wait_mark (below). */
int rsx_msd_count(rsx_engine* e, const void* d_keys, uint64_t n, int bits, int world, uint64_t* d_counts);
int rsx_msd_scatter(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, void* d_staging, uint32_t* d_staging_payload);
int rsx_msd_plan(rsx_engine* e, const uint64_t* d_table, uint32_t stride, uint32_t cap_at, int rank, int grouping, void* hip_stream);
int rsx_msd_plan_wait(rsx_engine* e, uint64_t* wave_start, uint64_t* wave_count, uint64_t* loads, uint64_t* verdict);
int rsx_msd_push(rsx_engine* e, int wave, const void* d_staging, const uint32_t* d_staging_payload, const uint64_t* d_peer_keys, const uint64_t* d_peer_payload, int parts,
                 void* hip_stream);
int rsx_copy_to_device(rsx_engine* e, void* d_dst, const void* host_src, uint64_t bytes);
int rsx_copy_from_device(rsx_engine* e, void* host_dst, const void* d_src, uint64_t bytes);
int rsx_copy_on_device(rsx_engine* e, void* d_dst, const void* d_src, uint64_t bytes);
int rsx_wait_for(rsx_engine* e, rsx_engine* other);
/* Named points of an engine's stream: rsx_record_mark(e, slot) marks "everything enqueued on e's stream so far" (slot 0..RSX_MAX_MARKS-1, re-recordable);
 * rsx_wait_mark(e, other, slot) makes e's stream wait for other's mark — at any later time, whatever has been enqueued on other's stream since
 * (rsx_wait_for = record + wait in one call).  Used by the C++ sharded driver: one mark per wave on the communication stream. */
#define RSX_MAX_MARKS 256
int rsx_record_mark(rsx_engine* e, int slot);
int rsx_wait_mark(rsx_engine* e, rsx_engine* other, int slot);
int rsx_key_range(rsx_engine* e, const void* d_keys, uint64_t n, uint64_t* lo, uint64_t* hi);
int rsx_partition_range(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, uint64_t lo, int shift, uint64_t mul,
                        void* d_keys_out, uint32_t* d_payload_out, uint64_t* bucket_offsets);
int rsx_result_device(rsx_engine* e, void** d_keys, uint32_t** d_payload);
int rsx_copy_result(rsx_engine* e, void* d_keys_out, uint32_t* d_payload_out);

/* Launch geometry of the tile kernels, as host arithmetic (no device is touched): workgroup b of a launch over
 * num_keys keys in tiles of tile_keys keys works on tile tiles_out[b] (values >= *ntiles mark workgroups that
 * exit at once); *blocks = workgroups launched.  xcd_remap / xcd_phase as RSX_OPT_XCD_REMAP / RSX_OPT_XCD_PHASE.
 * At most cap entries are written.  For tests of the mapping: every tile must appear exactly once. */
int rsx_tile_map(uint64_t num_keys, uint32_t tile_keys, int xcd_remap, int64_t xcd_phase, uint32_t* tiles_out, uint64_t cap,
                 uint32_t* blocks, uint32_t* ntiles);

/* ---- measurements ------------------------------------------------------------
 * rsx_timings synchronises, folds pending HIP-event pairs into the statistics and
 * copies them out (getRuntimes, src/RadixSortGPU.cpp:591-595); reset != 0 clears
 * them afterwards. */
int rsx_timings(rsx_engine* e, rsx_runtimes* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* RADIXSORT_HIP_H */
