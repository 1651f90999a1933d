// rsx_capi.hip — implementation of include/radixsort_hip.h for gfx950.
//
// The engine owns what ComputeDeviceData<T> owns in the reference
// (/root/reference/src/ComputeDeviceData.cpp:42-77): inputKeys/outputKeys,
// inputPermutations/outputPermutations, histograms, globsum, temp — and drives the
// kernels of rsx_kernels.hpp the way RadixSortGPU<T>::Histogram / ScanHistogram /
// Reorder / calculate do (src/RadixSortGPU.cpp:16-346), minus the per-launch
// `finish()`: everything is enqueued on one HIP stream and the host synchronises
// only in upload/download/timings, as the C ABI header states.
#include "radixsort_hip.h"
#ifdef RSX_EXPERIMENTS
#include "radixsort_hip_experiments.h"
#endif
#include "rsx_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int status, const char* what, hipError_t err = hipSuccess)
{
    g_last_error = what;
    if (err != hipSuccess) {
        g_last_error += ": ";
        g_last_error += hipGetErrorString(err);
    }
    std::fprintf(stderr, "[radixsort_hip] %s\n", g_last_error.c_str());
    return status;
}

#define RSX_TRY(expr, status)                                     \
    do {                                                          \
        const hipError_t rsx_err_ = (expr);                       \
        if (rsx_err_ != hipSuccess) {                             \
            return fail((status), #expr, rsx_err_);               \
        }                                                         \
    } while (0)

// Tile geometry.  One compiled shape per key width for now; the table layout
// depends on it, so it is fixed per engine.
#ifndef RSX_TILE_THREADS
#define RSX_TILE_THREADS 256
#endif
#ifndef RSX_KPT
#define RSX_KPT 16
#endif
constexpr int kTileThreads = RSX_TILE_THREADS;
constexpr int kKeysPerThread = RSX_KPT;
constexpr int kTileKeys = kTileThreads * kKeysPerThread;
constexpr int kSmallKeysPerThread = 4;      // small self-scan sorts: tiles of 1024 keys
// (probe builds with another tile shape, e.g. -DRSX_TILE_THREADS=512 -DRSX_KPT=8, leave out the one-launch tile sort, the inline scan and the 8-bit chain, which are written for 256 x 16)
#define RSX_PRODUCT_SHAPE (RSX_TILE_THREADS == 256 && RSX_KPT == 16)

enum Phase : int { PH_HISTO = 0, PH_SCAN = 1, PH_PASTE = 2, PH_REORDER = 3, PH_TOTAL = 4, PH_COUNT = 5 };

struct EventPair {
    int phase;
    hipEvent_t start, stop;
};

void stat_reset(rsx_phase_stat& s)
{
    s.min_ms = std::numeric_limits<double>::infinity();
    s.max_ms = -std::numeric_limits<double>::infinity();
    s.avg_ms = 0.0;
    s.sum_ms = 0.0;
    s.n = 0;
}

// Statistics::update (src/Statistics.h:21-31) with the first-sample-min slip fixed:
// the reference's `else if` never lets the first sample become the minimum.
void stat_update(rsx_phase_stat& s, double ms)
{
    s.n += 1;
    s.sum_ms += ms;
    s.avg_ms = s.sum_ms / static_cast<double>(s.n);
    s.max_ms = std::max(s.max_ms, ms);
    s.min_ms = std::min(s.min_ms, ms);
}

}  // namespace

struct GraphEntry {      // one captured rsx_sort chain
    const void* in;
    const uint32_t* pin;
    uint64_t n;
    int cur, first, last, flags;
    uint64_t options_epoch;      // rsx_set_option calls seen when the chain was captured: an option may change the launches
    hipStream_t stream;
    hipGraphExec_t exec;
    int end_cur;
    const void* end_last_in;
    int end_last_shift;
};

struct rsx_engine {
    int device = 0;
    int key_bytes = 4;
    bool is_signed = false;
    bool has_payload = false;
    uint64_t capacity = 0;
    uint64_t n = 0;

    void* keys[2] = {nullptr, nullptr};         // [cur] = "inputKeys", [cur^1] = "outputKeys"
    uint32_t* perm[2] = {nullptr, nullptr};     // inputPermutations / outputPermutations
    int cur = 0;
    // where the sorted data of the last rsx_sort / rsx_sort_from lives
    void* result_keys = nullptr;
    uint32_t* result_perm = nullptr;

    uint32_t* table = nullptr;                  // "histograms": [digit][tile]
    uint32_t* globsum = nullptr;                // block sums of the table scan
    uint32_t* globsum2 = nullptr;               // their scanned copy when scan #2 ran inside the paste
    uint32_t* globsum_live = nullptr;           // which of the two a download should read
    int paste_scan = 1;                         // rsx_sort: scan #2 + paste in one launch (env RSX_PASTE_SCAN)
    int fused_scan = 1;                         // rsx_sort: scan #1, scan #2 and paste in ONE launch, arrival counter inside (env RSX_FUSED_SCAN)
    unsigned long long* gsums = nullptr;        // fused scan: {epoch, raw group sum} granules [group][16]
    uint32_t* scan_timeout = nullptr;           // fused scan: set by a workgroup whose poll ran out — the DEVICE address of ...
    uint32_t* scan_timeout_host = nullptr;      // ... this word of mapped pinned host memory: the host reads it without a copy (check_scan_timeout)
    uint32_t fused_scan_resident = 0;           // workgroups of scan_fused_kernel the device holds at once (occupancy query x CU count, rsx_create)
    uint32_t fused_scan_limit = 0;              // largest table, in scan groups, that takes the fused scan (RSX_OPT_FUSED_SCAN_MAX_GROUPS; default: half of the above)
    bool table_valid = false;                   // e->table holds the last pass's table in the engine's own [digit][4096-key tile] geometry ...
    bool globsum_valid = false;                 // ... and globsum_live its group sums (rsx_download refuses to hand out anything else)
    uint32_t scan_epoch = 0;                    // launch count of the fused scan (tags the granules; never 0)
    uint32_t* temp = nullptr;                   // grand total of scan #2
    uint32_t* counts_next = nullptr;            // look-ahead histogram of the next pass, [tile][digit]
#ifdef RSX_EXPERIMENTS
    uint32_t* counts_next2 = nullptr;           // inline-scan chain: the second of two alternating count buffers (the scan of a launch reads one, its look-ahead adds into the other)
    uint32_t* scan_ready = nullptr;             // inline-scan chain: [group] = epoch of the launch whose table entries of that group are published
    int inline_scan = 0;                        // rsx_sort: mid-size sorts run the table scan inside the reorder launch (RSX_XOPT_INLINE_SCAN); measured SLOWER than the scan launch (profiles/r03_tuning_log.md §4)
    uint32_t inline_scan_max_groups = 64;       // ... for tables of at most this many scan groups (2^26 keys), never beyond the inline kernels' own co-residency limit
    uint32_t inline_scan_limit = 0;             // workgroups of the INLINE reorder kernels resident at once / 2 (occupancy query, rsx_create)
#endif
#ifdef RSX_STAMPS
    unsigned long long* stamps = nullptr;       // diagnostic build: 16 phase stamps per tile of ONE chosen launch
    int stamp_pass = -1;                        // env RSX_STAMP_PASS: the pass whose reorder launch writes them
#endif
    uint32_t* ref_table = nullptr;              // diagnostics in the reference's [digit][group][item] geometry
    uint32_t* ref_globsum = nullptr;
    const void* counted_keys = nullptr;         // rsx_partition_count left a raw table for exactly this input
    uint64_t counted_n = 0;
    int counted_shift = 0, counted_bits = 0;
    const void* last_in = nullptr;              // input buffer and shift of the most recent reorder
    int last_shift = 0;
    int ref_diag = 0;                           // RSX_OPT_REF_DIAGNOSTICS
    uint64_t splitters[rsx::kMaxSplitters] = {};   // splitters of the current split partition, unsigned sort order
    uint32_t nsplit = 0;
    bool result_external = false;               // the last sort wrote into the caller's buffer (rsx_sort_from_to): nothing to download
    void* final_keys_out = nullptr;             // rsx_sort_from_to: where the last pass writes
    uint32_t* final_perm_out = nullptr;
    unsigned long long* range_dev = nullptr;    // per-workgroup {min, max} of rsx_key_range
    unsigned long long* range_host = nullptr;   // pinned mirror
    uint32_t* starts_dev = nullptr;             // 16 bucket starts (rsx_partition)
    uint32_t* starts_host = nullptr;            // pinned mirror
    uint64_t table_cap = 0;
#ifdef RSX_EXPERIMENTS
    // Large buffers may be backed by separately created physical chunks mapped into one virtual range in a shuffled order
    // (env RSX_ALLOC_MODE, big_alloc): which physical pages a buffer gets decides how the scatter's 16..512 write
    // fronts — a power of two apart for 2^k uniform keys — fall onto L2 sets / HBM banks (profiles/r03_tuning_log.md §5).
    struct BigBuf {
        void* base = nullptr;
        size_t size = 0;
        std::vector<hipMemGenericAllocationHandle_t> chunks;     // empty: plain hipMalloc
    };
    std::vector<BigBuf> big;
    int alloc_mode = 0;         // 0 hipMalloc; 1 chunks mapped in creation order; 2 chunks mapped in a shuffled order
    size_t alloc_chunk = 0;     // chunk bytes (env RSX_ALLOC_CHUNK_MB; 0 = 32 MiB)
#endif
    // exchange step of the sharded sort on the top B <= 8 bits (capi_msd.inc; all allocated on first use)
    rsx::MsdPlan* msd_plan = nullptr;           // device: segments of this rank per (wave, destination), what it receives per wave, loads, verdict
    rsx::MsdPlan* msd_plan_host = nullptr;      // pinned mirror of the part the host needs
    uint32_t* msd_starts = nullptr;             // device: first staging slot of each of the 256 fine buckets, wave-major order
    hipEvent_t msd_event = nullptr;             // the plan (and its copy to the host) is complete
    hipEvent_t msd_scatter_event = nullptr;     // rsx_msd_scatter has filled the staging buffer (pushes on another stream wait for it)
    bool msd_scattered = false;
    hipEvent_t marks[RSX_MAX_MARKS] = {};        // rsx_record_mark / rsx_wait_mark
    std::map<int, hipEvent_t> order_events;     // rsx_wait_for: this engine's events, one per device of the engines it has waited for
    const void* msd_keys = nullptr;             // rsx_msd_count left table8 / cbase8 for exactly this input ...
    uint64_t msd_n = 0;
    int msd_bits = 0, msd_world = 0;            // ... partitioned on this many top bits for this many ranks
    bool msd_planned = false;

    hipStream_t stream = nullptr;
    bool own_stream = false;

    // end-to-end pipeline (rsx_pipeline_submit): two jobs in flight, each with its own device inbox and outbox,
    // uploads and downloads on streams of their own beside the sort stream
    struct Pipeline {
        bool ready = false;
        void* in[2] = {nullptr, nullptr};
        void* out[2] = {nullptr, nullptr};
        uint32_t* pin[2] = {nullptr, nullptr};
        uint32_t* pout[2] = {nullptr, nullptr};
        hipStream_t s_in = nullptr, s_out = nullptr;
        hipEvent_t in_ready[2] = {nullptr, nullptr}, sorted[2] = {nullptr, nullptr}, out_done[2] = {nullptr, nullptr};
        uint64_t submitted = 0;
    } pipe;

    int profile = 0;            // 0 off, 1 every launch, 2 reorder launches (+ whole sort) only
    int xcd_remap = 1;
    uint64_t options_epoch = 0;
    int64_t xcd_phase = -1;     // RSX_OPT_XCD_PHASE / env RSX_XCD_PHASE, in tiles: -1 = a range's eighth (the XCDs spread evenly over the walk), 0 = lockstep
    int lookahead = 1;          // rsx_sort builds pass p+1's histogram inside pass p's reorder
    int small_scan = 1;         // rsx_sort: one-workgroup scan+paste for tables of <= 1024 tiles (env RSX_SMALL_SCAN)
    uint64_t radix8_min_keys = 1u << 19;        // 8-bit passes only above this many keys (env RSX_RADIX8_MIN_KEYS; at least one tile)
    int radix_bits = 4;         // RSX_OPT_RADIX_BITS: 4 (the reference's configuration) or 8 (half the passes; rsx_sort chain only)
#ifdef RSX_EXPERIMENTS
    int reorder8_version = 1;   // RSX_XOPT_REORDER8_KERNEL: 3 = ranks from one returning LDS atomic per key (needs lds_atomics_ordered), 1 = two ranking rounds of the 4-bit machinery (the product's), 2 = its one-trip variant
    int r8_stay = -1;           // kernel 1 as a grid that stays (rsx::reorder8_stay_kernel): workgroups per CU of that grid, 0 / -1 = one workgroup per tile (RSX_XOPT_REORDER8_STAY)
    int lds_atomics_ordered = -1;               // -1 not probed yet; 1: ds_add_rtn serves lanes in ascending lane order on this device (lds_atomic_order_probe_kernel); 0: it does not, kernel 3 is refused
    uint32_t* tickets8 = nullptr;               // staying 8-bit scatter: [pass of the chain][XCD] tile tickets, zeroed at the start of every sort that uses them
#endif
    // Unused dynamic LDS per workgroup of the default 8-bit scatter = fewer workgroups per CU.  The 8-bit scatter leaves a tile as ~256 runs of ~16 keys whose first and last
    // sectors are completed by the NEIGHBOURING tile's runs; the halves merge only while the line is still in the XCD's L2, and the lines held open grow with the tiles in
    // flight: 64-bit elements (uint64 keys, or uint32 key + payload packed) overflow the 4 MiB at 3 workgroups per CU and are 10-27 % faster at 2
    // (profiles/r03_8bit_workgroups_per_cu.txt, r03_ab8_workgroups_policy.txt).  -1 = that policy (r8_extra_lds_for); env RSX_R8_EXTRA_LDS_KB = a fixed value for every variant.
    long r8_extra_lds = -1;
    // the same for the 4-bit reorder launches of 4096-key tiles (env RSX_REORDER_EXTRA_LDS_KB, <= 64; -1 = policy): only uint32 keys WITH a payload gain from
    // fewer workgroups per CU (three instead of four: six engines 6.26-6.84 -> 6.17-6.55 ms per sort); uint32 keys, uint64 keys and uint64 + payload lose 1-20 %
    // (profiles/r03_4bit_workgroups_per_cu.txt, r03_modes_u32pay4_workgroups_per_cu.txt)
    long reorder_extra_lds = -1;
    int reorder_wide = -1;      // 4-bit reorder of 64-bit keys WITH payload on 512 threads x 8 keys (env RSX_REORDER_WIDE: 0 / 1, -1 = policy: on)
    // kernel 1 on workgroups of 512 threads x 8 keys (the same 4096-key tiles, tables and LDS bytes in flight; twice the waves): env RSX_R8_WIDE, -1 = policy
    int r8_wide = -1;
    int num_cus = 0;
    int r8_packed = 1;          // 8-bit scatter of uint32 keys WITH payload: key and payload as one 64-bit element through the ranking rounds (env RSX_R8_PACKED; kernel 1 only)
    bool radix8_ready = false;                  // the five tables below exist and the reorder8 kernels may use their LDS
    uint32_t* counts8 = nullptr;                // 8-bit digits: raw counts [tile][256] (allocated on first use)
    uint32_t* table8 = nullptr;                 //   group-local exclusive prefixes [tile][256]
    uint32_t* gsum8 = nullptr;                  //   per scan group: totals, then prefixes inside the group's chunk [group][256]
    uint32_t* csum8 = nullptr;                  //   per chunk of groups: totals [chunk <= 16][256]
    uint32_t* cbase8 = nullptr;                 //   per chunk: smaller digits + this digit in earlier chunks [chunk][256]
    int self_scan = 1;          // rsx_sort: tables of at most self_scan_max tiles need no scan launch (env RSX_SELF_SCAN)
    uint32_t self_scan_max = 1024;              // env RSX_SELF_SCAN_MAX (<= 1024 tiles = 2^22 keys; measured: -36 % at 2^13..2^18, -24 % at 2^20, -11 % at 2^22)
    uint32_t* cnt3[3] = {nullptr, nullptr, nullptr};      // self-scan: three rotating [tile][16] count buffers
    uint64_t small_tile_max_keys = 1u << 19;              // self-scan sorts of at most this many keys use tiles of 256 x 4 keys (env RSX_SMALL_TILE_MAX_KEYS; measured: -20 % up to 2^16, -16 % at 2^18, -6 % at 2^19, +20 % at 2^20)
    int tile_sort = 1;          // rsx_sort: inputs of at most one tile are sorted by ONE workgroup in ONE launch, all passes in LDS (env RSX_TILE_SORT)
    int first_pass = 0;
    int last_pass = 0;

    int use_graph = 0;          // RSX_OPT_GRAPH: replay sorts of <= 2^22 keys from a captured hipGraph (measured: no gain, off)
    std::vector<GraphEntry> graphs;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> pool;
    rsx_phase_stat stats[PH_COUNT];

    uint32_t passes() const { return static_cast<uint32_t>(key_bytes * 8 / RSX_RADIX_BITS); }
    uint64_t ntiles(uint64_t count) const { return (count + kTileKeys - 1) / kTileKeys; }
};

namespace {

struct Bracket {   // optional HIP-event pair around one launch (profile mode)
    rsx_engine* e;
    int phase;
    hipEvent_t start = nullptr, stop = nullptr;
    bool on = false;
    Bracket(rsx_engine* eng, int ph) : e(eng), phase(ph)
    {
        if (e->profile == 0 || (e->profile == 2 && ph != PH_REORDER && ph != PH_TOTAL)) return;
        auto take = [&]() -> hipEvent_t {
            if (!e->pool.empty()) {
                hipEvent_t ev = e->pool.back();
                e->pool.pop_back();
                return ev;
            }
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) != hipSuccess) return nullptr;
            return ev;
        };
        start = take();
        stop = take();
        on = start && stop;
        if (on) (void)hipEventRecord(start, e->stream);
    }
    ~Bracket()
    {
        if (!on) return;
        (void)hipEventRecord(stop, e->stream);
        e->pending.push_back({phase, start, stop});
    }
};

int drain_events(rsx_engine* e)
{
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (const EventPair& p : e->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            stat_update(e->stats[p.phase], static_cast<double>(ms));
        }
        e->pool.push_back(p.start);
        e->pool.push_back(p.stop);
    }
    e->pending.clear();
    return RSX_OK;
}

template <typename Key>
Key flip_mask(const rsx_engine* e)
{
    return e->is_signed ? static_cast<Key>(Key{1} << (sizeof(Key) * 8 - 1)) : Key{0};
}

struct Grid {
    uint32_t ntiles, tiles_per_xcd, blocks;
    int remap;        // kernel argument: bit 0 = XCD-contiguous tile ranges, bits 8.. = phase stagger between the XCDs (tiles)
};

Grid grid_for(const rsx_engine* e, uint64_t count, uint32_t tile_keys = kTileKeys)
{
    Grid g;
    g.ntiles = static_cast<uint32_t>((count + tile_keys - 1) / tile_keys);
    g.tiles_per_xcd = (g.ntiles + rsx::kNumXcd - 1) / rsx::kNumXcd;
    g.blocks = e->xcd_remap ? g.tiles_per_xcd * rsx::kNumXcd : g.ntiles;
    // XCD x enters its range x * phase tiles in (rsx::tile_of_block): ranges that start n/8 apart walked in lockstep keep the
    // eight XCDs on the same HBM channels (-6 % at 2^28 keys, profiles/r02_tuning_log.md §6)
    const uint64_t want = e->xcd_phase < 0 ? g.tiles_per_xcd / rsx::kNumXcd : static_cast<uint64_t>(e->xcd_phase);
    const uint32_t phase = (e->xcd_remap && (rsx::kNumXcd - 1) * want < g.tiles_per_xcd) ? static_cast<uint32_t>(want) : 0u;
    g.remap = (e->xcd_remap ? 1 : 0) | static_cast<int>(phase << 8);
    return g;
}

// the first `nsplit` splitters of the engine narrowed to the key type (0 = not a splitter launch)
template <typename Key>
rsx::SplitSet<Key> split_set(const rsx_engine* e, uint32_t nsplit)
{
    rsx::SplitSet<Key> set{};
    for (uint32_t k = 0; k < nsplit && k < static_cast<uint32_t>(rsx::kMaxSplitters); ++k) set.s[k] = static_cast<Key>(e->splitters[k]);
    set.n = nsplit;
    return set;
}

template <typename Key, bool RANGED = false>
int launch_histogram(rsx_engine* e, const void* in, uint64_t count, int shift, uint32_t mask, Key lo = Key{0}, Key mul = Key{0}, uint32_t nsplit = 0)
{
    if (count == 0) return RSX_OK;
    e->counted_keys = nullptr;          // e->table is about to be overwritten: an earlier rsx_partition_count* is void
    e->table_valid = true;
    e->globsum_valid = false;
    const Grid g = grid_for(e, count);
    Bracket b(e, PH_HISTO);
    hipLaunchKernelGGL((rsx::histogram_kernel<Key, kTileThreads, kKeysPerThread, RANGED>), dim3(g.blocks), dim3(kTileThreads), 0, e->stream,
                       static_cast<const Key*>(in), e->table, count, g.ntiles, g.tiles_per_xcd, g.remap, shift,
                       flip_mask<Key>(e), mask, lo, mul, split_set<Key>(e, nsplit));
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// scan_level2 = false: only scan #1 runs and globsum keeps the RAW group sums (for launch_paste_scan)
int launch_scan(rsx_engine* e, uint64_t count, bool from_counts = false, bool scan_level2 = true)
{
    if (count == 0) return RSX_OK;
    e->counted_keys = nullptr;          // the raw counts turn into prefixes: a pending count is consumed or void
    const uint32_t ntiles = static_cast<uint32_t>(e->ntiles(count));
    const uint32_t ngroups = (ntiles + rsx::kScanTiles - 1) / rsx::kScanTiles;
    e->globsum_live = e->globsum;
    e->table_valid = true;
    e->globsum_valid = true;
    {
        Bracket b(e, PH_SCAN);
        if (from_counts) {
            hipLaunchKernelGGL((rsx::scan_blocks_kernel<true, true>), dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->globsum,
                               ntiles, ngroups, e->counts_next);
        } else {
            hipLaunchKernelGGL((rsx::scan_blocks_kernel<false, false>), dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->globsum,
                               ntiles, ngroups, static_cast<uint32_t*>(nullptr));
        }
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    if (scan_level2) {
        Bracket b(e, PH_SCAN);
        hipLaunchKernelGGL(rsx::scan_globsum_kernel, dim3(1), dim3(rsx::kGlobsumThreads), 0, e->stream, e->globsum, e->temp,
                           static_cast<uint32_t>(RSX_RADIX) * ngroups);
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// rsx_sort only: scan #2 and paste in one launch (every workgroup reduces the raw group sums itself)
int launch_paste_scan(rsx_engine* e, uint64_t count)
{
    if (count == 0) return RSX_OK;
    const uint32_t ntiles = static_cast<uint32_t>(e->ntiles(count));
    const uint32_t ngroups = (ntiles + rsx::kScanTiles - 1) / rsx::kScanTiles;
    e->globsum_live = e->globsum2;
    e->globsum_valid = true;
    Bracket b(e, PH_PASTE);
    hipLaunchKernelGGL(rsx::paste_scan_kernel, dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->globsum, e->globsum2, e->temp,
                       ntiles, ngroups);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// rsx_sort only: scan #1, scan #2 and paste in ONE launch (all its workgroups resident at once; an arrival counter in
// global memory stands where the two kernel boundaries stood).  Returns true when it took the pass.
bool launch_scan_fused(rsx_engine* e, uint64_t count, bool from_counts, int* rc)
{
    const uint32_t ntiles = static_cast<uint32_t>(e->ntiles(count));
    const uint32_t ngroups = (ntiles + rsx::kScanTiles - 1) / rsx::kScanTiles;
    // (a captured graph would replay a stale epoch: the graph path keeps the separate launches)
    // every workgroup of the fused scan waits for every other one's group sums: the whole grid must be resident at once.
    // fused_scan_limit is derived from the occupancy query at rsx_create (half of what the device holds, so that two
    // engines scanning at the same time still fit); larger tables take scan #1, then scan #2 + paste (two launches).
    if (!e->fused_scan || e->use_graph || count == 0 || ngroups > e->fused_scan_limit) return false;
    e->counted_keys = nullptr;
    e->globsum_live = e->globsum2;
    e->table_valid = true;
    e->globsum_valid = true;
    if (++e->scan_epoch == 0) e->scan_epoch = 1;
    {
        Bracket b(e, PH_SCAN);
        if (from_counts) {
            hipLaunchKernelGGL((rsx::scan_fused_kernel<true, true>), dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->gsums, e->globsum2, e->temp,
                               ntiles, ngroups, e->counts_next, e->scan_epoch, e->scan_timeout);
        } else {
            hipLaunchKernelGGL((rsx::scan_fused_kernel<false, false>), dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->gsums, e->globsum2, e->temp,
                               ntiles, ngroups, e->counts_next, e->scan_epoch, e->scan_timeout);
        }
    }
    *rc = hipGetLastError() == hipSuccess ? RSX_OK : fail(RSX_CALCULATION_FAILED, "scan_fused_kernel launch");
    return true;
}

// rsx_sort only: tables of at most 1024 tiles (2^22 keys) are scanned AND pasted by one workgroup
// in one launch.  Returns true when it took the pass (the caller then skips the paste).
bool launch_scan_small(rsx_engine* e, uint64_t count, bool from_counts, int* rc)
{
    const uint32_t ntiles = static_cast<uint32_t>(e->ntiles(count));
    if (!e->small_scan || count == 0 || ntiles > static_cast<uint32_t>(rsx::kSmallScanMaxTiles)) return false;
    e->counted_keys = nullptr;
    e->table_valid = true;
    e->globsum_valid = false;           // one workgroup scans the whole table: there are no group sums
    {
        Bracket b(e, PH_SCAN);
        if (from_counts) {
            hipLaunchKernelGGL((rsx::scan_small_kernel<true, true>), dim3(1), dim3(rsx::kSmallScanThreads), 0, e->stream, e->table, e->counts_next, e->temp, ntiles);
        } else {
            hipLaunchKernelGGL((rsx::scan_small_kernel<false, false>), dim3(1), dim3(rsx::kSmallScanThreads), 0, e->stream, e->table, e->counts_next, e->temp, ntiles);
        }
    }
    *rc = hipGetLastError() == hipSuccess ? RSX_OK : fail(RSX_CALCULATION_FAILED, "scan_small_kernel launch");
    return true;
}

int launch_paste(rsx_engine* e, uint64_t count)
{
    if (count == 0) return RSX_OK;
    const uint32_t ntiles = static_cast<uint32_t>(e->ntiles(count));
    const uint32_t ngroups = (ntiles + rsx::kScanTiles - 1) / rsx::kScanTiles;
    Bracket b(e, PH_PASTE);
    hipLaunchKernelGGL(rsx::paste_kernel, dim3(ngroups), dim3(rsx::kScanTiles), 0, e->stream, e->table, e->globsum, ntiles, ngroups);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// next_shift < 0: plain reorder.  next_shift >= 0: also count digit (key >> next_shift) & 15 per
// OUTPUT tile into e->counts_next (all zero on entry: zeroed at the start of the sort and handed
// back zeroed by the scan that consumes it).
template <typename Key, bool PAYLOAD, bool LOOKAHEAD, bool RANGED = false, int KPT = kKeysPerThread, int THREADS = kTileThreads>
int launch_reorder_t(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, int shift,
                     uint32_t mask, int next_shift, Key lo = Key{0}, Key mul = Key{0}, uint32_t nsplit = 0,
                     rsx::SelfScanArgs self = rsx::SelfScanArgs{nullptr, nullptr, nullptr}, uint32_t* next_counts = nullptr)
{
    using L = rsx::ReorderLayout<Key, THREADS, KPT>;
    const Grid g = grid_for(e, count, THREADS * KPT);
    e->last_in = in;
    e->last_shift = shift;
    Bracket b(e, PH_REORDER);
    hipLaunchKernelGGL((rsx::reorder_kernel<Key, THREADS, KPT, PAYLOAD, LOOKAHEAD, RANGED>), dim3(g.blocks), dim3(THREADS),
                       L::BYTES + (KPT != kKeysPerThread ? 0u : e->reorder_extra_lds >= 0 ? static_cast<size_t>(e->reorder_extra_lds) : (PAYLOAD && sizeof(Key) == 4 && !RANGED) ? (16u << 10) : 0u),
                       e->stream, static_cast<const Key*>(in), static_cast<Key*>(out), pin, pout, e->table, count,
                       g.ntiles, g.tiles_per_xcd, g.remap, shift, flip_mask<Key>(e), mask,
                       next_counts ? next_counts : e->counts_next, next_shift,
#ifdef RSX_STAMPS
                       (shift == e->stamp_pass * RSX_RADIX_BITS) ? reinterpret_cast<const uint32_t*>(e->stamps) : static_cast<const uint32_t*>(nullptr),
#else
                       static_cast<const uint32_t*>(nullptr),      // (folding the paste into the scatter — it adds globsum[group] itself — was measured 3 % slower and removed in round 4)
#endif
                       lo, mul, split_set<Key>(e, nsplit), self);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

template <typename Key>
int launch_reorder(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, int shift,
                   uint32_t mask, int next_shift = -1)
{
    if (count == 0) return RSX_OK;
    const bool payload = pin && pout;
#if RSX_PRODUCT_SHAPE
    if constexpr (sizeof(Key) == 8) {
        // 64-bit keys WITH payload: the same 4096-key tiles (same table) ranked by 512 threads x 8 keys — 140 VGPRs and three waves per SIMD become ~90 and four;
        // 1.149-1.184 -> 1.092-1.127 ms per launch (profiles/r03_ab_4bit_512x8.txt).  Every other 4-bit variant is 1-4 % slower that way and keeps 256 x 16.
        if (payload && (e->reorder_wide >= 0 ? e->reorder_wide != 0 : true)) {
            return next_shift >= 0 ? launch_reorder_t<Key, true, true, false, 8, 512>(e, in, out, pin, pout, count, shift, mask, next_shift)
                                   : launch_reorder_t<Key, true, false, false, 8, 512>(e, in, out, pin, pout, count, shift, mask, 0);
        }
    }
#endif
    if (next_shift >= 0) {
        return payload ? launch_reorder_t<Key, true, true>(e, in, out, pin, pout, count, shift, mask, next_shift)
                       : launch_reorder_t<Key, false, true>(e, in, out, nullptr, nullptr, count, shift, mask, next_shift);
    }
    return payload ? launch_reorder_t<Key, true, false>(e, in, out, pin, pout, count, shift, mask, 0)
                   : launch_reorder_t<Key, false, false>(e, in, out, nullptr, nullptr, count, shift, mask, 0);
}

// The scatter kernels address their staging image from LDS address 0 (rsx::lds_store_at): true when the kernel declares no static LDS
// in front of its dynamic array — checked here per kernel from its code-object attributes — and the device places a lone dynamic array
// at 0 — checked by one probe launch per engine (lds_base_probe, rsx_create).  Either failing is a build / device the kernels were
// not written for: rsx_create returns KERNEL_CREATION_FAILED (src/OperationStatus.h:12) instead of anything going wrong on the device.
int no_static_lds(const void* fn, const char* name)
{
    hipFuncAttributes attr;
    RSX_TRY(hipFuncGetAttributes(&attr, fn), RSX_KERNEL_CREATION_FAILED);
    if (attr.sharedSizeBytes != 0) return fail(RSX_KERNEL_CREATION_FAILED, (std::string(name) + ": static LDS in front of the dynamic array (the kernels address LDS from 0)").c_str());
    return RSX_OK;
}

int lds_base_probe(rsx_engine* e)
{
    uint32_t at = 1;
    RSX_TRY(hipMemsetAsync(e->temp + 4, 0xFF, 4, e->stream), RSX_KERNEL_CREATION_FAILED);
    hipLaunchKernelGGL(rsx::lds_base_probe_kernel, dim3(1), dim3(64), 256, e->stream, e->temp + 4);
    RSX_TRY(hipGetLastError(), RSX_KERNEL_CREATION_FAILED);
    RSX_TRY(hipMemcpyAsync(&at, e->temp + 4, 4, hipMemcpyDeviceToHost, e->stream), RSX_KERNEL_CREATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_KERNEL_CREATION_FAILED);
    if (at != 0) return fail(RSX_KERNEL_CREATION_FAILED, "a kernel's dynamic LDS does not start at address 0 on this device");
    return RSX_OK;
}

template <typename Key, bool PAYLOAD, bool LOOKAHEAD, bool RANGED = false, int KPT = kKeysPerThread, int THREADS = kTileThreads>
int allow_lds()
{
    using L = rsx::ReorderLayout<Key, THREADS, KPT>;
    const void* fn = reinterpret_cast<const void*>(&rsx::reorder_kernel<Key, THREADS, KPT, PAYLOAD, LOOKAHEAD, RANGED>);
    RSX_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L::BYTES) + (64 << 10)), RSX_INITIALIZATION_FAILED);
    return no_static_lds(fn, "reorder_kernel");
}

// One full pass chain on explicit buffers: histogram -> scan -> paste -> reorder.
template <typename Key>
int run_pass(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, int shift, uint32_t mask)
{
    int rc;
    if ((rc = launch_histogram<Key>(e, in, count, shift, mask)) != RSX_OK) return rc;
    if ((rc = launch_scan(e, count)) != RSX_OK) return rc;
    if ((rc = launch_paste(e, count)) != RSX_OK) return rc;
    return launch_reorder<Key>(e, in, out, pin, pout, count, shift, mask);
}

// One stable pass with the ranged bucket function (multi-GPU partition).
template <typename Key>
int run_ranged_pass(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, uint64_t lo, int shift,
                    uint64_t mul)
{
    const Key klo = static_cast<Key>(lo), kmul = static_cast<Key>(mul);
    int rc;
    if ((rc = launch_histogram<Key, true>(e, in, count, shift, RSX_RADIX - 1, klo, kmul)) != RSX_OK) return rc;
    if ((rc = launch_scan(e, count)) != RSX_OK) return rc;
    if ((rc = launch_paste(e, count)) != RSX_OK) return rc;
    if (pin && pout) return launch_reorder_t<Key, true, false, true>(e, in, out, pin, pout, count, shift, RSX_RADIX - 1, 0, klo, kmul);
    return launch_reorder_t<Key, false, false, true>(e, in, out, nullptr, nullptr, count, shift, RSX_RADIX - 1, 0, klo, kmul);
}

constexpr int kRangeBlocks = 2048;

template <typename Key>
int key_range(rsx_engine* e, const void* d_keys, uint64_t n, uint64_t* lo, uint64_t* hi)
{
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>(kRangeBlocks, (n + 4095) / 4096));
    hipLaunchKernelGGL(rsx::key_range_kernel<Key>, dim3(blocks), dim3(rsx::kRangeThreads), 0, e->stream, static_cast<const Key*>(d_keys), n,
                       flip_mask<Key>(e), e->range_dev);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->range_host, e->range_dev, static_cast<size_t>(blocks) * 16, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    unsigned long long l = ~0ull, h = 0ull;
    for (uint32_t b = 0; b < blocks; ++b) {
        l = std::min(l, e->range_host[2 * b]);
        h = std::max(h, e->range_host[2 * b + 1]);
    }
    *lo = l;
    *hi = h;
    return RSX_OK;
}

// Inputs of at most one tile: every pass inside LDS, one launch (rsx::tile_sort_kernel).  The buffers end up
// exactly as the pass chain would leave them: the result where the chain's last pass would have written it,
// the order before the last pass in the other ping-pong buffer, table / group sums / total of the last pass.
template <typename Key, bool PAYLOAD>
int launch_tile_sort_t(rsx_engine* e, const void* in, void* out, void* before_last, const uint32_t* pin, uint32_t* pout, uint32_t* pbefore_last, uint64_t count)
{
    using L = rsx::TileSortLayout<Key, kTileThreads, kKeysPerThread>;
    Bracket b(e, PH_REORDER);
    hipLaunchKernelGGL((rsx::tile_sort_kernel<Key, kTileThreads, kKeysPerThread, PAYLOAD>), dim3(1), dim3(kTileThreads), L::BYTES, e->stream,
                       static_cast<const Key*>(in), static_cast<Key*>(out), static_cast<Key*>(before_last), pin, pout, pbefore_last,
                       static_cast<uint32_t>(count), e->first_pass, e->last_pass, flip_mask<Key>(e), e->table, e->globsum, e->temp);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

template <typename Key>
int sort_tile_enqueue(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count)
{
    const int npasses = e->last_pass - e->first_pass;
    const void* in = ext_keys ? ext_keys : e->keys[e->cur];
    const uint32_t* pin = e->has_payload ? (ext_keys ? ext_perm : e->perm[e->cur]) : nullptr;
    const int first_dst = ext_keys ? e->cur : (e->cur ^ 1);
    const int final_buf = first_dst ^ ((npasses - 1) & 1);            // where the chain's last pass writes
    void* out = e->final_keys_out ? e->final_keys_out : e->keys[final_buf];
    uint32_t* pout = e->has_payload ? (e->final_keys_out ? e->final_perm_out : e->perm[final_buf]) : nullptr;
    void* before_last = npasses > 1 ? e->keys[final_buf ^ 1] : nullptr;
    uint32_t* pbefore_last = (npasses > 1 && e->has_payload) ? e->perm[final_buf ^ 1] : nullptr;
    // (with internal input and an odd pass count `before_last` is the input buffer itself: the one workgroup has
    // read all of it into registers before anything is written)
    Bracket whole(e, PH_TOTAL);
    e->counted_keys = nullptr;
    e->globsum_live = e->globsum;
    const int rc = e->has_payload ? launch_tile_sort_t<Key, true>(e, in, out, before_last, pin, pout, pbefore_last, count)
                                  : launch_tile_sort_t<Key, false>(e, in, out, before_last, nullptr, nullptr, nullptr, count);
    if (rc != RSX_OK) return rc;
    e->last_in = npasses > 1 ? before_last : in;
    e->last_shift = (e->last_pass - 1) * RSX_RADIX_BITS;
    e->table_valid = true;
    e->globsum_valid = true;
    e->result_external = e->final_keys_out != nullptr;
    if (e->final_keys_out) {
        e->result_keys = e->final_keys_out;
        e->result_perm = e->has_payload ? e->final_perm_out : nullptr;
    } else {
        e->cur = final_buf;
        e->result_keys = e->keys[e->cur];
        e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    }
    return RSX_OK;
}

// Small tables (2 .. self_scan_max tiles): no scan launch at all — one histogram launch for the first pass, then one
// reorder launch per pass; every reorder workgroup derives its own 16 bases from the raw [tile][16] counts (three
// rotating count buffers, see rsx::SelfScanArgs).  The sort is then `passes + 1` dependent launches instead of `2 passes + 2`.
template <typename Key, int KPT = kKeysPerThread>
int sort_selfscan_enqueue(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count)
{
    // KPT < 16: tiles of 256 x KPT keys for small inputs — a tile's trip through the reorder (what every pass of a small sort
    // waits for) is a chain of per-key steps, so a quarter of the keys per thread is a shorter chain; the engine's
    // own [digit][tile] table read-back is not produced in that geometry
    const Grid g = grid_for(e, count, kTileThreads * KPT);
    const void* in = ext_keys ? ext_keys : e->keys[e->cur];
    const uint32_t* pin = e->has_payload ? (ext_keys ? ext_perm : e->perm[e->cur]) : nullptr;
    int dst = ext_keys ? e->cur : (e->cur ^ 1);
    Bracket whole(e, PH_TOTAL);
    e->counted_keys = nullptr;
    {
        Bracket b(e, PH_HISTO);
        hipLaunchKernelGGL((rsx::histogram_kernel<Key, kTileThreads, KPT, false>), dim3(g.blocks), dim3(kTileThreads), 0, e->stream,
                           static_cast<const Key*>(in), e->table, count, g.ntiles, g.tiles_per_xcd, g.remap, e->first_pass * RSX_RADIX_BITS,
                           flip_mask<Key>(e), static_cast<uint32_t>(RSX_RADIX - 1), Key{0}, Key{0}, split_set<Key>(e, 0), e->cnt3[0], e->cnt3[1], e->cnt3[2]);
        RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    }
    int i = 0;
    for (int pass = e->first_pass; pass < e->last_pass; ++pass, ++i) {
        const bool last = pass + 1 == e->last_pass;
        const bool to_caller = e->final_keys_out && last;
        void* out = to_caller ? e->final_keys_out : e->keys[dst];
        uint32_t* pout = e->has_payload ? (to_caller ? e->final_perm_out : e->perm[dst]) : nullptr;
        const int shift = pass * RSX_RADIX_BITS;
        const rsx::SelfScanArgs self{e->cnt3[i % 3], e->cnt3[(i + 2) % 3], (last && KPT == kKeysPerThread) ? e->table : nullptr};
        uint32_t* next = e->cnt3[(i + 1) % 3];
        int rc;
        if (e->has_payload) {
            rc = last ? launch_reorder_t<Key, true, false, false, KPT>(e, in, out, pin, pout, count, shift, RSX_RADIX - 1, 0, Key{0}, Key{0}, 0, self, next)
                      : launch_reorder_t<Key, true, true, false, KPT>(e, in, out, pin, pout, count, shift, RSX_RADIX - 1, shift + RSX_RADIX_BITS, Key{0}, Key{0}, 0, self, next);
        } else {
            rc = last ? launch_reorder_t<Key, false, false, false, KPT>(e, in, out, nullptr, nullptr, count, shift, RSX_RADIX - 1, 0, Key{0}, Key{0}, 0, self, next)
                      : launch_reorder_t<Key, false, true, false, KPT>(e, in, out, nullptr, nullptr, count, shift, RSX_RADIX - 1, shift + RSX_RADIX_BITS, Key{0}, Key{0}, 0, self, next);
        }
        if (rc != RSX_OK) return rc;
        in = out;
        pin = pout;
        dst ^= 1;
    }
    // (tiles of 1024 keys leave no table in the engine's own geometry; the 4096-key form writes it in its last pass — but no group sums)
    e->table_valid = (KPT == kKeysPerThread);
    e->globsum_valid = false;
    if (in == e->keys[0] || in == e->keys[1]) e->cur = (in == e->keys[0]) ? 0 : 1;
    e->result_external = e->final_keys_out != nullptr;
    if (e->final_keys_out) {
        e->result_keys = e->final_keys_out;
        e->result_perm = e->has_payload ? e->final_perm_out : nullptr;
    } else {
        e->result_keys = e->keys[e->cur];
        e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    }
    return RSX_OK;
}

#if RSX_TILE_THREADS == 256      // (the 8-bit kernels are written for 256-thread tiles: one thread per digit)
// extra dynamic LDS of the 8-bit scatter, by variant: 2 workgroups per CU instead of 3 for 64-bit elements (see rsx_engine::r8_extra_lds)
size_t r8_extra_lds_for(const rsx_engine* e, bool elem64, bool separate_payload)
{
    if (e->r8_extra_lds >= 0) return static_cast<size_t>(e->r8_extra_lds);
    // uint32 keys only: 29 KiB + 4 = four workgroups per CU instead of five — 1 % slower on random keys (0.586 -> 0.579 of peak), 4-8 % faster on Range / InvertedRange
    // (0.55-0.57 -> 0.584-0.594; r03_8bit_range_workgroups_per_cu.txt); with a separate payload array (RSX_R8_PACKED=0) the registers allow four anyway
    if (!elem64) return separate_payload ? 0 : (4u << 10);
    // uint64 keys / packed uint32 key + payload: 45 KiB + 8 = two workgroups per CU (+10 % / +27 % on random keys, +15 % on Range), and the spread between
    // engines of one process — the "modes" of rounds 1-2 — shrinks from 7-11 % to 2 % (profiles/r03_modes_vs_workgroups_per_cu.txt).  uint64 keys with a payload
    // array likewise (six engines: 14.9-16.6 ms per sort at three, 14.4-14.8 at two), at the price of 9 % on constant data, which has nothing to merge
    if (separate_payload) return 16u << 10;
    return 8u << 10;
}

// 512-thread workgroups for the 8-bit scatter, by variant
bool r8_wide_for(const rsx_engine* e, bool key64, bool has_payload)
{
    if (e->r8_wide >= 0) return e->r8_wide != 0;
    // 64-bit keys without payload run at two waves per SIMD on 256-thread tiles (two workgroups per CU is all the L2 takes, r8_extra_lds_for) and wait: twice the waves
    // on the same bytes in flight, 0.896 -> 0.807 ms per launch on uniform keys, 0.776 -> 0.720 on constant ones.  Every other variant loses 2-12 % to the doubled
    // per-thread overhead (counter words, raking scan; 8-wave barriers): profiles/r03_ab_wide.txt
    return key64 && !has_payload;
}

template <typename K, int THREADS, int KPT, bool PAYLOAD, bool PACKED32 = false>
int allow_lds8(int extra)
{
    using L = rsx::Reorder8Layout<K, THREADS, KPT, PAYLOAD>;
    const void* fn = reinterpret_cast<const void*>(&rsx::reorder8_kernel<K, THREADS, KPT, PAYLOAD, PACKED32>);
    RSX_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L::BYTES) + extra), RSX_INITIALIZATION_FAILED);
    return no_static_lds(fn, "reorder8_kernel");
}

#ifdef RSX_EXPERIMENTS
template <typename Key> int ensure_radix8_experiments(rsx_engine* e, int extra);
template <typename Key> bool launch_reorder8_experiment(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, const Grid& g,
                                                        uint32_t chunk_groups, int shift, int byte_pass, bool first_of_chain, int* rc);
#endif

// The 8-bit chain's tables, allocated on first use — by sort_chain BEFORE any stream capture begins (an allocation inside
// hipStreamBeginCapture invalidates the capture) and by rsx_set_option(RSX_OPT_RADIX_BITS, 8).  A call that failed half-way
// keeps what it got and the next one asks for the rest.
template <typename Key>
int ensure_radix8(rsx_engine* e)
{
    if (e->radix8_ready) return RSX_OK;
    const int extra = e->r8_extra_lds >= 0 ? static_cast<int>(e->r8_extra_lds) : (16 << 10);      // the most any variant asks for
    const size_t rows = static_cast<size_t>(e->ntiles(e->capacity)) * rsx::kRadix8 * 4;
    const size_t groups = ((e->ntiles(e->capacity) + rsx::kScan8Tiles - 1) / rsx::kScan8Tiles) * rsx::kRadix8 * 4;
    if (!e->counts8) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&e->counts8), rows), RSX_INITIALIZATION_FAILED);
    if (!e->table8) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&e->table8), rows), RSX_INITIALIZATION_FAILED);
    if (!e->gsum8) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&e->gsum8), groups), RSX_INITIALIZATION_FAILED);
    if (!e->csum8) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&e->csum8), rsx::kScan8MaxChunks * rsx::kRadix8 * 4), RSX_INITIALIZATION_FAILED);
    if (!e->cbase8) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&e->cbase8), rsx::kScan8MaxChunks * rsx::kRadix8 * 4), RSX_INITIALIZATION_FAILED);
    int rc = allow_lds8<Key, kTileThreads, kKeysPerThread, false>(extra);
    if (rc == RSX_OK) rc = allow_lds8<Key, kTileThreads, kKeysPerThread, true>(extra);
    if (rc == RSX_OK) rc = allow_lds8<Key, 512, 8, false>(extra);
    if (rc == RSX_OK) rc = allow_lds8<Key, 512, 8, true>(extra);
    if constexpr (sizeof(Key) == 4) {          // uint32 key + payload as one 64-bit element
        if (rc == RSX_OK) rc = allow_lds8<uint64_t, kTileThreads, kKeysPerThread, false, true>(extra);
        if (rc == RSX_OK) rc = allow_lds8<uint64_t, 512, 8, false, true>(extra);
    }
#ifdef RSX_EXPERIMENTS
    if (rc == RSX_OK) rc = ensure_radix8_experiments<Key>(e, extra);
#endif
    if (rc != RSX_OK) return rc;
    e->radix8_ready = true;
    return RSX_OK;
}

// One scatter launch of an 8-bit pass: by variant (keys only / separate payload array / uint32 key + payload packed into one 64-bit
// element) on 256 x 16 or 512 x 8 threads x keys; the same 4096-key tiles and tables for all of them.
template <typename Key>
int launch_reorder8(rsx_engine* e, const void* in, void* out, const uint32_t* pin, uint32_t* pout, uint64_t count, const Grid& g, uint32_t chunk_groups, int shift)
{
    const Key flip = flip_mask<Key>(e);
    e->last_in = nullptr;            // an 8-bit pass has no counterpart in the reference's geometry: RSX_OPT_REF_DIAGNOSTICS refuses after it (a later 4-bit pass sets it again)
    const bool packed = sizeof(Key) == 4 && e->has_payload && e->r8_packed;
    const bool wide = r8_wide_for(e, sizeof(Key) == 8, e->has_payload);
    const size_t wide_extra = e->r8_extra_lds >= 0 ? static_cast<size_t>(e->r8_extra_lds) : 0;      // (57-62 KiB per workgroup: two per CU as they stand)
    const dim3 grid(g.blocks);
    // (dynamic LDS per variant, named here: template commas do not survive inside the launch macro)
    constexpr size_t lds_packed = rsx::Reorder8Layout<uint64_t, kTileThreads, kKeysPerThread>::BYTES, lds_packed_wide = rsx::Reorder8Layout<uint64_t, 512, 8, false>::BYTES;
    constexpr size_t lds_pay = rsx::Reorder8Layout<Key, kTileThreads, kKeysPerThread, true>::BYTES, lds_pay_wide = rsx::Reorder8Layout<Key, 512, 8, true>::BYTES;
    constexpr size_t lds_keys = rsx::Reorder8Layout<Key, kTileThreads, kKeysPerThread>::BYTES, lds_keys_wide = rsx::Reorder8Layout<Key, 512, 8, false>::BYTES;
    if (packed) {
        // uint32 key + payload as one 64-bit element (rsx::reorder8_kernel<.., PACKED32>)
        if (wide) {
            hipLaunchKernelGGL((rsx::reorder8_kernel<uint64_t, 512, 8, false, true>), grid, dim3(512), lds_packed_wide + wide_extra, e->stream,
                               static_cast<const uint64_t*>(in), static_cast<uint64_t*>(out), pin, pout, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, static_cast<uint64_t>(flip));
        } else {
            hipLaunchKernelGGL((rsx::reorder8_kernel<uint64_t, kTileThreads, kKeysPerThread, false, true>), grid, dim3(kTileThreads),
                               lds_packed + r8_extra_lds_for(e, true, false), e->stream,
                               static_cast<const uint64_t*>(in), static_cast<uint64_t*>(out), pin, pout, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, static_cast<uint64_t>(flip));
        }
    } else if (e->has_payload) {
        if (wide) {
            hipLaunchKernelGGL((rsx::reorder8_kernel<Key, 512, 8, true>), grid, dim3(512), lds_pay_wide + wide_extra, e->stream,
                               static_cast<const Key*>(in), static_cast<Key*>(out), pin, pout, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, flip);
        } else {
            hipLaunchKernelGGL((rsx::reorder8_kernel<Key, kTileThreads, kKeysPerThread, true>), grid, dim3(kTileThreads),
                               lds_pay + r8_extra_lds_for(e, sizeof(Key) == 8, true), e->stream,
                               static_cast<const Key*>(in), static_cast<Key*>(out), pin, pout, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, flip);
        }
    } else {
        if (wide) {
            hipLaunchKernelGGL((rsx::reorder8_kernel<Key, 512, 8, false>), grid, dim3(512), lds_keys_wide + wide_extra, e->stream,
                               static_cast<const Key*>(in), static_cast<Key*>(out), nullptr, nullptr, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, flip);
        } else {
            hipLaunchKernelGGL((rsx::reorder8_kernel<Key, kTileThreads, kKeysPerThread, false>), grid, dim3(kTileThreads),
                               lds_keys + r8_extra_lds_for(e, sizeof(Key) == 8, false), e->stream,
                               static_cast<const Key*>(in), static_cast<Key*>(out), nullptr, nullptr, e->table8, e->gsum8, e->cbase8, chunk_groups,
                               count, g.ntiles, g.tiles_per_xcd, g.remap, shift, flip);
        }
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// chunks of the second scan level for a table of `ngroups` scan groups
struct Scan8Shape {
    uint32_t ngroups, chunk_groups, nchunks;
};
Scan8Shape scan8_shape(uint32_t ntiles)
{
    Scan8Shape s;
    s.ngroups = (ntiles + rsx::kScan8Tiles - 1) / rsx::kScan8Tiles;
    s.chunk_groups = std::max<uint32_t>(64u, (s.ngroups + rsx::kScan8MaxChunks - 1) / rsx::kScan8MaxChunks);
    s.nchunks = (s.ngroups + s.chunk_groups - 1) / s.chunk_groups;
    return s;
}

// histogram8 -> scan8 (blocks, chunks[, top]) of one 8-bit pass: leaves table8 / gsum8 / cbase8 ready for the scatter
template <typename Key>
int launch_count8(rsx_engine* e, const void* in, uint64_t count, const Grid& g, const Scan8Shape& s, int shift)
{
    e->msd_keys = nullptr;              // the 8-bit tables are about to be overwritten: a pending rsx_msd_count is void
    {
        Bracket b(e, PH_HISTO);
        hipLaunchKernelGGL((rsx::histogram8_kernel<Key, kTileThreads, kKeysPerThread>), dim3(g.blocks), dim3(kTileThreads), 0, e->stream,
                           static_cast<const Key*>(in), e->counts8, count, g.ntiles, g.tiles_per_xcd, g.remap, shift, flip_mask<Key>(e));
    }
    {
        Bracket b(e, PH_SCAN);
        hipLaunchKernelGGL(rsx::scan8_blocks_kernel, dim3(s.ngroups), dim3(rsx::kRadix8), 0, e->stream, e->counts8, e->table8, e->gsum8, g.ntiles);
    }
    {
        Bracket b(e, PH_SCAN);
        if (s.nchunks == 1) {
            hipLaunchKernelGGL(rsx::scan8_chunks_kernel<true>, dim3(1), dim3(rsx::kRadix8), 0, e->stream, e->gsum8, e->csum8, s.ngroups, s.chunk_groups, e->cbase8, e->temp);
        } else {
            hipLaunchKernelGGL(rsx::scan8_chunks_kernel<false>, dim3(s.nchunks), dim3(rsx::kRadix8), 0, e->stream, e->gsum8, e->csum8, s.ngroups, s.chunk_groups, e->cbase8, e->temp);
            hipLaunchKernelGGL(rsx::scan8_top_kernel, dim3(1), dim3(rsx::kRadix8), 0, e->stream, e->csum8, e->cbase8, e->temp, s.nchunks);
        }
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// 8-bit digits: per pass histogram8 -> scan8 (two launches) -> reorder8, half as many passes.  Taken by the sort
// chain when RSX_OPT_RADIX_BITS is 8 and the pass range [first_pass, last_pass) — counted in 4-bit passes, as
// everywhere in this API — covers whole bytes.
template <typename Key>
int sort8_chain_enqueue(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count)
{
    const Grid g = grid_for(e, count);
    const Scan8Shape s = scan8_shape(g.ntiles);
    if (!e->radix8_ready) return fail(RSX_INITIALIZATION_FAILED, "sort8_chain_enqueue: the 8-bit tables were not allocated (ensure_radix8)");
    const void* in = ext_keys ? ext_keys : e->keys[e->cur];
    const uint32_t* pin = e->has_payload ? (ext_keys ? ext_perm : e->perm[e->cur]) : nullptr;
    int dst = ext_keys ? e->cur : (e->cur ^ 1);
    Bracket whole(e, PH_TOTAL);
    e->counted_keys = nullptr;
    for (int pass = e->first_pass; pass < e->last_pass; pass += 2) {
        const bool to_caller = e->final_keys_out && pass + 2 == e->last_pass;
        void* out = to_caller ? e->final_keys_out : e->keys[dst];
        uint32_t* pout = e->has_payload ? (to_caller ? e->final_perm_out : e->perm[dst]) : nullptr;
        const int shift = pass * RSX_RADIX_BITS;
        int rc = launch_count8<Key>(e, in, count, g, s, shift);
        if (rc != RSX_OK) return rc;
        {
            Bracket b(e, PH_REORDER);
#ifdef RSX_EXPERIMENTS
            if (!launch_reorder8_experiment<Key>(e, in, out, pin, pout, count, g, s.chunk_groups, shift, pass / 2, pass == e->first_pass, &rc))
#endif
                rc = launch_reorder8<Key>(e, in, out, pin, pout, count, g, s.chunk_groups, shift);
        }
        if (rc != RSX_OK) return rc;
        in = out;                                     // (launch_reorder8 cleared last_in: the reference-geometry diagnostics describe 4-bit passes, rsx_download refuses them after this chain)
        pin = pout;
        dst ^= 1;
    }
    e->table_valid = false;             // the 8-bit tables are [tile][256]: nothing in the engine's [digit][tile] geometry
    e->globsum_valid = false;
    if (in == e->keys[0] || in == e->keys[1]) e->cur = (in == e->keys[0]) ? 0 : 1;
    e->result_external = e->final_keys_out != nullptr;
    if (e->final_keys_out) {
        e->result_keys = e->final_keys_out;
        e->result_perm = e->has_payload ? e->final_perm_out : nullptr;
    } else {
        e->result_keys = e->keys[e->cur];
        e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    }
    return RSX_OK;
}

#endif

#ifdef RSX_EXPERIMENTS
template <typename Key> int sort_inline_enqueue(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count);
bool inline_scan_takes(const rsx_engine* e, uint64_t count);
#endif

template <typename Key>
int sort_chain_enqueue(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count)
{
    // (8-bit digits pay off from 2^19 keys: below that the 4-bit self-scan chain — 9 launches of a 1024- or 4096-key tile's
    // latency — is faster than 4 passes of four launches; same result either way)
#if RSX_TILE_THREADS == 256
    if (e->radix_bits == 8 && count > e->radix8_min_keys && e->first_pass < e->last_pass && (e->first_pass & 1) == 0 && (e->last_pass & 1) == 0) {
        return sort8_chain_enqueue<Key>(e, ext_keys, ext_perm, count);
    }
    if (e->radix_bits == 8 && count > e->radix8_min_keys && (e->first_pass & 1) == 0 && (e->last_pass & 1) == 1 && e->last_pass - e->first_pass >= 3) {
        // an odd range from a byte boundary (the sharded sort's local passes 0 .. P-2): whole bytes with 8-bit digits inside the
        // engine's buffers, then the last nibble as one 4-bit pass into wherever the result was asked for
        const int first = e->first_pass, last = e->last_pass;
        void* const keys_out = e->final_keys_out;
        uint32_t* const perm_out = e->final_perm_out;
        e->last_pass = last - 1;
        e->final_keys_out = nullptr;
        e->final_perm_out = nullptr;
        int rc = sort8_chain_enqueue<Key>(e, ext_keys, ext_perm, count);
        e->last_pass = last;
        e->final_keys_out = keys_out;
        e->final_perm_out = perm_out;
        if (rc != RSX_OK) return rc;
        e->first_pass = last - 1;
        rc = sort_chain_enqueue<Key>(e, nullptr, nullptr, count);
        e->first_pass = first;
        return rc;
    }
#endif
#if RSX_PRODUCT_SHAPE
    if (e->tile_sort && e->profile != 1 && count > 0 && count <= static_cast<uint64_t>(kTileKeys) && e->first_pass < e->last_pass) {
        return sort_tile_enqueue<Key>(e, ext_keys, ext_perm, count);
    }
#endif
    if (e->self_scan && e->lookahead && count > static_cast<uint64_t>(kTileKeys) && e->ntiles(count) <= e->self_scan_max && e->first_pass < e->last_pass) {
        if (count <= e->small_tile_max_keys) return sort_selfscan_enqueue<Key, kSmallKeysPerThread>(e, ext_keys, ext_perm, count);
        return sort_selfscan_enqueue<Key>(e, ext_keys, ext_perm, count);
    }
#if defined(RSX_EXPERIMENTS) && RSX_PRODUCT_SHAPE
    if (inline_scan_takes(e, count)) return sort_inline_enqueue<Key>(e, ext_keys, ext_perm, count);
#endif
    // Ping-pong.  With external input the first pass reads the caller's buffer (never
    // written) and the chain continues inside the engine's two buffers.
    const void* in = ext_keys ? ext_keys : e->keys[e->cur];
    const uint32_t* pin = e->has_payload ? (ext_keys ? ext_perm : e->perm[e->cur]) : nullptr;
    int dst = ext_keys ? e->cur : (e->cur ^ 1);
    Bracket whole(e, PH_TOTAL);
    if (e->lookahead && count > 0) {
        // look-ahead counters start from zero (a previous sort that failed midway may have left some)
        RSX_TRY(hipMemsetAsync(e->counts_next, 0, static_cast<size_t>(e->ntiles(count)) * RSX_RADIX * 4, e->stream), RSX_CALCULATION_FAILED);
    }
    for (int pass = e->first_pass; pass < e->last_pass; ++pass) {
        const bool to_caller = e->final_keys_out && pass + 1 == e->last_pass;      // rsx_sort_from_to
        void* out = to_caller ? e->final_keys_out : e->keys[dst];
        uint32_t* pout = e->has_payload ? (to_caller ? e->final_perm_out : e->perm[dst]) : nullptr;
        const int shift = pass * RSX_RADIX_BITS;
        int rc;
        if (!e->lookahead) {
            rc = run_pass<Key>(e, in, out, pin, pout, count, shift, RSX_RADIX - 1);
        } else {
            // only the first pass reads the keys for a histogram; every later table was
            // counted by the previous pass's reorder while it scattered (look-ahead)
            const bool first = pass == e->first_pass;
            rc = first ? launch_histogram<Key>(e, in, count, shift, RSX_RADIX - 1) : RSX_OK;
            bool pasted = false;
            if (rc == RSX_OK) pasted = launch_scan_small(e, count, /*from_counts=*/!first, &rc);
            if (rc == RSX_OK && !pasted) pasted = launch_scan_fused(e, count, /*from_counts=*/!first, &rc);
            const bool merged = e->paste_scan != 0;                  // scan #2 inside the paste launch
            if (rc == RSX_OK && !pasted) rc = launch_scan(e, count, /*from_counts=*/!first, /*scan_level2=*/!merged);
            // small tables were scanned and pasted in one launch; otherwise the paste kernel runs
            const bool last = pass + 1 == e->last_pass;
            if (rc == RSX_OK && !pasted) rc = merged ? launch_paste_scan(e, count) : launch_paste(e, count);
            const int next_shift = last ? -1 : shift + RSX_RADIX_BITS;
            if (rc == RSX_OK) rc = launch_reorder<Key>(e, in, out, pin, pout, count, shift, RSX_RADIX - 1, next_shift);
        }
        if (rc != RSX_OK) return rc;
        in = out;
        pin = pout;
        dst ^= 1;
    }
    // `in` now names the buffer holding the result; make it the engine's "inputKeys"
    // (the reference's even pass count guarantees the same, src/RadixSortGPU.cpp:263-266,394-400)
    if (in == e->keys[0] || in == e->keys[1]) {
        e->cur = (in == e->keys[0]) ? 0 : 1;
    }
    e->result_external = e->final_keys_out != nullptr;
    if (e->final_keys_out) {
        e->result_keys = e->final_keys_out;
        e->result_perm = e->has_payload ? e->final_perm_out : nullptr;
    } else {
        e->result_keys = e->keys[e->cur];
        e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    }
    return RSX_OK;
}

// Small sorts are launch-bound (about 30 launches of a few microseconds each): capture the
// whole pass loop once into a hipGraph and replay it.  A graph is keyed by everything its
// nodes baked in; replaying it also replays the chain's effect on the engine's buffer names.
constexpr uint64_t kGraphMaxKeys = 1ull << 22;

template <typename Key>
int sort_chain(rsx_engine* e, const void* ext_keys, const uint32_t* ext_perm, uint64_t count)
{
#if RSX_TILE_THREADS == 256
    if (e->radix_bits == 8 && count > e->radix8_min_keys) {
        const int rc8 = ensure_radix8<Key>(e);          // before any capture begins
        if (rc8 != RSX_OK) return rc8;
    }
#endif
    // (the legacy null stream cannot be captured: PyTorch's default stream is that one)
    const bool graphable = e->use_graph && e->profile == 0 && count > 0 && count <= kGraphMaxKeys && e->stream != nullptr && !e->final_keys_out;
    if (!graphable) return sort_chain_enqueue<Key>(e, ext_keys, ext_perm, count);
    GraphEntry key{};
    key.in = ext_keys;
    key.pin = ext_perm;
    key.n = count;
    key.cur = e->cur;
    key.first = e->first_pass;
    key.last = e->last_pass;
    key.flags = (e->lookahead ? 1 : 0) | (e->xcd_remap ? 2 : 0) | (e->small_scan ? 16 : 0);
    key.options_epoch = e->options_epoch;
    key.stream = e->stream;
    for (const GraphEntry& g : e->graphs) {
        if (g.in == key.in && g.pin == key.pin && g.n == key.n && g.cur == key.cur && g.first == key.first && g.last == key.last &&
            g.flags == key.flags && g.options_epoch == key.options_epoch && g.stream == key.stream) {
            RSX_TRY(hipGraphLaunch(g.exec, e->stream), RSX_CALCULATION_FAILED);
            e->cur = g.end_cur;
            e->last_in = g.end_last_in;
            e->last_shift = g.end_last_shift;
            e->result_external = false;
            e->result_keys = e->keys[e->cur];
            e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
            return RSX_OK;
        }
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        e->use_graph = 0;                        // this stream cannot be captured: stay eager from now on
        return sort_chain_enqueue<Key>(e, ext_keys, ext_perm, count);
    }
    const int rc = sort_chain_enqueue<Key>(e, ext_keys, ext_perm, count);
    const hipError_t end = hipStreamEndCapture(e->stream, &graph);
    if (rc != RSX_OK || end != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc != RSX_OK ? rc : fail(RSX_CALCULATION_FAILED, "hipStreamEndCapture", end);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (inst != hipSuccess) return fail(RSX_CALCULATION_FAILED, "hipGraphInstantiate", inst);
    key.exec = exec;
    key.end_cur = e->cur;
    key.end_last_in = e->last_in;
    key.end_last_shift = e->last_shift;
    if (e->graphs.size() >= 8) {                 // tiny cache: drop the oldest
        (void)hipGraphExecDestroy(e->graphs.front().exec);
        e->graphs.erase(e->graphs.begin());
    }
    e->graphs.push_back(key);
    RSX_TRY(hipGraphLaunch(exec, e->stream), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

// a fused-scan workgroup whose poll ran out leaves a flag: surfaced at the host's next synchronisation point
// (the flag is a word of mapped pinned host memory that the kernel stores to at system scope: reading it costs no copy and no
// synchronisation, so it is also looked at by the asynchronous calls — there it reports a time-out of work that has already
// finished; the synchronising calls look after their hipStreamSynchronize.)  Reported once, then cleared: the engine stays usable.
int check_scan_timeout(rsx_engine* e, int status)
{
    if (!e->scan_timeout_host) return RSX_OK;
    volatile uint32_t* flag = e->scan_timeout_host;
    if (*flag != 0) {
        *flag = 0;
        return fail(status, "the fused table scan timed out waiting for a group sum: the result of that sort is undefined (reported once; the engine remains usable)");
    }
    return RSX_OK;
}
bool aligned16(const void* p)
{
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

bool overlaps(const void* a, uint64_t abytes, const void* b, uint64_t bbytes)
{
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), b0 = reinterpret_cast<uintptr_t>(b);
    return a && b && abytes && bbytes && a0 < b0 + bbytes && b0 < a0 + abytes;
}

// How a caller's device range relates to one of the engine's two ping-pong buffers:
// 0 = disjoint from both, 1 = starts exactly at buffer *which (the pointer rsx_result_device hands out), -1 = any other overlap.
int alias_of(const void* p, uint64_t bytes, void* const bufs[2], uint64_t buf_bytes, int* which)
{
    for (int i = 0; i < 2; ++i) {
        if (!bufs[i] || !overlaps(p, bytes, bufs[i], buf_bytes)) continue;
        if (p == bufs[i]) {
            *which = i;
            return 1;
        }
        return -1;
    }
    return 0;
}

int bind_device(const rsx_engine* e, int status)
{
    RSX_TRY(hipSetDevice(e->device), status);
    return RSX_OK;
}

// ---- large-buffer allocation ----------------------------------------------------------------------------------------------
#ifdef RSX_EXPERIMENTS
hipError_t big_alloc(rsx_engine* e, void** out, size_t bytes);       // capi_experiments.inc: optionally VMM chunks mapped in a shuffled order
bool big_free(rsx_engine* e, void* p);
#else
hipError_t big_alloc(rsx_engine*, void** out, size_t bytes)
{
    *out = nullptr;
    return hipMalloc(out, bytes);
}

bool big_free(rsx_engine*, void* p)
{
    return !p || hipFree(p) == hipSuccess;
}
#endif

#ifdef RSX_EXPERIMENTS
#include "capi_experiments.inc"
#endif

// ---- environment --------------------------------------------------------------------------------------------------------------
// Every environment variable the library reads, parsed ONCE per engine at rsx_create (rsx_set_option overrides afterwards where an
// option exists).  They select between code paths that give identical results; defaults are the measured policies.
struct EnvKnob {
    const char* name;
    const char* meaning;
    void (*apply)(rsx_engine* e, const char* value);
};
constexpr long kb_or_policy(const char* v, int max_kb)
{
    return std::atoi(v) < 0 ? -1L : static_cast<long>(std::min(max_kb, std::atoi(v))) * 1024;
}
const EnvKnob kEnvKnobs[] = {
    {"RSX_XCD_REMAP", "0: workgroup b takes tile b (default 1: XCD-contiguous tile ranges)", [](rsx_engine* e, const char* v) { e->xcd_remap = std::atoi(v) != 0; }},
    {"RSX_XCD_PHASE", "tiles by which consecutive XCDs enter their range staggered (-1: an eighth of a range)", [](rsx_engine* e, const char* v) { e->xcd_phase = std::atoll(v); }},
    {"RSX_LOOKAHEAD", "0: every pass runs histogram -> scan -> paste -> reorder (default 1: RSX_OPT_LOOKAHEAD)", [](rsx_engine* e, const char* v) { e->lookahead = std::atoi(v) != 0; }},
    {"RSX_GRAPH", "1: small sorts replay a captured hipGraph (RSX_OPT_GRAPH, default 0)", [](rsx_engine* e, const char* v) { e->use_graph = std::atoi(v) != 0; }},
    {"RSX_SMALL_SCAN", "0: no one-workgroup table scan (RSX_OPT_SMALL_SCAN)", [](rsx_engine* e, const char* v) { e->small_scan = std::atoi(v) != 0; }},
    {"RSX_TILE_SORT", "0: no one-launch sort of inputs up to one tile (RSX_OPT_TILE_SORT)", [](rsx_engine* e, const char* v) { e->tile_sort = std::atoi(v) != 0; }},
    {"RSX_SELF_SCAN", "0: small tables get scan launches (RSX_OPT_SELF_SCAN)", [](rsx_engine* e, const char* v) { e->self_scan = std::atoi(v) != 0; }},
    {"RSX_SELF_SCAN_MAX", "largest table, in tiles, of the self-scan chain (RSX_OPT_SELF_SCAN_MAX_TILES)",
     [](rsx_engine* e, const char* v) { e->self_scan_max = std::min<uint32_t>(static_cast<uint32_t>(std::atoi(v)), rsx::kSelfScanMaxTiles); }},
    {"RSX_SMALL_TILE_MAX_KEYS", "self-scan sorts up to this many keys use 1024-key tiles (RSX_OPT_SMALL_TILE_MAX_KEYS)",
     [](rsx_engine* e, const char* v) { e->small_tile_max_keys = std::min<uint64_t>(std::strtoull(v, nullptr, 10), static_cast<uint64_t>(rsx::kSelfScanMaxTiles) * kTileThreads * kSmallKeysPerThread); }},
    {"RSX_PASTE_SCAN", "0: scan #2 gets a launch of its own in the two-launch table scan", [](rsx_engine* e, const char* v) { e->paste_scan = std::atoi(v) != 0; }},
    {"RSX_FUSED_SCAN", "0: never the one-launch table scan (RSX_OPT_FUSED_SCAN)", [](rsx_engine* e, const char* v) { e->fused_scan = std::atoi(v) != 0; }},
    {"RSX_RADIX_BITS", "8: byte-wide passes in the sort chain (RSX_OPT_RADIX_BITS)", [](rsx_engine* e, const char* v) { e->radix_bits = std::atoi(v) == 8 ? 8 : 4; }},
    {"RSX_RADIX8_MIN_KEYS", "8-bit passes only above this many keys (default 2^19)", [](rsx_engine* e, const char* v) { e->radix8_min_keys = std::max<uint64_t>(std::strtoull(v, nullptr, 10), kTileKeys); }},
    {"RSX_REORDER_EXTRA_LDS_KB", "unused dynamic LDS per 4-bit scatter workgroup = fewer per CU (-1: per-variant policy)", [](rsx_engine* e, const char* v) { e->reorder_extra_lds = kb_or_policy(v, 64); }},
    {"RSX_REORDER_WIDE", "4-bit scatter of 64-bit keys + payload on 512 x 8 (0 / 1, -1: policy = on)", [](rsx_engine* e, const char* v) { e->reorder_wide = std::max(-1, std::min(1, std::atoi(v))); }},
    {"RSX_R8_EXTRA_LDS_KB", "the same for the 8-bit scatter (-1: per-variant policy)", [](rsx_engine* e, const char* v) { e->r8_extra_lds = kb_or_policy(v, 96); }},
    {"RSX_R8_PACKED", "0: uint32 key and payload travel apart through the 8-bit scatter (default 1: one 64-bit element)", [](rsx_engine* e, const char* v) { e->r8_packed = std::atoi(v) != 0; }},
    {"RSX_R8_WIDE", "8-bit scatter on 512 x 8 (0 / 1, -1: policy = 64-bit keys without payload)", [](rsx_engine* e, const char* v) { e->r8_wide = std::max(-1, std::min(1, std::atoi(v))); }},
#ifdef RSX_EXPERIMENTS
    {"RSX_ALLOC_MODE", "experiments: key buffers as VMM chunks mapped in creation (1) or shuffled (2) order", [](rsx_engine* e, const char* v) { e->alloc_mode = std::max(0, std::min(2, std::atoi(v))); }},
    {"RSX_ALLOC_CHUNK_MB", "experiments: chunk size of RSX_ALLOC_MODE", [](rsx_engine* e, const char* v) { e->alloc_chunk = static_cast<size_t>(std::max(0, std::atoi(v))) << 20; }},
#endif
};
// (RSX_FUSED_SCAN_MAX_GROUPS is applied where the occupancy query has produced its bound; RSX_DEBUG_PTRS prints the buffer addresses)

void apply_environment(rsx_engine* e)
{
    for (const EnvKnob& k : kEnvKnobs) {
        if (const char* v = std::getenv(k.name)) k.apply(e, v);
    }
}

}  // namespace

#define RSX_BY_KEY(e, call32, call64) ((e)->key_bytes == 4 ? (call32) : (call64))

extern "C" {

const char* rsx_last_error(void)
{
    return g_last_error.c_str();
}

const char* rsx_version(void)
{
    return "radixsort_hip 0.1 (gfx950; tile 256x16; 4-bit LSD)";
}

int rsx_device_count(int* count)
{
    if (!count) return fail(RSX_INITIALIZATION_FAILED, "rsx_device_count: null argument");
    *count = 0;
    int c = 0;
    const hipError_t err = hipGetDeviceCount(&c);
    if (err != hipSuccess) {
        return fail(RSX_INITIALIZATION_FAILED, "hipGetDeviceCount", err);
    }
    *count = c;
    return RSX_OK;
}

int rsx_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(RSX_INITIALIZATION_FAILED, "rsx_device_name: null buffer");
    hipDeviceProp_t prop;
    RSX_TRY(hipGetDeviceProperties(&prop, device), RSX_INITIALIZATION_FAILED);
    std::snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return RSX_OK;
}

int rsx_create(rsx_engine** out, int device, int key_bytes, int is_signed, int has_payload, uint64_t capacity)
{
    if (!out) return fail(RSX_INITIALIZATION_FAILED, "rsx_create: null out pointer");
    *out = nullptr;
    if (key_bytes != 4 && key_bytes != 8) return fail(RSX_INITIALIZATION_FAILED, "rsx_create: key_bytes must be 4 or 8");
    if (capacity == 0 || capacity > 0xFFFFFFFFull) {
        return fail(RSX_RESIZE_FAILED, "rsx_create: capacity must be in [1, 2^32-1] (32-bit slots)");
    }
    int count = 0;
    if (rsx_device_count(&count) != RSX_OK || count <= 0) {
        return fail(RSX_INITIALIZATION_FAILED, "rsx_create: no HIP device (this library has no CPU fallback)");
    }
    if (device < 0 || device >= count) return fail(RSX_INITIALIZATION_FAILED, "rsx_create: device ordinal out of range");

    rsx_engine* e = new (std::nothrow) rsx_engine();
    if (!e) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_create: out of host memory");
    e->device = device;
    e->key_bytes = key_bytes;
    e->is_signed = is_signed != 0;
    e->has_payload = has_payload != 0;
    e->capacity = capacity;
    e->n = 0;
    e->last_pass = static_cast<int>(e->passes());
    for (auto& s : e->stats) stat_reset(s);
    apply_environment(e);

    auto bail = [&](int status, const char* what, hipError_t err) {
        rsx_destroy(e);
        return fail(status, what, err);
    };
    hipError_t err;
    if ((err = hipSetDevice(device)) != hipSuccess) return bail(RSX_INITIALIZATION_FAILED, "hipSetDevice", err);
    if ((err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipStreamCreate", err);
    e->own_stream = true;

    const size_t key_buf = static_cast<size_t>(capacity) * key_bytes;
    for (int i = 0; i < 2; ++i) {
        if ((err = big_alloc(e, &e->keys[i], key_buf)) != hipSuccess) return bail(RSX_INITIALIZATION_FAILED, "allocation of a key buffer", err);
        if (e->has_payload) {
            if ((err = big_alloc(e, reinterpret_cast<void**>(&e->perm[i]), static_cast<size_t>(capacity) * 4)) != hipSuccess)
                return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(permutations)", err);
        }
    }
    e->table_cap = static_cast<uint64_t>(RSX_RADIX) * e->ntiles(capacity);
    if ((e->ntiles(capacity) + rsx::kScanTiles - 1) / rsx::kScanTiles > rsx::kMaxScanGroups) {
        return bail(RSX_RESIZE_FAILED, "rsx_create: capacity exceeds the two-level table scan", hipSuccess);
    }
    // rounded up to whole scan groups (keeps 16-byte row accesses of the last group in bounds)
    const size_t table_alloc = ((e->ntiles(capacity) + rsx::kScanTiles - 1) / rsx::kScanTiles) * rsx::kScanTiles * RSX_RADIX * 4;
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->table), table_alloc)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(histograms)", err);
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->counts_next), table_alloc)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(look-ahead counts)", err);
#ifdef RSX_EXPERIMENTS
    {
        // the inline-scan chain serves tables of at most kFusedScanMaxGroups groups: its second count buffer need not be larger
        const size_t second = std::min<size_t>(table_alloc, static_cast<size_t>(rsx::kFusedScanMaxGroups) * rsx::kScanTiles * RSX_RADIX * 4);
        if ((err = hipMalloc(reinterpret_cast<void**>(&e->counts_next2), second)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(second look-ahead counts)", err);
        if ((err = hipMemsetAsync(e->counts_next2, 0, second, e->stream)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipMemsetAsync(second look-ahead counts)", err);
        if ((err = hipMalloc(reinterpret_cast<void**>(&e->scan_ready), rsx::kFusedScanMaxGroups * 4)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(scan ready words)", err);
        if ((err = hipMemsetAsync(e->scan_ready, 0, rsx::kFusedScanMaxGroups * 4, e->stream)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipMemsetAsync(scan ready words)", err);
    }
#endif
#ifdef RSX_STAMPS
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->stamps), e->ntiles(capacity) * 16 * 8)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(stamps)", err);
    if (const char* env = std::getenv("RSX_STAMP_PASS")) e->stamp_pass = std::atoi(env);
#endif
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->globsum), rsx::kMaxScanBlocks * 4)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(globsum)", err);
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->globsum2), rsx::kMaxScanBlocks * 4)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(globsum2)", err);
    e->globsum_live = e->globsum;
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->temp), 64)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(temp)", err);
    for (int i = 0; i < 3; ++i) {
        if ((err = hipMalloc(reinterpret_cast<void**>(&e->cnt3[i]), rsx::kSelfScanMaxTiles * RSX_RADIX * 4)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(self-scan counts)", err);
    }
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->gsums), rsx::kFusedScanMaxGroups * RSX_RADIX * 8)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(group sums)", err);
    if ((err = hipMemsetAsync(e->gsums, 0, rsx::kFusedScanMaxGroups * RSX_RADIX * 8, e->stream)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMemsetAsync(group sums)", err);
    if ((err = hipHostMalloc(reinterpret_cast<void**>(&e->scan_timeout_host), 64, hipHostMallocMapped)) != hipSuccess)
        return bail(RSX_HOST_BUFFERS_FAILED, "hipHostMalloc(scan timeout flag)", err);
    e->scan_timeout_host[0] = 0;
    if ((err = hipHostGetDevicePointer(reinterpret_cast<void**>(&e->scan_timeout), e->scan_timeout_host, 0)) != hipSuccess)
        return bail(RSX_HOST_BUFFERS_FAILED, "hipHostGetDevicePointer(scan timeout flag)", err);
    {
        // Co-residency of the fused scan: its workgroups wait for each other inside one launch, so a table may take it only
        // if the device holds the whole grid at once.  The occupancy query is advisory (ROCm 7.2 over-reports by one
        // workgroup per CU for SGPR-heavy kernels) and other work may hold CUs: half of its answer is the limit, which also
        // leaves room for a second engine scanning on another stream.  On a partitioned device (CPX: 32 CUs) the same
        // arithmetic gives a correspondingly smaller limit; RSX_OPT_FUSED_SCAN_MAX_GROUPS / env RSX_FUSED_SCAN_MAX_GROUPS override it.
        int per_cu = 0, cus = 0;
        if ((err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&rsx::scan_fused_kernel<true, true>), rsx::kScanTiles, 0)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipOccupancyMaxActiveBlocksPerMultiprocessor(scan_fused_kernel)", err);
        if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess)
            return bail(RSX_INITIALIZATION_FAILED, "hipDeviceGetAttribute(multiProcessorCount)", err);
        e->num_cus = cus;
        e->fused_scan_resident = static_cast<uint32_t>(std::max(per_cu, 0)) * static_cast<uint32_t>(std::max(cus, 0));
        e->fused_scan_limit = std::min<uint32_t>(e->fused_scan_resident / 2, rsx::kFusedScanMaxGroups);
        if (const char* env = std::getenv("RSX_FUSED_SCAN_MAX_GROUPS"))
            e->fused_scan_limit = std::min<uint32_t>(static_cast<uint32_t>(std::max(0, std::atoi(env))), std::min<uint32_t>(e->fused_scan_resident, rsx::kFusedScanMaxGroups));
    }
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->ref_table), rsx::kRefTable * 4)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(ref table)", err);
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->ref_globsum), rsx::kRefSplit * 4)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(ref globsum)", err);
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->range_dev), kRangeBlocks * 16)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(range)", err);
    if ((err = hipHostMalloc(reinterpret_cast<void**>(&e->range_host), kRangeBlocks * 16, hipHostMallocDefault)) != hipSuccess)
        return bail(RSX_HOST_BUFFERS_FAILED, "hipHostMalloc(range)", err);
    if ((err = hipMalloc(reinterpret_cast<void**>(&e->starts_dev), RSX_RADIX * 4)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMalloc(starts)", err);
    if ((err = hipHostMalloc(reinterpret_cast<void**>(&e->starts_host), RSX_RADIX * 4, hipHostMallocDefault)) != hipSuccess)
        return bail(RSX_HOST_BUFFERS_FAILED, "hipHostMalloc(starts)", err);
    if ((err = hipMemsetAsync(e->globsum, 0, rsx::kMaxScanBlocks * 4, e->stream)) != hipSuccess)
        return bail(RSX_INITIALIZATION_FAILED, "hipMemsetAsync(globsum)", err);

    int rc = lds_base_probe(e);
#if RSX_PRODUCT_SHAPE
    if (rc == RSX_OK) {
        auto allow_tile = [](const void* fn, size_t bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)); };
        using L32 = rsx::TileSortLayout<uint32_t, kTileThreads, kKeysPerThread>;
        using L64 = rsx::TileSortLayout<uint64_t, kTileThreads, kKeysPerThread>;
        if (allow_tile(reinterpret_cast<const void*>(&rsx::tile_sort_kernel<uint32_t, kTileThreads, kKeysPerThread, false>), L32::BYTES) != hipSuccess ||
            allow_tile(reinterpret_cast<const void*>(&rsx::tile_sort_kernel<uint32_t, kTileThreads, kKeysPerThread, true>), L32::BYTES) != hipSuccess ||
            allow_tile(reinterpret_cast<const void*>(&rsx::tile_sort_kernel<uint64_t, kTileThreads, kKeysPerThread, false>), L64::BYTES) != hipSuccess ||
            allow_tile(reinterpret_cast<const void*>(&rsx::tile_sort_kernel<uint64_t, kTileThreads, kKeysPerThread, true>), L64::BYTES) != hipSuccess)
            rc = fail(RSX_INITIALIZATION_FAILED, "hipFuncSetAttribute(tile_sort_kernel)");
    }
#endif
#if RSX_PRODUCT_SHAPE
    if (rc == RSX_OK) rc = allow_lds<uint64_t, true, false, false, 8, 512>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, true, true, false, 8, 512>();
#endif
    if (rc == RSX_OK) rc = allow_lds<uint32_t, false, false>();
    if (rc == RSX_OK) rc = allow_lds<uint32_t, true, false>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, false, false>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, true, false>();
    if (rc == RSX_OK) rc = allow_lds<uint32_t, false, true>();
    if (rc == RSX_OK) rc = allow_lds<uint32_t, true, true>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, false, true>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, true, true>();
    if (rc == RSX_OK) rc = allow_lds<uint32_t, false, false, true>();
    if (rc == RSX_OK) rc = allow_lds<uint32_t, true, false, true>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, false, false, true>();
    if (rc == RSX_OK) rc = allow_lds<uint64_t, true, false, true>();
    if (rc != RSX_OK) {
        rsx_destroy(e);
        return rc;
    }
    e->result_keys = e->keys[0];
    e->result_perm = e->perm[0];
    if (std::getenv("RSX_DEBUG_PTRS")) std::fprintf(stderr, "[radixsort_hip] keys[0]=%p keys[1]=%p table=%p counts_next=%p\n", e->keys[0], e->keys[1], static_cast<void*>(e->table), static_cast<void*>(e->counts_next));
    *out = e;
    return RSX_OK;
}

int rsx_destroy(rsx_engine* e)
{
    if (!e) return RSX_OK;
    int status = RSX_OK;
    if (hipSetDevice(e->device) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (const EventPair& p : e->pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (hipEvent_t ev : e->pool) (void)hipEventDestroy(ev);
    for (const GraphEntry& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    for (int i = 0; i < 2; ++i) {
        if (!big_free(e, e->keys[i])) status = RSX_CLEANUP_FAILED;
        if (!big_free(e, e->perm[i])) status = RSX_CLEANUP_FAILED;
    }
    if (e->table && hipFree(e->table) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->counts_next && hipFree(e->counts_next) != hipSuccess) status = RSX_CLEANUP_FAILED;
#ifdef RSX_EXPERIMENTS
    if (e->counts_next2 && hipFree(e->counts_next2) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->scan_ready && hipFree(e->scan_ready) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->tickets8 && hipFree(e->tickets8) != hipSuccess) status = RSX_CLEANUP_FAILED;
#endif
    if (e->globsum && hipFree(e->globsum) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->globsum2 && hipFree(e->globsum2) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->temp && hipFree(e->temp) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->gsums && hipFree(e->gsums) != hipSuccess) status = RSX_CLEANUP_FAILED;
    for (int i = 0; i < 3; ++i) {
        if (e->cnt3[i] && hipFree(e->cnt3[i]) != hipSuccess) status = RSX_CLEANUP_FAILED;
    }
    if (e->counts8 && hipFree(e->counts8) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->table8 && hipFree(e->table8) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->gsum8 && hipFree(e->gsum8) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->csum8 && hipFree(e->csum8) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->cbase8 && hipFree(e->cbase8) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->scan_timeout_host && hipHostFree(e->scan_timeout_host) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->starts_dev && hipFree(e->starts_dev) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->msd_plan && hipFree(e->msd_plan) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->msd_starts && hipFree(e->msd_starts) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->msd_plan_host && hipHostFree(e->msd_plan_host) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->msd_event) (void)hipEventDestroy(e->msd_event);
    if (e->msd_scatter_event) (void)hipEventDestroy(e->msd_scatter_event);
    for (auto& kv : e->order_events) if (kv.second) (void)hipEventDestroy(kv.second);
    for (hipEvent_t ev : e->marks) if (ev) (void)hipEventDestroy(ev);
    if (e->range_dev && hipFree(e->range_dev) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->ref_table && hipFree(e->ref_table) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->ref_globsum && hipFree(e->ref_globsum) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->range_host && hipHostFree(e->range_host) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->starts_host && hipHostFree(e->starts_host) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->pipe.s_in) (void)hipStreamSynchronize(e->pipe.s_in);
    if (e->pipe.s_out) (void)hipStreamSynchronize(e->pipe.s_out);
    for (int i = 0; i < 2; ++i) {
        if (e->pipe.in[i] && hipFree(e->pipe.in[i]) != hipSuccess) status = RSX_CLEANUP_FAILED;
        if (e->pipe.out[i] && hipFree(e->pipe.out[i]) != hipSuccess) status = RSX_CLEANUP_FAILED;
        if (e->pipe.pin[i] && hipFree(e->pipe.pin[i]) != hipSuccess) status = RSX_CLEANUP_FAILED;
        if (e->pipe.pout[i] && hipFree(e->pipe.pout[i]) != hipSuccess) status = RSX_CLEANUP_FAILED;
        if (e->pipe.in_ready[i]) (void)hipEventDestroy(e->pipe.in_ready[i]);
        if (e->pipe.sorted[i]) (void)hipEventDestroy(e->pipe.sorted[i]);
        if (e->pipe.out_done[i]) (void)hipEventDestroy(e->pipe.out_done[i]);
    }
    if (e->pipe.s_in && hipStreamDestroy(e->pipe.s_in) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->pipe.s_out && hipStreamDestroy(e->pipe.s_out) != hipSuccess) status = RSX_CLEANUP_FAILED;
    if (e->own_stream && e->stream && hipStreamDestroy(e->stream) != hipSuccess) status = RSX_CLEANUP_FAILED;
    delete e;
    return status;
}

int rsx_set_stream(rsx_engine* e, void* hip_stream)
{
    if (!e) return fail(RSX_INITIALIZATION_FAILED, "rsx_set_stream: null engine");
    if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
    const int rc = drain_events(e);
    if (rc != RSX_OK) return rc;
    for (const GraphEntry& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
    if (e->own_stream && e->stream) {
        RSX_TRY(hipStreamDestroy(e->stream), RSX_INITIALIZATION_FAILED);
    }
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->own_stream = false;
    return RSX_OK;
}

int rsx_get_stream(const rsx_engine* e, void** hip_stream)
{
    if (!e || !hip_stream) return fail(RSX_INITIALIZATION_FAILED, "rsx_get_stream: null argument");
    *hip_stream = static_cast<void*>(e->stream);
    return RSX_OK;
}

int rsx_set_option(rsx_engine* e, int option, int64_t value)
{
    if (!e) return fail(RSX_INITIALIZATION_FAILED, "rsx_set_option: null engine");
    ++e->options_epoch;          // captured chains of the old settings are not replayed (sort_chain)
    switch (option) {
    case RSX_OPT_PROFILE: e->profile = value < 0 || value > 2 ? 1 : static_cast<int>(value); return RSX_OK;
    case RSX_OPT_XCD_REMAP: e->xcd_remap = value != 0; return RSX_OK;
    case RSX_OPT_LOOKAHEAD: e->lookahead = value != 0; return RSX_OK;
    case RSX_OPT_REF_DIAGNOSTICS: e->ref_diag = value != 0; return RSX_OK;
    case RSX_OPT_GRAPH: e->use_graph = value != 0; return RSX_OK;
    case RSX_OPT_SMALL_SCAN: e->small_scan = value != 0; return RSX_OK;
    case RSX_OPT_TILE_SORT: e->tile_sort = value != 0; return RSX_OK;
    case RSX_OPT_FUSED_SCAN: e->fused_scan = value != 0; return RSX_OK;
#ifdef RSX_EXPERIMENTS
    case RSX_XOPT_REORDER8_KERNEL:
        if (value < 1 || value > 3) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: the 8-bit scatter kernel is 1, 2 or 3");
        e->reorder8_version = static_cast<int>(value);
        return RSX_OK;
    case RSX_XOPT_REORDER8_STAY:
        if (value < -1 || value > 8) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: workgroups per CU of the staying 8-bit scatter: -1 / 0 (off) .. 8");
        e->r8_stay = static_cast<int>(value);
        return RSX_OK;
    case RSX_XOPT_INLINE_SCAN:
        e->inline_scan = value != 0;
        if (e->inline_scan && e->inline_scan_limit == 0) {
            if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
            return RSX_BY_KEY(e, inline_scan_prepare<uint32_t>(e), inline_scan_prepare<uint64_t>(e));
        }
        return RSX_OK;
    case RSX_XOPT_INLINE_SCAN_MAX_GROUPS:
        if (value < 0) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: negative group count");
        e->inline_scan_max_groups = static_cast<uint32_t>(std::min<int64_t>(value, rsx::kFusedScanMaxGroups));
        return RSX_OK;
    case RSX_XOPT_DEBUG_RAISE_SCAN_TIMEOUT:
        // tests only: stores to the time-out word exactly as a fused-scan workgroup whose poll ran out does
        if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
        hipLaunchKernelGGL(rsx::raise_flag_kernel, dim3(1), dim3(64), 0, e->stream, e->scan_timeout);
        RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
        return RSX_OK;
#endif
    case RSX_OPT_FUSED_SCAN_MAX_GROUPS:
        // never beyond what the occupancy query says is resident at once (nor the granule buffer): -1 restores the default
        if (value < -1) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: negative group count");
        e->fused_scan_limit = value < 0 ? std::min<uint32_t>(e->fused_scan_resident / 2, rsx::kFusedScanMaxGroups)
                                        : static_cast<uint32_t>(std::min<int64_t>(value, std::min<uint32_t>(e->fused_scan_resident, rsx::kFusedScanMaxGroups)));
        return RSX_OK;
    case RSX_OPT_SELF_SCAN: e->self_scan = value != 0; return RSX_OK;
    case RSX_OPT_SMALL_TILE_MAX_KEYS:
        if (value < 0) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: negative key count");
        e->small_tile_max_keys = std::min<uint64_t>(static_cast<uint64_t>(value), static_cast<uint64_t>(rsx::kSelfScanMaxTiles) * kTileThreads * kSmallKeysPerThread);
        return RSX_OK;
    case RSX_OPT_SELF_SCAN_MAX_TILES:
        if (value < 0) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: negative tile count");
        e->self_scan_max = static_cast<uint32_t>(std::min<int64_t>(value, rsx::kSelfScanMaxTiles));
        return RSX_OK;
    case RSX_OPT_XCD_PHASE:
        if (value < -1 || value > (1 << 22)) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: XCD phase out of range");
        e->xcd_phase = value;
        return RSX_OK;
    case RSX_OPT_RADIX_BITS:
        if (value != 4 && value != 8) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: digit width must be 4 or 8 bits");
        e->radix_bits = static_cast<int>(value);
#if RSX_TILE_THREADS == 256
        if (value == 8 && e->capacity > e->radix8_min_keys) {
            if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
            return RSX_BY_KEY(e, ensure_radix8<uint32_t>(e), ensure_radix8<uint64_t>(e));
        }
#endif
        return RSX_OK;
    case RSX_OPT_FIRST_PASS:
        if (value < 0 || value > e->passes()) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: first pass out of range");
        e->first_pass = static_cast<int>(value);
        return RSX_OK;
    case RSX_OPT_LAST_PASS:
        if (value < 0 || value > e->passes()) return fail(RSX_CALCULATION_FAILED, "rsx_set_option: last pass out of range");
        e->last_pass = static_cast<int>(value);
        return RSX_OK;
    default: return fail(RSX_INITIALIZATION_FAILED, "rsx_set_option: unknown option");
    }
}

int rsx_get_geometry(const rsx_engine* e, rsx_geometry* out)
{
    if (!e || !out) return fail(RSX_INITIALIZATION_FAILED, "rsx_get_geometry: null argument");
    out->tile_threads = kTileThreads;
    out->keys_per_thread = kKeysPerThread;
    out->tile_keys = kTileKeys;
    out->scan_block = rsx::kScanBlock;
    out->num_keys = e->n;
    out->capacity = e->capacity;
    out->num_tiles = e->ntiles(e->n);
    out->table_len = static_cast<uint64_t>(RSX_RADIX) * out->num_tiles;
    out->num_scan_blocks = static_cast<uint64_t>(RSX_RADIX) * ((out->num_tiles + rsx::kScanTiles - 1) / rsx::kScanTiles);
    out->num_passes = e->passes();
    out->key_bytes = static_cast<uint32_t>(e->key_bytes);
    out->fused_scan_resident = e->fused_scan_resident;
    out->fused_scan_max_groups = e->fused_scan_limit;
    return RSX_OK;
}

int rsx_resize(rsx_engine* e, uint64_t num_keys)
{
    if (!e) return fail(RSX_RESIZE_FAILED, "rsx_resize: null engine");
    if (num_keys > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_resize: beyond capacity");
    e->n = num_keys;
    return RSX_OK;
}

int rsx_upload(rsx_engine* e, const void* host_keys, const uint32_t* host_perm, uint64_t n)
{
    if (!e) return fail(RSX_DATA_UPLOAD_FAILED, "rsx_upload: null engine");
    if (n > e->capacity) return fail(RSX_DATA_UPLOAD_FAILED, "rsx_upload: beyond capacity");
    if (n > 0 && !host_keys) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_upload: null host keys");
    if (n > 0 && e->has_payload && !host_perm) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_upload: payload engine needs a permutation buffer");
    if (bind_device(e, RSX_DATA_UPLOAD_FAILED) != RSX_OK) return RSX_DATA_UPLOAD_FAILED;
    e->n = n;
    if (n > 0) {
        RSX_TRY(hipMemcpyAsync(e->keys[e->cur], host_keys, static_cast<size_t>(n) * e->key_bytes, hipMemcpyHostToDevice, e->stream),
                RSX_DATA_UPLOAD_FAILED);
        if (e->has_payload) {
            RSX_TRY(hipMemcpyAsync(e->perm[e->cur], host_perm, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, e->stream),
                    RSX_DATA_UPLOAD_FAILED);
        }
    }
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_DATA_UPLOAD_FAILED);   // "wait until end of write" (RadixSortGPU.cpp:305)
    e->result_external = false;
    e->result_keys = e->keys[e->cur];
    e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    return RSX_OK;
}

int rsx_fill_pad(rsx_engine* e, uint64_t byte_offset)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_fill_pad: null engine");
    const uint64_t first = byte_offset / e->key_bytes;
    if (byte_offset % e->key_bytes != 0 || first > e->n) return fail(RSX_CALCULATION_FAILED, "rsx_fill_pad: bad offset");
    const uint64_t count = e->n - first;
    if (count == 0) return RSX_OK;
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((count + 255) / 256, 2048));
    if (e->key_bytes == 4) {
        // numeric_limits<T>::max() - 1 (RadixSortGPU.cpp:274-276)
        const uint32_t v = e->is_signed ? 0x7FFFFFFEu : 0xFFFFFFFEu;
        hipLaunchKernelGGL(rsx::fill_kernel<uint32_t>, dim3(blocks), dim3(256), 0, e->stream, static_cast<uint32_t*>(e->keys[e->cur]),
                           first, count, v);
    } else {
        const uint64_t v = e->is_signed ? 0x7FFFFFFFFFFFFFFEull : 0xFFFFFFFFFFFFFFFEull;
        hipLaunchKernelGGL(rsx::fill_kernel<uint64_t>, dim3(blocks), dim3(256), 0, e->stream, static_cast<uint64_t*>(e->keys[e->cur]),
                           first, count, v);
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    return RSX_OK;
}

int rsx_download(rsx_engine* e, void* host_keys_out, uint32_t* host_perm_out, uint32_t* hist_out, uint64_t hist_cap,
                 uint32_t* globsum_out, uint64_t globsum_cap)
{
    if (!e) return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_download: null engine");
    if (bind_device(e, RSX_DATA_DOWNLOAD_FAILED) != RSX_OK) return RSX_DATA_DOWNLOAD_FAILED;
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_DATA_DOWNLOAD_FAILED);
    if (check_scan_timeout(e, RSX_DATA_DOWNLOAD_FAILED) != RSX_OK) return RSX_DATA_DOWNLOAD_FAILED;
    if (e->result_external && e->n > 0 && (host_keys_out || host_perm_out))
        return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_download: the last sort (rsx_sort_from_to) wrote into the caller's buffer; the engine holds no result");
    if (e->n > 0 && host_keys_out) {
        RSX_TRY(hipMemcpyAsync(host_keys_out, e->result_keys, static_cast<size_t>(e->n) * e->key_bytes, hipMemcpyDeviceToHost, e->stream),
                RSX_DATA_DOWNLOAD_FAILED);
    }
    if (e->n > 0 && host_perm_out && e->has_payload) {
        RSX_TRY(hipMemcpyAsync(host_perm_out, e->result_perm, static_cast<size_t>(e->n) * 4, hipMemcpyDeviceToHost, e->stream),
                RSX_DATA_DOWNLOAD_FAILED);
    }
    if (e->ref_diag && ((hist_out && hist_cap) || (globsum_out && globsum_cap))) {
        // the reference's m_hHistograms / m_hGlobsum: recomputed in its own geometry from the last pass's input
        if (e->n == 0 || e->n % rsx::kRefVps != 0 || !e->last_in)
            return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_download: reference-geometry diagnostics (RSX_OPT_REF_DIAGNOSTICS) need a finished 4-BIT pass over a multiple of 1024 keys as the last pass of the sort");
        if (e->key_bytes == 4) {
            hipLaunchKernelGGL(rsx::ref_histogram_kernel<uint32_t>, dim3(rsx::kRefVps), dim3(256), 0, e->stream,
                               static_cast<const uint32_t*>(e->last_in), e->ref_table, e->n, e->last_shift, flip_mask<uint32_t>(e));
        } else {
            hipLaunchKernelGGL(rsx::ref_histogram_kernel<uint64_t>, dim3(rsx::kRefVps), dim3(256), 0, e->stream,
                               static_cast<const uint64_t*>(e->last_in), e->ref_table, e->n, e->last_shift, flip_mask<uint64_t>(e));
        }
        hipLaunchKernelGGL(rsx::ref_scan_kernel, dim3(1), dim3(1024), 0, e->stream, e->ref_table, e->ref_globsum);
        RSX_TRY(hipGetLastError(), RSX_DATA_DOWNLOAD_FAILED);
        if (hist_out && hist_cap) {
            RSX_TRY(hipMemcpyAsync(hist_out, e->ref_table, std::min<uint64_t>(hist_cap, rsx::kRefTable) * 4, hipMemcpyDeviceToHost, e->stream),
                    RSX_DATA_DOWNLOAD_FAILED);
        }
        if (globsum_out && globsum_cap) {
            RSX_TRY(hipMemcpyAsync(globsum_out, e->ref_globsum, std::min<uint64_t>(globsum_cap, rsx::kRefSplit) * 4, hipMemcpyDeviceToHost, e->stream),
                    RSX_DATA_DOWNLOAD_FAILED);
        }
        RSX_TRY(hipStreamSynchronize(e->stream), RSX_DATA_DOWNLOAD_FAILED);
        return RSX_OK;
    }
    // The engine's own table exists only after a pass in the 4096-key [digit][tile] geometry (the step API, the chains with scan
    // launches, the 4096-key self-scan chain, the one-workgroup sort); sorts on 1024-key tiles and 8-bit passes leave none, and
    // handing out what an earlier sort left would be silently wrong.  RSX_OPT_REF_DIAGNOSTICS rebuilds its tables for any path.
    if (hist_out && hist_cap && e->n > 0 && !e->table_valid)
        return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_download: the last sort left no table in the engine's [digit][tile] geometry (tiles of 1024 keys, or 8-bit digits); with 4-bit digits use RSX_OPT_REF_DIAGNOSTICS (the reference-geometry tables, 4-bit passes only) or RSX_OPT_SMALL_TILE_MAX_KEYS = 0; 8-bit passes have no such table");
    if (globsum_out && globsum_cap && e->n > 0 && !e->globsum_valid)
        return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_download: the last sort's table scan produced no group sums (self-scan, one-workgroup scan, or 8-bit digits); with 4-bit digits use RSX_OPT_REF_DIAGNOSTICS (4-bit passes only) or the chain with scan launches; 8-bit passes have no group sums in this geometry");
    if (hist_out && hist_cap) {
        const uint64_t live = static_cast<uint64_t>(RSX_RADIX) * e->ntiles(e->n);
        const uint64_t take = std::min(hist_cap, live);
        if (take) {
            RSX_TRY(hipMemcpyAsync(hist_out, e->table, take * 4, hipMemcpyDeviceToHost, e->stream), RSX_DATA_DOWNLOAD_FAILED);
        }
    }
    if (globsum_out && globsum_cap) {
        const uint64_t take = std::min<uint64_t>(globsum_cap, rsx::kMaxScanBlocks);
        RSX_TRY(hipMemcpyAsync(globsum_out, e->globsum_live, take * 4, hipMemcpyDeviceToHost, e->stream), RSX_DATA_DOWNLOAD_FAILED);
    }
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_DATA_DOWNLOAD_FAILED);
    return RSX_OK;
}

int rsx_pin_host(rsx_engine* e, void* host_ptr, uint64_t bytes)
{
    if (!e || !host_ptr || bytes == 0) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_pin_host: null argument");
    if (bind_device(e, RSX_HOST_BUFFERS_FAILED) != RSX_OK) return RSX_HOST_BUFFERS_FAILED;
    RSX_TRY(hipHostRegister(host_ptr, static_cast<size_t>(bytes), hipHostRegisterMapped), RSX_HOST_BUFFERS_FAILED);
    return RSX_OK;
}

int rsx_unpin_host(rsx_engine* e, void* host_ptr)
{
    if (!e || !host_ptr) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_unpin_host: null argument");
    if (bind_device(e, RSX_HOST_BUFFERS_FAILED) != RSX_OK) return RSX_HOST_BUFFERS_FAILED;
    RSX_TRY(hipHostUnregister(host_ptr), RSX_HOST_BUFFERS_FAILED);
    return RSX_OK;
}

int rsx_host_device_pointer(rsx_engine* e, void* host_ptr, void** device_ptr)
{
    if (!e || !host_ptr || !device_ptr) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_host_device_pointer: null argument");
    if (bind_device(e, RSX_HOST_BUFFERS_FAILED) != RSX_OK) return RSX_HOST_BUFFERS_FAILED;
    RSX_TRY(hipHostGetDevicePointer(device_ptr, host_ptr, 0), RSX_HOST_BUFFERS_FAILED);
    return RSX_OK;
}

namespace {
int pipeline_init(rsx_engine* e)
{
    rsx_engine::Pipeline& p = e->pipe;
    if (p.ready) return RSX_OK;
    // (a call that failed half-way keeps what it got — rsx_destroy frees it — and the next one asks only for the rest)
    const size_t key_buf = static_cast<size_t>(e->capacity) * e->key_bytes;
    for (int i = 0; i < 2; ++i) {
        if (!p.in[i]) RSX_TRY(hipMalloc(&p.in[i], key_buf), RSX_INITIALIZATION_FAILED);
        if (!p.out[i]) RSX_TRY(hipMalloc(&p.out[i], key_buf), RSX_INITIALIZATION_FAILED);
        if (e->has_payload) {
            if (!p.pin[i]) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&p.pin[i]), static_cast<size_t>(e->capacity) * 4), RSX_INITIALIZATION_FAILED);
            if (!p.pout[i]) RSX_TRY(hipMalloc(reinterpret_cast<void**>(&p.pout[i]), static_cast<size_t>(e->capacity) * 4), RSX_INITIALIZATION_FAILED);
        }
        if (!p.in_ready[i]) RSX_TRY(hipEventCreateWithFlags(&p.in_ready[i], hipEventDisableTiming), RSX_INITIALIZATION_FAILED);
        if (!p.sorted[i]) RSX_TRY(hipEventCreateWithFlags(&p.sorted[i], hipEventDisableTiming), RSX_INITIALIZATION_FAILED);
        if (!p.out_done[i]) RSX_TRY(hipEventCreateWithFlags(&p.out_done[i], hipEventDisableTiming), RSX_INITIALIZATION_FAILED);
    }
    if (!p.s_in) RSX_TRY(hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking), RSX_INITIALIZATION_FAILED);
    if (!p.s_out) RSX_TRY(hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking), RSX_INITIALIZATION_FAILED);
    p.ready = true;
    return RSX_OK;
}
}  // namespace

int rsx_pipeline_submit(rsx_engine* e, const void* host_keys, const uint32_t* host_perm, uint64_t n, void* host_keys_out, uint32_t* host_perm_out)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_pipeline_submit: null engine");
    if (n == 0 || n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_pipeline_submit: key count must be in [1, capacity]");
    if (!host_keys || !host_keys_out) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_pipeline_submit: null host buffer");
    if (e->has_payload && (!host_perm || !host_perm_out)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_pipeline_submit: payload engine needs permutation buffers");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    int rc = pipeline_init(e);
    if (rc != RSX_OK) return rc;
    rsx_engine::Pipeline& p = e->pipe;
    const int s = static_cast<int>(p.submitted & 1u);
    const size_t kb = static_cast<size_t>(n) * e->key_bytes;
    // upload: the inbox is free once the job two submissions ago has been sorted out of it
    if (p.submitted >= 2) RSX_TRY(hipStreamWaitEvent(p.s_in, p.sorted[s], 0), RSX_DATA_UPLOAD_FAILED);
    RSX_TRY(hipMemcpyAsync(p.in[s], host_keys, kb, hipMemcpyHostToDevice, p.s_in), RSX_DATA_UPLOAD_FAILED);
    if (e->has_payload) RSX_TRY(hipMemcpyAsync(p.pin[s], host_perm, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, p.s_in), RSX_DATA_UPLOAD_FAILED);
    RSX_TRY(hipEventRecord(p.in_ready[s], p.s_in), RSX_DATA_UPLOAD_FAILED);
    // sort: behind its upload, and behind the download that last read this outbox
    RSX_TRY(hipStreamWaitEvent(e->stream, p.in_ready[s], 0), RSX_CALCULATION_FAILED);
    if (p.submitted >= 2) RSX_TRY(hipStreamWaitEvent(e->stream, p.out_done[s], 0), RSX_CALCULATION_FAILED);
    rc = rsx_sort_from_to(e, p.in[s], p.pin[s], n, 0, static_cast<int>(e->passes()), p.out[s], p.pout[s]);
    if (rc != RSX_OK) return rc;
    RSX_TRY(hipEventRecord(p.sorted[s], e->stream), RSX_CALCULATION_FAILED);
    // download
    RSX_TRY(hipStreamWaitEvent(p.s_out, p.sorted[s], 0), RSX_DATA_DOWNLOAD_FAILED);
    RSX_TRY(hipMemcpyAsync(host_keys_out, p.out[s], kb, hipMemcpyDeviceToHost, p.s_out), RSX_DATA_DOWNLOAD_FAILED);
    if (e->has_payload) RSX_TRY(hipMemcpyAsync(host_perm_out, p.pout[s], static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, p.s_out), RSX_DATA_DOWNLOAD_FAILED);
    RSX_TRY(hipEventRecord(p.out_done[s], p.s_out), RSX_DATA_DOWNLOAD_FAILED);
    p.submitted += 1;
    return RSX_OK;
}

int rsx_pipeline_wait(rsx_engine* e)
{
    if (!e) return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_pipeline_wait: null engine");
    if (!e->pipe.ready) return RSX_OK;
    if (bind_device(e, RSX_DATA_DOWNLOAD_FAILED) != RSX_OK) return RSX_DATA_DOWNLOAD_FAILED;
    RSX_TRY(hipStreamSynchronize(e->pipe.s_in), RSX_DATA_DOWNLOAD_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_DATA_DOWNLOAD_FAILED);
    RSX_TRY(hipStreamSynchronize(e->pipe.s_out), RSX_DATA_DOWNLOAD_FAILED);
    return check_scan_timeout(e, RSX_DATA_DOWNLOAD_FAILED);
}

int rsx_histogram(rsx_engine* e, int pass)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_histogram: null engine");
    if (pass < 0 || pass >= static_cast<int>(e->passes())) return fail(RSX_CALCULATION_FAILED, "rsx_histogram: pass out of range");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    return RSX_BY_KEY(e, launch_histogram<uint32_t>(e, e->keys[e->cur], e->n, pass * RSX_RADIX_BITS, RSX_RADIX - 1),
                      launch_histogram<uint64_t>(e, e->keys[e->cur], e->n, pass * RSX_RADIX_BITS, RSX_RADIX - 1));
}

int rsx_scan(rsx_engine* e)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_scan: null engine");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    return launch_scan(e, e->n);
}

int rsx_paste(rsx_engine* e)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_paste: null engine");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    return launch_paste(e, e->n);
}

int rsx_reorder(rsx_engine* e, int pass)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_reorder: null engine");
    if (pass < 0 || pass >= static_cast<int>(e->passes())) return fail(RSX_CALCULATION_FAILED, "rsx_reorder: pass out of range");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    const int src = e->cur, dst = e->cur ^ 1;
    const uint32_t* pin = e->has_payload ? e->perm[src] : nullptr;
    uint32_t* pout = e->has_payload ? e->perm[dst] : nullptr;
    const int rc = RSX_BY_KEY(e, launch_reorder<uint32_t>(e, e->keys[src], e->keys[dst], pin, pout, e->n, pass * RSX_RADIX_BITS, RSX_RADIX - 1),
                              launch_reorder<uint64_t>(e, e->keys[src], e->keys[dst], pin, pout, e->n, pass * RSX_RADIX_BITS, RSX_RADIX - 1));
    if (rc != RSX_OK) return rc;
    e->cur = dst;   // swap of the buffer names (RadixSortGPU.cpp:263-266)
    e->result_external = false;
    e->result_keys = e->keys[e->cur];
    e->result_perm = e->has_payload ? e->perm[e->cur] : nullptr;
    return RSX_OK;
}

int rsx_sort(rsx_engine* e)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_sort: null engine");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    return RSX_BY_KEY(e, sort_chain<uint32_t>(e, nullptr, nullptr, e->n), sort_chain<uint64_t>(e, nullptr, nullptr, e->n));
}

int rsx_sync(rsx_engine* e)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_sync: null engine");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    return check_scan_timeout(e, RSX_CALCULATION_FAILED);
}

int rsx_check_status(rsx_engine* e)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_check_status: null engine");
    return check_scan_timeout(e, RSX_CALCULATION_FAILED);
}

int rsx_sort_from(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_sort_from: null engine");
    if (n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_sort_from: beyond capacity");
    if (n > 0 && (!d_keys || !aligned16(d_keys))) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from: keys must be a 16-byte aligned device pointer");
    if (e->has_payload && n > 0 && (!d_payload || !aligned16(d_payload)))
        return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from: payload engine needs a 16-byte aligned payload pointer");
    if (e->first_pass >= e->last_pass) return fail(RSX_CALCULATION_FAILED, "rsx_sort_from: empty pass range");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    if (n == 0) {
        e->n = 0;
        return RSX_OK;
    }
    // The first pass of an external sort writes into the engine's own inputKeys buffer.  A caller who hands
    // back one of the engine's buffers (e.g. the pointer rsx_result_device returned) would have that pass read
    // and write the same memory: such input is sorted through the internal ping-pong instead; any other
    // overlap with the engine's buffers is refused.
    int wk = -1, wp = -1;
    void* const perm_bufs[2] = {e->perm[0], e->perm[1]};
    const int ak = alias_of(d_keys, n * static_cast<uint64_t>(e->key_bytes), e->keys, e->capacity * static_cast<uint64_t>(e->key_bytes), &wk);
    const int ap = e->has_payload ? alias_of(d_payload, n * 4, perm_bufs, e->capacity * 4, &wp) : 0;
    if (ak < 0 || ap < 0) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from: input overlaps the engine's own buffers (only their start addresses are accepted)");
    if (ak == 1 || ap == 1) {
        if (e->has_payload && !(ak == 1 && ap == 1 && wk == wp))
            return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from: keys and payload must name the same one of the engine's buffer pairs");
        e->cur = wk;
        d_keys = nullptr;            // internal ping-pong: reads keys[cur], never writes what it reads
        d_payload = nullptr;
    }
    e->n = n;
    return RSX_BY_KEY(e, sort_chain<uint32_t>(e, d_keys, d_payload, n), sort_chain<uint64_t>(e, d_keys, d_payload, n));
}

int rsx_sort_from_to(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, int first_pass, int last_pass, void* d_keys_out,
                     uint32_t* d_payload_out)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_sort_from_to: null engine");
    if (first_pass < 0 || last_pass > e->key_bytes * 2 || first_pass >= last_pass) return fail(RSX_CALCULATION_FAILED, "rsx_sort_from_to: pass range out of bounds");
    if (n > 0 && !d_keys_out) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from_to: no output buffer");
    if (e->has_payload && n > 0 && !d_payload_out) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from_to: payload engine needs a payload output buffer");
    for (int i = 0; i < 2 && n > 0; ++i) {
        if (overlaps(d_keys_out, n * static_cast<uint64_t>(e->key_bytes), e->keys[i], e->capacity * static_cast<uint64_t>(e->key_bytes)) ||
            (e->has_payload && overlaps(d_payload_out, n * 4, e->perm[i], e->capacity * 4)))
            return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from_to: the output overlaps the engine's own buffers");
    }
    if (n > 0 && (overlaps(d_keys_out, n * static_cast<uint64_t>(e->key_bytes), d_keys, n * static_cast<uint64_t>(e->key_bytes)) ||
                  (e->has_payload && overlaps(d_payload_out, n * 4, d_payload, n * 4))))
        return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sort_from_to: input and output overlap");
    const int saved_first = e->first_pass, saved_last = e->last_pass;
    e->first_pass = first_pass;
    e->last_pass = last_pass;
    e->final_keys_out = d_keys_out;
    e->final_perm_out = d_payload_out;
    const int rc = rsx_sort_from(e, d_keys, d_payload, n);
    e->final_keys_out = nullptr;
    e->final_perm_out = nullptr;
    e->first_pass = saved_first;
    e->last_pass = saved_last;
    return rc;
}

int rsx_partition(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, int shift, int bits, void* d_keys_out,
                  uint32_t* d_payload_out, uint64_t* bucket_offsets)
{
    if (!e || !bucket_offsets) return fail(RSX_CALCULATION_FAILED, "rsx_partition: null argument");
    if (bits < 0 || bits > RSX_RADIX_BITS || shift < 0 || shift + bits > e->key_bytes * 8)
        return fail(RSX_CALCULATION_FAILED, "rsx_partition: bit field out of range (at most 4 bits)");
    if (n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_partition: beyond capacity");
    const uint32_t buckets = 1u << bits;
    if (n > 0 && (!d_keys || !d_keys_out || !aligned16(d_keys) || !aligned16(d_keys_out)))
        return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition: key buffers must be 16-byte aligned device pointers");
    const bool with_payload = e->has_payload && d_payload && d_payload_out;
    if (e->has_payload && !with_payload && n > 0) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition: payload engine needs payload buffers");
    if (with_payload && !aligned16(d_payload)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition: the payload input must be 16-byte aligned (it is read 16 bytes per lane)");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    if (n == 0) {
        for (uint32_t d = 0; d <= buckets; ++d) bucket_offsets[d] = 0;
        return RSX_OK;
    }
    const uint32_t mask = buckets - 1;
    const uint32_t* pin = with_payload ? d_payload : nullptr;
    uint32_t* pout = with_payload ? d_payload_out : nullptr;
    const int rc = RSX_BY_KEY(e, run_pass<uint32_t>(e, d_keys, d_keys_out, pin, pout, n, shift, mask),
                              run_pass<uint64_t>(e, d_keys, d_keys_out, pin, pout, n, shift, mask));
    if (rc != RSX_OK) return rc;
    hipLaunchKernelGGL(rsx::bucket_starts_kernel, dim3(1), dim3(64), 0, e->stream, e->table, static_cast<uint32_t>(e->ntiles(n)), e->starts_dev);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->starts_host, e->starts_dev, RSX_RADIX * 4, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (uint32_t d = 0; d < buckets; ++d) bucket_offsets[d] = e->starts_host[d];
    bucket_offsets[buckets] = n;
    return RSX_OK;
}

int rsx_partition_count(rsx_engine* e, const void* d_keys, uint64_t n, int shift, int bits, uint64_t* bucket_counts)
{
    if (!e || !bucket_counts) return fail(RSX_CALCULATION_FAILED, "rsx_partition_count: null argument");
    if (bits < 0 || bits > RSX_RADIX_BITS || shift < 0 || shift + bits > e->key_bytes * 8)
        return fail(RSX_CALCULATION_FAILED, "rsx_partition_count: bit field out of range (at most 4 bits)");
    if (n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_partition_count: beyond capacity");
    const uint32_t buckets = 1u << bits;
    for (uint32_t d = 0; d < buckets; ++d) bucket_counts[d] = 0;
    if (n == 0) return RSX_OK;
    if (!d_keys || !aligned16(d_keys)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_count: keys must be a 16-byte aligned device pointer");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    const int rc = RSX_BY_KEY(e, launch_histogram<uint32_t>(e, d_keys, n, shift, buckets - 1), launch_histogram<uint64_t>(e, d_keys, n, shift, buckets - 1));
    if (rc != RSX_OK) return rc;
    hipLaunchKernelGGL(rsx::digit_totals_kernel, dim3(RSX_RADIX), dim3(256), 0, e->stream, e->table, static_cast<uint32_t>(e->ntiles(n)), e->range_dev);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->range_host, e->range_dev, RSX_RADIX * 8, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (uint32_t d = 0; d < buckets; ++d) bucket_counts[d] = e->range_host[d];
    e->counted_keys = d_keys;
    e->counted_n = n;
    e->counted_shift = shift;
    e->counted_bits = bits;
    return RSX_OK;
}

int rsx_partition_scatter(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, int shift, int bits,
                          void* d_keys_out, uint32_t* d_payload_out)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_partition_scatter: null engine");
    if (n == 0) {
        e->counted_keys = nullptr;
        return RSX_OK;
    }
    if (d_keys != e->counted_keys || n != e->counted_n || shift != e->counted_shift || bits != e->counted_bits)
        return fail(RSX_CALCULATION_FAILED, "rsx_partition_scatter: must follow rsx_partition_count on the same keys and bit field");
    e->counted_keys = nullptr;       // the table is consumed by this call
    if (n == 0) return RSX_OK;
    if (!d_keys_out || !aligned16(d_keys_out)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter: output must be a 16-byte aligned device pointer");
    const bool with_payload = e->has_payload && d_payload && d_payload_out;
    if (e->has_payload && !with_payload) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter: payload engine needs payload buffers");
    if (with_payload && !aligned16(d_payload)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter: the payload input must be 16-byte aligned (it is read 16 bytes per lane)");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    int rc = launch_scan(e, n);
    if (rc == RSX_OK) rc = launch_paste(e, n);
    if (rc != RSX_OK) return rc;
    const uint32_t mask = (1u << bits) - 1;
    const uint32_t* pin = with_payload ? d_payload : nullptr;
    uint32_t* pout = with_payload ? d_payload_out : nullptr;
    return RSX_BY_KEY(e, launch_reorder<uint32_t>(e, d_keys, d_keys_out, pin, pout, n, shift, mask),
                      launch_reorder<uint64_t>(e, d_keys, d_keys_out, pin, pout, n, shift, mask));
}

int rsx_sample_keys(rsx_engine* e, const void* d_keys, uint64_t n, uint32_t count, uint64_t* samples)
{
    if (!e || !samples) return fail(RSX_CALCULATION_FAILED, "rsx_sample_keys: null argument");
    if (count == 0 || count > static_cast<uint32_t>(kRangeBlocks) * 2) return fail(RSX_CALCULATION_FAILED, "rsx_sample_keys: count must be in [1, 4096]");
    if (n == 0 || !d_keys) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_sample_keys: no keys");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    if (e->key_bytes == 4) {
        hipLaunchKernelGGL(rsx::sample_keys_kernel<uint32_t>, dim3((count + 255) / 256), dim3(256), 0, e->stream, static_cast<const uint32_t*>(d_keys), n,
                           count, flip_mask<uint32_t>(e), e->range_dev);
    } else {
        hipLaunchKernelGGL(rsx::sample_keys_kernel<uint64_t>, dim3((count + 255) / 256), dim3(256), 0, e->stream, static_cast<const uint64_t*>(d_keys), n,
                           count, flip_mask<uint64_t>(e), e->range_dev);
    }
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->range_host, e->range_dev, static_cast<size_t>(count) * 8, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (uint32_t i = 0; i < count; ++i) samples[i] = e->range_host[i];
    return RSX_OK;
}

int rsx_partition_count_split(rsx_engine* e, const void* d_keys, uint64_t n, const uint64_t* splitters, int nsplit, uint64_t* bucket_counts)
{
    if (!e || !bucket_counts || (nsplit > 0 && !splitters)) return fail(RSX_CALCULATION_FAILED, "rsx_partition_count_split: null argument");
    if (nsplit < 1 || nsplit > 7) return fail(RSX_CALCULATION_FAILED, "rsx_partition_count_split: 1..7 splitters");
    for (int k = 1; k < nsplit; ++k) {
        if (splitters[k] <= splitters[k - 1]) return fail(RSX_CALCULATION_FAILED, "rsx_partition_count_split: splitters must be strictly increasing");
    }
    if (n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_partition_count_split: beyond capacity");
    const uint32_t buckets = 2u * static_cast<uint32_t>(nsplit) + 1u;
    for (uint32_t d = 0; d < buckets; ++d) bucket_counts[d] = 0;
    if (n == 0) return RSX_OK;
    if (!d_keys || !aligned16(d_keys)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_count_split: keys must be a 16-byte aligned device pointer");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    for (int k = 0; k < nsplit; ++k) {
        if (e->key_bytes == 4 && splitters[k] > 0xFFFFFFFFull) return fail(RSX_CALCULATION_FAILED, "rsx_partition_count_split: splitter exceeds the key width");
        e->splitters[k] = splitters[k];
    }
    e->nsplit = static_cast<uint32_t>(nsplit);
    const int rc = RSX_BY_KEY(e, (launch_histogram<uint32_t, true>(e, d_keys, n, 0, RSX_RADIX - 1, 0u, 0u, e->nsplit)),
                              (launch_histogram<uint64_t, true>(e, d_keys, n, 0, RSX_RADIX - 1, 0ull, 0ull, e->nsplit)));
    if (rc != RSX_OK) return rc;
    hipLaunchKernelGGL(rsx::digit_totals_kernel, dim3(RSX_RADIX), dim3(256), 0, e->stream, e->table, static_cast<uint32_t>(e->ntiles(n)), e->range_dev);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->range_host, e->range_dev, RSX_RADIX * 8, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (uint32_t d = 0; d < buckets; ++d) bucket_counts[d] = e->range_host[d];
    e->counted_keys = d_keys;
    e->counted_n = n;
    e->counted_shift = -1;            // marks a splitter count
    e->counted_bits = nsplit;
    return RSX_OK;
}

int rsx_partition_scatter_split(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, void* d_keys_out, uint32_t* d_payload_out)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_partition_scatter_split: null engine");
    if (n == 0) {
        e->counted_keys = nullptr;
        return RSX_OK;
    }
    if (d_keys != e->counted_keys || n != e->counted_n || e->counted_shift != -1)
        return fail(RSX_CALCULATION_FAILED, "rsx_partition_scatter_split: must follow rsx_partition_count_split on the same keys");
    e->counted_keys = nullptr;
    if (n == 0) return RSX_OK;
    if (!d_keys_out || !aligned16(d_keys_out)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter_split: output must be a 16-byte aligned device pointer");
    const bool with_payload = e->has_payload && d_payload && d_payload_out;
    if (e->has_payload && !with_payload) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter_split: payload engine needs payload buffers");
    if (with_payload && !aligned16(d_payload)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_scatter_split: the payload input must be 16-byte aligned (it is read 16 bytes per lane)");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    int rc = launch_scan(e, n);
    if (rc == RSX_OK) rc = launch_paste(e, n);
    if (rc != RSX_OK) return rc;
    const uint32_t* pin = with_payload ? d_payload : nullptr;
    uint32_t* pout = with_payload ? d_payload_out : nullptr;
    if (e->key_bytes == 4) {
        return with_payload ? launch_reorder_t<uint32_t, true, false, true>(e, d_keys, d_keys_out, pin, pout, n, 0, RSX_RADIX - 1, 0, 0u, 0u, e->nsplit)
                            : launch_reorder_t<uint32_t, false, false, true>(e, d_keys, d_keys_out, nullptr, nullptr, n, 0, RSX_RADIX - 1, 0, 0u, 0u, e->nsplit);
    }
    return with_payload ? launch_reorder_t<uint64_t, true, false, true>(e, d_keys, d_keys_out, pin, pout, n, 0, RSX_RADIX - 1, 0, 0ull, 0ull, e->nsplit)
                        : launch_reorder_t<uint64_t, false, false, true>(e, d_keys, d_keys_out, nullptr, nullptr, n, 0, RSX_RADIX - 1, 0, 0ull, 0ull, e->nsplit);
}

// ---- peer-visible buffers (the receive side of the peer-store exchange) ----------------------------------------------------
int rsx_peer_alloc(rsx_engine* e, uint64_t bytes, void** d_ptr, void* ipc_handle)
{
    if (!e || !d_ptr || bytes == 0) return fail(RSX_INITIALIZATION_FAILED, "rsx_peer_alloc: null argument");
    if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
    *d_ptr = nullptr;
    void* p = nullptr;
    RSX_TRY(hipMalloc(&p, static_cast<size_t>(bytes)), RSX_INITIALIZATION_FAILED);
    if (ipc_handle) {
        static_assert(sizeof(hipIpcMemHandle_t) == RSX_IPC_HANDLE_BYTES, "handle size of the C ABI");
        hipIpcMemHandle_t h;
        const hipError_t err = hipIpcGetMemHandle(&h, p);
        if (err != hipSuccess) {
            (void)hipFree(p);
            return fail(RSX_INITIALIZATION_FAILED, "hipIpcGetMemHandle", err);
        }
        std::memcpy(ipc_handle, &h, sizeof(h));
    }
    *d_ptr = p;
    return RSX_OK;
}

int rsx_peer_free(rsx_engine* e, void* d_ptr)
{
    if (!e) return fail(RSX_CLEANUP_FAILED, "rsx_peer_free: null engine");
    if (!d_ptr) return RSX_OK;
    if (bind_device(e, RSX_CLEANUP_FAILED) != RSX_OK) return RSX_CLEANUP_FAILED;
    RSX_TRY(hipFree(d_ptr), RSX_CLEANUP_FAILED);
    return RSX_OK;
}

int rsx_peer_open(rsx_engine* e, const void* ipc_handle, void** d_ptr)
{
    if (!e || !ipc_handle || !d_ptr) return fail(RSX_INITIALIZATION_FAILED, "rsx_peer_open: null argument");
    if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle, sizeof(h));
    *d_ptr = nullptr;
    RSX_TRY(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess), RSX_INITIALIZATION_FAILED);
    return RSX_OK;
}

int rsx_peer_close(rsx_engine* e, void* d_ptr)
{
    if (!e) return fail(RSX_CLEANUP_FAILED, "rsx_peer_close: null engine");
    if (!d_ptr) return RSX_OK;
    if (bind_device(e, RSX_CLEANUP_FAILED) != RSX_OK) return RSX_CLEANUP_FAILED;
    RSX_TRY(hipIpcCloseMemHandle(d_ptr), RSX_CLEANUP_FAILED);
    return RSX_OK;
}

int rsx_peer_enable(rsx_engine* e, int peer_device)
{
    if (!e) return fail(RSX_INITIALIZATION_FAILED, "rsx_peer_enable: null engine");
    if (peer_device == e->device) return RSX_OK;
    if (bind_device(e, RSX_INITIALIZATION_FAILED) != RSX_OK) return RSX_INITIALIZATION_FAILED;
    int can = 0;
    RSX_TRY(hipDeviceCanAccessPeer(&can, e->device, peer_device), RSX_INITIALIZATION_FAILED);
    if (!can) return fail(RSX_INITIALIZATION_FAILED, "rsx_peer_enable: the device cannot access that peer");
    const hipError_t err = hipDeviceEnablePeerAccess(peer_device, 0);
    if (err != hipSuccess && err != hipErrorPeerAccessAlreadyEnabled) return fail(RSX_INITIALIZATION_FAILED, "hipDeviceEnablePeerAccess", err);
    (void)hipGetLastError();
    return RSX_OK;
}

#include "capi_msd.inc"

int rsx_key_range(rsx_engine* e, const void* d_keys, uint64_t n, uint64_t* lo, uint64_t* hi)
{
    if (!e || !lo || !hi) return fail(RSX_CALCULATION_FAILED, "rsx_key_range: null argument");
    *lo = ~0ull;
    *hi = 0ull;
    if (n == 0) return RSX_OK;
    if (!d_keys || !aligned16(d_keys)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_key_range: keys must be a 16-byte aligned device pointer");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    return RSX_BY_KEY(e, key_range<uint32_t>(e, d_keys, n, lo, hi), key_range<uint64_t>(e, d_keys, n, lo, hi));
}

int rsx_partition_range(rsx_engine* e, const void* d_keys, const uint32_t* d_payload, uint64_t n, uint64_t lo, int shift, uint64_t mul,
                        void* d_keys_out, uint32_t* d_payload_out, uint64_t* bucket_offsets)
{
    if (!e || !bucket_offsets) return fail(RSX_CALCULATION_FAILED, "rsx_partition_range: null argument");
    if (shift < 0 || shift >= e->key_bytes * 8) return fail(RSX_CALCULATION_FAILED, "rsx_partition_range: shift out of range");
    if (n > e->capacity) return fail(RSX_RESIZE_FAILED, "rsx_partition_range: beyond capacity");
    if (n > 0 && (!d_keys || !d_keys_out || !aligned16(d_keys) || !aligned16(d_keys_out)))
        return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_range: key buffers must be 16-byte aligned device pointers");
    const bool with_payload = e->has_payload && d_payload && d_payload_out;
    if (e->has_payload && !with_payload && n > 0) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_range: payload engine needs payload buffers");
    if (with_payload && !aligned16(d_payload)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_partition_range: the payload input must be 16-byte aligned (it is read 16 bytes per lane)");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    if (n == 0) {
        for (uint32_t d = 0; d <= RSX_RADIX; ++d) bucket_offsets[d] = 0;
        return RSX_OK;
    }
    const uint32_t* pin = with_payload ? d_payload : nullptr;
    uint32_t* pout = with_payload ? d_payload_out : nullptr;
    if (e->key_bytes == 4 && mul > 0xFFFFFFFFull) return fail(RSX_CALCULATION_FAILED, "rsx_partition_range: multiplier exceeds the key width");
    const int rc = RSX_BY_KEY(e, run_ranged_pass<uint32_t>(e, d_keys, d_keys_out, pin, pout, n, lo, shift, mul),
                              run_ranged_pass<uint64_t>(e, d_keys, d_keys_out, pin, pout, n, lo, shift, mul));
    if (rc != RSX_OK) return rc;
    hipLaunchKernelGGL(rsx::bucket_starts_kernel, dim3(1), dim3(64), 0, e->stream, e->table, static_cast<uint32_t>(e->ntiles(n)), e->starts_dev);
    RSX_TRY(hipGetLastError(), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpyAsync(e->starts_host, e->starts_dev, RSX_RADIX * 4, hipMemcpyDeviceToHost, e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    for (uint32_t d = 0; d < RSX_RADIX; ++d) bucket_offsets[d] = e->starts_host[d];
    bucket_offsets[RSX_RADIX] = n;
    return RSX_OK;
}

int rsx_tile_map(uint64_t num_keys, uint32_t tile_keys, int xcd_remap, int64_t xcd_phase, uint32_t* tiles_out, uint64_t cap, uint32_t* blocks, uint32_t* ntiles)
{
    if (!blocks || !ntiles || tile_keys == 0 || (cap > 0 && !tiles_out)) return fail(RSX_HOST_BUFFERS_FAILED, "rsx_tile_map: null argument");
    if (xcd_phase < -1 || (num_keys + tile_keys - 1) / tile_keys > (1ull << 24)) return fail(RSX_RESIZE_FAILED, "rsx_tile_map: out of range");
    rsx_engine probe;                    // no device is touched: the launch geometry is host arithmetic
    probe.xcd_remap = xcd_remap != 0;
    probe.xcd_phase = xcd_phase;
    const Grid g = grid_for(&probe, num_keys, tile_keys);
    *blocks = g.blocks;
    *ntiles = g.ntiles;
    for (uint64_t b = 0; b < g.blocks && b < cap; ++b) tiles_out[b] = rsx::tile_of_block(static_cast<uint32_t>(b), g.tiles_per_xcd, g.remap);
    return RSX_OK;
}

int rsx_result_device(rsx_engine* e, void** d_keys, uint32_t** d_payload)
{
    if (!e) return fail(RSX_CALCULATION_FAILED, "rsx_result_device: null engine");
    if (d_keys) *d_keys = e->result_external ? nullptr : e->result_keys;
    if (d_payload) *d_payload = e->result_external ? nullptr : e->result_perm;
    return RSX_OK;
}

int rsx_copy_result(rsx_engine* e, void* d_keys_out, uint32_t* d_payload_out)
{
    if (!e) return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_copy_result: null engine");
    if (bind_device(e, RSX_DATA_DOWNLOAD_FAILED) != RSX_OK) return RSX_DATA_DOWNLOAD_FAILED;
    if (e->n == 0) return RSX_OK;
    if (check_scan_timeout(e, RSX_DATA_DOWNLOAD_FAILED) != RSX_OK) return RSX_DATA_DOWNLOAD_FAILED;      // (of sorts that have finished: this call does not synchronise)
    if (e->result_external)
        return fail(RSX_DATA_DOWNLOAD_FAILED, "rsx_copy_result: the last sort (rsx_sort_from_to) wrote into the caller's buffer; the engine holds no result");
    if (d_keys_out) {
        RSX_TRY(hipMemcpyAsync(d_keys_out, e->result_keys, static_cast<size_t>(e->n) * e->key_bytes, hipMemcpyDeviceToDevice, e->stream),
                RSX_DATA_DOWNLOAD_FAILED);
    }
    if (d_payload_out && e->has_payload) {
        RSX_TRY(hipMemcpyAsync(d_payload_out, e->result_perm, static_cast<size_t>(e->n) * 4, hipMemcpyDeviceToDevice, e->stream),
                RSX_DATA_DOWNLOAD_FAILED);
    }
    return RSX_OK;
}

#ifdef RSX_STAMPS
// diagnostic build only: the 16 stamps of the first `tiles` tiles of the chosen launch
int rsx_debug_stamps(rsx_engine* e, unsigned long long* host_out, uint64_t tiles)
{
    if (!e || !host_out) return RSX_CALCULATION_FAILED;
    RSX_TRY(hipStreamSynchronize(e->stream), RSX_CALCULATION_FAILED);
    RSX_TRY(hipMemcpy(host_out, e->stamps, std::min<uint64_t>(tiles, e->ntiles(e->capacity)) * 16 * 8, hipMemcpyDeviceToHost), RSX_CALCULATION_FAILED);
    return RSX_OK;
}
#endif

int rsx_timings(rsx_engine* e, rsx_runtimes* out, int reset)
{
    if (!e || !out) return fail(RSX_CALCULATION_FAILED, "rsx_timings: null argument");
    if (bind_device(e, RSX_CALCULATION_FAILED) != RSX_OK) return RSX_CALCULATION_FAILED;
    const int rc = drain_events(e);
    if (rc != RSX_OK) return rc;
    out->histogram = e->stats[PH_HISTO];
    out->scan = e->stats[PH_SCAN];
    out->paste = e->stats[PH_PASTE];
    out->reorder = e->stats[PH_REORDER];
    out->total = e->stats[PH_TOTAL];
    if (reset) {
        for (auto& s : e->stats) stat_reset(s);
    }
    return RSX_OK;
}

}  // extern "C"
