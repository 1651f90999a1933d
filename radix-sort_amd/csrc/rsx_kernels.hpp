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
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsx {

constexpr int kRadixBits = 4;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kWave = 64;      // CDNA wavefront
constexpr int kNumXcd = 8;     // MI355X: 8 XCDs, each with a private L2

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

// 16-byte vector of keys: 4 x u32 or 2 x u64 -> one global_load_dwordx4 / ds_*_b128.
template <typename Key>
struct alignas(16) KeyVec {
    static constexpr int N = 16 / sizeof(Key);
    Key k[N];
};

struct alignas(16) U32x4 {
    uint32_t v[4];
};

// Keys are read exactly once per pass: RSX_STREAM_LOADS=1 marks those 16-byte loads non-temporal (experiment, tuning log §7).
#ifndef RSX_STREAM_LOADS
#define RSX_STREAM_LOADS 0
#endif
template <typename Key>
__device__ __forceinline__ KeyVec<Key> load_keys16(const Key* p)
{
#if RSX_STREAM_LOADS
    typedef uint32_t u32x4_native __attribute__((ext_vector_type(4)));
    const u32x4_native x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_native*>(p));
    KeyVec<Key> v;
    __builtin_memcpy(&v, &x, 16);
    return v;
#else
    return *reinterpret_cast<const KeyVec<Key>*>(p);
#endif
}

// The packed counters in LDS are touched as 16-bit halves, 32-bit words and 16-byte
// vectors; these typedefs keep type-based alias analysis from reordering them.
typedef uint16_t __attribute__((may_alias)) u16_alias;
typedef uint32_t __attribute__((may_alias)) u32_alias;

template <typename Key>
__device__ __forceinline__ uint32_t digit_of(Key key, int shift, Key flip, uint32_t mask)
{
    // `flip` is the sign bit for signed key types, 0 otherwise: the reference's
    // `key + OFFSET` with OFFSET = -numeric_limits<T>::min() (RadixSortGPU.cpp:436-440,
    // RadixSort.cl:51) is exactly an XOR of the sign bit.
    return static_cast<uint32_t>((key ^ flip) >> shift) & mask;
}

// Bucket of the multi-GPU partition pass: x = (key ^ sign) - lo, 16 equal-width buckets over the
// global key range.  mul != 0: floor(x * 16 / (hi - lo + 1)) as a multiply-high by
// mul = floor(16 * 2^W / (hi - lo + 1)); mul == 0 (ranges of at most 16 values): x >> shift.
// Monotone in the key either way; the clamp only ever catches the padding key.
__device__ __forceinline__ uint32_t ranged_bucket(uint32_t x, int shift, uint32_t mul, uint32_t mask)
{
    const uint32_t q = mul ? __umulhi(x, mul) : (x >> shift);
    return q < mask ? q : mask;
}
__device__ __forceinline__ uint32_t ranged_bucket(uint64_t x, int shift, uint64_t mul, uint32_t mask)
{
    const uint64_t q = mul ? __umul64hi(x, mul) : (x >> shift);
    return q < mask ? static_cast<uint32_t>(q) : mask;
}

// Bucket by splitters s_0 < s_1 < ... (at most 7, distinct, unsigned order):
//   bucket(x) = 2 * #{s_k < x} + [x == some s_k]
// even buckets are the open intervals between splitters, odd buckets hold exactly the keys EQUAL to
// a splitter — the host may cut those anywhere (ties split by (rank, index)), which is what keeps the
// ranks balanced when one key value dominates.  Monotone in x; at most 15 buckets.
// The set travels by value in the kernel arguments, so the splitters sit in scalar registers.
constexpr int kMaxSplitters = 7;
template <typename Key>
struct SplitSet {
    Key s[kMaxSplitters];
    uint32_t n;
    uint32_t rot;      // n == 0 only: rotate the 4-bit range bucket right by `rot` (wave-major bucket order, see wave_major)
};

// Wave-major order of the 16 top-nibble buckets for world = 16 >> rot ranks owning 1 << rot
// consecutive buckets each: bucket b = rank * k + wave  ->  wave * world + rank, which for these
// powers of two is a rotation of the nibble.  All the buckets of one wave then sit next to each
// other, in rank order, so that wave can leave in one all-to-all while the next is still being sorted.
__device__ __forceinline__ uint32_t wave_major(uint32_t b, uint32_t rot)
{
    return ((b >> rot) | (b << (kRadixBits - rot))) & static_cast<uint32_t>(kRadix - 1);
}

template <typename Key>
__device__ __forceinline__ uint32_t splitter_bucket(Key x, const SplitSet<Key>& set)
{
    uint32_t b = 0;
#pragma unroll
    for (int k = 0; k < kMaxSplitters; ++k) {
        if (k < static_cast<int>(set.n)) {            // wave-uniform
            b += (x > set.s[k] ? 1u : 0u) + (x >= set.s[k] ? 1u : 0u);
        }
    }
    return b;
}

// Workgroup -> tile.  Hardware deals consecutive workgroup ids round-robin over the 8
// XCDs (observed, speed only).  With the remap every XCD walks its own contiguous range
// of tiles, so the seam between the output runs of tiles t and t+1 (same digit,
// adjacent addresses, usually inside one 128-B line) meets in ONE L2 and is merged
// before it goes to HBM; the [digit][tile] table rows are written the same way.
//
// Phase (bits 8.. of `remap`, in tiles): XCD x starts its walk `x * phase` tiles into its range and wraps round.  Without it the
// eight XCDs advance in lockstep through ranges that start exactly n/8 apart (128 MiB at 2^28 uint32 keys), i.e. at any moment
// their eight read streams (and the 8 x 16 write streams) sit on identical low address bits and pile onto the same HBM channels:
// measured 3.57-3.69 ms per sort in lockstep against 3.34-3.40 ms staggered (profiles/r02_tuning_log.md §6).
__device__ __forceinline__ uint32_t tile_of_block(uint32_t bid, uint32_t tiles_per_xcd, int remap)
{
    if (!(remap & 1)) {
        return bid;
    }
    const uint32_t x = bid % kNumXcd;
    uint32_t j = bid / kNumXcd + x * (static_cast<uint32_t>(remap) >> 8);      // host keeps 7 * phase < tiles_per_xcd
    j = j >= tiles_per_xcd ? j - tiles_per_xcd : j;
    return x * tiles_per_xcd + j;
}

// Inclusive prefix sum across the 64 lanes of a wave with DPP only (no LDS traffic):
// Hillis-Steele inside each row of 16 lanes (row_shr 1,2,4,8), then the row totals are
// carried with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
    return v;
}

// Exclusive prefix over the workgroup of one value per thread; `total` gets the sum.
// `wtot` is LDS scratch of THREADS/64 words.  Contains two barriers.
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wtot, uint32_t& total)
{
    constexpr int WAVES = THREADS / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t incl = wave_inclusive_scan(v);
    if (lane == kWave - 1) {
        wtot[wave] = incl;
    }
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t t = wtot[w];
        before += (static_cast<uint32_t>(w) < wave) ? t : 0u;
        all += t;
    }
    total = all;
    __syncthreads();   // wtot may be reused by the caller
    return before + incl - v;
}

// ---------------------------------------------------------------------------
// histogram: table[digit][tile] = number of keys of the tile with that digit
// ---------------------------------------------------------------------------
// HBM traffic: reads n*sizeof(Key) once, coalesced 16 B/lane; writes 64 B per tile.
// Counting uses LDS atomics on 32 lane-private replicas of the 16 counters (row stride
// 17 words): a wave whose keys all share one digit (Zeros, Range) still spreads over 32
// banks instead of serialising on one address.
// RANGED (multi-GPU partition only): bucket = ranged_bucket((key ^ flip) - lo) — 16 equal-width
// buckets over the global key range [lo, hi], a monotone function of the key.
template <typename Key, int THREADS, int KPT, bool RANGED = false>
__global__ __launch_bounds__(THREADS) void histogram_kernel(const Key* __restrict__ keys, uint32_t* __restrict__ table,
                                                             uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd,
                                                             int remap, int shift, Key flip, uint32_t mask, Key lo, Key mul,
                                                             SplitSet<Key> split, uint32_t* __restrict__ rows_out = nullptr,
                                                             uint32_t* __restrict__ zero_a = nullptr, uint32_t* __restrict__ zero_b = nullptr)
{
    auto dig = [=](Key key) -> uint32_t {
        if constexpr (RANGED) {
            if (split.n) {
                return splitter_bucket(static_cast<Key>(key ^ flip), split);
            }
            const uint32_t b = ranged_bucket(static_cast<Key>((key ^ flip) - lo), shift, mul, mask);
            return split.rot ? wave_major(b, split.rot) : b;
        } else {
            return digit_of(key, shift, flip, mask);
        }
    };
    constexpr int TILE = THREADS * KPT;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    constexpr int REP = 32, RSTRIDE = kRadix + 1;
    __shared__ uint32_t cnt[REP * RSTRIDE];

    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    for (uint32_t i = tid; i < REP * RSTRIDE; i += THREADS) {
        cnt[i] = 0;
    }
    __syncthreads();

    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    uint32_t* mine = cnt + (tid & (REP - 1)) * RSTRIDE;

    if (valid == TILE) {
        KeyVec<Key> v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = load_keys16(keys + base + static_cast<uint32_t>(j) * THREADS * VEC + tid * VEC);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                atomicAdd(&mine[dig(v[j].k[e])], 1u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const uint32_t li = static_cast<uint32_t>(j) * THREADS * VEC + tid * VEC + e;
                if (li < valid) {
                    atomicAdd(&mine[dig(keys[base + li])], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t s = 0;
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            s += cnt[r * RSTRIDE + tid];
        }
        if (rows_out) {
            // self-scan sorts (small tables): raw counts as one [tile][16] row, and this tile's rows of the two
            // other rotating count buffers start from zero
            rows_out[tile * kRadix + tid] = s;
            zero_a[tile * kRadix + tid] = 0;
            zero_b[tile * kRadix + tid] = 0;
        } else {
            table[static_cast<uint64_t>(tid) * ntiles + tile] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// scan of the [digit][tile] table (ScanHistogram / PasteHistogram)
// ---------------------------------------------------------------------------
// Two levels like the reference (512 groups of 32 entries -> scan of the group sums -> paste,
// RadixSort.cl:125-197), cut differently: a scan group is 256 consecutive TILES of all 16
// digits (4096 entries), so the raw counts are read as whole [tile][16] rows when they come
// from the look-ahead buffer and as 16 coalesced row segments when they come from the
// histogram kernel.  Group sums live in globsum[digit][group]; their exclusive scan in that
// (digit-major) order is the global offset of each group.
constexpr int kScanTiles = 256;                               // tiles per scan group = threads per workgroup
constexpr int kScanBlock = kScanTiles;                        // entries of ONE digit per scan group
constexpr int kGlobsumThreads = 1024;
constexpr int kMaxScanGroups = 4096;                          // 2^20 tiles
constexpr int kMaxScanBlocks = kRadix * kMaxScanGroups;       // entries of globsum

// scan #1: exclusive scan over the 256 tiles of the group, per digit; group total -> globsum[d][group]
template <bool FROM_COUNTS, bool ZERO_BACK = true>
__global__ __launch_bounds__(kScanTiles) void scan_blocks_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ globsum,
                                                                  uint32_t ntiles, uint32_t ngroups, uint32_t* __restrict__ counts)
{
    constexpr int WAVES = kScanTiles / kWave;
    __shared__ uint32_t wsum[WAVES][kRadix];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t group = blockIdx.x;
    const uint32_t tile = group * kScanTiles + tid;
    const bool live = tile < ntiles;
    uint32_t c[kRadix];
    if constexpr (FROM_COUNTS) {
        // one [tile][16] row per thread: 64 contiguous bytes; handed back zeroed so that the next
        // look-ahead pass needs no memset
        U32x4* row = reinterpret_cast<U32x4*>(counts + static_cast<uint64_t>(tile) * kRadix);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            U32x4 x = {{0u, 0u, 0u, 0u}};
            if (live) {
                x = row[q];
                if constexpr (ZERO_BACK) {
                    row[q] = U32x4{{0u, 0u, 0u, 0u}};
                }
            }
            c[q * 4 + 0] = x.v[0];
            c[q * 4 + 1] = x.v[1];
            c[q * 4 + 2] = x.v[2];
            c[q * 4 + 3] = x.v[3];
        }
    } else {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            c[d] = live ? table[static_cast<uint64_t>(d) * ntiles + tile] : 0u;
        }
    }
    uint32_t incl[kRadix];
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        incl[d] = wave_inclusive_scan(c[d]);
    }
    if (lane == kWave - 1) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            wsum[wave][d] = incl[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            before += (static_cast<uint32_t>(w) < wave) ? wsum[w][d] : 0u;
        }
        if (live) {
            table[static_cast<uint64_t>(d) * ntiles + tile] = before + incl[d] - c[d];
        }
    }
    if (tid < kRadix) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            total += wsum[w][tid];
        }
        globsum[static_cast<uint64_t>(tid) * ngroups + group] = total;
    }
}

// scan #2: exclusive scan of the group sums in place (digit-major), grand total -> temp[0]
__global__ __launch_bounds__(kGlobsumThreads) void scan_globsum_kernel(uint32_t* __restrict__ globsum, uint32_t* __restrict__ temp,
                                                                        uint32_t nentries)
{
    __shared__ uint32_t wtot[kGlobsumThreads / kWave];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nentries + kGlobsumThreads - 1) / kGlobsumThreads;     // <= 64
    const uint32_t first = tid * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i) {
        sum += (first + i < nentries) ? globsum[first + i] : 0u;
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<kGlobsumThreads>(sum, wtot, total);
    for (uint32_t i = 0; i < per; ++i) {
        if (first + i < nentries) {
            const uint32_t cnt = globsum[first + i];
            globsum[first + i] = run;
            run += cnt;
        }
    }
    if (tid == 0) {
        temp[0] = total;
    }
}

// paste: every entry of (digit d, group g) += scanned globsum[d][g] -> global exclusive prefix
__global__ __launch_bounds__(kScanTiles) void paste_kernel(uint32_t* __restrict__ table, const uint32_t* __restrict__ globsum,
                                                            uint32_t ntiles, uint32_t ngroups)
{
    const uint32_t group = blockIdx.x;
    const uint32_t tile = group * kScanTiles + threadIdx.x;
    if (tile >= ntiles) {
        return;
    }
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        table[static_cast<uint64_t>(d) * ntiles + tile] += globsum[static_cast<uint64_t>(d) * ngroups + group];
    }
}

// paste with scan #2 folded in (rsx_sort path): every workgroup derives the 16 global offsets of ITS
// group straight from the RAW group sums — sum of all groups of smaller digits plus the groups
// before it in its own digit row — and applies them.  All workgroups redo the (tiny, L2-resident)
// reduction instead of waiting for a one-workgroup scan kernel: one launch less per pass.  The
// scanned values are also written to `scanned` so that a downloaded globsum looks the same.
__global__ __launch_bounds__(kScanTiles) void paste_scan_kernel(uint32_t* __restrict__ table, const uint32_t* __restrict__ raw_sums,
                                                                 uint32_t* __restrict__ scanned, uint32_t* __restrict__ temp,
                                                                 uint32_t ntiles, uint32_t ngroups)
{
    constexpr int WAVES = kScanTiles / kWave;
    __shared__ uint32_t part[WAVES][2 * kRadix];
    __shared__ uint32_t dtot[kRadix];
    __shared__ uint32_t off[kRadix];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t group = blockIdx.x;
    uint32_t pre[kRadix], tot[kRadix];
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        pre[d] = 0;
        tot[d] = 0;
    }
    for (uint32_t g2 = tid; g2 < ngroups; g2 += kScanTiles) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            const uint32_t v = raw_sums[static_cast<uint64_t>(d) * ngroups + g2];
            tot[d] += v;
            pre[d] += (g2 < group) ? v : 0u;
        }
    }
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        const uint32_t a = wave_inclusive_scan(pre[d]);
        const uint32_t b = wave_inclusive_scan(tot[d]);
        if (lane == kWave - 1) {
            part[wave][d] = a;
            part[wave][kRadix + d] = b;
        }
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            t += part[w][kRadix + tid];
        }
        dtot[tid] = t;
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t base = 0;
#pragma unroll 1
        for (uint32_t d2 = 0; d2 < tid; ++d2) {
            base += dtot[d2];
        }
        uint32_t p = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            p += part[w][tid];
        }
        off[tid] = base + p;
        scanned[static_cast<uint64_t>(tid) * ngroups + group] = base + p;
        if (group == 0 && tid == kRadix - 1) {
            temp[0] = base + dtot[tid];
        }
    }
    __syncthreads();
    const uint32_t tile = group * kScanTiles + tid;
    if (tile < ntiles) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            table[static_cast<uint64_t>(d) * ntiles + tile] += off[d];
        }
    }
}

// scan #1, scan #2 and paste in ONE launch for tables of up to kFusedScanMaxGroups groups (all of its workgroups
// are then resident at once).  Every workgroup scans its 256 tiles per digit as scan_blocks_kernel does and
// KEEPS the block-local prefixes in registers.  Its 16 group sums leave as 8-byte {epoch, value} granules —
// one aligned write-through (agent-scope, sc1) store each, the data is its own flag — into sums[group][16];
// then every workgroup sweeps ALL granules with agent-scope loads, re-reading a granule until its tag is
// this launch's epoch (relaxed polls with s_sleep, bounded), derives its 16 global offsets and writes the
// finished table once.  Both sides of the hand-off bypass the non-coherent L1/L2 path, so no release/acquire
// fence (which would write back the 8 MiB of counters just zeroed) is needed, and a granule of an earlier
// launch can never be taken for a current one.  Against the two-launch form this drops a kernel boundary,
// the table's second read and write, and the 32 wave scans of the paste.  `epoch` is the engine's launch
// count (never 0).  A poll that runs out sets *timeout and lets the workgroup finish with garbage rather
// than hang the GPU (rsx_sync / rsx_download report it).
constexpr int kFusedScanMaxGroups = 512;      // 2^29 keys; 256-thread workgroups, <= 2 per CU: all resident with room to spare (its registers allow 4)
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

template <bool FROM_COUNTS, bool ZERO_BACK>
__global__ __launch_bounds__(kScanTiles) void scan_fused_kernel(uint32_t* __restrict__ table, unsigned long long* sums, uint32_t* __restrict__ scanned,
                                                                 uint32_t* __restrict__ temp, uint32_t ntiles, uint32_t ngroups,
                                                                 uint32_t* __restrict__ counts, uint32_t epoch, uint32_t* timeout)
{
    constexpr int WAVES = kScanTiles / kWave;
    __shared__ uint32_t wsum[WAVES][kRadix];
    __shared__ uint32_t part[WAVES][2][kRadix];
    __shared__ uint32_t dtot[kRadix], off[kRadix];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t group = blockIdx.x;
    const uint32_t tile = group * kScanTiles + tid;
    const bool live = tile < ntiles;
    uint32_t c[kRadix];
    if constexpr (FROM_COUNTS) {
        U32x4* row = reinterpret_cast<U32x4*>(counts + static_cast<uint64_t>(tile) * kRadix);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            U32x4 x = {{0u, 0u, 0u, 0u}};
            if (live) {
                x = row[q];
                if constexpr (ZERO_BACK) {
                    row[q] = U32x4{{0u, 0u, 0u, 0u}};
                }
            }
            c[q * 4 + 0] = x.v[0];
            c[q * 4 + 1] = x.v[1];
            c[q * 4 + 2] = x.v[2];
            c[q * 4 + 3] = x.v[3];
        }
    } else {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            c[d] = live ? table[static_cast<uint64_t>(d) * ntiles + tile] : 0u;
        }
    }
    uint32_t ex[kRadix];          // block-local exclusive prefix of this tile, per digit (stays in registers)
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        ex[d] = wave_inclusive_scan(c[d]);
    }
    if (lane == kWave - 1) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            wsum[wave][d] = ex[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            before += (static_cast<uint32_t>(w) < wave) ? wsum[w][d] : 0u;
        }
        ex[d] = before + ex[d] - c[d];
    }
    // ---- publish the 16 group sums as {epoch, value} granules ---------------------------------------
    if (tid < kRadix) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            total += wsum[w][tid];
        }
        __hip_atomic_store((gu64*)(sums) + static_cast<uint64_t>(group) * kRadix + tid, (static_cast<unsigned long long>(epoch) << 32) | total,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- sweep all group sums -> the 16 global offsets of this group ---------------------------------
    // thread (digit d = tid & 15, slice p = tid >> 4) walks the groups p, p+16, ...: a wave reads 4 whole
    // 128-byte rows per load; a granule whose tag is not yet this launch's is simply read again
    {
        const uint32_t d = tid & 15u, p = tid >> 4;
        uint32_t tot = 0, pre = 0, spins = 0;
        constexpr uint32_t SLICES = kScanTiles / kRadix;
        constexpr int BATCH = 8;                 // loads in flight per thread: the sweep is a chain of dependent L2 trips otherwise (0.0146 -> 0.0122 ms at 65,536 tiles)
        const unsigned long long absent = static_cast<unsigned long long>(epoch) << 32;
        for (uint32_t g0 = p; g0 < ngroups; g0 += SLICES * BATCH) {
            unsigned long long x[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const uint32_t g2 = g0 + b * SLICES;
                x[b] = g2 < ngroups ? __hip_atomic_load((gu64*)(sums) + static_cast<uint64_t>(g2) * kRadix + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : absent;
            }
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const uint32_t g2 = g0 + b * SLICES;
                while (static_cast<uint32_t>(x[b] >> 32) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {              // seconds: something is badly wrong; do not hang the device
                        __hip_atomic_store((gu32*)(timeout), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    x[b] = __hip_atomic_load((gu64*)(sums) + static_cast<uint64_t>(g2) * kRadix + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const uint32_t v = static_cast<uint32_t>(x[b]);
                tot += v;
                pre += (g2 < group) ? v : 0u;
            }
        }
        tot += __shfl_xor(tot, 16);
        pre += __shfl_xor(pre, 16);
        tot += __shfl_xor(tot, 32);
        pre += __shfl_xor(pre, 32);
        if (lane < kRadix) {
            part[wave][0][lane] = tot;
            part[wave][1][lane] = pre;
        }
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            t += part[w][0][tid];
        }
        dtot[tid] = t;
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t base = 0;
#pragma unroll 1
        for (uint32_t d2 = 0; d2 < tid; ++d2) {
            base += dtot[d2];
        }
        uint32_t pr = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            pr += part[w][1][tid];
        }
        off[tid] = base + pr;
        scanned[static_cast<uint64_t>(tid) * ngroups + group] = base + pr;
        if (group == 0 && tid == kRadix - 1) {
            temp[0] = base + dtot[tid];
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            table[static_cast<uint64_t>(d) * ntiles + tile] = ex[d] + off[d];
        }
    }
}

// Whole table scan in ONE workgroup — scan #1, scan #2 and paste of a small table in a single
// launch.  Up to 2^22 keys a pass is so short that the three tiny kernels above and their launch
// boundaries (~15 us together) dominate it; one 1024-thread workgroup walks a table of at most
// 1024 tiles in a few microseconds (measured: 0.081 vs 0.107 ms per sort at 2^16 keys, 0.098 vs
// 0.126 at 2^20; beyond 2^22 keys the single workgroup loses to the three launches).  Thread t owns the consecutive tiles
// [t*tpt, (t+1)*tpt): it sums its rows per digit, the 16 per-digit sums are scanned across the
// workgroup, digit d starts after all keys of smaller digits, and a second walk over the same
// rows writes the global exclusive prefix.
constexpr int kSmallScanThreads = 1024;
constexpr int kSmallScanMaxTiles = 1024;   // one tile per thread; beyond ~2^22 keys one workgroup is slower than the three launches

template <bool FROM_COUNTS, bool ZERO_BACK>
__global__ __launch_bounds__(kSmallScanThreads) void scan_small_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ counts,
                                                                        uint32_t* __restrict__ temp, uint32_t ntiles)
{
    constexpr int WAVES = kSmallScanThreads / kWave;
    __shared__ uint32_t wsum[WAVES][kRadix], wpre[WAVES][kRadix], dtot[kRadix];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t tpt = (ntiles + kSmallScanThreads - 1) / kSmallScanThreads;
    const uint32_t t0 = tid * tpt;
    const uint32_t t1 = (t0 + tpt < ntiles) ? t0 + tpt : ntiles;

    auto load_row = [&](uint32_t tile, uint32_t (&c)[kRadix]) {
        if constexpr (FROM_COUNTS) {
            const U32x4* row = reinterpret_cast<const U32x4*>(counts + static_cast<uint64_t>(tile) * kRadix);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const U32x4 x = row[q];
                c[q * 4 + 0] = x.v[0];
                c[q * 4 + 1] = x.v[1];
                c[q * 4 + 2] = x.v[2];
                c[q * 4 + 3] = x.v[3];
            }
        } else {
#pragma unroll
            for (int d = 0; d < kRadix; ++d) {
                c[d] = table[static_cast<uint64_t>(d) * ntiles + tile];
            }
        }
    };

    uint32_t sums[kRadix];
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        sums[d] = 0;
    }
    for (uint32_t tile = t0; tile < t1; ++tile) {
        uint32_t c[kRadix];
        load_row(tile, c);
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            sums[d] += c[d];
        }
    }
    uint32_t start[kRadix];
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        start[d] = wave_inclusive_scan(sums[d]);
    }
    if (lane == kWave - 1) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            wsum[wave][d] = start[d];
        }
    }
    __syncthreads();
    // cross-wave combine by 256 threads (wave w, digit d): sums of the waves before w; kept out
    // of registers on purpose (16 x 16 values per thread would spill)
    if (tid < WAVES * kRadix) {
        const uint32_t w = tid / kRadix, d = tid % kRadix;
        uint32_t acc = 0;
#pragma unroll 1
        for (uint32_t w2 = 0; w2 < w; ++w2) {
            acc += wsum[w2][d];
        }
        wpre[w][d] = acc;
        if (w == WAVES - 1) {
            dtot[d] = acc + wsum[w][d];
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;             // keys with a smaller digit, whole array
#pragma unroll 1
        for (int d = 0; d < kRadix; ++d) {
            const uint32_t t = dtot[d];
            dtot[d] = run;
            run += t;
        }
        temp[0] = run;                // grand total, as scan #2 leaves it
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        start[d] = dtot[d] + wpre[wave][d] + start[d] - sums[d];
    }
    for (uint32_t tile = t0; tile < t1; ++tile) {
        uint32_t c[kRadix];
        load_row(tile, c);
        if constexpr (FROM_COUNTS && ZERO_BACK) {
            U32x4* row = reinterpret_cast<U32x4*>(counts + static_cast<uint64_t>(tile) * kRadix);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                row[q] = U32x4{{0u, 0u, 0u, 0u}};
            }
        }
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            table[static_cast<uint64_t>(d) * ntiles + tile] = start[d];
            start[d] += c[d];
        }
    }
}

// ---------------------------------------------------------------------------
// reorder: the stable scatter (the graded pass)
// ---------------------------------------------------------------------------
// LDS plan of one workgroup (dwords):
//   xbuf  : the tile in locally sorted order (one pad element every 2^PADSH so that the
//           stride-KPT writes of a single-digit tile do not pile on two banks)
//   cnt   : 8 x THREADS packed counters, word [d&7][thread] holds digit d in its low
//           (d<8) or high (d>=8) 16 bits
//   wtot  : wave totals of the raking scan
//   runs  : per digit, {(global slot of the tile's first key of that digit) - (its local slot), look-ahead base}
//   la    : look-ahead counters [digit][segment 0/1][next digit] + one dummy
//
// Instruction count matters as much as bytes here: measured on MI355X the fused kernel's time follows
// the shader clock (0.42 ms at 2.4 GHz, 0.50 ms at 1.9 GHz — the clock the power controller drops to for a
// few milliseconds when a sort starts on an idle GPU, which is every sort in the reference's upload ->
// sort -> download order), while the plain kernel stays at its HBM time.  Hence the hand-placed address
// arithmetic below: every per-key step is written so that it compiles to the fewest VALU instructions
// (profiles/r02_tuning_log.md has the before/after ISA counts).
struct alignas(8) RunBase {
    uint32_t gbase;      // (global slot of the tile's first key of this digit) - (its tile-local slot)
    uint32_t la_base;    // (digit << 5) - (output tile of that global slot << 4): la index of a key = la_base + (tile of ITS slot << 4) + next digit
};

// Look-ahead histogram: one key's contribution to la[(digit, segment)][next digit].
// `idx` is the counter index (kLaDummy for a slot that holds no key).  On random data the 64
// lanes of a wave spread over 16 counters (4 lanes each) and simply add 1.  When the whole
// wave targets ONE counter (constant or sorted data: every pass of Zeros, most passes of
// Range) the uniform branch lets lane 0 add 64 instead of 64 lanes serialising on one address.
constexpr uint32_t kLaDummy = 2 * kRadix * kRadix;   // one spare counter past the 512 real ones
// RSX_LA_REPLICAS=2 (default): every counter in two adjacent copies, odd and even lanes adding to different ones, so that the 32
// lanes of one LDS pass hit 32 different words instead of piling two deep on 16 addresses.  Before the XCD stagger this made no
// difference (the kernel waited for HBM); with it, interleaved A/B: 3.345-3.392 against 3.369-3.423 ms per sort back to back,
// and 0.417-0.424 against 0.430-0.447 ms per scatter launch right after an upload, when the shader clock is low and the waves
// wait for LDS issue (SQ counters: 23 % of their cycles, bank conflicts on 49 % of the LDS cycles with one copy).
#ifndef RSX_EARLY_RANK
#define RSX_EARLY_RANK 1
#endif
#ifndef RSX_LA_REPLICAS
#define RSX_LA_REPLICAS 2
#endif
constexpr int kLaReplicas = RSX_LA_REPLICAS;
static_assert(kLaReplicas == 1 || kLaReplicas == 2, "odd/even-lane replicas");

// (the payload kernels keep one copy: their A/B showed nothing beyond run-to-run noise, and they are the ones short of registers;
// so do the 64-bit keys-only kernels: 13.03 against 13.15 ms per 2^28-key sort with one copy)
template <typename Key, bool PAYLOAD>
constexpr int la_replicas()
{
    return (PAYLOAD || sizeof(Key) != 4) ? 1 : kLaReplicas;
}
template <int REPL>
__device__ __forceinline__ void lookahead_count(uint32_t* la, uint32_t idx)
{
    const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(idx)));
    if (__builtin_expect(__ballot(idx != first) == 0ull, 0)) {
        if ((threadIdx.x & (kWave - 1)) == 0) {
            atomicAdd(&la[first * REPL], static_cast<uint32_t>(kWave));
        }
    } else {
        atomicAdd(&la[idx * REPL + (threadIdx.x & (REPL - 1))], 1u);
    }
}

// (a + b) << SH in ONE instruction.  hipcc lowers `(slot + (slot >> 5)) * 4` to shift, shift, and, add3 (it
// distributes the multiplication); the staging address of every key is exactly this expression.
template <int SH>
__device__ __forceinline__ uint32_t add_lshl(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_add_lshl_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(SH));
    return r;
}

// A store to the workgroup's LDS at a BYTE OFFSET from its start.  The kernels below carve everything out of one
// `extern __shared__` array and declare no static LDS, so that array starts at LDS address 0 — but hipcc
// only learns this after instruction selection and otherwise spends one `v_add_u32 addr, 0, addr` per
// computed address.  reorder_kernel checks the assumption once per workgroup (lds_base_is_zero).
template <typename T>
__device__ __forceinline__ void lds_store_at(uint32_t byte_offset, T value)
{
    *reinterpret_cast<__attribute__((address_space(3))) T*>(static_cast<uintptr_t>(byte_offset)) = value;
}
__device__ __forceinline__ bool lds_base_is_zero(const void* dynamic_lds)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)dynamic_lds)) == 0u;
}

// The 32-bit word of a key that holds the bit field starting at `shift` (fields never straddle the two
// halves of a 64-bit key: the sort's digits are 4-bit aligned); `hi` is wave-uniform.
__device__ __forceinline__ uint32_t field_word(uint32_t key, bool) { return key; }
__device__ __forceinline__ uint32_t field_word(uint64_t key, bool hi) { return hi ? static_cast<uint32_t>(key >> 32) : static_cast<uint32_t>(key); }

// Diagnostic build only (-DRSX_STAMPS, tools/stamp_probe.py): wave 0 of every workgroup writes the shader-cycle
// counter at the phase boundaries of reorder_kernel into a buffer of its own (16 words per tile) that no other
// code reads; the pointer travels in the otherwise unused `globsum` argument.  The product build has no stamp.
#ifdef RSX_STAMPS
#define RSX_STAMP(k)                                                                                      \
    do {                                                                                                  \
        if (stamp_buf && tid == 0) {                                                                      \
            unsigned long long t_;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            stamp_buf[static_cast<uint64_t>(tile) * 16 + (k)] = t_;                                       \
        }                                                                                                 \
    } while (0)
#else
#define RSX_STAMP(k) do { } while (0)
#endif

// ALIAS (optional, -DRSX_ALIAS_COUNTERS=1): the packed counters share LDS with the staging image (they are dead once every thread
// has read its 16 "first slot of my digit" values, which is when the image starts to fill; one extra barrier in between).  That
// takes a uint32 tile from 28 to 20 KiB and a uint64 tile from 46 to 37 KiB: 6 instead of 5, and 4 instead of 3, resident
// workgroups per CU.  Measured with six interleaved runs per build (the runs are bimodal, 2-3 % apart, so pairs mislead):
// uint32 3.67 / 3.76 ms against 3.62 / 3.71 without, uint64 13.15-13.43 against 12.92-13.34, uint64+payload 18.04-18.20 against
// 18.19-18.29 — the barrier costs more than the occupancy gives.  Off.
#ifndef RSX_ALIAS_COUNTERS
#define RSX_ALIAS_COUNTERS 0
#endif
template <typename Key, int THREADS, int KPT, bool ALIAS = (RSX_ALIAS_COUNTERS != 0)>
struct ReorderLayout {
    static constexpr int TILE = THREADS * KPT;
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int PADSH = (KD == 1) ? 5 : 4;
    static constexpr int XELEMS = TILE + (TILE >> PADSH);
    static constexpr int XBUF_DW = XELEMS * KD;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int CNT_AT = ALIAS ? 0 : XBUF_DW;                              // dword offset of the counters
    static constexpr int IMAGE_DW = ALIAS ? (XBUF_DW > CNT_DW ? XBUF_DW : CNT_DW) : XBUF_DW + CNT_DW;
    static constexpr int WTOT_DW = 16;
    static constexpr int GBASE_DW = 2 * kRadix;             // per digit {gbase, la_base}: one ds_read_b64
    static constexpr int LA_DW = kLaReplicas * (kRadix * 2 * kRadix + 8);  // look-ahead counters [digit][segment 0/1][next digit][replica] + dummies
    static constexpr int SELF_DW = (THREADS / kWave) * 2 * kRadix + kRadix;      // self-scan: per-wave partial sums + the 16 bases
    static constexpr int TOTAL_DW = IMAGE_DW + WTOT_DW + GBASE_DW + LA_DW + SELF_DW;
    static constexpr int TILE_SHIFT = __builtin_ctz(TILE);
    static_assert((TILE & (TILE - 1)) == 0, "tile size must be a power of two (slot -> output tile by shift)");
    static_assert(THREADS % (1 << PADSH) == 0, "the padded index of slot r*THREADS+t must split into a per-thread base and a constant");
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    // Workgroups one CU can hold by LDS (160 KiB) -> waves per SIMD the register allocator must leave room for (second
    // __launch_bounds__ argument = waves per SIMD, not blocks per CU); never asked beyond 6 (80 VGPRs: what the keys-only
    // kernels need; 8 would mean 64 and spills).
    static constexpr int WGS_PER_CU = static_cast<int>((160 * 1024) / BYTES);
#ifndef RSX_REORDER_WAVES_CAP
#define RSX_REORDER_WAVES_CAP 6      // 7 (72 VGPRs) measured twice, before and after the XCD stagger: see the tuning log
#endif
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > RSX_REORDER_WAVES_CAP ? RSX_REORDER_WAVES_CAP : (WGS_PER_CU * THREADS / 256);
    static_assert(TILE <= 32768, "16-bit packed counters");
    static_assert(KPT % (16 / sizeof(Key)) == 0 && THREADS % 64 == 0 && THREADS % 8 == 0, "geometry");
};

// Register budget: keys-only kernels are held to the occupancy LDS allows; payload kernels carry
// twice the per-key state (key, slot, payload, target) and are given 128 VGPRs instead of spilling.
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool RANGED = false>
constexpr int reorder_min_waves()
{
    constexpr int w = ReorderLayout<Key, THREADS, KPT, (!RANGED && RSX_ALIAS_COUNTERS != 0)>::MIN_WAVES;
    constexpr int cap = 4 * THREADS / 256;
    return (PAYLOAD && w > cap) ? cap : w;
}

// Self-scan (tables of at most kSelfScanMaxTiles tiles): there is no scan launch — `counts` holds the RAW counts of this
// pass as [tile][16] rows (written by the histogram kernel for the first pass, by the previous reorder's look-ahead
// afterwards) and every workgroup derives the 16 first slots of ITS tile itself while its keys are on their way:
// keys with a smaller digit anywhere + keys with the digit in earlier tiles.  Three count buffers rotate: this pass
// reads one, adds the next pass's counts into the second and zeroes its tile's row of the third.
constexpr int kSelfScanMaxTiles = 1024;
struct SelfScanArgs {
    const uint32_t* counts;      // nullptr: the table comes scanned (the normal path)
    uint32_t* zero_rows;
    uint32_t* table_out;         // last pass: leave the tile's 16 first slots in table[digit][tile] as the scan would
};

// LOOKAHEAD: while a key leaves for its slot g, the kernel also counts the key's NEXT
// digit for the output tile g / TILE — i.e. it builds the next pass's per-tile histogram
// (layout [tile][digit] in `next_counts`, zeroed by the host) without another pass over
// HBM.  A run (one digit of one source tile) covers at most two output tiles, so the
// counts are first gathered in LDS as [digit][segment 0/1][next digit] and then flushed
// with one global atomic per non-zero counter (16 consecutive lanes -> one 64-B segment).
// The LOOKAHEAD variant serves rsx_sort's passes only: its digit is exactly the 4-bit field at `shift`
// (mask 15) and the next digit the field at `next_shift`.  It works on RAW fields (no sign flip per
// key): the sign bit only ever toggles the top bit of the top digit, which is folded into where the
// counters, the run bases and the flushed counts are PLACED (flip_cur / flip_next below).
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool LOOKAHEAD, bool RANGED = false>
__global__ __launch_bounds__(THREADS, (reorder_min_waves<Key, THREADS, KPT, PAYLOAD, RANGED>())) void reorder_kernel(const Key* __restrict__ in, Key* __restrict__ out,
                                                           const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
                                                           const uint32_t* __restrict__ table, uint64_t n, uint32_t ntiles,
                                                           uint32_t tiles_per_xcd, int remap, int shift, Key flip, uint32_t mask,
                                                           uint32_t* __restrict__ next_counts, int next_shift,
                                                           const uint32_t* __restrict__ globsum, Key lo, Key mul,
                                                           SplitSet<Key> split, SelfScanArgs self)
{
    using L = ReorderLayout<Key, THREADS, KPT, (!RANGED && RSX_ALIAS_COUNTERS != 0)>;
    static_assert(!(RANGED && LOOKAHEAD), "the ranged bucket function is for the one-pass partition only");
    constexpr bool RAW = LOOKAHEAD;                 // digits are raw 4-bit fields; the sign flip lives in the placement
    constexpr int TILE = L::TILE;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    constexpr uint32_t CNT_ROW_BYTES = THREADS * 4;           // one [digit&7] row of packed counters

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* xbuf = smem;
    uint32_t* cnt = smem + L::CNT_AT;
    uint32_t* wtot = smem + L::IMAGE_DW;
    RunBase* runs = reinterpret_cast<RunBase*>(wtot + L::WTOT_DW);
    uint32_t* la = wtot + L::WTOT_DW + L::GBASE_DW;
    uint32_t* self_part = la + L::LA_DW;                                   // [wave][total / before][digit]
    uint32_t* self_base = self_part + (THREADS / kWave) * 2 * kRadix;      // [digit]

    const uint32_t tid = threadIdx.x;
    const uint32_t slot_tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap & ~2);
    if (slot_tile >= ntiles) {
        return;
    }
    if (!lds_base_is_zero(smem)) {
        __builtin_trap();           // lds_store_at addresses the staging image from LDS address 0
    }
    // bit 1 of `remap`: walk the tiles from the back (experiment: start with what the previous pass wrote last)
    const uint32_t tile = (remap & 2) ? ntiles - 1 - slot_tile : slot_tile;
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    const bool full = (valid == TILE);
    // Slots past `valid` hold a key whose digit is 15 in every pass; being last in index
    // order as well they land in local slots [valid, TILE) and are never stored.
    const Key pad_key = static_cast<Key>(~flip);
#ifdef RSX_STAMPS
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(globsum));
    globsum = nullptr;
    if (stamp_buf && tid == 0) {
        unsigned long long rt_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory");
        stamp_buf[static_cast<uint64_t>(tile) * 16 + 14] = rt_;
    }
    RSX_STAMP(0);
#endif

    // RAW placement constants (wave-uniform, scalar registers): which 32-bit word of the key holds the
    // digit, the field position inside it, and whether the sign bit is the digit's top bit.
    const bool hi_cur = sizeof(Key) == 8 && shift >= 32;
    const bool hi_next = sizeof(Key) == 8 && next_shift >= 32;
    const uint32_t sh = static_cast<uint32_t>(shift) & 31u;
    const uint32_t nsh = static_cast<uint32_t>(next_shift) & 31u;
    // RAW passes never sort by the digit that holds the sign bit (the most significant pass of a sort is always
    // its last one, which runs the plain variant), so a raw current digit IS the true digit; only the NEXT digit
    // may be the sign digit, and that is settled where the counts are flushed (flip_next).
    constexpr uint32_t flip_cur = 0u;
    const uint32_t flip_next = LOOKAHEAD ? static_cast<uint32_t>((flip >> next_shift) & Key{kRadix - 1}) : 0u;

    // digit of a key as phases 2, 4 and 5 index with it: RAW -> the raw field; otherwise the true digit / bucket
    auto dig = [=](Key key) -> uint32_t {
        if constexpr (RANGED) {
            if (split.n) {
                return splitter_bucket(static_cast<Key>(key ^ flip), split);
            }
            const uint32_t b = ranged_bucket(static_cast<Key>((key ^ flip) - lo), shift, mul, mask);
            return split.rot ? wave_major(b, split.rot) : b;
        } else if constexpr (RAW) {
            return __builtin_amdgcn_ubfe(field_word(key, hi_cur), sh, 4u);
        } else {
            return digit_of(key, shift, flip, mask);
        }
    };

    // The 8 raking threads whose first scan word belongs to thread 0 (digits hl and hl+8)
    // fetch table[digit][tile] for those two digits now, so the latency hides under the key loads.
    constexpr uint32_t RAKE_STRIDE = THREADS / 8;
    const bool rake_head = (tid % RAKE_STRIDE) == 0;
    const uint32_t hl = tid / RAKE_STRIDE;
    uint32_t first_lo = 0, first_hi = 0;
#ifndef RSX_SELF_SCAN_KERNEL
#define RSX_SELF_SCAN_KERNEL 1
#endif
    const bool self_scan = RSX_SELF_SCAN_KERNEL && !RANGED && self.counts != nullptr;      // wave-uniform
    if (rake_head && !self_scan) {
        const uint64_t e_lo = static_cast<uint64_t>(hl) * ntiles + tile;
        const uint64_t e_hi = static_cast<uint64_t>(hl + 8) * ntiles + tile;
        first_lo = table[e_lo];
        first_hi = table[e_hi];
        if (globsum) {
            // PasteHistogram folded in: the table holds block-local prefixes, add the scanned
            // sum of the scan group (256 tiles of one digit) each entry lives in (RadixSort.cl:185-197)
            const uint32_t ngroups = (ntiles + kScanTiles - 1) / kScanTiles;
            first_lo += globsum[static_cast<uint64_t>(hl) * ngroups + tile / kScanTiles];
            first_hi += globsum[static_cast<uint64_t>(hl + 8) * ngroups + tile / kScanTiles];
        }
    }

    // ---- 1. every lane fetches its own KPT consecutive keys (64 contiguous bytes, 16-byte loads) ----
    Key k[KPT];
    if (full) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const KeyVec<Key> v = load_keys16(in + base + tid * KPT + j * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                k[j * VEC + e] = v.k[e];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t li = tid * KPT + i;
            k[i] = li < valid ? in[base + li] : pad_key;
        }
    }
    // payload of the thread's blocked keys straight from HBM (64 B contiguous per lane)
    uint32_t pl[PAYLOAD ? KPT : 1];
    if constexpr (PAYLOAD) {
        if (full) {
#pragma unroll
            for (int q = 0; q < KPT / 4; ++q) {
                const U32x4 x = *reinterpret_cast<const U32x4*>(pin + base + tid * KPT + q * 4);
                pl[q * 4 + 0] = x.v[0];
                pl[q * 4 + 1] = x.v[1];
                pl[q * 4 + 2] = x.v[2];
                pl[q * 4 + 3] = x.v[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t li = tid * KPT + i;
                pl[i] = li < valid ? pin[base + li] : 0u;
            }
        }
    }

    if (self_scan) {
        // (the key loads above are in flight; this is L2-resident table work under their latency)
        // thread (q = tid & 3, r = tid >> 2) reads digits 4q..4q+3 of the rows r, r + 64, ... with 16-byte loads: a wave covers
        // 16 rows per instruction and a table of 1024 tiles is 4 rounds of 4 loads in flight
        const uint32_t q = tid & 3u, r = tid >> 2;
        constexpr uint32_t RS = THREADS / 4;
        const U32x4* rows = reinterpret_cast<const U32x4*>(self.counts);
        uint32_t tot[4] = {0u, 0u, 0u, 0u}, pre[4] = {0u, 0u, 0u, 0u};
        uint32_t t2 = r;
        for (; t2 + 3u * RS < ntiles; t2 += 4u * RS) {
            U32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = rows[(t2 + static_cast<uint32_t>(u) * RS) * 4u + q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool earlier = t2 + static_cast<uint32_t>(u) * RS < tile;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    tot[c] += v[u].v[c];
                    pre[c] += earlier ? v[u].v[c] : 0u;
                }
            }
        }
        for (; t2 < ntiles; t2 += RS) {
            const U32x4 v = rows[t2 * 4u + q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tot[c] += v.v[c];
                pre[c] += (t2 < tile) ? v.v[c] : 0u;
            }
        }
        // the 16 lanes of a wave with the same q: lanes q, q+4, q+8, q+12 of each row of 16 (row_ror:4, row_ror:8), then the four rows
        auto same_q_sum = [](uint32_t x) -> uint32_t {
            x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x124, 0xf, 0xf, false));
            x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x128, 0xf, 0xf, false));
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            return x;
        };
        const uint32_t lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tot[c] = same_q_sum(tot[c]);
            pre[c] = same_q_sum(pre[c]);
        }
        if (lane < 4u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                self_part[(wave * 2 + 0) * kRadix + lane * 4u + static_cast<uint32_t>(c)] = tot[c];
                self_part[(wave * 2 + 1) * kRadix + lane * 4u + static_cast<uint32_t>(c)] = pre[c];
            }
        }
        __syncthreads();
        if (tid < kRadix) {
            uint32_t total = 0, before = 0;
#pragma unroll
            for (int w = 0; w < THREADS / kWave; ++w) {
                total += self_part[(w * 2 + 0) * kRadix + tid];
                before += self_part[(w * 2 + 1) * kRadix + tid];
            }
            const uint32_t first = wave_inclusive_scan(total) - total + before;     // smaller digits anywhere + this digit in earlier tiles
            self_base[tid] = first;
            self.zero_rows[tile * kRadix + tid] = 0;
            if (self.table_out) {
                self.table_out[static_cast<uint64_t>(tid) * ntiles + tile] = first;
            }
        }
        __syncthreads();
        if (rake_head) {
            first_lo = self_base[hl];
            first_hi = self_base[hl + 8];
        }
    }

    // ---- 2. each thread = one virtual processor: KPT consecutive keys, private counters ----------
    // The 16 digit counters of a thread live in ONE 64-bit register while it ranks its keys (nibble d = keys
    // seen so far with digit d; at most KPT-1 = 15 before the last key, so a nibble never overflows) and reach
    // LDS only once, as the 8 packed words of the raking scan.  Counting in LDS instead — read, add, write per
    // key on a counter that the next key may hit again — is a chain of 16 dependent LDS round trips: measured
    // with in-kernel stamps it was 5,100-5,800 of a tile's 17,000 cycles of residency, and the kernel's
    // throughput is residency-bound (4.7 tiles per CU in flight).
    static_assert(KPT <= 16, "nibble counters: a thread's count of one digit must fit 4 bits before its last key");
    u32_alias* cnt32 = reinterpret_cast<u32_alias*>(cnt);
    if constexpr (LOOKAHEAD) {
        for (uint32_t c = tid; c < static_cast<uint32_t>(L::LA_DW); c += THREADS) {
            la[c] = 0;
        }
    }
#ifdef RSX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSX_STAMP(1);
#endif
    // RANGED: the bucket function costs tens of instructions per key, so it is evaluated once:
    // the thread's 16 buckets are kept as nibbles, and travel to step 5 as bytes next to the
    // staged keys (in the counter area, which is free by then)
    uint32_t nib[RANGED ? KPT / 8 : 1];
    auto bucket_at = [&](int i) -> uint32_t {
        if constexpr (RANGED) {
            return (nib[i >> 3] >> ((i & 7) * 4)) & 15u;
        } else {
            return dig(k[i]);
        }
    };
    if constexpr (RANGED) {
#pragma unroll
        for (int w = 0; w < KPT / 8; ++w) {
            nib[w] = 0;
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            nib[i >> 3] |= dig(k[i]) << ((i & 7) * 4);
        }
    }
    uint32_t slot[KPT];      // first: rank among the thread's own equal-digit keys; later: tile-local slot
    {
        uint64_t seen = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t sh4 = bucket_at(i) << 2;
            slot[i] = static_cast<uint32_t>(seen >> sh4) & 15u;
#if RSX_EARLY_RANK
            asm volatile("" : "+v"(slot[i]));      // materialise the rank now: otherwise hipcc keeps all 16 intermediate `seen` values (32 VGPRs) and extracts the ranks after the loop
#endif
            if (i + 1 < KPT) {
                seen += 1ull << sh4;
            }
        }
        const uint32_t seen_lo = static_cast<uint32_t>(seen), seen_hi = static_cast<uint32_t>(seen >> 32);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            // word [l][tid]: digit l in the low half, digit l+8 in the high half
            cnt32[l * THREADS + tid] = __builtin_amdgcn_ubfe(seen_lo, 4u * l, 4u) | (__builtin_amdgcn_ubfe(seen_hi, 4u * l, 4u) << 16);
        }
        // the last key: one add without return on the thread's own word, behind the store above (LDS
        // operations of a wave execute in order)
        const uint32_t d_last = bucket_at(KPT - 1);
        atomicAdd(reinterpret_cast<uint32_t*>(cnt) + (d_last & 7u) * THREADS + tid, 1u << ((d_last >> 3) * 16u));
    }
    // Byte address (inside the counter area) of the 16-bit counter of (digit d, this thread):
    // word [d&7][tid], half d>>3  ->  (d&7) * THREADS*4 + tid*4 + (d>>3)*2.  RAW: two bit-field extracts and
    // two shift-adds per key (LOOKAHEAD passes never sort by the sign digit: raw digit = true digit).
    unsigned char* cbytes = reinterpret_cast<unsigned char*>(cnt);
    auto counter_at = [&](int i) -> u16_alias* {
        if constexpr (RAW) {
            const uint32_t w = field_word(k[i], hi_cur);
            const uint32_t l3 = __builtin_amdgcn_ubfe(w, sh, 3u);
            const uint32_t h = __builtin_amdgcn_ubfe(w, sh + 3u, 1u);
            return reinterpret_cast<u16_alias*>(cbytes + (l3 * CNT_ROW_BYTES + tid * 4u) + h * 2u);
        } else {
            const uint32_t d = bucket_at(i);
            return reinterpret_cast<u16_alias*>(cbytes + ((d & 7u) * CNT_ROW_BYTES + tid * 4u) + (d >> 3) * 2u);
        }
    };
    RSX_STAMP(2);
    __syncthreads();
    RSX_STAMP(3);

    // ---- 3. raking scan over the 8*THREADS packed words in [digit&7][thread] order ---
    {
        U32x4 a = *reinterpret_cast<const U32x4*>(cnt + tid * 8);
        U32x4 b = *reinterpret_cast<const U32x4*>(cnt + tid * 8 + 4);
        const uint32_t sum = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
        uint32_t total;
        uint32_t run = block_exclusive_scan<THREADS>(sum, wtot, total);
        // low halves now prefix digits 0..7, high halves digits 8..15; the latter start
        // after ALL keys with digit < 8, i.e. after total.low
        run += total << 16;
        if (rake_head) {
            // `run` is the scanned word of (true digit hl | hl+8, thread 0): the tile-local slot of the
            // tile's first key with that digit.  Stored where phase 5 looks it up: at the RAW digit.
            const uint32_t g_lo = first_lo - (run & 0xFFFFu), g_hi = first_hi - (run >> 16);
            const uint32_t r_lo = hl ^ flip_cur, r_hi = (hl + 8u) ^ flip_cur;
            runs[r_lo] = RunBase{g_lo, (r_lo << 5) - ((first_lo >> L::TILE_SHIFT) << 4)};
            runs[r_hi] = RunBase{g_hi, (r_hi << 5) - ((first_hi >> L::TILE_SHIFT) << 4)};
        }
        uint32_t t;
        t = a.v[0]; a.v[0] = run; run += t;
        t = a.v[1]; a.v[1] = run; run += t;
        t = a.v[2]; a.v[2] = run; run += t;
        t = a.v[3]; a.v[3] = run; run += t;
        t = b.v[0]; b.v[0] = run; run += t;
        t = b.v[1]; b.v[1] = run; run += t;
        t = b.v[2]; b.v[2] = run; run += t;
        t = b.v[3]; b.v[3] = run;
        *reinterpret_cast<U32x4*>(cnt + tid * 8) = a;
        *reinterpret_cast<U32x4*>(cnt + tid * 8 + 4) = b;
    }
    __syncthreads();
    RSX_STAMP(4);

    // ---- 4. tile-local slot of every key; stage the tile in sorted order -------------
    // Written as "all reads, then all writes" on purpose: the compiler cannot prove that the
    // staging writes do not alias the counters, so a fused loop waits for every LDS read
    // before the next one is issued (16 exposed LDS latencies instead of one).
    Key* xk = reinterpret_cast<Key*>(xbuf);
    {
        uint32_t first_of_digit[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            first_of_digit[i] = *counter_at(i);
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            slot[i] += first_of_digit[i];
        }
        if constexpr (L::CNT_AT == 0) {
            __syncthreads();                 // the image overlays the counters: nobody may still be reading them
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            // xk[slot + (slot >> PADSH)] = k  (xbuf is the first thing in the workgroup's LDS)
            lds_store_at<Key>(add_lshl<(sizeof(Key) == 4 ? 2 : 3)>(slot[i], slot[i] >> L::PADSH), k[i]);
        }
        if constexpr (RANGED) {
            __syncthreads();                 // every thread has read its counters: reuse the area
            unsigned char* staged_bucket = reinterpret_cast<unsigned char*>(cnt);
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                staged_bucket[slot[i]] = static_cast<unsigned char>(bucket_at(i));
            }
        }
    }
    __syncthreads();
    RSX_STAMP(5);

    // ---- 5. leave as runs: consecutive lanes -> consecutive addresses inside a run ---
    // Same batching: 16 key reads in flight, then 16 run-base reads, then 16 stores.  Slot
    // i = r*THREADS + tid sits at padded index i + (i >> PADSH) = (tid + (tid >> PADSH)) + r*RSTRIDE: one
    // per-thread base and compile-time offsets, no address arithmetic per key.
    constexpr uint32_t RSTRIDE = THREADS + (THREADS >> L::PADSH);
    const uint32_t rd_base = tid + (tid >> L::PADSH);
    Key okey[KPT];
    uint32_t g[KPT];
    uint32_t la_idx[LOOKAHEAD ? KPT : 1];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
        okey[r] = xk[rd_base + static_cast<uint32_t>(r) * RSTRIDE];
    }
    {
        RunBase rb[KPT];
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            if constexpr (RANGED) {
                rb[r] = runs[reinterpret_cast<const unsigned char*>(cnt)[static_cast<uint32_t>(r) * THREADS + tid]];
            } else {
                rb[r] = runs[dig(okey[r])];
            }
        }
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            g[r] = rb[r].gbase + tid + static_cast<uint32_t>(r) * THREADS;
            if constexpr (LOOKAHEAD) {
                // counter [digit][segment][next digit]: la_base = (digit << 5) - (first output tile << 4)
                la_idx[r] = rb[r].la_base + ((g[r] >> L::TILE_SHIFT) << 4) + __builtin_amdgcn_ubfe(field_word(okey[r], hi_next), nsh, 4u);
            }
        }
    }
    RSX_STAMP(6);
    // keys leave first, then the look-ahead counts: both free their registers before the
    // payload takes its own trip through the staging image
    if (full) {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            out[g[r]] = okey[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            const uint32_t i = static_cast<uint32_t>(r) * THREADS + tid;
            if (i < valid) {
                out[g[r]] = okey[r];
            } else if constexpr (LOOKAHEAD) {
                la_idx[r] = kLaDummy;
            }
        }
    }
    RSX_STAMP(7);
    if constexpr (LOOKAHEAD) {
        // Wave-uniform counters (constant or sorted data) must not become 64 lanes serialising on one LDS
        // address, but testing every key for it costs a scalar branch and an LDS drain per key.  Round 0
        // stands for the wave: where its 64 slots already disagree (any data with entropy in these two
        // digits) the other rounds simply add; otherwise every round is tested.
        const uint32_t first0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(la_idx[0])));
        if (__builtin_expect(__ballot(la_idx[0] != first0) != 0ull, 1)) {
            constexpr int REPL = la_replicas<Key, PAYLOAD>();
            uint32_t* la_mine = la + (tid & (REPL - 1));             // this lane's copy of every counter
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                atomicAdd(&la_mine[la_idx[r] * REPL], 1u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                lookahead_count<la_replicas<Key, PAYLOAD>()>(la, la_idx[r]);
            }
        }
    }
    if constexpr (PAYLOAD) {
        __syncthreads();      // every wave has read its keys: the image may be overwritten
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            lds_store_at<uint32_t>(add_lshl<2>(slot[i], slot[i] >> 5), pl[i]);
        }
        __syncthreads();
        constexpr uint32_t PSTRIDE = THREADS + (THREADS >> 5);
        const uint32_t pd_base = tid + (tid >> 5);
        uint32_t pay[KPT];
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            pay[r] = xbuf[pd_base + static_cast<uint32_t>(r) * PSTRIDE];
        }
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            const uint32_t i = static_cast<uint32_t>(r) * THREADS + tid;
            if (full || i < valid) {
                pout[g[r]] = pay[r];
            }
        }
    }
    RSX_STAMP(8);
    if constexpr (LOOKAHEAD) {
        __syncthreads();
        RSX_STAMP(9);
        // an opaque copy of the thread id: otherwise the compiler shares `tid >> 5` address arithmetic with
        // the ranking phase, keeps it alive through the whole kernel and spills it at the 96-VGPR budget
        uint32_t first = tid;
        asm volatile("" : "+v"(first));
        for (uint32_t c = first; c < kLaDummy; c += THREADS) {
            constexpr int REPL = la_replicas<Key, PAYLOAD>();
            uint32_t v = la[c * REPL];
            if constexpr (REPL == 2) {
                v += la[c * REPL + 1];
            }
            if (v) {
                // counter c = [raw digit d][segment][raw next digit]; the counts table is indexed by the TRUE next digit
                const uint32_t d = c >> 5, seg = (c >> 4) & 1u, d2 = (c & 15u) ^ flip_next;
                const uint32_t run_tile = ((d << 5) - runs[d].la_base) >> 4;
                atomicAdd(&next_counts[static_cast<uint64_t>(run_tile + seg) * kRadix + d2], v);
            }
        }
    }
#ifdef RSX_STAMPS
    RSX_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // all stores and atomics of wave 0 acknowledged
    RSX_STAMP(11);
    if (stamp_buf && tid == 0) {
        unsigned long long rt_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory");
        stamp_buf[static_cast<uint64_t>(tile) * 16 + 15] = rt_;
    }
#endif
}

// ---------------------------------------------------------------------------
// tile_sort: inputs of at most ONE tile, every pass inside LDS, one launch
// ---------------------------------------------------------------------------
// The reference publishes timings from 2^1 keys upwards (Performance/performance.csv); for such inputs the
// pass chain above is 2 launches per pass whose cost is pure latency.  Here one workgroup keeps the keys in
// registers (thread t = the KPT consecutive keys 16t.. of the current order), ranks them exactly like
// reorder_kernel does (nibble counters in a register, packed [digit&7][thread] words, raking DPP scan),
// scatters them into an LDS image in sorted order and reads its next KPT consecutive keys back — the same
// stable pass, `last_pass - first_pass` times, with no HBM traffic in between.  The keys as they stood before
// the LAST pass are also written out (`before_last`): that buffer is what rsx_download's reference-geometry
// diagnostics recompute from, exactly as after a multi-launch sort; the [digit][tile] table (one tile: the 16
// digit starts), the group sums and the grand total of the last pass are written as the chain would leave them.
template <typename Key, int THREADS, int KPT>
struct TileSortLayout {
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int ROW_DW = KPT * KD + 4;                 // one thread's KPT keys + 16 bytes: rows stay 16-byte aligned and
                                                                // 16 consecutive rows start on 16 different bank quads
    static constexpr int XBUF_DW = THREADS * ROW_DW;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int TOTAL_DW = XBUF_DW + CNT_DW + 16 + kRadix;
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    static_assert(ROW_DW % 4 == 0 && KPT == 16, "row geometry (slot >> 4 selects the row)");
};

template <typename Key, int THREADS, int KPT, bool PAYLOAD>
__global__ __launch_bounds__(THREADS) void tile_sort_kernel(const Key* in, Key* out, Key* before_last,        // may alias each other: everything
                                                            const uint32_t* pin, uint32_t* pout, uint32_t* pbefore_last,            // is read before anything is written
                                                            uint32_t n, int first_pass, int last_pass,
                                                            Key flip, uint32_t* __restrict__ table, uint32_t* __restrict__ globsum,
                                                            uint32_t* __restrict__ temp)
{
    using L = TileSortLayout<Key, THREADS, KPT>;
    constexpr int KD = L::KD;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* xbuf = smem;
    uint32_t* cnt = smem + L::XBUF_DW;
    uint32_t* wtot = cnt + L::CNT_DW;
    uint32_t* dstart = wtot + 16;                // tile-local first slot of every digit (last pass: the table)
    const uint32_t tid = threadIdx.x;
    const Key pad_key = static_cast<Key>(~flip);  // digit 15 in every pass: pads stay behind every real key
    // dword index of slot s in the image: row s/KPT, KPT keys per row, 4 dwords of padding per row
    auto image_dw = [](uint32_t s) { return s * KD + ((s >> 4) << 2); };

    Key k[KPT];
    uint32_t pl[PAYLOAD ? KPT : 1];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t li = tid * KPT + i;
        k[i] = li < n ? in[li] : pad_key;
        if constexpr (PAYLOAD) {
            pl[i] = li < n ? pin[li] : 0u;
        }
    }
    u32_alias* cnt32 = reinterpret_cast<u32_alias*>(cnt);
    const u16_alias* cnt16 = reinterpret_cast<const u16_alias*>(cnt);
    constexpr uint32_t RAKE_STRIDE = THREADS / 8;
    const bool rake_head = (tid % RAKE_STRIDE) == 0;
    const uint32_t hl = tid / RAKE_STRIDE;

#pragma unroll 1
    for (int pass = first_pass; pass < last_pass; ++pass) {
        const int shift = pass * kRadixBits;
        const bool last = pass + 1 == last_pass;
        if (last && before_last && pass > first_pass) {
            // the order the last pass starts from: thread t holds slots 16t .. 16t+15
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t li = tid * KPT + i;
                if (li < n) {
                    before_last[li] = k[i];
                    if constexpr (PAYLOAD) {
                        pbefore_last[li] = pl[i];
                    }
                }
            }
        }
        // ranks among the thread's own keys + its 16 counters (nibbles of one 64-bit register)
        uint32_t slot[KPT], dg[KPT];
        uint64_t seen = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            dg[i] = digit_of(k[i], shift, flip, static_cast<uint32_t>(kRadix - 1));
            const uint32_t sh4 = dg[i] << 2;
            slot[i] = static_cast<uint32_t>(seen >> sh4) & 15u;
            if (i + 1 < KPT) {
                seen += 1ull << sh4;
            }
        }
        const uint32_t seen_lo = static_cast<uint32_t>(seen), seen_hi = static_cast<uint32_t>(seen >> 32);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            cnt32[l * THREADS + tid] = __builtin_amdgcn_ubfe(seen_lo, 4u * l, 4u) | (__builtin_amdgcn_ubfe(seen_hi, 4u * l, 4u) << 16);
        }
        atomicAdd(cnt + (dg[KPT - 1] & 7u) * THREADS + tid, 1u << ((dg[KPT - 1] >> 3) * 16u));
        __syncthreads();
        {
            U32x4 a = *reinterpret_cast<const U32x4*>(cnt + tid * 8);
            U32x4 b = *reinterpret_cast<const U32x4*>(cnt + tid * 8 + 4);
            const uint32_t sum = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
            uint32_t total;
            uint32_t run = block_exclusive_scan<THREADS>(sum, wtot, total);
            run += total << 16;
            if (rake_head) {
                dstart[hl] = run & 0xFFFFu;
                dstart[hl + 8] = run >> 16;
            }
            uint32_t t;
            t = a.v[0]; a.v[0] = run; run += t;
            t = a.v[1]; a.v[1] = run; run += t;
            t = a.v[2]; a.v[2] = run; run += t;
            t = a.v[3]; a.v[3] = run; run += t;
            t = b.v[0]; b.v[0] = run; run += t;
            t = b.v[1]; b.v[1] = run; run += t;
            t = b.v[2]; b.v[2] = run; run += t;
            t = b.v[3]; b.v[3] = run;
            *reinterpret_cast<U32x4*>(cnt + tid * 8) = a;
            *reinterpret_cast<U32x4*>(cnt + tid * 8 + 4) = b;
        }
        __syncthreads();
        {
            uint32_t first_of_digit[KPT];
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                first_of_digit[i] = cnt16[(((dg[i] & 7u) * THREADS + tid) << 1) + (dg[i] >> 3)];
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                slot[i] += first_of_digit[i];
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                *reinterpret_cast<Key*>(xbuf + image_dw(slot[i])) = k[i];
            }
        }
        if (last && tid < kRadix) {
            // what the chain's scan + paste leave behind for one tile: table[d][0], the scanned group sums, the total
            // (pad keys of a partial tile count as digit 15 locally but are not keys: starts are clamped to n)
            const uint32_t s = dstart[tid] < n ? dstart[tid] : n;
            table[tid] = s;
            globsum[tid] = s;
            if (tid == 0) {
                temp[0] = n;
            }
        }
        __syncthreads();
        if (!last) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const KeyVec<Key> v = *reinterpret_cast<const KeyVec<Key>*>(xbuf + tid * L::ROW_DW + j * 4);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    k[j * VEC + e] = v.k[e];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                const uint32_t i = static_cast<uint32_t>(r) * THREADS + tid;
                if (i < n) {
                    out[i] = *reinterpret_cast<const Key*>(xbuf + image_dw(i));
                }
            }
        }
        if constexpr (PAYLOAD) {
            __syncthreads();           // every thread has taken its keys: the image carries the payload now
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                xbuf[slot[i] + ((slot[i] >> 4) << 2)] = pl[i];
            }
            __syncthreads();
            if (!last) {
#pragma unroll
                for (int q = 0; q < KPT / 4; ++q) {
                    const U32x4 x = *reinterpret_cast<const U32x4*>(xbuf + tid * (KPT + 4) + q * 4);
                    pl[q * 4 + 0] = x.v[0];
                    pl[q * 4 + 1] = x.v[1];
                    pl[q * 4 + 2] = x.v[2];
                    pl[q * 4 + 3] = x.v[3];
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    const uint32_t i = static_cast<uint32_t>(r) * THREADS + tid;
                    if (i < n) {
                        pout[i] = xbuf[i + ((i >> 4) << 2)];
                    }
                }
            }
        }
        __syncthreads();               // the image and the counters are free for the next pass
    }
}

// ---------------------------------------------------------------------------
// 8-bit digits: half the passes (RSX_OPT_RADIX_BITS = 8, reported separately from the 4-bit configuration)
// ---------------------------------------------------------------------------
// The reference's digit width is a parameter (_NUM_BITS_PER_RADIX, src/Parameters.h:25, pushed into the kernels at
// src/RadixSortGPU.cpp:569-584).  A pass over an 8-bit digit is built from the 4-bit machinery above: the tile is
// sorted locally by the low nibble and then by the high nibble of the digit — two stable rounds through LDS, the
// second one starting from 16 consecutive keys of the first one's order per thread — and leaves as up to 256 runs.
// Tables are [tile][256] (a tile's 256 counters are one contiguous 1 KiB row):
//   histogram8_kernel   counts8[tile][d]   = keys of the tile with digit d
//   scan8_blocks_kernel table8[tile][d]    = keys with digit d in EARLIER tiles of the tile's group (G tiles); gsum8[group][d] = group total
//   scan8_chunks_kernel gsum8[group][d]    = keys with digit d in earlier groups of the group's chunk; csum8[chunk][d] = chunk total
//   reorder8_kernel     slot of a key      = (keys with smaller digits) + (digit d in earlier chunks) + gsum8[group][d] + table8[tile][d]
//                                            + (its rank inside the tile's run of digit d)
constexpr int kRadix8 = 256;
constexpr int kScan8Tiles = 64;               // tiles per scan group

template <typename Key>
__device__ __forceinline__ uint32_t digit8_of(Key key, int shift, Key flip)
{
    return static_cast<uint32_t>((key ^ flip) >> shift) & 255u;
}

template <typename Key, int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void histogram8_kernel(const Key* __restrict__ keys, uint32_t* __restrict__ counts8, uint64_t n, uint32_t ntiles,
                                                              uint32_t tiles_per_xcd, int remap, int shift, Key flip)
{
    static_assert(THREADS == kRadix8, "one thread per digit writes the tile's row");
    constexpr int TILE = THREADS * KPT;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    __shared__ uint32_t cnt[kRadix8];
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    cnt[tid] = 0;
    __syncthreads();
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    if (valid == TILE) {
        KeyVec<Key> v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = load_keys16(keys + base + static_cast<uint32_t>(j) * THREADS * VEC + tid * VEC);
        }
        // a wave whose keys all share the digit (constant or sorted data) would serialise 64 lanes on one LDS
        // address per key: the first key stands for the wave, as in reorder_kernel's look-ahead
        const uint32_t d0 = digit8_of(v[0].k[0], shift, flip);
        const bool spread = __ballot(d0 != static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(d0)))) != 0ull;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const uint32_t d = digit8_of(v[j].k[e], shift, flip);
                if (spread) {
                    atomicAdd(&cnt[d], 1u);
                } else {
                    const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(d)));
                    if (__ballot(d != first) == 0ull) {
                        if ((tid & (kWave - 1)) == 0) {
                            atomicAdd(&cnt[first], static_cast<uint32_t>(kWave));
                        }
                    } else {
                        atomicAdd(&cnt[d], 1u);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const uint32_t li = static_cast<uint32_t>(j) * THREADS * VEC + tid * VEC + e;
                if (li < valid) {
                    atomicAdd(&cnt[digit8_of(keys[base + li], shift, flip)], 1u);
                }
            }
        }
    }
    __syncthreads();
    counts8[static_cast<uint64_t>(tile) * kRadix8 + tid] = cnt[tid];
}

// one workgroup per group of kScan8Tiles tiles; thread d walks the group's rows (1 KiB each, coalesced)
__global__ __launch_bounds__(kRadix8) void scan8_blocks_kernel(const uint32_t* __restrict__ counts8, uint32_t* __restrict__ table8, uint32_t* __restrict__ gsum8,
                                                                uint32_t ntiles)
{
    const uint32_t d = threadIdx.x, group = blockIdx.x;
    const uint32_t t0 = group * kScan8Tiles;
    const uint32_t t1 = t0 + kScan8Tiles < ntiles ? t0 + kScan8Tiles : ntiles;
    uint32_t run = 0;
    uint32_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c[u] = counts8[static_cast<uint64_t>(t + u) * kRadix8 + d];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            table8[static_cast<uint64_t>(t + u) * kRadix8 + d] = run;
            run += c[u];
        }
    }
    for (; t < t1; ++t) {
        const uint32_t c = counts8[static_cast<uint64_t>(t) * kRadix8 + d];
        table8[static_cast<uint64_t>(t) * kRadix8 + d] = run;
        run += c;
    }
    gsum8[static_cast<uint64_t>(group) * kRadix8 + d] = run;
}

// Second level: the groups are cut into at most kScan8MaxChunks chunks of `chunk_groups` consecutive groups; one
// workgroup per chunk turns its groups' totals into exclusive prefixes INSIDE the chunk (thread d walks the rows,
// eight loads in flight) and leaves the chunk total in csum8[chunk][d].  Third level (scan8_top_kernel, one
// workgroup, a few microseconds): cbase8[chunk][d] = keys with a smaller digit + keys with digit d in earlier chunks.
constexpr int kScan8MaxChunks = 16;

__global__ __launch_bounds__(kRadix8) void scan8_top_kernel(const uint32_t* __restrict__ csum8, uint32_t* __restrict__ cbase8, uint32_t* __restrict__ temp,
                                                            uint32_t nchunks)
{
    __shared__ uint32_t wtot[kRadix8 / kWave];
    const uint32_t d = threadIdx.x;
    uint32_t cs[kScan8MaxChunks];
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kScan8MaxChunks; ++w) {
        cs[w] = static_cast<uint32_t>(w) < nchunks ? csum8[w * kRadix8 + d] : 0u;
        total += cs[w];
    }
    uint32_t all;
    uint32_t run = block_exclusive_scan<kRadix8>(total, wtot, all);
#pragma unroll
    for (int w = 0; w < kScan8MaxChunks; ++w) {
        if (static_cast<uint32_t>(w) < nchunks) {
            cbase8[w * kRadix8 + d] = run;
        }
        run += cs[w];
    }
    if (d == 0) {
        temp[0] = all;                        // grand total, as the 4-bit scan leaves it
    }
}

// ONLY_CHUNK (a table of one chunk, i.e. up to 2^24 keys): the workgroup is also the top level — the digit bases
// go straight to cbase8[0][d] and the grand total to temp[0]; no scan8_top_kernel launch.
template <bool ONLY_CHUNK>
__global__ __launch_bounds__(kRadix8) void scan8_chunks_kernel(uint32_t* __restrict__ gsum8, uint32_t* __restrict__ csum8, uint32_t ngroups, uint32_t chunk_groups,
                                                               uint32_t* __restrict__ cbase8, uint32_t* __restrict__ temp)
{
    __shared__ uint32_t wtot[kRadix8 / kWave];
    const uint32_t d = threadIdx.x, chunk = blockIdx.x;
    const uint32_t g0 = chunk * chunk_groups;
    const uint32_t g1 = g0 + chunk_groups < ngroups ? g0 + chunk_groups : ngroups;
    uint32_t run = 0;
    uint32_t g = g0;
    for (; g + 8 <= g1; g += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c[u] = gsum8[static_cast<uint64_t>(g + u) * kRadix8 + d];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            gsum8[static_cast<uint64_t>(g + u) * kRadix8 + d] = run;
            run += c[u];
        }
    }
    for (; g < g1; ++g) {
        const uint32_t c = gsum8[static_cast<uint64_t>(g) * kRadix8 + d];
        gsum8[static_cast<uint64_t>(g) * kRadix8 + d] = run;
        run += c;
    }
    csum8[static_cast<uint64_t>(chunk) * kRadix8 + d] = run;
    if constexpr (ONLY_CHUNK) {
        uint32_t all;
        cbase8[d] = block_exclusive_scan<kRadix8>(run, wtot, all);
        if (d == 0) {
            temp[0] = all;
        }
    }
}

template <typename Key, int THREADS, int KPT>
struct Reorder8Layout {
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int ROW_DW = KPT * KD + 4;                 // as TileSortLayout: 16-byte aligned rows on distinct bank quads
    static constexpr int XBUF_DW = THREADS * ROW_DW;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int TOTAL_DW = XBUF_DW + CNT_DW + 16 + kRadix8;
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    static constexpr int WGS_PER_CU = static_cast<int>((160 * 1024) / BYTES);
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > 8 ? 8 : (WGS_PER_CU * THREADS / 256);
    static_assert(KPT == 16 && THREADS == kRadix8, "row geometry; one thread per digit handles the tile's table row");
};

template <typename Key, int THREADS, int KPT, bool PAYLOAD>
__global__ __launch_bounds__(THREADS, (PAYLOAD ? 2 : (Reorder8Layout<Key, THREADS, KPT>::MIN_WAVES > 4 ? 4 : Reorder8Layout<Key, THREADS, KPT>::MIN_WAVES))) void reorder8_kernel(
    const Key* __restrict__ in, Key* __restrict__ out, const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
    const uint32_t* __restrict__ counts8, const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
    uint32_t chunk_groups, uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd, int remap, int shift, Key flip)
{
    using L = Reorder8Layout<Key, THREADS, KPT>;
    constexpr int TILE = THREADS * KPT;
    constexpr int KD = L::KD;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    constexpr uint32_t CNT_ROW_BYTES = THREADS * 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* xbuf = smem;
    uint32_t* cnt = smem + L::XBUF_DW;
    uint32_t* wtot = cnt + L::CNT_DW;
    uint32_t* gb = wtot + 16;                     // per 8-bit digit: (global slot of the tile's first key with it) - (its local slot)
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap & ~2);
    if (tile >= ntiles) {
        return;
    }
    if (!lds_base_is_zero(smem)) {
        __builtin_trap();           // lds_store_at addresses the image from LDS address 0
    }
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    const bool full = (valid == TILE);
    // Inside the kernel keys are held with the sign bit flipped (k ^ flip: unsigned order = numeric order), so the
    // digits are plain bit fields; the flip is undone on the way out.  Unsigned types skip both (flip == 0, uniform).
    const Key pad_key = static_cast<Key>(~Key{0});        // digit 255, behind every real key of the tile
    const bool hi = sizeof(Key) == 8 && shift >= 32;      // the byte never straddles the halves of a 64-bit key
    const uint32_t sh = static_cast<uint32_t>(shift) & 31u;

    // this thread's digit of the tile's table row (latency hides under the key loads)
    const uint32_t my_count = counts8[static_cast<uint64_t>(tile) * kRadix8 + tid];
    const uint32_t group = tile / kScan8Tiles;
    const uint32_t my_first = table8[static_cast<uint64_t>(tile) * kRadix8 + tid] + gsum8[static_cast<uint64_t>(group) * kRadix8 + tid] +
                              cbase8[static_cast<uint64_t>(group / chunk_groups) * kRadix8 + tid];      // smaller digits + this digit in earlier chunks

    Key k[KPT];
    uint32_t pl[PAYLOAD ? KPT : 1];
    if (full) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const KeyVec<Key> v = load_keys16(in + base + tid * KPT + j * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                k[j * VEC + e] = v.k[e];
            }
        }
        if constexpr (PAYLOAD) {
#pragma unroll
            for (int q = 0; q < KPT / 4; ++q) {
                const U32x4 x = *reinterpret_cast<const U32x4*>(pin + base + tid * KPT + q * 4);
                pl[q * 4 + 0] = x.v[0];
                pl[q * 4 + 1] = x.v[1];
                pl[q * 4 + 2] = x.v[2];
                pl[q * 4 + 3] = x.v[3];
            }
        }
        if (flip != Key{0}) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                k[i] ^= flip;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t li = tid * KPT + i;
            k[i] = li < valid ? static_cast<Key>(in[base + li] ^ flip) : pad_key;
            if constexpr (PAYLOAD) {
                pl[i] = li < valid ? pin[base + li] : 0u;
            }
        }
    }
    // local first slot of every digit = exclusive scan of the tile's 256 counts; gb = global first - local first
    {
        uint32_t all;
        const uint32_t local_first = block_exclusive_scan<THREADS>(my_count, wtot, all);
        gb[tid] = my_first - local_first;
    }
    u32_alias* cnt32 = reinterpret_cast<u32_alias*>(cnt);
    unsigned char* cbytes = reinterpret_cast<unsigned char*>(cnt);
    // image: slot s at dword s*KD + 4*(s/16) (rows of KPT keys + 16 bytes); slot i = r*THREADS + tid -> per-thread base + r * OUT_STRIDE
    constexpr uint32_t OUT_STRIDE_DW = THREADS * KD + (THREADS / 16) * 4;
    const uint32_t out_base_dw = tid * KD + ((tid >> 4) << 2);

#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        const uint32_t rsh = sh + static_cast<uint32_t>(round) * kRadixBits;      // sh is a multiple of 8: rsh + 4 <= 32
        uint32_t slot[KPT];
        {
            uint64_t seen = 0;
            uint32_t d_last = 0;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = __builtin_amdgcn_ubfe(field_word(k[i], hi), rsh, 4u);
                const uint32_t sh4 = d << 2;
                slot[i] = static_cast<uint32_t>(seen >> sh4) & 15u;
                if (i + 1 < KPT) {
                    seen += 1ull << sh4;
                } else {
                    d_last = d;
                }
            }
            const uint32_t seen_lo = static_cast<uint32_t>(seen), seen_hi = static_cast<uint32_t>(seen >> 32);
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                cnt32[l * THREADS + tid] = __builtin_amdgcn_ubfe(seen_lo, 4u * l, 4u) | (__builtin_amdgcn_ubfe(seen_hi, 4u * l, 4u) << 16);
            }
            atomicAdd(cnt + (d_last & 7u) * THREADS + tid, 1u << ((d_last >> 3) * 16u));
        }
        __syncthreads();
        {
            U32x4 a = *reinterpret_cast<const U32x4*>(cnt + tid * 8);
            U32x4 b = *reinterpret_cast<const U32x4*>(cnt + tid * 8 + 4);
            const uint32_t sum = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
            uint32_t total;
            uint32_t run = block_exclusive_scan<THREADS>(sum, wtot, total);
            run += total << 16;
            uint32_t t;
            t = a.v[0]; a.v[0] = run; run += t;
            t = a.v[1]; a.v[1] = run; run += t;
            t = a.v[2]; a.v[2] = run; run += t;
            t = a.v[3]; a.v[3] = run; run += t;
            t = b.v[0]; b.v[0] = run; run += t;
            t = b.v[1]; b.v[1] = run; run += t;
            t = b.v[2]; b.v[2] = run; run += t;
            t = b.v[3]; b.v[3] = run;
            *reinterpret_cast<U32x4*>(cnt + tid * 8) = a;
            *reinterpret_cast<U32x4*>(cnt + tid * 8 + 4) = b;
        }
        __syncthreads();
        {
            uint32_t first_of_digit[KPT];
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t w = field_word(k[i], hi);
                const uint32_t l3 = __builtin_amdgcn_ubfe(w, rsh, 3u);
                const uint32_t h = __builtin_amdgcn_ubfe(w, rsh + 3u, 1u);
                first_of_digit[i] = *reinterpret_cast<const u16_alias*>(cbytes + (l3 * CNT_ROW_BYTES + tid * 4u) + h * 2u);
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                slot[i] += first_of_digit[i];
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                // byte offset of slot s: (s*KD + 4*(s>>4)) * 4
                if constexpr (KD == 1) {
                    lds_store_at<Key>(add_lshl<2>(slot[i], (slot[i] >> 2) & ~3u), k[i]);
                } else {
                    lds_store_at<Key>(add_lshl<2>(slot[i] << 1, (slot[i] >> 2) & ~3u), k[i]);
                }
            }
        }
        __syncthreads();
        if (round == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const KeyVec<Key> v = *reinterpret_cast<const KeyVec<Key>*>(xbuf + tid * L::ROW_DW + j * 4);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    k[j * VEC + e] = v.k[e];
                }
            }
        } else {
            // leave as runs: slot i = r*THREADS + tid, its global slot = gb[digit] + i
            Key okey[KPT];
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                okey[r] = *reinterpret_cast<const Key*>(xbuf + out_base_dw + static_cast<uint32_t>(r) * OUT_STRIDE_DW);
            }
            uint32_t g[KPT];
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                g[r] = gb[__builtin_amdgcn_ubfe(field_word(okey[r], hi), sh, 8u)];
            }
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                g[r] += tid + static_cast<uint32_t>(r) * THREADS;
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    out[g[r]] = static_cast<Key>(okey[r] ^ flip);
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if (static_cast<uint32_t>(r) * THREADS + tid < valid) {
                        out[g[r]] = static_cast<Key>(okey[r] ^ flip);
                    }
                }
            }
            if constexpr (PAYLOAD) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    k[r] = static_cast<Key>(g[r]);          // keys are gone; keep each slot's destination for its payload
                }
            }
        }
        if constexpr (PAYLOAD) {
            __syncthreads();           // every thread has taken its keys: the image carries the payload now
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                lds_store_at<uint32_t>(add_lshl<2>(slot[i], (slot[i] >> 2) & ~3u), pl[i]);
            }
            __syncthreads();
            if (round == 0) {
#pragma unroll
                for (int q = 0; q < KPT / 4; ++q) {
                    const U32x4 x = *reinterpret_cast<const U32x4*>(xbuf + tid * (KPT + 4) + q * 4);
                    pl[q * 4 + 0] = x.v[0];
                    pl[q * 4 + 1] = x.v[1];
                    pl[q * 4 + 2] = x.v[2];
                    pl[q * 4 + 3] = x.v[3];
                }
            } else {
                const uint32_t pbase = tid + ((tid >> 4) << 2);
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if (full || static_cast<uint32_t>(r) * THREADS + tid < valid) {
                        pout[static_cast<uint32_t>(k[r])] = xbuf[pbase + static_cast<uint32_t>(r) * (THREADS + (THREADS / 16) * 4)];
                    }
                }
            }
        }
        __syncthreads();               // image and counters are free for the second round
    }
}

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
// Diagnostics in the REFERENCE's geometry (RadixSortGPU.cpp:412-428 downloads them after every
// sort): 1024 virtual processors with contiguous sub-lists of n/1024 keys, counter table
// [digit][group][item] = [digit][vp], its global exclusive scan ("pasted" table, 16384 words)
// and the scanned sums of the 512 blocks of 32 entries (globsum).  Recomputed on request from the
// input of the last pass, which still sits in the other ping-pong buffer.
constexpr int kRefVps = 1024;
constexpr int kRefTable = kRadix * kRefVps;      // _RADIX * _NUM_ITEMS = 16384
constexpr int kRefSplit = 512;                   // _NUM_HISTOSPLIT

template <typename Key>
__global__ __launch_bounds__(256) void ref_histogram_kernel(const Key* __restrict__ keys, uint32_t* __restrict__ ref_table,
                                                             uint64_t n, int shift, Key flip)
{
    __shared__ uint32_t cnt[kRadix];
    const uint32_t vp = blockIdx.x;
    const uint64_t sub = n / kRefVps;
    if (threadIdx.x < kRadix) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t local[kRadix] = {};
    for (uint64_t j = threadIdx.x; j < sub; j += blockDim.x) {
        const uint32_t d = digit_of(keys[vp * sub + j], shift, flip, static_cast<uint32_t>(kRadix - 1));
#pragma unroll
        for (int v = 0; v < kRadix; ++v) {
            local[v] += (d == static_cast<uint32_t>(v)) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int v = 0; v < kRadix; ++v) {
        if (local[v]) atomicAdd(&cnt[v], local[v]);
    }
    __syncthreads();
    if (threadIdx.x < kRadix) {
        ref_table[threadIdx.x * kRefVps + vp] = cnt[threadIdx.x];     // items*(ir*groups+gr)+it == ir*1024 + vp
    }
}

// exclusive scan of the 16384 counters in place (= the table after scan #1, scan #2 and paste);
// globsum[b] = scanned sum of block b = the pasted value of the block's first entry
__global__ __launch_bounds__(1024) void ref_scan_kernel(uint32_t* __restrict__ ref_table, uint32_t* __restrict__ ref_globsum)
{
    __shared__ uint32_t wtot[1024 / kWave];
    const uint32_t tid = threadIdx.x;
    uint32_t v[kRadix];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kRadix; ++i) {
        const uint32_t c = ref_table[tid * kRadix + i];
        v[i] = sum;
        sum += c;
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan<1024>(sum, wtot, total);
#pragma unroll
    for (int i = 0; i < kRadix; ++i) {
        ref_table[tid * kRadix + i] = v[i] + before;
    }
    // blocks of kRefTable / kRefSplit = 32 entries: thread tid owns entries [16 tid, 16 tid + 16)
    if ((tid & 1u) == 0) {
        ref_globsum[tid >> 1] = before;
    }
}

// `count` keys picked one per stratum of n/count consecutive keys, at a hashed position inside the
// stratum; written in unsigned sort order (key ^ flip) as uint64 (splitter selection, multi-GPU)
template <typename Key>
__global__ void sample_keys_kernel(const Key* __restrict__ keys, uint64_t n, uint32_t count, Key flip, unsigned long long* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    const uint64_t lo = static_cast<uint64_t>(i) * n / count, hi = static_cast<uint64_t>(i + 1) * n / count;
    const uint64_t width = hi > lo ? hi - lo : 1;
    uint64_t h = (static_cast<uint64_t>(i) + 1) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    const uint64_t pos = lo + h % width;
    out[i] = static_cast<unsigned long long>(static_cast<Key>(keys[pos < n ? pos : n - 1] ^ flip));
}

// min / max of the keys in unsigned order (key ^ flip); one {min, max} pair per workgroup,
// reduced on the host (multi-GPU partition: 16 equal-width buckets over the global range)
constexpr int kRangeThreads = 256;
template <typename Key>
__global__ __launch_bounds__(kRangeThreads) void key_range_kernel(const Key* __restrict__ keys, uint64_t n, Key flip,
                                                                   unsigned long long* __restrict__ partial)
{
    __shared__ unsigned long long smin[kRangeThreads / kWave], smax[kRangeThreads / kWave];
    constexpr int VEC = KeyVec<Key>::N;
    unsigned long long lo = ~0ull, hi = 0ull;
    const uint64_t nvec = n / VEC;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const KeyVec<Key> v = *reinterpret_cast<const KeyVec<Key>*>(keys + i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const unsigned long long u = static_cast<unsigned long long>(static_cast<Key>(v.k[e] ^ flip));
            lo = u < lo ? u : lo;
            hi = u > hi ? u : hi;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * VEC) {      // ragged tail
        const unsigned long long u = static_cast<unsigned long long>(static_cast<Key>(keys[nvec * VEC + threadIdx.x] ^ flip));
        lo = u < lo ? u : lo;
        hi = u > hi ? u : hi;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const unsigned long long ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) {
        smin[wave] = lo;
        smax[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kRangeThreads / kWave; ++w) {
            lo = smin[w] < lo ? smin[w] : lo;
            hi = smax[w] > hi ? smax[w] : hi;
        }
        partial[2 * blockIdx.x] = lo;
        partial[2 * blockIdx.x + 1] = hi;
    }
}

template <typename Key>
__global__ void fill_kernel(Key* __restrict__ dst, uint64_t first, uint64_t count, Key value)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        dst[first + i] = value;
    }
}

// per-digit totals of a RAW counter table: out[d] = sum over tiles of table[d][tile] (one workgroup per digit)
__global__ __launch_bounds__(256) void digit_totals_kernel(const uint32_t* __restrict__ table, uint32_t ntiles, unsigned long long* __restrict__ out)
{
    __shared__ unsigned long long wsum[256 / kWave];
    const uint32_t d = blockIdx.x;
    unsigned long long acc = 0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += blockDim.x) {
        acc += table[static_cast<uint64_t>(d) * ntiles + t];
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        wsum[threadIdx.x / kWave] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[d] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// bucket start offsets of a finished (scanned + pasted) table: out[d] = table[d][0]
__global__ void bucket_starts_kernel(const uint32_t* __restrict__ table, uint32_t ntiles, uint32_t* __restrict__ out)
{
    if (threadIdx.x < kRadix) {
        out[threadIdx.x] = table[static_cast<uint64_t>(threadIdx.x) * ntiles];
    }
}

}  // namespace rsx
