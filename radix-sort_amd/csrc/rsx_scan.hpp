// rsx_scan.hpp — scans of the [digit][tile] table (the reference's `scanhistograms` x2 + `pastehistograms`, RadixSort.cl:125-197): two-launch, paste-scan, fused and one-workgroup forms.
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"

namespace rsx {

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
// kFusedScanMaxGroups sizes the granule buffer only.  Which tables actually take this kernel is decided per engine from the
// occupancy query (workgroups resident at once = per-CU answer x CU count; half of that is the limit, rsx_create): on a whole
// MI355X that is the full 512 groups (2^29 keys), on a partitioned device or under a CU mask correspondingly fewer.
constexpr int kFusedScanMaxGroups = 512;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

// LDS scratch of one fused-scan workgroup (1 KiB).  The stand-alone kernel keeps it in static LDS; reorder_kernel's inline scan
// (below) lays it over the start of its dynamic LDS, which is free until the tile's own work begins.
struct FusedScanLds {
    uint32_t wsum[kScanTiles / kWave][kRadix];
    uint32_t part[kScanTiles / kWave][2][kRadix];
    uint32_t dtot[kRadix], off[kRadix];
};

// One workgroup's share of the fused scan: scan group `group` (256 tiles, all 16 digits).  PUBLISH (the inline form): the finished
// table entries leave as write-through (sc1) stores, every storing wave drains them, and after the workgroup's barrier ONE lane
// stores `epoch` to ready[group] — the R1 hand-off of the CDNA guide (sc1 payload -> vmcnt(0) -> barrier -> sc1 flag); consumers poll
// that word with sc1 loads and read the entries with sc1 loads.  FROM_COUNTS is a run-time (wave-uniform) flag here.
template <bool ZERO_BACK, bool PUBLISH>
__device__ __forceinline__ void fused_scan_group(FusedScanLds& lds, uint32_t group, uint32_t* __restrict__ table, unsigned long long* sums,
                                                 uint32_t* __restrict__ scanned, uint32_t* __restrict__ temp, uint32_t ntiles, uint32_t ngroups,
                                                 uint32_t* __restrict__ counts, bool from_counts, uint32_t epoch, uint32_t* timeout, uint32_t* ready)
{
    constexpr int WAVES = kScanTiles / kWave;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t tile = group * kScanTiles + tid;
    const bool live = tile < ntiles;
    uint32_t c[kRadix];
    if (from_counts) {
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
            lds.wsum[wave][d] = ex[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kRadix; ++d) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            before += (static_cast<uint32_t>(w) < wave) ? lds.wsum[w][d] : 0u;
        }
        ex[d] = before + ex[d] - c[d];
    }
    // ---- publish the 16 group sums as {epoch, value} granules ---------------------------------------
    if (tid < kRadix) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            total += lds.wsum[w][tid];
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
                        __hip_atomic_store((gu32*)(timeout), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // mapped host memory
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
            lds.part[wave][0][lane] = tot;
            lds.part[wave][1][lane] = pre;
        }
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            t += lds.part[w][0][tid];
        }
        lds.dtot[tid] = t;
    }
    __syncthreads();
    if (tid < kRadix) {
        uint32_t base = 0;
#pragma unroll 1
        for (uint32_t d2 = 0; d2 < tid; ++d2) {
            base += lds.dtot[d2];
        }
        uint32_t pr = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            pr += lds.part[w][1][tid];
        }
        lds.off[tid] = base + pr;
        scanned[static_cast<uint64_t>(tid) * ngroups + group] = base + pr;
        if (group == 0 && tid == kRadix - 1) {
            temp[0] = base + lds.dtot[tid];
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int d = 0; d < kRadix; ++d) {
            if constexpr (PUBLISH) {
                __hip_atomic_store((gu32*)(table) + static_cast<uint64_t>(d) * ntiles + tile, ex[d] + lds.off[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                table[static_cast<uint64_t>(d) * ntiles + tile] = ex[d] + lds.off[d];
            }
        }
    }
    if constexpr (PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores ...
        __syncthreads();                                       // ... before ONE lane signals for the workgroup
        if (tid == 0) {
            __hip_atomic_store((gu32*)(ready) + group, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <bool FROM_COUNTS, bool ZERO_BACK>
__global__ __launch_bounds__(kScanTiles) void scan_fused_kernel(uint32_t* __restrict__ table, unsigned long long* sums, uint32_t* __restrict__ scanned,
                                                                 uint32_t* __restrict__ temp, uint32_t ntiles, uint32_t ngroups,
                                                                 uint32_t* __restrict__ counts, uint32_t epoch, uint32_t* timeout)
{
    __shared__ FusedScanLds lds;
    fused_scan_group<ZERO_BACK, false>(lds, blockIdx.x, table, sums, scanned, temp, ntiles, ngroups, counts, FROM_COUNTS, epoch, timeout, nullptr);
}

#ifdef RSX_EXPERIMENTS
// tests only (RSX_XOPT_DEBUG_RAISE_SCAN_TIMEOUT, experiments build): the store a timed-out sweep makes, without the sweep
__global__ void raise_flag_kernel(uint32_t* flag)
{
    if (threadIdx.x == 0) {
        __hip_atomic_store((gu32*)(flag), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

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

}  // namespace rsx
