// rsx_radix8.hpp — 8-bit digits: histogram8 / scan8 / reorder8 (RSX_OPT_RADIX_BITS = 8).
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"
#include "rsx_reorder.hpp"

namespace rsx {

// ---------------------------------------------------------------------------
// 8-bit digits: half the passes (RSX_OPT_RADIX_BITS = 8, reported separately from the 4-bit configuration)
// ---------------------------------------------------------------------------
// The reference's digit width is a parameter (_NUM_BITS_PER_RADIX, src/Parameters.h:25, pushed into the kernels at
// src/RadixSortGPU.cpp:569-584).  A pass over an 8-bit digit is built from the 4-bit machinery above: the tile is
// sorted locally by the low nibble and then by the high nibble of the digit — two stable rounds through LDS, the
// second one starting from 16 consecutive keys of the first one's order per thread — and leaves as up to 256 runs.
// Tables are [tile][256] (a tile's 256 counters are one contiguous 1 KiB row):
//   histogram8_kernel   counts8[tile][d]   = (keys of the tile with digit d) | (keys of the tile with a SMALLER digit) << 16   — both <= 4096
//   scan8_blocks_kernel table8[tile][d]    = (keys with digit d in EARLIER tiles of the tile's group of G tiles) - (keys of the tile with a smaller digit);
//                                            gsum8[group][d] = group total
//   scan8_chunks_kernel gsum8[group][d]    = keys with digit d in earlier groups of the group's chunk; csum8[chunk][d] = chunk total
//   reorder8_kernel     slot of a key      = cbase8[chunk][d] (keys with smaller digits + digit d in earlier chunks) + gsum8[group][d] + table8[tile][d]
//                                            + (its slot in the tile-local sorted order)
// (round 3: the tile's own exclusive scan over its 256 counts is done once, by the histogram kernel, which has the counts in LDS anyway, and rides
// in the upper half of the count word — the scatter kernel then needs one table row instead of two, no block scan and two barriers fewer.)
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
    __shared__ uint32_t wtot[kRadix8 / kWave];
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
    {
        const uint32_t c = cnt[tid];
        uint32_t all;
        const uint32_t local_first = block_exclusive_scan<THREADS, false>(c, wtot, all);       // keys of the tile with a smaller digit
        counts8[static_cast<uint64_t>(tile) * kRadix8 + tid] = c | (local_first << 16);
    }
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
            // word = count | (keys of the tile with a smaller digit) << 16: the entry is what the scatter adds to a key's tile-local slot
            table8[static_cast<uint64_t>(t + u) * kRadix8 + d] = run - (c[u] >> 16);
            run += c[u] & 0xFFFFu;
        }
    }
    for (; t < t1; ++t) {
        const uint32_t c = counts8[static_cast<uint64_t>(t) * kRadix8 + d];
        table8[static_cast<uint64_t>(t) * kRadix8 + d] = run - (c >> 16);
        run += c & 0xFFFFu;
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

// (measured and removed in round 4, see profiles/r03_tuning_log.md §10 and the r03 history: a separate payload array riding in the same LDS image as its keys
// — 7 % slower on random uint64 + payload —, a padded second-round image, and key / payload stores of the packed variant issued apart)
template <typename Key, int THREADS, int KPT, bool PAYLOAD = false>
struct Reorder8Layout {
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int ROW_DW = KPT * KD + 4;                 // as TileSortLayout: 16-byte aligned rows on distinct bank quads
    static constexpr int XBUF_DW = THREADS * ROW_DW;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int TOTAL_DW = XBUF_DW + CNT_DW + 16 + kRadix8;
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    static constexpr int WGS_PER_CU = static_cast<int>((160 * 1024) / BYTES);
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > 8 ? 8 : (WGS_PER_CU * THREADS / 256);
    static constexpr int LG = (KPT == 16) ? 4 : 3;              // log2 of the slots per row
    static_assert((KPT == 16 || KPT == 8) && THREADS * KPT == 4096 && THREADS >= kRadix8, "4096-key tiles in rows of KPT slots; the first 256 threads handle the tile's table row");
};

// One tile's keys (and payloads) in registers as loaded — sign flip not yet applied — with this thread's entry of the tile's table row.
template <typename Key, int KPT, bool PAYLOAD>
struct Reorder8Regs {
    Key k[KPT];
    uint32_t pl[PAYLOAD ? KPT : 1];
    uint32_t base3[3];          // this thread's digit: table row entry, group sum, chunk base — added up where they are used, not where they are loaded
    uint32_t valid;
};

// Issue the loads of tile `tile` (nothing here waits for them; reorder8_sort_tile's first use does).
// End of the ragged tile's masked loads (one tile per launch): wait for them here.  Left pending, they reach the join with the whole-tile path, where
// the compiler then drains vmcnt to 0 BEFORE it issues the whole tile's key loads (a register of theirs is a destination of the pending ones) — and
// with that waits for the table-row loads first: +13 % on the packed uint32 + payload scatter.  s_waitcnt vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15).
__device__ __forceinline__ void reorder8_partial_tile_loaded()
{
    __builtin_amdgcn_s_waitcnt(0x0f70);
}

// FULL_ONLY: the tile is known to be whole — vector loads and no branch (what a prefetch across a loop iteration needs: the wait counters of
// loads issued under a branch are merged conservatively at the join, and the first use behind it would wait for everything).
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool PACKED32, bool FULL_ONLY = false>
__device__ __forceinline__ void reorder8_fetch(Reorder8Regs<Key, KPT, PAYLOAD>& t, const Key* __restrict__ in, const uint32_t* __restrict__ pin,
                                               const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
                                               uint32_t chunk_groups, uint64_t n, uint32_t tile, Key flip)
{
    constexpr int TILE = THREADS * KPT;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    const uint32_t tid = threadIdx.x;
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = (FULL_ONLY || left >= static_cast<uint64_t>(TILE)) ? static_cast<uint32_t>(TILE) : static_cast<uint32_t>(left);
    const bool full = FULL_ONLY || (valid == TILE);
    t.valid = valid;
    // pads sort behind every real key of the tile (digit 255 once the flip below is applied)
    const Key pad_raw = static_cast<Key>(~Key{0}) ^ flip;
    // this thread's digit of the tile's table row (latency hides under the key loads): global slot of the tile's first key with
    // that digit minus its tile-local slot
    const uint32_t group = tile / kScan8Tiles;
    const uint32_t dg = (THREADS == kRadix8) ? tid : (tid & static_cast<uint32_t>(kRadix8 - 1));      // (wider workgroups: the upper threads load the same row again and do not use it)
    t.base3[0] = table8[static_cast<uint64_t>(tile) * kRadix8 + dg];
    t.base3[1] = gsum8[static_cast<uint64_t>(group) * kRadix8 + dg];
    t.base3[2] = cbase8[static_cast<uint64_t>(group / chunk_groups) * kRadix8 + dg];      // smaller digits + this digit in earlier chunks
    if constexpr (PACKED32) {
        const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in);
        if (full) {
#pragma unroll
            for (int q = 0; q < KPT / 4; ++q) {
                const U32x4 a = *reinterpret_cast<const U32x4*>(in32 + base + tid * KPT + q * 4);
                const U32x4 b = *reinterpret_cast<const U32x4*>(pin + base + tid * KPT + q * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    t.k[q * 4 + c] = (static_cast<Key>(b.v[c]) << 32) | static_cast<Key>(a.v[c]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t li = tid * KPT + i;
                t.k[i] = li < valid ? ((static_cast<Key>(pin[base + li]) << 32) | static_cast<Key>(in32[base + li])) : pad_raw;
            }
            reorder8_partial_tile_loaded();
        }
    } else if (full) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const KeyVec<Key> v = load_keys16(in + base + tid * KPT + j * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                t.k[j * VEC + e] = v.k[e];
            }
        }
        if constexpr (PAYLOAD) {
#pragma unroll
            for (int q = 0; q < KPT / 4; ++q) {
                const U32x4 x = *reinterpret_cast<const U32x4*>(pin + base + tid * KPT + q * 4);
                t.pl[q * 4 + 0] = x.v[0];
                t.pl[q * 4 + 1] = x.v[1];
                t.pl[q * 4 + 2] = x.v[2];
                t.pl[q * 4 + 3] = x.v[3];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t li = tid * KPT + i;
            t.k[i] = li < valid ? in[base + li] : pad_raw;
            if constexpr (PAYLOAD) {
                t.pl[i] = li < valid ? pin[base + li] : 0u;
            }
        }
        reorder8_partial_tile_loaded();
    }
}

// Rank the tile's keys on the 8-bit digit at `shift` (two stable 4-bit rounds through the LDS image), then store them — and their
// payloads — as runs at their global slots.  Ends with a barrier: the image, the counters and gb[] are free for the next tile.
// UNROLL_ROUNDS: both rounds written out.  The staying kernel needs it: on gfx9 (stores count in vmcnt) the compiler drains vmcnt to 0 in the
// preheader of a loop that stores and uses registers loaded before it — which would end the prefetch of the next tile right there.
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool PACKED32, bool UNROLL_ROUNDS = false>
__device__ __forceinline__ void reorder8_sort_tile(Reorder8Regs<Key, KPT, PAYLOAD>& t, uint32_t* smem, Key* __restrict__ out, uint32_t* __restrict__ pout, int shift, Key flip)
{
    using L = Reorder8Layout<Key, THREADS, KPT, PAYLOAD>;
    constexpr int TILE = THREADS * KPT;
    constexpr int KD = L::KD;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    constexpr uint32_t CNT_ROW_BYTES = THREADS * 4;
    uint32_t* xbuf = smem;
    uint32_t* cnt = smem + L::XBUF_DW;
    uint32_t* wtot = cnt + L::CNT_DW;
    uint32_t* gb = wtot + 16;                     // per 8-bit digit: (global slot of the tile's first key with it) - (its local slot)
    const uint32_t tid = threadIdx.x;
    const uint32_t valid = t.valid;
    const bool full = (valid == TILE);
    // Inside the kernel keys are held with the sign bit flipped (k ^ flip: unsigned order = numeric order), so the
    // digits are plain bit fields; the flip is undone on the way out.  Unsigned types skip both (flip == 0, uniform).
    const bool hi = !PACKED32 && sizeof(Key) == 8 && shift >= 32;      // the byte never straddles the halves of a 64-bit key
    const uint32_t sh = static_cast<uint32_t>(shift) & 31u;
    if (flip != Key{0}) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            t.k[i] ^= flip;
        }
    }
    if (THREADS == kRadix8 || tid < static_cast<uint32_t>(kRadix8)) {
        gb[tid] = t.base3[0] + t.base3[1] + t.base3[2];          // (read after several barriers)
    }

    u32_alias* cnt32 = reinterpret_cast<u32_alias*>(cnt);
    unsigned char* cbytes = reinterpret_cast<unsigned char*>(cnt);
    // image: slot s in row s/KPT (KPT keys + 16 bytes) at key j = s%KPT; slot i = r*THREADS + tid -> per-thread base + r * OUT_STRIDE
    constexpr int LG = L::LG;
    constexpr uint32_t OUT_STRIDE_DW = (THREADS / KPT) * L::ROW_DW;
    const uint32_t out_base_dw = (tid >> LG) * L::ROW_DW + (tid & static_cast<uint32_t>(KPT - 1)) * KD;

    constexpr int kRoundsUnrolled = UNROLL_ROUNDS ? 2 : 1;
#pragma unroll kRoundsUnrolled
    for (int round = 0; round < 2; ++round) {
        const uint32_t rsh = sh + static_cast<uint32_t>(round) * kRadixBits;      // sh is a multiple of 8: rsh + 4 <= 32
        uint32_t slot[KPT];
        {
            uint64_t seen = 0;
            uint32_t d_last = 0;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = __builtin_amdgcn_ubfe(field_word(t.k[i], hi), rsh, 4u);
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
            uint32_t run = block_exclusive_scan<THREADS, false>(sum, wtot, total);       // (wtot is next written two barriers on)
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
                const uint32_t w = field_word(t.k[i], hi);
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
                    lds_store_at<Key>(add_lshl<2>(slot[i], (slot[i] >> (LG - 2)) & ~3u), t.k[i]);
                } else {
                    lds_store_at<Key>(add_lshl<2>(slot[i] << 1, (slot[i] >> (LG - 2)) & ~3u), t.k[i]);
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
                    t.k[j * VEC + e] = v.k[e];
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
            if (flip != Key{0}) {        // (uniform: unsigned keys skip the flip on the way out as on the way in)
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    okey[r] ^= flip;
                }
            }
            if constexpr (PACKED32) {
                uint32_t* out32 = reinterpret_cast<uint32_t*>(out);
                // (key and payload of a slot together: all keys first, then all payloads, is 7 % slower — r03_ab_split_stores.txt)
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if (full || static_cast<uint32_t>(r) * THREADS + tid < valid) {
                        out32[g[r]] = static_cast<uint32_t>(okey[r]);
                        pout[g[r]] = static_cast<uint32_t>(okey[r] >> 32);
                    }
                }
            } else if (full) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    out[g[r]] = okey[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if (static_cast<uint32_t>(r) * THREADS + tid < valid) {
                        out[g[r]] = okey[r];
                    }
                }
            }
            if constexpr (PAYLOAD) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    t.k[r] = static_cast<Key>(g[r]);          // keys are gone; keep each slot's destination for its payload
                }
            }
        }
        if constexpr (PAYLOAD) {
            __syncthreads();           // every thread has taken its keys: the image carries the payload now
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                lds_store_at<uint32_t>(add_lshl<2>(slot[i], (slot[i] >> (LG - 2)) & ~3u), t.pl[i]);
            }
            __syncthreads();
            if (round == 0) {
#pragma unroll
                for (int q = 0; q < KPT / 4; ++q) {
                    const U32x4 x = *reinterpret_cast<const U32x4*>(xbuf + tid * (KPT + 4) + q * 4);
                    t.pl[q * 4 + 0] = x.v[0];
                    t.pl[q * 4 + 1] = x.v[1];
                    t.pl[q * 4 + 2] = x.v[2];
                    t.pl[q * 4 + 3] = x.v[3];
                }
            } else {
                const uint32_t pbase = tid + ((tid >> LG) << 2);
                constexpr uint32_t pstride = THREADS + (THREADS / KPT) * 4;
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if (full || static_cast<uint32_t>(r) * THREADS + tid < valid) {
                        pout[static_cast<uint32_t>(t.k[r])] = xbuf[pbase + static_cast<uint32_t>(r) * pstride];
                    }
                }
            }
        }
        __syncthreads();               // image and counters are free for the second round
    }
}

// PACKED32 (uint32 keys WITH payload, instantiated as Key = uint64_t, PAYLOAD = false): key and payload travel through the two ranking
// rounds as ONE 64-bit element (key in the low word, where the digit is taken; payload in the high word) — `in` / `out` really point at
// uint32 keys, `pin` / `pout` at the payloads.  One 8-byte LDS access per element and round instead of two 4-byte ones, and none of the
// payload's own trips (4 barriers fewer per tile): the uint32 + payload scatter then costs what the uint64 keys-only one costs.
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool PACKED32 = false>
__global__ __launch_bounds__(THREADS, (PAYLOAD ? (THREADS > 256 ? 4 : 2) : (Reorder8Layout<Key, THREADS, KPT>::MIN_WAVES > 4 ? 4 : Reorder8Layout<Key, THREADS, KPT>::MIN_WAVES))) void reorder8_kernel(
    const Key* __restrict__ in, Key* __restrict__ out, const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
    const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
    uint32_t chunk_groups, uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd, int remap, int shift, Key flip)
{
    static_assert(!PACKED32 || (sizeof(Key) == 8 && !PAYLOAD), "packed (uint32 key, payload) elements are 64-bit and carry their payload themselves");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    // (lds_store_at addresses the image from LDS address 0: checked once on the host, rsx_create)
    Reorder8Regs<Key, KPT, PAYLOAD> t;
    reorder8_fetch<Key, THREADS, KPT, PAYLOAD, PACKED32>(t, in, pin, table8, gsum8, cbase8, chunk_groups, n, tile, flip);
    reorder8_sort_tile<Key, THREADS, KPT, PAYLOAD, PACKED32>(t, smem, out, pout, shift, flip);
}

}  // namespace rsx
