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

}  // namespace rsx
