// rsx_histogram.hpp — histogram_kernel: per-tile digit counts (the reference's `histogram`, RadixSort.cl:16-71).
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"

namespace rsx {

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
            return ranged_bucket(static_cast<Key>((key ^ flip) - lo), shift, mul, mask);
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

}  // namespace rsx
