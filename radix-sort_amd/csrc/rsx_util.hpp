// rsx_util.hpp — reference-geometry diagnostics, key sampling, key range, padding fill, bucket totals.
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"

namespace rsx {

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

// where does a kernel's lone dynamic LDS array start?  (rsx_create: the scatter kernels address their image from LDS address 0)
__global__ void lds_base_probe_kernel(uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t probe_smem[];
    if (threadIdx.x == 0) {
        probe_smem[0] = 1;
        out[0] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)probe_smem));
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
