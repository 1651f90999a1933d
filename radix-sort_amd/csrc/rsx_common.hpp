// rsx_common.hpp — constants and small device helpers shared by every kernel header (included through rsx_kernels.hpp).
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

template <typename Key>
__device__ __forceinline__ KeyVec<Key> load_keys16(const Key* p)
{
    return *reinterpret_cast<const KeyVec<Key>*>(p);
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
};

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
__host__ __device__ __forceinline__ uint32_t tile_of_block(uint32_t bid, uint32_t tiles_per_xcd, int remap)
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
// `wtot` is LDS scratch of THREADS/64 words.  Contains two barriers; TRAILING_BARRIER = false drops the second one for callers
// that do not touch `wtot` again before their own next barrier.
template <int THREADS, bool TRAILING_BARRIER = true>
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
    if constexpr (TRAILING_BARRIER) {
        __syncthreads();   // wtot may be reused by the caller
    }
    return before + incl - v;
}

}  // namespace rsx
