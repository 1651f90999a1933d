// rsx_tile_sort.hpp — tile_sort_kernel: inputs of at most one tile, every pass inside LDS, one launch.
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"
#include "rsx_reorder.hpp"

namespace rsx {

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

}  // namespace rsx
