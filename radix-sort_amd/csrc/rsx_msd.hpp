// rsx_msd.hpp — the sharded sort's exchange step on the top B <= 8 key bits: bucket placement in wave-major order, the device-side
// exchange plan, and the per-wave push of one rank's buckets into the owners' receive buffers (peer stores over xGMI).
// Part of rsx_kernels.hpp.  Nothing in the reference corresponds to this (single in-order queue on one device,
// /root/reference/Common/ComputeState.cpp:88-101); it is SURVEY §8(e)'s "bucket exchange", re-designed for a point-to-point fabric.
//
// One MSD pass = the 8-bit pass kernels on the field [keybits - B, keybits): histogram8 -> scan8_blocks -> scan8_chunks, then
//   msd_top_kernel      bucket totals (for the all_gather) + bucket starts in WAVE-MAJOR order: with world = 2^r ranks owning k = 2^B / world
//                       consecutive (coarse) buckets each, bucket b = rank * k + wave sits at position wave * world + rank, so wave w holds one
//                       bucket per rank, contiguous and in rank order (the B = 4 form of round 2, now for any B <= 8)
//   reorder8_kernel     unchanged: it adds cbase8[chunk][digit], which msd_top_kernel filled with the wave-major starts — the shard lands in the
//                       staging buffer grouped by (wave, destination rank), stable
//   msd_layout_kernel   from the gathered [source rank][bucket] count table: where each (wave, destination) segment of THIS rank lands in the
//                       destination's receive buffer, what this rank receives per wave, and whether every rank's buffers hold their share — on
//                       the device, so that no host round trip stands between the all_gather and the first push
//   msd_push_kernel     wave w: copies this rank's `world` segments of the wave from staging into the destinations' receive buffers
//                       (peer-mapped memory of the other GPUs; 16-byte stores) — a link-bound copy on a small grid, which leaves the CUs to
//                       the local sort of wave w - 1
#pragma once

#include "rsx_common.hpp"
#include "rsx_radix8.hpp"

namespace rsx {

constexpr int kMsdMaxWorld = 16;

// The scatter always works on the key's top BYTE (256 fine buckets d, byte-aligned digit: the 8-bit pass kernels as they are); the partition's
// B = `bits` top bits name the COARSE bucket c = d >> (8 - B) that decides owner and wave: with world = 2^r ranks owning k = 2^B / world consecutive
// coarse buckets each, c = rank * k + wave.  Position of fine bucket d in the staging buffer = [wave][rank][low 8 - B bits of d]: a (wave, rank)
// segment is contiguous, and inside it the keys are already grouped by the remaining bits of the byte.
__host__ __device__ __forceinline__ uint32_t msd_position(uint32_t d, uint32_t bits, uint32_t world)
{
    const uint32_t sub_shift = 8u - bits;
    const uint32_t c = d >> sub_shift, sub = d & ((1u << sub_shift) - 1u);
    const uint32_t k = (1u << bits) / world;      // coarse buckets (= waves) per rank
    return (((c % k) * world + c / k) << sub_shift) | sub;
}

// One workgroup of 256 threads (thread d = fine bucket d), after scan8_chunks_kernel<false>: csum8[chunk][d] = keys of the chunk with digit d.
//   totals[c]          keys of the shard in COARSE bucket c < 2^bits, 0 beyond (uint64: the row that is all_gathered)
//   starts[d]          first staging slot of fine bucket d, wave-major order (a coarse bucket c begins at starts[c << (8 - bits)])
//   cbase8[chunk][d]   starts[d] + keys with digit d in earlier chunks — what reorder8_kernel adds per key
__global__ __launch_bounds__(kRadix8) void msd_top_kernel(const uint32_t* __restrict__ csum8, uint32_t* __restrict__ cbase8, uint32_t nchunks, uint32_t bits,
                                                          uint32_t world, unsigned long long* __restrict__ totals, uint32_t* __restrict__ starts, uint32_t* __restrict__ temp)
{
    __shared__ uint32_t by_pos[kRadix8];
    __shared__ uint32_t fine[kRadix8];
    __shared__ uint32_t wtot[kRadix8 / kWave];
    const uint32_t d = threadIdx.x;
    uint32_t cs[kScan8MaxChunks];
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kScan8MaxChunks; ++w) {
        cs[w] = static_cast<uint32_t>(w) < nchunks ? csum8[w * kRadix8 + d] : 0u;
        total += cs[w];
    }
    const uint32_t pos = msd_position(d, bits, world);
    by_pos[pos] = total;
    fine[d] = total;
    __syncthreads();
    uint32_t all;
    const uint32_t before = block_exclusive_scan<kRadix8>(by_pos[d], wtot, all);      // thread p scans position p
    by_pos[d] = before;
    {
        const uint32_t sub_shift = 8u - bits;
        unsigned long long coarse = 0;
        if (d < (1u << bits)) {
            for (uint32_t j = 0; j < (1u << sub_shift); ++j) {
                coarse += fine[(d << sub_shift) + j];
            }
        }
        totals[d] = coarse;
    }
    __syncthreads();
    uint32_t run = by_pos[pos];
    starts[d] = run;
#pragma unroll
    for (int w = 0; w < kScan8MaxChunks; ++w) {
        if (static_cast<uint32_t>(w) < nchunks) {
            cbase8[w * kRadix8 + d] = run;
        }
        run += cs[w];
    }
    if (d == 0) {
        temp[0] = all;
    }
}

// The exchange plan of one rank, in device memory (all counts and offsets in KEYS).
struct MsdSegment {
    uint32_t src;        // first slot of the segment in this rank's staging buffer
    uint32_t count;
    uint32_t dst;        // first slot in the destination's receive buffer
    uint32_t pad;
};
struct MsdPlan {
    MsdSegment seg[kRadix8];                 // [wave * world + destination]
    unsigned long long wave_start[kRadix8];  // what THIS rank receives: first slot of wave w in its receive buffer (16-byte aligned) ...
    unsigned long long wave_count[kRadix8];  // ... and its keys
    unsigned long long load[kMsdMaxWorld];   // keys every rank ends up with
    unsigned long long verdict;              // 0: go.  Bit r (r < 16) = rank r's buffers are too small; bit 32 + r = rank r reported a non-zero status word
                                             // (table[r * stride + cap_at + 2]: an engine error of an earlier step).  Non-zero: no push writes anything.
    unsigned long long waves;                // 2^bits / world
};

// table[src * stride + b] = keys of source rank src in coarse bucket b (b < nbuckets = 2^bits, natural order; sub_shift = 8 - bits); table[src * stride + cap_at] / [cap_at + 1] = that
// rank's receive / output capacity in keys, [cap_at + 2] = its status word (non-zero: that rank's engine reported an error; everybody stops together).  Receive layout at every destination: the waves follow each other, each starting on a 16-byte
// boundary (`align` keys: the local sort loads 16 bytes per lane), and inside a wave the sources follow each other in rank order.
__global__ __launch_bounds__(kRadix8) void msd_layout_kernel(const unsigned long long* __restrict__ table, uint32_t stride, uint32_t cap_at, uint32_t nbuckets,
                                                             uint32_t world, uint32_t rank, uint32_t align, uint32_t sub_shift, uint32_t grouping,
                                                             const uint32_t* __restrict__ starts, MsdPlan* __restrict__ plan)
{
    __shared__ unsigned long long bad;
    const uint32_t dst = threadIdx.x;
    const uint32_t waves = nbuckets / world;
    if (dst == 0) {
        bad = 0ull;
    }
    __syncthreads();
    if (dst < world) {
        unsigned long long at = 0, total = 0;
        for (uint32_t w = 0; w < waves; ++w) {
            const uint32_t b = dst * waves + w;
            if (grouping == 0u || (w & (w - 1u)) == 0u) {      // grouping 1 ("doubling groups"): only waves 0, 1, 2, 4, 8, ... start aligned (ShardPlanner.h)
                at = (at + align - 1) / align * align;
            }
            const unsigned long long wave_at = at;
            unsigned long long mine_at = 0, wave_total = 0;
            for (uint32_t src = 0; src < world; ++src) {
                const unsigned long long c = table[static_cast<uint64_t>(src) * stride + b];
                if (src == rank) {
                    mine_at = at;
                }
                at += c;
                wave_total += c;
            }
            total += wave_total;
            MsdSegment s;
            s.src = starts[b << sub_shift];
            s.count = static_cast<uint32_t>(table[static_cast<uint64_t>(rank) * stride + b]);
            s.dst = static_cast<uint32_t>(mine_at);
            s.pad = 0;
            plan->seg[w * world + dst] = s;
            if (dst == rank) {
                plan->wave_start[w] = wave_at;
                plan->wave_count[w] = wave_total;
            }
        }
        plan->load[dst] = total;
        // `at` = slots the destination's receive buffer needs (alignment gaps included); its output buffer takes the keys alone
        const unsigned long long recv_cap = table[static_cast<uint64_t>(dst) * stride + cap_at], out_cap = table[static_cast<uint64_t>(dst) * stride + cap_at + 1];
        if (at > recv_cap || total > out_cap || at > 0xFFFFFFFFull) {
            atomicOr(&bad, 1ull << dst);
        }
        if (table[static_cast<uint64_t>(dst) * stride + cap_at + 2] != 0ull) {
            atomicOr(&bad, 1ull << (32 + dst));
        }
    }
    __syncthreads();
    if (dst == 0) {
        plan->verdict = bad;
        plan->waves = waves;
    }
}

// 16 bytes at 4-byte alignment (gfx9 global accesses need dword alignment only)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));
typedef __attribute__((address_space(1))) uint32_t msd_gu32;                  // destinations arrive as integers: say "global", or the stores are flat_store
typedef __attribute__((address_space(1))) u32x4_a16 msd_gu32x4;

// dwords from src to dst: single dwords up to dst's next 16-byte boundary, 16-byte stores from there, single dwords for the tail
__device__ __forceinline__ void msd_copy_dwords(msd_gu32* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t ndw, uint32_t part, uint32_t parts)
{
    const uint32_t tid = threadIdx.x, threads = blockDim.x;
    const uint64_t head = (ndw < 4) ? ndw : ((16u - (reinterpret_cast<uintptr_t>((uint32_t*)dst) & 15u)) & 15u) / 4u;
    const uint64_t nvec = (ndw - head) / 4;
    if (part == 0) {
        if (tid < head) {
            dst[tid] = src[tid];
        }
        const uint64_t tail_at = head + nvec * 4;
        if (tid < ndw - tail_at) {
            dst[tail_at + tid] = src[tail_at + tid];
        }
    }
    const u32x4_a4* s4 = reinterpret_cast<const u32x4_a4*>(src + head);
    msd_gu32x4* d4 = (msd_gu32x4*)(dst + head);
    // each workgroup takes a contiguous share of the vectors, four loads in flight per thread
    const uint64_t per = (nvec + parts - 1) / parts;
    const uint64_t v0 = static_cast<uint64_t>(part) * per;
    const uint64_t v1 = v0 + per < nvec ? v0 + per : nvec;
    uint64_t v = v0 + tid;
    for (; v + 3ull * threads < v1; v += 4ull * threads) {
        const u32x4_a4 a = s4[v], b = s4[v + threads], c = s4[v + 2ull * threads], d = s4[v + 3ull * threads];
        d4[v] = a;
        d4[v + threads] = b;
        d4[v + 2ull * threads] = c;
        d4[v + 3ull * threads] = d;
    }
    for (; v < v1; v += threads) {
        d4[v] = s4[v];
    }
}

// Wave `wave` of this rank's staging buffer into the receive buffers: grid (parts, world) — workgroups (*, dst) copy segment [wave][dst].
// key_dw = dwords per key (1 or 2); peer_keys[dst] / peer_pays[dst] = base addresses of rank dst's receive buffers as THIS rank addresses them.
__global__ __launch_bounds__(256) void msd_push_kernel(const MsdPlan* __restrict__ plan, uint32_t wave, uint32_t world, uint32_t key_dw,
                                                       const uint32_t* __restrict__ staging, const uint32_t* __restrict__ staging_pay,
                                                       const unsigned long long* __restrict__ peer_keys, const unsigned long long* __restrict__ peer_pays)
{
    if (plan->verdict != 0ull) {
        return;                  // some rank's buffers are too small: nobody writes anything (every rank computed the same verdict)
    }
    const uint32_t dst = blockIdx.y;
    const MsdSegment s = plan->seg[wave * world + dst];
    if (s.count == 0) {
        return;
    }
    msd_gu32* out = (msd_gu32*)(peer_keys[dst]) + static_cast<uint64_t>(s.dst) * key_dw;
    msd_copy_dwords(out, staging + static_cast<uint64_t>(s.src) * key_dw, static_cast<uint64_t>(s.count) * key_dw, blockIdx.x, gridDim.x);
    if (staging_pay) {
        msd_gu32* pout = (msd_gu32*)(peer_pays[dst]) + s.dst;
        msd_copy_dwords(pout, staging_pay + s.src, s.count, blockIdx.x, gridDim.x);
    }
}

}  // namespace rsx
