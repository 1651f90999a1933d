// rsx_radix8_experiments.hpp — measured and REJECTED alternatives of the 8-bit scatter, kept for A/B runs: built only with
// -DRSX_EXPERIMENTS (tools/build_variant.sh experiments ...), never part of the product library.
//   reorder8_stay_kernel    the default ranking as a grid that stays and prefetches its next tile          (1.5x slower, tuning log r03 §8)
//   reorder8v2_kernel       keys and payload make ONE trip through LDS                                      (better on Range, worse elsewhere, §1)
//   reorder8v3_kernel       ranks from one returning LDS atomic per key + lds_atomic_order_probe_kernel     (1.6x slower on random keys, §1)
#pragma once

#include "rsx_radix8.hpp"

namespace rsx {

// The same scatter as a grid that stays: gridDim.x = 8 * (workgroups per XCD) workgroups take the whole tiles of their XCD's range one after the
// other and each issues the loads of its NEXT tile before it ranks the current one, so that a tile's HBM latency hides under the previous
// tile's LDS work instead of under other workgroups.  Tiles are handed out by a ticket counter per XCD (tickets[x], zero at launch): the tiles
// in flight on an XCD stay neighbours, as under the hardware's own dispatch order — with a fixed stride per workgroup they drift apart and the
// partial sectors of neighbouring tiles no longer meet in the L2 (measured: 0.89 -> 1.44 ms per launch).  A ticket is drawn one tile ahead
// (the returning atomic's latency hides under the ranking as well) and handed to the workgroup through one LDS word.  The array's ragged tile
// goes to workgroup 0 afterwards.
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool PACKED32 = false>
__global__ __launch_bounds__(THREADS, 2) void reorder8_stay_kernel(
    const Key* __restrict__ in, Key* __restrict__ out, const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
    const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
    uint32_t chunk_groups, uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd, int remap, int shift, Key flip, uint32_t* __restrict__ tickets)
{
    using L = Reorder8Layout<Key, THREADS, KPT>;
    static_assert(!PACKED32 || (sizeof(Key) == 8 && !PAYLOAD), "packed (uint32 key, payload) elements are 64-bit and carry their payload themselves");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // (LDS base 0: checked on the host, rsx_create)
    constexpr int TILE = THREADS * KPT;
    uint32_t* slot = smem + L::XBUF_DW + L::CNT_DW + 12;      // (a word of the wave-total area that the block scan does not use)
    const uint32_t tid = threadIdx.x;
    const uint32_t nfull = static_cast<uint32_t>(n / TILE);
    const bool by_xcd = (remap & 1) != 0;
    const uint32_t x = by_xcd ? blockIdx.x % kNumXcd : 0u;
    const uint32_t first = by_xcd ? x * tiles_per_xcd : 0u;
    const uint32_t range = by_xcd ? tiles_per_xcd : ntiles;
    const uint32_t count = nfull <= first ? 0u : (nfull - first < range ? nfull - first : range);        // whole tiles of this XCD's range
    const uint32_t phase = by_xcd ? x * (static_cast<uint32_t>(remap) >> 8) : 0u;
    const auto tile_at = [&](uint32_t q) { return first + (q + phase) % count; };          // ticket q < count
    uint32_t ticket = 0;
    const auto draw = [&]() {
        if (tid == 0) {
            // (an increment, not an add: the compiler's atomic optimizer rewrites a uniform-address add into add + readfirstlane and would wait for it here)
            ticket = __builtin_amdgcn_atomic_inc32(tickets + x, 0xffffffffu, __ATOMIC_RELAXED, "agent");
        }
    };
    const auto hand_over = [&]() {          // (callers keep a barrier between the last read of the word and this)
        if (tid == 0) {
            *slot = ticket;
        }
        __syncthreads();
        return *slot;
    };
    if (count != 0) {
        draw();
        const uint32_t q0 = hand_over();
        if (q0 < count) {
            Reorder8Regs<Key, KPT, PAYLOAD> cur;
            reorder8_fetch<Key, THREADS, KPT, PAYLOAD, PACKED32, true>(cur, in, pin, table8, gsum8, cbase8, chunk_groups, n, tile_at(q0), flip);
            draw();
            __syncthreads();
            uint32_t qn = hand_over();
            for (;;) {
                const bool more = qn < count;
                Reorder8Regs<Key, KPT, PAYLOAD> nxt;
                // (past the end: the range's first tile once more, never used)
                reorder8_fetch<Key, THREADS, KPT, PAYLOAD, PACKED32, true>(nxt, in, pin, table8, gsum8, cbase8, chunk_groups, n, more ? tile_at(qn) : first, flip);
                if (more) {
                    draw();
                }
                reorder8_sort_tile<Key, THREADS, KPT, PAYLOAD, PACKED32, true>(cur, smem, out, pout, shift, flip);
                if (!more) {
                    break;
                }
                qn = hand_over();
                cur = nxt;
            }
        }
    }
    if (blockIdx.x == 0 && nfull < ntiles) {
        Reorder8Regs<Key, KPT, PAYLOAD> last;
        reorder8_fetch<Key, THREADS, KPT, PAYLOAD, PACKED32>(last, in, pin, table8, gsum8, cbase8, chunk_groups, n, nfull, flip);
        reorder8_sort_tile<Key, THREADS, KPT, PAYLOAD, PACKED32>(last, smem, out, pout, shift, flip);
    }
}

// ---------------------------------------------------------------------------
// reorder8 v2 (round 3): the keys make ONE trip through LDS instead of two
// ---------------------------------------------------------------------------
// v1 above sorts the tile by the low nibble, stages it, re-reads it as 16 consecutive slots per thread, sorts by the high nibble
// and stages it again — and with a payload every staging is followed by a second trip of the payload through the same image
// (18 workgroup barriers per tile).  Here keys and payload stay in the registers of the thread that loaded them until the
// final slot of every key is known:
//   round 0  thread t ranks its 16 keys by the LOW nibble (nibble counters in a 64-bit register, packed words, raking scan: the
//            4-bit machinery) -> s0 = slot in the order (low nibble, index); it leaves the key's HIGH nibble (<< 2, one byte) at H[s0]
//   round 1  thread t owns slots 16t .. 16t+15 of that order: one ds_read_b128 of H gives their 16 high nibbles, which it ranks the
//            same way -> s1 = slot in the order (high nibble, low nibble, index) = the tile-local sorted order; M[16t + j] = s1
//            (16 x u16 = two ds_write_b128)
//   final    the loading thread reads f = M[s0] for its 16 keys and stages key (and payload, into a second image, in the SAME trip)
//            at f; the tile leaves as runs exactly as in reorder_kernel (padded image, per-thread read base + constant offsets).
// Per key 1 byte + 2 bytes of hand-off instead of a whole second trip of key and payload; 10 barriers per tile with or without a
// payload.  The work area (packed counters 8 KiB, H 4 KiB, M 8 KiB) is dead when the image is written and shares its LDS.
template <typename Key, int THREADS, int KPT, bool PAYLOAD>
struct Reorder8V2Layout {
    static constexpr int TILE = THREADS * KPT;
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int PADSH = (KD == 1) ? 5 : 4;                      // as ReorderLayout: one pad element every 2^PADSH
    static constexpr int XBUF_DW = (TILE + (TILE >> PADSH)) * KD;        // key image
    static constexpr int PBUF_DW = PAYLOAD ? TILE + (TILE >> 5) : 0;     // payload image
    static constexpr int IMAGE_DW = XBUF_DW + PBUF_DW;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int H_DW = TILE / 4;                                // one byte per slot
    static constexpr int M_DW = TILE / 2;                                // one u16 per slot
    static constexpr int CNT_AT = 0, H_AT = CNT_DW, M_AT = CNT_DW + H_DW;
    static constexpr int WORK_DW = CNT_DW + H_DW + M_DW;
    static constexpr int BODY_DW = IMAGE_DW > WORK_DW ? IMAGE_DW : WORK_DW;
    static constexpr int TOTAL_DW = BODY_DW + 16 + kRadix8;              // + wave totals + run bases
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    static constexpr int WGS_PER_CU = static_cast<int>((160 * 1024) / BYTES);
    static constexpr int WAVES_CAP = 5;      // 6 (80 VGPRs) spills 12 bytes per lane in the uint32 keys-only kernel
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > 5 ? 5 : (WGS_PER_CU * THREADS / 256);
    static_assert(KPT == 16 && THREADS == kRadix8, "16 slots per thread = one 16-byte row of H; one thread per digit handles the tile's table row");
    static_assert((H_AT * 4) % 16 == 0 && (M_AT * 4) % 16 == 0 && (XBUF_DW * 4) % 16 == 0, "16-byte aligned rows");
};

template <typename Key, int THREADS, int KPT, bool PAYLOAD>
__global__ __launch_bounds__(THREADS, (Reorder8V2Layout<Key, THREADS, KPT, PAYLOAD>::MIN_WAVES)) void reorder8v2_kernel(
    const Key* __restrict__ in, Key* __restrict__ out, const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
    const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
    uint32_t chunk_groups, uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd, int remap, int shift, Key flip)
{
    using L = Reorder8V2Layout<Key, THREADS, KPT, PAYLOAD>;
    constexpr int TILE = L::TILE;
    constexpr int VEC = KeyVec<Key>::N;
    constexpr int NV = KPT / VEC;
    constexpr uint32_t CNT_ROW_BYTES = THREADS * 4;
    constexpr uint32_t H_BYTES = L::H_AT * 4, M_BYTES = L::M_AT * 4, PBUF_BYTES = L::XBUF_DW * 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* cnt = smem + L::CNT_AT;
    uint32_t* wtot = smem + L::BODY_DW;
    uint32_t* gb = wtot + 16;                     // per 8-bit digit: (global slot of the tile's first key with it) - (its tile-local slot)
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    // (LDS base 0: checked on the host, rsx_create)
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    const bool full = (valid == TILE);
    // keys are held with the sign bit flipped (unsigned order = numeric order): digits are plain bit fields
    const Key pad_key = static_cast<Key>(~Key{0});        // digit 255, behind every real key of the tile
    const bool hi = sizeof(Key) == 8 && shift >= 32;      // the byte never straddles the halves of a 64-bit key
    const uint32_t sh = static_cast<uint32_t>(shift) & 31u;

    // this thread's digit of the tile's table row (latency hides under the key loads)
    const uint32_t group = tile / kScan8Tiles;
    const uint32_t my_base = table8[static_cast<uint64_t>(tile) * kRadix8 + tid] + gsum8[static_cast<uint64_t>(group) * kRadix8 + tid] +
                             cbase8[static_cast<uint64_t>(group / chunk_groups) * kRadix8 + tid];

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
    gb[tid] = my_base;           // (read after several barriers)
    u32_alias* cnt32 = reinterpret_cast<u32_alias*>(cnt);
    unsigned char* cbytes = reinterpret_cast<unsigned char*>(cnt);

    // One ranking round of the 4-bit machinery over 16 values per thread given as `x4[i]` = digit << 2 (0 .. 60): on return
    // slot[i] = position of (thread, i) in the workgroup-wide order (digit, thread, i).  Three barriers.
    auto rank_round = [&](const uint32_t (&x4)[KPT], uint32_t (&slot)[KPT]) {
        uint64_t seen = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            slot[i] = static_cast<uint32_t>(seen >> x4[i]) & 15u;
            asm volatile("" : "+v"(slot[i]));
            if (i + 1 < KPT) {
                seen += 1ull << x4[i];
            }
        }
        const uint32_t seen_lo = static_cast<uint32_t>(seen), seen_hi = static_cast<uint32_t>(seen >> 32);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            cnt32[l * THREADS + tid] = __builtin_amdgcn_ubfe(seen_lo, 4u * l, 4u) | (__builtin_amdgcn_ubfe(seen_hi, 4u * l, 4u) << 16);
        }
        const uint32_t d_last = x4[KPT - 1] >> 2;
        atomicAdd(cnt + (d_last & 7u) * THREADS + tid, 1u << ((d_last >> 3) * 16u));
        __syncthreads();
        {
            U32x4 a = *reinterpret_cast<const U32x4*>(cnt + tid * 8);
            U32x4 b = *reinterpret_cast<const U32x4*>(cnt + tid * 8 + 4);
            const uint32_t sum = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
            uint32_t total;
            uint32_t run = block_exclusive_scan<THREADS, false>(sum, wtot, total);       // (wtot is next written several barriers on)
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
        uint32_t first_of_digit[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            // 16-bit counter of (digit d, this thread): word [d&7][tid], half d>>3; x4 = d << 2
            const uint32_t l3 = __builtin_amdgcn_ubfe(x4[i], 2u, 3u);
            const uint32_t h = x4[i] >> 5;
            first_of_digit[i] = *reinterpret_cast<const u16_alias*>(cbytes + (l3 * CNT_ROW_BYTES + tid * 4u) + h * 2u);
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            slot[i] += first_of_digit[i];
        }
    };

    // ---- round 0: by the low nibble, keys in registers ----------------------------------------------------------
    uint32_t s0[KPT];
    {
        uint32_t x4[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            x4[i] = __builtin_amdgcn_ubfe(field_word(k[i], hi), sh, 4u) << 2;
        }
        rank_round(x4, s0);
        // the key's high nibble (<< 2) goes to whoever owns slot s0
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            lds_store_at<unsigned char>(H_BYTES + s0[i], static_cast<unsigned char>(__builtin_amdgcn_ubfe(field_word(k[i], hi), sh + 4u, 4u) << 2));
        }
    }
    __syncthreads();
    // ---- round 1: slots 16t .. 16t+15 of that order, by the high nibble -----------------------------------------
    {
        const U32x4 hrow = *reinterpret_cast<const U32x4*>(smem + L::H_AT + tid * 4);
        uint32_t x4[KPT], s1[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            x4[j] = __builtin_amdgcn_ubfe(hrow.v[j >> 2], 8u * (j & 3), 8u);
        }
        rank_round(x4, s1);
        U32x4 ma, mb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ma.v[q] = s1[2 * q] | (s1[2 * q + 1] << 16);
            mb.v[q] = s1[8 + 2 * q] | (s1[8 + 2 * q + 1] << 16);
        }
        *reinterpret_cast<U32x4*>(smem + L::M_AT + tid * 8) = ma;
        *reinterpret_cast<U32x4*>(smem + L::M_AT + tid * 8 + 4) = mb;
    }
    __syncthreads();
    // ---- final slot of this thread's own keys; one trip through the image ----------------------------------------
    uint32_t f[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        f[i] = *reinterpret_cast<const u16_alias*>(reinterpret_cast<const unsigned char*>(smem) + M_BYTES + s0[i] * 2u);
    }
    __syncthreads();                 // the image overlays the work area: nobody may still be reading M
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        lds_store_at<Key>(add_lshl<(sizeof(Key) == 4 ? 2 : 3)>(f[i], f[i] >> L::PADSH), k[i]);
    }
    if constexpr (PAYLOAD) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            lds_store_at<uint32_t>(PBUF_BYTES + add_lshl<2>(f[i], f[i] >> 5), pl[i]);
        }
    }
    __syncthreads();
    // leave as runs: slot i = r*THREADS + tid, its global slot = gb[digit] + i
    constexpr uint32_t RSTRIDE = THREADS + (THREADS >> L::PADSH);
    const uint32_t rd_base = tid + (tid >> L::PADSH);
    const Key* xk = reinterpret_cast<const Key*>(smem);
    Key okey[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
        okey[r] = xk[rd_base + static_cast<uint32_t>(r) * RSTRIDE];
    }
    uint32_t pay[PAYLOAD ? KPT : 1];
    if constexpr (PAYLOAD) {
        constexpr uint32_t PSTRIDE = THREADS + (THREADS >> 5);
        const uint32_t pd_base = L::XBUF_DW + tid + (tid >> 5);
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            pay[r] = smem[pd_base + static_cast<uint32_t>(r) * PSTRIDE];
        }
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
    if (flip != Key{0}) {        // (uniform)
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            okey[r] ^= flip;
        }
    }
    if (full) {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            out[g[r]] = okey[r];
        }
        if constexpr (PAYLOAD) {
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                pout[g[r]] = pay[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            if (static_cast<uint32_t>(r) * THREADS + tid < valid) {
                out[g[r]] = okey[r];
                if constexpr (PAYLOAD) {
                    pout[g[r]] = pay[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// reorder8 v3 (round 3): ranks from ONE returning LDS atomic per key
// ---------------------------------------------------------------------------
// The two kernels above rank by the 4-bit machinery twice and are issue-bound (737 VALU per wave of 1,024 keys, vector ALUs busy
// 78 % of a launch; profiles/r03_tuning_log.md §1).  Here a wave holds its 1,024 consecutive keys STRIPED — instruction i of lane l
// is key i*64 + l — and every key takes one `ds_add_rtn_u32` on its wave's own 256 counters: LDS operations of a wave execute in
// issue order, and lanes of one instruction that meet on an address are served in ascending lane order, so the returned value is
// the key's rank among the wave's earlier keys of the same digit in INDEX order — a stable rank for ≈ 2 VALU.  (The lane order is
// measured behaviour of this hardware, not an architectural promise: rsx_create's first use of the 8-bit path runs
// lds_atomic_order_probe_kernel and the engine falls back to the two-round kernel if it ever fails; the stable-argsort tests run
// on this kernel.)  A block scan over the 4 x 256 counts in (digit, wave) order turns counts into bases; slot = base + rank; keys and
// payload make one trip through the padded image and leave as runs.  5 barriers, ≈ 300 VALU per wave.
// A wave whose 64 keys of an instruction all share the digit (constant / long-run data) would serialise 64 lanes on one address:
// instruction 0 stands for the wave as in reorder_kernel's look-ahead — only where it is uniform is every instruction tested and a
// uniform one handled by lane 0 alone (+64, rank = returned value + lane).
template <typename Key, int THREADS, int KPT, bool PAYLOAD>
struct Reorder8V3Layout {
    static constexpr int TILE = THREADS * KPT;
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int PADSH = (KD == 1) ? 5 : 4;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int XBUF_DW = (TILE + (TILE >> PADSH)) * KD;
    static constexpr int PBUF_DW = PAYLOAD ? TILE + (TILE >> 5) : 0;
    static constexpr int IMAGE_DW = XBUF_DW + PBUF_DW;
    static constexpr int CNT_DW = WAVES * kRadix8;                       // [wave][digit]; dead before the image is written: shares its LDS
    static constexpr int BODY_DW = IMAGE_DW > CNT_DW ? IMAGE_DW : CNT_DW;
    static constexpr int TOTAL_DW = BODY_DW + 16 + kRadix8;              // + wave totals + run bases
    static constexpr size_t BYTES = static_cast<size_t>(TOTAL_DW) * 4;
    static constexpr int WGS_PER_CU = static_cast<int>((160 * 1024) / BYTES);
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > 5 ? 5 : (WGS_PER_CU * THREADS / 256);
    static_assert(THREADS == kRadix8 && KPT * kWave * WAVES == TILE, "one thread per digit scans the counts; a wave's keys are KPT rows of 64");
};

__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t* p, uint32_t v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Does `ds_add_rtn_u32` hand lanes that meet on one address their old values in ascending lane order, and do a wave's LDS atomics
// execute in issue order?  out[0] = number of (instruction, lane) pairs whose returned rank differs from the rank computed with
// ballots; one workgroup of 256 threads, 64 rounds of 16 instructions with digit patterns from all-equal to all-distinct.
__global__ __launch_bounds__(256) void lds_atomic_order_probe_kernel(uint32_t* out)
{
    __shared__ uint32_t cnt[4][kRadix8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t bad = 0;
    for (uint32_t round = 0; round < 64; ++round) {
        for (uint32_t c = tid; c < 4 * kRadix8; c += 256) {
            (&cnt[0][0])[c] = 0;
        }
        __syncthreads();
        const uint32_t spread = 1u + (round * 37u) % 255u;          // number of distinct digits in play
        uint32_t got[16], dig[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t h = (tid * 2654435761u) ^ (round * 40503u) ^ (static_cast<uint32_t>(i) * 97u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            dig[i] = h % spread;
            got[i] = lds_add_rtn(&cnt[wave][dig[i]], 1u);
        }
        // reference ranks without atomics: per instruction, the count the earlier instructions left in a second table + the lower lanes
        // of this instruction with the same digit (ballots over the digit's 8 bits); the highest lane of each group writes the new count
        __syncthreads();
        __shared__ uint32_t ref[4][kRadix8];
        for (uint32_t c = tid; c < 4 * kRadix8; c += 256) {
            (&ref[0][0])[c] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t before = ref[wave][dig[i]];               // all lanes read the count left by instructions 0..i-1
            __builtin_amdgcn_wave_barrier();
            unsigned long long same = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (dig[i] >> b) & 1u;
                const unsigned long long vote = __ballot(bit);
                same &= bit ? vote : ~vote;
            }
            const uint32_t lower = static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
            if (before + lower != got[i]) {
                ++bad;
            }
            const bool leader = (same >> lane) == 1ull;
            if (leader) {
                ref[wave][dig[i]] = before + static_cast<uint32_t>(__popcll(same));
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (bad) {
        atomicAdd(out, bad);
    }
}

template <typename Key, int THREADS, int KPT, bool PAYLOAD>
__global__ __launch_bounds__(THREADS, (Reorder8V3Layout<Key, THREADS, KPT, PAYLOAD>::MIN_WAVES)) void reorder8v3_kernel(
    const Key* __restrict__ in, Key* __restrict__ out, const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
    const uint32_t* __restrict__ table8, const uint32_t* __restrict__ gsum8, const uint32_t* __restrict__ cbase8,
    uint32_t chunk_groups, uint64_t n, uint32_t ntiles, uint32_t tiles_per_xcd, int remap, int shift, Key flip)
{
    using L = Reorder8V3Layout<Key, THREADS, KPT, PAYLOAD>;
    constexpr int TILE = L::TILE;
    constexpr uint32_t PBUF_BYTES = L::XBUF_DW * 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* cnt = smem;                         // [wave][digit] counts, then bases
    uint32_t* wtot = smem + L::BODY_DW;
    uint32_t* gb = wtot + 16;                     // per digit: (global slot of the tile's first key with it) - (its tile-local slot)
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    // (LDS base 0: checked on the host, rsx_create)
    const uint64_t base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t left = n - base;
    const uint32_t valid = left < static_cast<uint64_t>(TILE) ? static_cast<uint32_t>(left) : static_cast<uint32_t>(TILE);
    const bool full = (valid == TILE);
    const Key pad_key = static_cast<Key>(~Key{0});        // digit 255, behind every real key of the tile (keys are held sign-flipped)
    const bool hi = sizeof(Key) == 8 && shift >= 32;
    const uint32_t sh = static_cast<uint32_t>(shift) & 31u;

    const uint32_t group = tile / kScan8Tiles;
    const uint32_t my_base = table8[static_cast<uint64_t>(tile) * kRadix8 + tid] + gsum8[static_cast<uint64_t>(group) * kRadix8 + tid] +
                             cbase8[static_cast<uint64_t>(group / chunk_groups) * kRadix8 + tid];

    // striped: element i of this thread is key wave*1024 + i*64 + lane of the tile (a wave-instruction reads 64 consecutive keys)
    const uint32_t first_li = wave * (KPT * kWave) + lane;
    Key k[KPT];
    uint32_t pl[PAYLOAD ? KPT : 1];
    if (full) {
        const Key* src = in + base + first_li;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            k[i] = src[i * kWave];
        }
        if constexpr (PAYLOAD) {
            const uint32_t* psrc = pin + base + first_li;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                pl[i] = psrc[i * kWave];
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
            const uint32_t li = first_li + static_cast<uint32_t>(i) * kWave;
            k[i] = li < valid ? static_cast<Key>(in[base + li] ^ flip) : pad_key;
            if constexpr (PAYLOAD) {
                pl[i] = li < valid ? pin[base + li] : 0u;
            }
        }
    }
    gb[tid] = my_base;           // (read after several barriers)
#pragma unroll
    for (int q = 0; q < L::WAVES; ++q) {
        cnt[q * kRadix8 + tid] = 0;
    }
    __syncthreads();
    // ---- ranks: one returning LDS atomic per key on the wave's own counters -------------------------------------------------------
    uint32_t* wcnt = cnt + wave * kRadix8;
    uint32_t slot[KPT];          // first the rank inside (wave, digit), then the tile-local slot
    {
        const uint32_t d0 = __builtin_amdgcn_ubfe(field_word(k[0], hi), sh, 8u);
        if (__builtin_expect(__ballot(d0 != static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(d0)))) != 0ull, 1)) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                slot[i] = lds_add_rtn(wcnt + __builtin_amdgcn_ubfe(field_word(k[i], hi), sh, 8u), 1u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = __builtin_amdgcn_ubfe(field_word(k[i], hi), sh, 8u);
                const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(d)));
                if (__ballot(d != first) == 0ull) {
                    uint32_t old = 0;
                    if (lane == 0) {
                        old = lds_add_rtn(wcnt + first, static_cast<uint32_t>(kWave));
                    }
                    slot[i] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(old))) + lane;
                } else {
                    slot[i] = lds_add_rtn(wcnt + d, 1u);
                }
            }
        }
    }
    __syncthreads();
    // ---- counts -> bases in (digit, wave) order: thread d owns digit d ----------------------------------------------------------
    {
        uint32_t c[L::WAVES];
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < L::WAVES; ++q) {
            c[q] = cnt[q * kRadix8 + tid];
            tot += c[q];
        }
        uint32_t all;
        uint32_t run = block_exclusive_scan<THREADS, false>(tot, wtot, all);      // keys of the tile with a smaller digit
#pragma unroll
        for (int q = 0; q < L::WAVES; ++q) {
            cnt[q * kRadix8 + tid] = run;
            run += c[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        slot[i] += wcnt[__builtin_amdgcn_ubfe(field_word(k[i], hi), sh, 8u)];
    }
    __syncthreads();                 // the image overlays the counters: nobody may still be reading them
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        lds_store_at<Key>(add_lshl<(sizeof(Key) == 4 ? 2 : 3)>(slot[i], slot[i] >> L::PADSH), k[i]);
    }
    if constexpr (PAYLOAD) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            lds_store_at<uint32_t>(PBUF_BYTES + add_lshl<2>(slot[i], slot[i] >> 5), pl[i]);
        }
    }
    __syncthreads();
    // leave as runs: slot i = r*THREADS + tid, its global slot = gb[digit] + i
    constexpr uint32_t RSTRIDE = THREADS + (THREADS >> L::PADSH);
    const uint32_t rd_base = tid + (tid >> L::PADSH);
    const Key* xk = reinterpret_cast<const Key*>(smem);
    Key okey[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
        okey[r] = xk[rd_base + static_cast<uint32_t>(r) * RSTRIDE];
    }
    uint32_t pay[PAYLOAD ? KPT : 1];
    if constexpr (PAYLOAD) {
        constexpr uint32_t PSTRIDE = THREADS + (THREADS >> 5);
        const uint32_t pd_base = L::XBUF_DW + tid + (tid >> 5);
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            pay[r] = smem[pd_base + static_cast<uint32_t>(r) * PSTRIDE];
        }
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
    if (flip != Key{0}) {        // (uniform)
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            okey[r] ^= flip;
        }
    }
    if (full) {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            out[g[r]] = okey[r];
        }
        if constexpr (PAYLOAD) {
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                pout[g[r]] = pay[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            if (static_cast<uint32_t>(r) * THREADS + tid < valid) {
                out[g[r]] = okey[r];
                if constexpr (PAYLOAD) {
                    pout[g[r]] = pay[r];
                }
            }
        }
    }
}

}  // namespace rsx
