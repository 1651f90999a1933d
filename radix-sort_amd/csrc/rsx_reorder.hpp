// rsx_reorder.hpp — reorder_kernel: the stable scatter (the reference's `reorder`, RadixSort.cl:74-119), with the look-ahead histogram, the self-scan and the ranged (multi-GPU partition) variants.
// Part of rsx_kernels.hpp (the overview of all kernels and their reference counterparts is there).
#pragma once

#include "rsx_common.hpp"
#include "rsx_scan.hpp"

namespace rsx {

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
// Two replicas: every counter in two adjacent copies, odd and even lanes adding to different ones, so that the 32
// lanes of one LDS pass hit 32 different words instead of piling two deep on 16 addresses.  Before the XCD stagger this made no
// difference (the kernel waited for HBM); with it, interleaved A/B: 3.345-3.392 against 3.369-3.423 ms per sort back to back,
// and 0.417-0.424 against 0.430-0.447 ms per scatter launch right after an upload, when the shader clock is low and the waves
// wait for LDS issue (SQ counters: 23 % of their cycles, bank conflicts on 49 % of the LDS cycles with one copy).
// (non-temporal key loads / scatter stores were measured and rejected: profiles/r03_tuning_log.md §6-7)
constexpr int kLaReplicas = 2;

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
// computed address.  The assumption is checked on the host, once per engine: every kernel that uses this is asked for its static LDS size
// (must be 0) and a probe kernel reports where a lone dynamic array starts (rsx_create -> RSX_KERNEL_CREATION_FAILED otherwise).
template <typename T>
__device__ __forceinline__ void lds_store_at(uint32_t byte_offset, T value)
{
    *reinterpret_cast<__attribute__((address_space(3))) T*>(static_cast<uintptr_t>(byte_offset)) = value;
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

// (the packed counters sharing LDS with the staging image — one workgroup more per CU for one barrier more per tile — was measured
// and rejected: profiles/r01_tuning_log.md)
template <typename Key, int THREADS, int KPT>
struct ReorderLayout {
    static constexpr int TILE = THREADS * KPT;
    static constexpr int KD = sizeof(Key) / 4;
    static constexpr int PADSH = (KD == 1) ? 5 : 4;
    static constexpr int XELEMS = TILE + (TILE >> PADSH);
    static constexpr int XBUF_DW = XELEMS * KD;
    static constexpr int CNT_DW = 8 * THREADS;
    static constexpr int CNT_AT = XBUF_DW;                                          // dword offset of the counters
    static constexpr int IMAGE_DW = XBUF_DW + CNT_DW;
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
    static constexpr int WAVES_CAP = 6;      // 7 (72 VGPRs) measured twice, before and after the XCD stagger: see the tuning log
    static constexpr int MIN_WAVES = (WGS_PER_CU * THREADS / 256) > WAVES_CAP ? WAVES_CAP : (WGS_PER_CU * THREADS / 256);
    static_assert(TILE <= 32768, "16-bit packed counters");
    static_assert(KPT % (16 / sizeof(Key)) == 0 && THREADS % 64 == 0 && THREADS % 8 == 0, "geometry");
};

// Register budget: keys-only kernels are held to the occupancy LDS allows; payload kernels carry
// twice the per-key state (key, slot, payload, target) and are given 128 VGPRs instead of spilling.
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool RANGED = false>
constexpr int reorder_min_waves()
{
    constexpr int w = ReorderLayout<Key, THREADS, KPT>::MIN_WAVES;
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

// Inline table scan (mid-size sorts, INLINE_SCAN kernels): the launch has no scan kernel in front of it.  Its first `ngroups`
// workgroups (in dispatch order the first to start) each scan one group of 256 tiles of THIS pass's raw counts before turning to their
// own tile — the fused scan's workgroup body (fused_scan_group), with the finished table entries published write-through and a
// per-group `ready` word; every workgroup's lane 0 polls the word of its tile's group while the other lanes rank the tile's keys, and
// the 16 table entries of the tile are then read with sc1 loads.  One dependent launch per pass instead of two.  The scan reads (and
// zeroes) `counts`, the look-ahead of the same launch adds into the OTHER count buffer: the two alternate from pass to pass.
struct InlineScanArgs {
    unsigned long long* sums;      // granules [group][16] (the fused scan's)
    uint32_t* scanned;             // scanned group sums, as the fused scan leaves them
    uint32_t* temp;
    uint32_t* counts;              // raw [tile][16] counts of this pass (from_counts), handed back zeroed
    uint32_t* ready;               // [group] = epoch once the group's table entries are published
    uint32_t* timeout;
    uint32_t epoch;
    uint32_t ngroups;
    int from_counts;               // 0: the raw counts sit in the table itself ([digit][tile], after the histogram kernel)
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
// INLINE_SCAN kernels are instantiated by the experiments build only (-DRSX_EXPERIMENTS: measured slower than the scan launch they remove); the fused-scan
// body they carry needs registers of its own (142 VGPRs): three waves per SIMD are asked of them, otherwise the uint32 variants spill (52-180 bytes per lane in round 3).
template <typename Key, int THREADS, int KPT, bool PAYLOAD, bool LOOKAHEAD, bool RANGED = false, bool INLINE_SCAN = false>
__global__ __launch_bounds__(THREADS, (INLINE_SCAN ? 3 : reorder_min_waves<Key, THREADS, KPT, PAYLOAD, RANGED>())) void reorder_kernel(const Key* __restrict__ in, Key* __restrict__ out,
                                                           const uint32_t* __restrict__ pin, uint32_t* __restrict__ pout,
                                                           const uint32_t* table, uint64_t n, uint32_t ntiles,
                                                           uint32_t tiles_per_xcd, int remap, int shift, Key flip, uint32_t mask,
                                                           uint32_t* __restrict__ next_counts, int next_shift,
                                                           const uint32_t* __restrict__ globsum, Key lo, Key mul,
                                                           SplitSet<Key> split, SelfScanArgs self, InlineScanArgs iscan = InlineScanArgs{})
{
    static_assert(!INLINE_SCAN || (!RANGED && THREADS == kScanTiles), "the inline scan is the fused scan's workgroup: 256 threads, rsx_sort's passes only");
    using L = ReorderLayout<Key, THREADS, KPT>;
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
    // (before the surplus workgroups of the XCD mapping leave: a scanning workgroup scans whether or not it has a tile of its own)
    if constexpr (INLINE_SCAN) {
        if (blockIdx.x < iscan.ngroups) {           // workgroup-uniform
            fused_scan_group<true, true>(*reinterpret_cast<FusedScanLds*>(smem), blockIdx.x, const_cast<uint32_t*>(table), iscan.sums, iscan.scanned, iscan.temp,
                                         ntiles, iscan.ngroups, iscan.counts, iscan.from_counts != 0, iscan.epoch, iscan.timeout, iscan.ready);
            __syncthreads();                        // the scratch is this workgroup's dynamic LDS again
        }
    }
    const uint32_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, remap);
    if (tile >= ntiles) {
        return;
    }
    // (lds_store_at addresses the staging image from LDS address 0: rsx_create checks once, on the host, that this kernel has no static
    // LDS in front of its dynamic array and that such an array starts at 0 on this device — KERNEL_CREATION_FAILED otherwise)
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
            return ranged_bucket(static_cast<Key>((key ^ flip) - lo), shift, mul, mask);
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
    // (the pieces of the two table entries stay apart until phase 3 adds them up: summed here, inside the raking threads' branch, the adds and with them
    // the wait for these loads would stand BEFORE the key loads of the wave)
    uint32_t first_lo = 0, first_hi = 0, group_lo = 0, group_hi = 0;
    const bool self_scan = !RANGED && self.counts != nullptr;      // wave-uniform
    if (rake_head && !self_scan && !INLINE_SCAN) {
        const uint64_t e_lo = static_cast<uint64_t>(hl) * ntiles + tile;
        const uint64_t e_hi = static_cast<uint64_t>(hl + 8) * ntiles + tile;
        first_lo = table[e_lo];
        first_hi = table[e_hi];
        if (globsum) {
            // PasteHistogram folded in: the table holds block-local prefixes, add the scanned
            // sum of the scan group (256 tiles of one digit) each entry lives in (RadixSort.cl:185-197)
            const uint32_t ngroups = (ntiles + kScanTiles - 1) / kScanTiles;
            group_lo = globsum[static_cast<uint64_t>(hl) * ngroups + tile / kScanTiles];
            group_hi = globsum[static_cast<uint64_t>(hl + 8) * ngroups + tile / kScanTiles];
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
            asm volatile("" : "+v"(slot[i]));      // materialise the rank now: otherwise hipcc keeps all 16 intermediate `seen` values (32 VGPRs) and extracts the ranks after the loop
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
    if constexpr (INLINE_SCAN) {
        // lane 0 waits for the table entries of the tile's scan group while the other waves are still ranking (one poller per
        // workgroup, relaxed sc1 loads, bounded); the barrier below is the one the other waves join before they load
        if (tid == 0) {
            const uint32_t* word = iscan.ready + tile / kScanTiles;
            uint32_t spins = 0;
            while (__hip_atomic_load((gu32*)(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != iscan.epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {
                    __hip_atomic_store((gu32*)(iscan.timeout), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
    }
    __syncthreads();
    if constexpr (INLINE_SCAN) {
        if (rake_head) {
            first_lo = __hip_atomic_load((gu32*)(table) + static_cast<uint64_t>(hl) * ntiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            first_hi = __hip_atomic_load((gu32*)(table) + static_cast<uint64_t>(hl + 8) * ntiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    RSX_STAMP(3);

    // ---- 3. raking scan over the 8*THREADS packed words in [digit&7][thread] order ---
    {
        U32x4 a = *reinterpret_cast<const U32x4*>(cnt + tid * 8);
        U32x4 b = *reinterpret_cast<const U32x4*>(cnt + tid * 8 + 4);
        const uint32_t sum = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
        uint32_t total;
        // (no trailing barrier: `wtot` is not written again in this kernel, and the barrier after the packed words below orders the rest)
        uint32_t run = block_exclusive_scan<THREADS, false>(sum, wtot, total);
        // low halves now prefix digits 0..7, high halves digits 8..15; the latter start
        // after ALL keys with digit < 8, i.e. after total.low
        run += total << 16;
        if (rake_head) {
            // `run` is the scanned word of (true digit hl | hl+8, thread 0): the tile-local slot of the
            // tile's first key with that digit.  Stored where phase 5 looks it up: at the RAW digit.
            first_lo += group_lo;
            first_hi += group_hi;
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

}  // namespace rsx
