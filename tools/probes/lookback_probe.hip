// lookback_probe.hip — what would a chained (decoupled look-back) scan cost on this chip?
//
// VERDICT r01 #5 asked for a measured A/B of resolving per-tile bases with a look-back inside the scatter launch instead of
// the look-ahead table + scan launch.  This probe isolates exactly that mechanism: `ntiles` workgroups of 256 threads (the
// reorder kernel's shape, same dynamic LDS so the same 5 workgroups per CU are resident), tile order = ticket order (a tile's
// predecessors have always started), every tile spends `pre` + `post` "work" (s_sleep loops sized like the reorder's phases),
// publishes its 16 per-digit aggregates as {epoch, state, value} granules (agent-scope write-through stores), then 16 lanes
// walk back over the predecessors — `window` granules in flight per lane — until they meet an inclusive prefix, and publish
// their own.  Reported: kernel time with and without the look-back, the mean number of predecessors a lane had to read, and
// the result is checked (exclusive prefix of known values).
//
// Round 3: `per_xcd` = 1 runs EIGHT chains instead of one: each XCD (read from HW_REG_XCC_ID, so that a chain's
// granules really stay in one L2 wherever the dispatcher put the workgroup) draws tickets from its own word and chains only over
// its own contiguous eighth of the tiles — the first tile of a range starts from a known base, as it would if the per-range digit
// offsets came from a kernel boundary.  A workgroup whose XCD's range is used up takes a ticket of the emptiest other range.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lookback_probe.hip -o gpurun_out/lookback_probe && gpurun_out/lookback_probe [ntiles] [lds KiB] [per_xcd]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr unsigned kAgg = 1u, kInc = 2u;

__device__ __forceinline__ void spin_work(int iters)
{
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_s_sleep(8);      // ~8 x 64 cycles
    }
}

template <int WINDOW>
__global__ __launch_bounds__(256) void lookback_kernel(unsigned long long* status, unsigned* ticket, unsigned epoch, unsigned* excl_out, unsigned long long* depth_sum,
                                                        int pre, int post, int do_lookback, unsigned* timeout, unsigned per_range)
{
    extern __shared__ unsigned smem[];
    __shared__ unsigned s_tile;
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        if (per_range == 0) {
            s_tile = atomicAdd(ticket, 1u);
        } else {
            // one ticket word per XCD (64 bytes apart); a range that is used up sends the workgroup to the next one
            unsigned x = __builtin_amdgcn_s_getreg((20u /* HW_REG_XCC_ID */) | (0u << 6) | ((4u - 1u) << 11)) & 7u;
            unsigned t = per_range;
            for (int tries = 0; tries < 8 && t >= per_range; ++tries, x = (x + 1u) & 7u) {
                t = atomicAdd(ticket + x * 16u, 1u);
                if (t < per_range) {
                    t += x * per_range;
                    break;
                }
                t = per_range;
            }
            s_tile = t;
        }
    }
    smem[tid] = tid;                      // touch the dynamic LDS so that the allocation is real
    __syncthreads();
    const unsigned tile = s_tile;
    const bool chain_head = per_range ? (tile % per_range) == 0 : tile == 0;      // starts from a known base
    spin_work(pre);
    if (tid < 16 && do_lookback) {
        const unsigned value = tile * 16u + tid + 1u;
        __hip_atomic_store((gu64*)status + (unsigned long long)tile * 16 + tid, ((unsigned long long)(epoch * 4u + (chain_head ? kInc : kAgg)) << 32) | value,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned sum = 0, depth = 0, spins = 0;
        bool done = chain_head;
        long long j = (long long)tile - 1;
        while (!done) {
            unsigned long long x[WINDOW];
#pragma unroll
            for (int w = 0; w < WINDOW; ++w) {
                x[w] = (j - w >= 0) ? __hip_atomic_load((gu64*)status + (unsigned long long)(j - w) * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int w = 0; w < WINDOW; ++w) {
                if (done || j - w < 0) {
                    continue;
                }
                const unsigned tag = (unsigned)(x[w] >> 32);
                if ((tag >> 2) != epoch) {       // not published yet: re-read from this predecessor on
                    j -= w;
                    goto again;
                }
                sum += (unsigned)x[w];
                ++depth;
                if ((tag & 3u) == kInc) {
                    done = true;
                }
            }
            j -= WINDOW;
            if (j < 0) {
                done = true;
            }
            continue;
        again:
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                *timeout = 1u;
                break;
            }
        }
        if (!chain_head) {
            __hip_atomic_store((gu64*)status + (unsigned long long)tile * 16 + tid, ((unsigned long long)(epoch * 4u + kInc) << 32) | (sum + value),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        excl_out[tile * 16 + tid] = sum;
        if (tid == 0) {
            atomicAdd(depth_sum, (unsigned long long)depth);
        }
    }
    __syncthreads();
    spin_work(post);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int g_lds_bytes = 27 * 1024;
static unsigned g_per_xcd = 0;

template <int WINDOW>
int run(unsigned ntiles, int pre, int post, unsigned long long* status, unsigned* ticket, unsigned* excl, unsigned long long* depth, unsigned* timeout, unsigned& epoch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int lb = 0; lb < 2; ++lb) {
        float best = 1e9f;
        unsigned long long dsum = 0;
        for (int rep = 0; rep < 3; ++rep) {
            ++epoch;
            CK(hipMemset(ticket, 0, 8 * 64));
            CK(hipMemset(depth, 0, 8));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(lookback_kernel<WINDOW>, dim3(ntiles), dim3(256), g_lds_bytes, 0, status, ticket, epoch, excl, depth, pre, post, lb, timeout,
                               g_per_xcd ? ntiles / 8 : 0u);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            CK(hipMemcpy(&dsum, depth, 8, hipMemcpyDeviceToHost));
        }
        unsigned to = 0;
        CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        bool ok = true;
        if (lb) {
            std::vector<unsigned> h(static_cast<size_t>(ntiles) * 16);
            CK(hipMemcpy(h.data(), excl, h.size() * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned> run16(16, 0);
            for (unsigned t = 0; t < ntiles && ok; ++t) {
                if (g_per_xcd && t % (ntiles / 8) == 0) run16.assign(16, 0);      // every range's chain starts from its own base
                for (unsigned d = 0; d < 16; ++d) {
                    ok = ok && h[static_cast<size_t>(t) * 16 + d] == run16[d];
                    run16[d] += t * 16u + d + 1u;
                }
            }
        }
        std::printf("  window %2d  look-back %s: %.3f ms%s", WINDOW, lb ? "on " : "off", best, lb ? "" : "\n");
        if (lb) std::printf("  (mean predecessors read per lane %.1f, prefixes %s%s)\n", double(dsum) / ntiles, ok ? "correct" : "WRONG", to ? ", TIMEOUT" : "");
    }
    return 0;
}

int main(int argc, char** argv)
{
    const unsigned ntiles = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 65536;
    if (argc > 2) g_lds_bytes = std::atoi(argv[2]) * 1024;     // e.g. 108: one workgroup per CU, a 16k-key "supertile" of four 4k tiles
    if (argc > 3) g_per_xcd = std::atoi(argv[3]) != 0;         // eight chains, one per XCD (ntiles must be a multiple of 8)
    if (g_per_xcd && ntiles % 8) return 2;
    std::printf("%s\n", g_per_xcd ? "EIGHT chains: one ticket word and one contiguous tile range per XCD (HW_REG_XCC_ID)" : "ONE chain over all tiles");
    unsigned long long *status, *depth;
    unsigned *ticket, *excl, *timeout;
    CK(hipMalloc(&status, static_cast<size_t>(ntiles) * 16 * 8));
    CK(hipMemset(status, 0, static_cast<size_t>(ntiles) * 16 * 8));
    CK(hipMalloc(&depth, 8));
    CK(hipMalloc(&ticket, 8 * 64));
    CK(hipMalloc(&excl, static_cast<size_t>(ntiles) * 16 * 4));
    CK(hipMalloc(&timeout, 4));
    CK(hipMemset(timeout, 0, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lookback_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lookback_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lookback_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_bytes));
    unsigned epoch = 0;
    // work sized like the reorder's phases at 5 resident workgroups per CU: ~3 us before the counts are known, ~4 us after
    const int shapes[3][2] = {{0, 0}, {12, 16}, {24, 32}};
    for (const auto& s : shapes) {
        std::printf("%u tiles, work before/after the look-back: %d / %d x s_sleep(8)\n", ntiles, s[0], s[1]);
        if (run<1>(ntiles, s[0], s[1], status, ticket, excl, depth, timeout, epoch)) return 1;
        if (run<4>(ntiles, s[0], s[1], status, ticket, excl, depth, timeout, epoch)) return 1;
        if (run<16>(ntiles, s[0], s[1], status, ticket, excl, depth, timeout, epoch)) return 1;
    }
    return 0;
}
