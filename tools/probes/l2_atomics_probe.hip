// l2_atomics_probe.hip — can counters that many workgroups add to be kept per XCD and updated by atomics that stay in that XCD's L2?
//
// Round 2 found (tuning log §7) that a few thousand same-address DEVICE-scope atomic adds per counter serialise at the memory side
// (~20 ns each across XCDs), which killed the two-level self-scan.  This probe times the same traffic three ways and checks the sums:
//   agent     one table, __HIP_MEMORY_SCOPE_AGENT adds (what the look-ahead flush uses; executes at the memory side)
//   agent/xcc eight tables indexed by HW_REG_XCC_ID, agent-scope adds (contention / 8, still at the memory side)
//   l2/xcc    eight tables indexed by HW_REG_XCC_ID, __HIP_MEMORY_SCOPE_WORKGROUP adds (no sc1: performed in the XCD's own L2;
//             a table is only ever touched through one L2, and the kernel boundary writes it back)
// Shape: `wgs` workgroups of 256 threads, each adds 1 to `per_wg` counters out of `slots` (consecutive lanes -> consecutive counters).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l2_atomics_probe.hip -o /tmp/l2a && /tmp/l2a [wgs] [slots] [per_wg]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void add_kernel(unsigned* tables, unsigned slots, unsigned per_wg)
{
    unsigned xcc = 0;
    if (MODE != 0) {
        xcc = __builtin_amdgcn_s_getreg((20u /* HW_REG_XCC_ID */) | (0u << 6) | ((4u - 1u) << 11)) & 7u;
    }
    unsigned* t = tables + static_cast<size_t>(xcc) * slots;
    for (unsigned i = threadIdx.x; i < per_wg; i += 256) {
        const unsigned s = (blockIdx.x * 37u + i) % slots;
        if (MODE == 2) {
            __hip_atomic_fetch_add(t + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(t + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv)
{
    const unsigned wgs = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 16384;
    const unsigned slots = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 2048;
    const unsigned per_wg = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 512;
    unsigned* tables;
    CK(hipMalloc(&tables, static_cast<size_t>(8) * slots * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const char* names[3] = {"agent    ", "agent/xcc", "l2/xcc   "};
    std::printf("%u workgroups x %u adds onto %u counters (%.0f adds per counter)\n", wgs, per_wg, slots, double(wgs) * per_wg / slots);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        bool ok = true;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(tables, 0, static_cast<size_t>(8) * slots * 4));
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(add_kernel<0>, dim3(wgs), dim3(256), 0, 0, tables, slots, per_wg);
            if (mode == 1) hipLaunchKernelGGL(add_kernel<1>, dim3(wgs), dim3(256), 0, 0, tables, slots, per_wg);
            if (mode == 2) hipLaunchKernelGGL(add_kernel<2>, dim3(wgs), dim3(256), 0, 0, tables, slots, per_wg);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            std::vector<unsigned> h(static_cast<size_t>(8) * slots);
            CK(hipMemcpy(h.data(), tables, h.size() * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> want(slots, 0), got(slots, 0);
            for (unsigned w = 0; w < wgs; ++w)
                for (unsigned i = 0; i < per_wg; ++i) want[(w * 37u + i) % slots] += 1;
            for (unsigned x = 0; x < 8; ++x)
                for (unsigned s = 0; s < slots; ++s) got[s] += h[static_cast<size_t>(x) * slots + s];
            for (unsigned s = 0; s < slots; ++s) ok = ok && got[s] == want[s];
        }
        std::printf("  %s  %.4f ms  (%.1f G adds/s)  sums %s\n", names[mode], best, double(wgs) * per_wg / best * 1e-6, ok ? "correct" : "WRONG");
    }
    return 0;
}
