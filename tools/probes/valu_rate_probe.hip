// valu_rate_probe.hip — issue cost of the integer VALU instructions the ranking phase of the reorder kernel is made of.
//
// The fused reorder kernel is issue-bound at the clock the power controller visits after idle: what an instruction
// costs decides which formulation of the per-thread digit counters is cheapest.  Every workgroup of 256 threads runs ITER
// iterations of 8 independent copies of one instruction (inline asm, nothing for the compiler to fold); with 8 waves per SIMD
// resident the SIMD is never short of ready waves, so time / (ITER * 8 * waves per SIMD) = cycles per wave64 instruction.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o tools/_variants/valu_rate_probe && tools/_variants/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define OP8_64(text)                                                                                                    \
    asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7)                                          \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])      \
                 : "v"(sh), "v"(one) : "vcc")
#define OP8_32(text)                                                                                                    \
    asm volatile(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7)                                          \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])      \
                 : "v"(sh), "v"(one) : "vcc", "s20", "s21")

#define LSHL32(i) "v_lshlrev_b32 %" #i ", %8, %" #i "\n\t"
#define ADD32(i) "v_add_u32 %" #i ", %8, %" #i "\n\t"
#define BFE32(i) "v_bfe_u32 %" #i ", %" #i ", %8, 4\n\t"
#define LSHL64(i) "v_lshlrev_b64 %" #i ", %8, %" #i "\n\t"
#define LSHR64(i) "v_lshrrev_b64 %" #i ", %8, %" #i "\n\t"
#define LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %9\n\t"
#define ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n\t"
#define LSHLADD32(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n\t"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", %8\n\t"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n\t"
#define ADD32E64(i) "v_add_u32_e64 %" #i ", %8, %" #i "\n\t"
#define AND32(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n\t"
#define LSHR32(i) "v_lshrrev_b32_e32 %" #i ", %8, %" #i "\n\t"
#define SUB32(i) "v_sub_u32_e32 %" #i ", %" #i ", %8\n\t"
#define MOV32(i) "v_mov_b32_e32 %" #i ", %8\n\t"
#define CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define CMPNE(i) "v_cmp_ne_u32_e32 vcc, %8, %" #i "\n\t"
#define CMPNE64(i) "v_cmp_ne_u32_e64 s[20:21], %8, %" #i "\n\t"
#define XOR32(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n\t"
#define PAIRVCC(i) "v_cmp_ne_u32_e32 vcc, %8, %" #i "\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n\t"
#define PAIRSGPR(i) "v_cmp_ne_u32_e64 s[20:21], %8, %" #i "\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define ADDC(i) "v_addc_co_u32_e32 %" #i ", vcc, %8, %" #i ", vcc\n\t"
#define DPPADD(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"

template <int WHICH>
__global__ __launch_bounds__(256) void probe(uint64_t* out, int iters, uint32_t sh_in)
{
    uint64_t a[8];
    uint32_t b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
        b[i] = threadIdx.x * 0x9E3779B9u + i;
    }
    uint32_t sh = sh_in;
    uint64_t one = 1;
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHICH == 0) OP8_32(LSHL32);
        if constexpr (WHICH == 1) OP8_32(ADD32);
        if constexpr (WHICH == 2) OP8_32(BFE32);
        if constexpr (WHICH == 3) OP8_64(LSHL64);
        if constexpr (WHICH == 4) OP8_64(LSHR64);
        if constexpr (WHICH == 5) OP8_64(LSHLADD64);
        if constexpr (WHICH == 6) OP8_32(ADD3);
        if constexpr (WHICH == 7) OP8_32(LSHLADD32);
        if constexpr (WHICH == 8) OP8_32(CNDMASK);
        if constexpr (WHICH == 9) OP8_32(ALIGNBIT);
        if constexpr (WHICH == 10) OP8_32(ANDOR);
        if constexpr (WHICH == 11) OP8_32(ADD32E64);
        if constexpr (WHICH == 12) OP8_32(AND32);
        if constexpr (WHICH == 13) OP8_32(LSHR32);
        if constexpr (WHICH == 14) OP8_32(SUB32);
        if constexpr (WHICH == 15) OP8_32(MOV32);
        if constexpr (WHICH == 16) { asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21"); OP8_32(CNDMASK64); }
        if constexpr (WHICH == 17) OP8_32(CMPNE);
        if constexpr (WHICH == 18) OP8_32(CMPNE64);
        if constexpr (WHICH == 19) OP8_32(XOR32);
        if constexpr (WHICH == 20) OP8_32(DPPADD);
        if constexpr (WHICH == 21) OP8_32(PAIRVCC);
        if constexpr (WHICH == 22) OP8_32(PAIRSGPR);
        if constexpr (WHICH == 23) OP8_32(ADDC);
        if constexpr (WHICH == 24) { asm volatile("s_mov_b64 vcc, exec" ::: "vcc"); OP8_32(CNDMASK); }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ b[i];
    if (s == 0x1234567ull) out[0] = s;       // keeps the loop alive
}

template <int WHICH>
void run(const char* name, uint64_t* d_out, double ghz)
{
    const int iters = 20000;
    const int blocks = 256 * 8;              // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0);
    (void)hipEventCreate(&t1);
    hipLaunchKernelGGL(probe<WHICH>, dim3(blocks), dim3(256), 0, 0, d_out, 1000, 4u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(t0, 0);
    hipLaunchKernelGGL(probe<WHICH>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 4u);
    (void)hipEventRecord(t1, 0);
    (void)hipEventSynchronize(t1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    const double instr_per_simd = static_cast<double>(iters) * 8.0 * 8.0;      // 8 per iteration x 8 waves per SIMD
    const double cycles = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
    std::printf("%-18s %8.3f ms  = %5.2f cycles per wave64 instruction at %.2f GHz\n", name, ms, cycles, ghz);
}

int main()
{
    uint64_t* d_out = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&d_out), 64);
    const double ghz = 2.4;
    for (int rep = 0; rep < 1; ++rep) {
        run<0>("v_lshlrev_b32", d_out, ghz);
        run<1>("v_add_u32", d_out, ghz);
        run<2>("v_bfe_u32", d_out, ghz);
        run<6>("v_add3_u32", d_out, ghz);
        run<7>("v_lshl_add_u32", d_out, ghz);
        run<8>("v_cndmask_b32", d_out, ghz);
        run<9>("v_alignbit_b32", d_out, ghz);
        run<10>("v_and_or_b32", d_out, ghz);
        run<11>("v_add_u32_e64", d_out, ghz);
        run<12>("v_and_b32_e32", d_out, ghz);
        run<13>("v_lshrrev_b32_e32", d_out, ghz);
        run<14>("v_sub_u32_e32", d_out, ghz);
        run<15>("v_mov_b32_e32", d_out, ghz);
        run<16>("v_cndmask_e64 sgpr", d_out, ghz);
        run<17>("v_cmp_ne_u32 vcc", d_out, ghz);
        run<18>("v_cmp_ne_u32 sgpr", d_out, ghz);
        run<19>("v_xor_b32_e32", d_out, ghz);
        run<20>("v_add_u32_dpp", d_out, ghz);
        run<21>("cmp+cndmask vcc (2)", d_out, ghz);
        run<22>("cmp+cndmask sgpr (2)", d_out, ghz);
        run<23>("v_addc_co_u32 vcc", d_out, ghz);
        run<24>("cndmask vcc, vcc set", d_out, ghz);
        run<3>("v_lshlrev_b64", d_out, ghz);
        run<4>("v_lshrrev_b64", d_out, ghz);
        run<5>("v_lshl_add_u64", d_out, ghz);
    }
    (void)hipFree(d_out);
    return 0;
}
