// oracle/ref_shim.cpp — C ABI over the REFERENCE's own headers.
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// This translation unit contains no reference source: it `#include`s
// CRadixSortCPU.h and Dataset.h from where they lie under /root/reference
// (include path given by oracle/Makefile) and wraps them for ctypes.  The
// result, oracle/_ref/libref_oracle.so, is git-ignored; it pins the restatement
// in radix_sort_cpu.hpp and may serve as bench.py's cpu_baseline ("reference").
#include "CRadixSortCPU.h"   // /root/reference/src/CRadixSortCPU.h
#include "Dataset.h"         // /root/reference/src/Dataset.h

#include <chrono>
#include <cstring>
#include <span>

namespace {

enum : int { U32 = 0, I32 = 1, U64 = 2, I64 = 3 };

template <typename F>
int by_type(int dtype, F&& f)
{
    switch (dtype) {
    case U32: f(static_cast<std::uint32_t*>(nullptr)); return 0;
    case I32: f(static_cast<std::int32_t*>(nullptr)); return 0;
    case U64: f(static_cast<std::uint64_t*>(nullptr)); return 0;
    case I64: f(static_cast<std::int64_t*>(nullptr)); return 0;
    default: return -1;
    }
}

template <typename T, typename DS>
void emit(void* out, std::uint64_t n)
{
    DS ds(static_cast<std::size_t>(n));
    std::memcpy(out, ds.dataset.data(), static_cast<std::size_t>(n) * sizeof(T));
}

}  // namespace

extern "C" {

int ref_radix_sort(int dtype, void* keys, std::uint64_t n)
{
    if (n == 0) return 0;   // the reference dereferences max_element of an empty span
    return by_type(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        std::span<T> view(static_cast<T*>(keys), static_cast<std::size_t>(n));
        RadixSortCPU<T>::sort(view);
    });
}

/// kind: 0 Zeros, 1 Range, 2 InvertedRange, 3 Random (the clock-seeded
/// RandomDistributed is not reproducible and is not exposed).
int ref_dataset(int kind, int dtype, void* out, std::uint64_t n)
{
    if (kind < 0 || kind > 3) return -2;
    return by_type(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        switch (kind) {
        case 0: emit<T, Zeros<T>>(out, n); break;
        case 1: emit<T, Range<T>>(out, n); break;
        case 2: emit<T, InvertedRange<T>>(out, n); break;
        case 3: emit<T, Random<T>>(out, n); break;
        }
    });
}

int ref_time_radix_sort(int dtype, const void* in, void* scratch, std::uint64_t n, int iters, double* mean_ms)
{
    return by_type(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            std::memcpy(scratch, in, static_cast<std::size_t>(n) * sizeof(T));
            std::span<T> view(static_cast<T*>(scratch), static_cast<std::size_t>(n));
            RadixSortCPU<T>::sort(view);
        }
        const auto t1 = std::chrono::steady_clock::now();
        *mean_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
    });
}

}  // extern "C"
