// oracle/oracle_capi.cpp — C ABI over oracle/radix_sort_cpu.hpp for ctypes.
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see the header of radix_sort_cpu.hpp).
// Built by oracle/Makefile into oracle/liboracle.so.
#include "radix_sort_cpu.hpp"

#include <chrono>

namespace {

enum : int { U32 = 0, I32 = 1, U64 = 2, I64 = 3 };

template <typename F>
int dispatch(int dtype, F&& f)
{
    switch (dtype) {
    case U32: f(static_cast<std::uint32_t*>(nullptr)); return 0;
    case I32: f(static_cast<std::int32_t*>(nullptr)); return 0;
    case U64: f(static_cast<std::uint64_t*>(nullptr)); return 0;
    case I64: f(static_cast<std::int64_t*>(nullptr)); return 0;
    default: return -1;
    }
}

}  // namespace

extern "C" {

/// RadixSortCPU<T>::sort restated; in place.  payload may be NULL.
int oracle_radix_sort(int dtype, void* keys, std::uint32_t* payload, std::uint64_t n)
{
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        oracle::radix_sort(static_cast<T*>(keys), static_cast<std::size_t>(n), payload);
    });
}

/// How many counting rounds the reference would run on this input.
std::uint64_t oracle_round_count(int dtype, const void* keys, std::uint64_t n)
{
    std::uint64_t r = 0;
    if (n == 0) return 0;
    dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        r = oracle::round_count(static_cast<const T*>(keys), static_cast<std::size_t>(n));
    });
    return r;
}

int oracle_std_sort(int dtype, void* keys, std::uint64_t n)
{
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        oracle::std_sort(static_cast<T*>(keys), static_cast<std::size_t>(n));
    });
}

int oracle_stable_argsort(int dtype, const void* keys, std::uint32_t* payload_inout, std::uint64_t n)
{
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        oracle::stable_argsort(static_cast<const T*>(keys), payload_inout, static_cast<std::size_t>(n));
    });
}

int oracle_dataset(int kind, int dtype, void* out, std::uint64_t n, std::uint64_t seed)
{
    if (kind < 0 || kind > 4) return -2;
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        oracle::fill_dataset(static_cast<oracle::DatasetKind>(kind), static_cast<T*>(out), static_cast<std::size_t>(n), seed);
    });
}

std::uint64_t oracle_fnv1a64(const void* bytes, std::uint64_t nbytes)
{
    return oracle::fnv1a64(bytes, static_cast<std::size_t>(nbytes));
}

/// Host emulation of the reference's GPU pass structure (n must be a multiple of 1024).
int oracle_emulate_reference_gpu(int dtype, void* keys, std::uint64_t n, std::uint32_t* table_out, std::uint32_t* globsum_out)
{
    if (n % 1024 != 0) return -3;
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        oracle::emulate_reference_gpu_sort(static_cast<T*>(keys), static_cast<std::size_t>(n), table_out, globsum_out);
    });
}

/// CPU baseline timing, shaped like SortDataRadix (CRadixSortTask.cpp:50-58,210-220):
/// copy-in + sort inside the timed region, `iters` repetitions, mean milliseconds.
int oracle_time_radix_sort(int dtype, const void* in, void* scratch, std::uint64_t n, int iters, double* mean_ms)
{
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            std::memcpy(scratch, in, static_cast<std::size_t>(n) * sizeof(T));
            oracle::radix_sort(static_cast<T*>(scratch), static_cast<std::size_t>(n));
        }
        const auto t1 = std::chrono::steady_clock::now();
        *mean_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
    });
}

/// The second referee of the reference, timed the same way (SortDataSTL, :32-43,189-199).
int oracle_time_std_sort(int dtype, const void* in, void* scratch, std::uint64_t n, int iters, double* mean_ms)
{
    return dispatch(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            std::memcpy(scratch, in, static_cast<std::size_t>(n) * sizeof(T));
            oracle::std_sort(static_cast<T*>(scratch), static_cast<std::size_t>(n));
        }
        const auto t1 = std::chrono::steady_clock::now();
        *mean_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
    });
}

}  // extern "C"
