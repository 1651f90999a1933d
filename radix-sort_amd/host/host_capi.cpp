// host_capi.cpp — tiny C ABI over the host-side pieces that non-C++ callers need:
// the reference's input generators (Dataset.h) and Resize().  bench.py and the Python
// tests use it so that synthetic inputs are produced by the product's own Dataset code.
#include "Dataset.h"
#include "RadixSortGPU.h"

#include <cstring>

namespace {

template <typename T>
int fill(int kind, void* out, std::uint64_t n, std::uint64_t seed)
{
    std::vector<T> v(static_cast<std::size_t>(n));
    switch (kind) {
    case 0: dataset_detail::fill_zeros(v); break;
    case 1: dataset_detail::fill_range(v); break;
    case 2: dataset_detail::fill_inverted_range(v); break;
    case 3: dataset_detail::fill_random(v); break;
    case 4: dataset_detail::fill_uniform(v, seed); break;
    default: return -2;
    }
    std::memcpy(out, v.data(), v.size() * sizeof(T));
    return 0;
}

}  // namespace

extern "C" {

/// kind: 0 Zeros, 1 Range, 2 InvertedRange, 3 Random, 4 RandomDistributed(seed);
/// dtype: 0 uint32, 1 int32, 2 uint64, 3 int64.
int rsxh_dataset_fill(int kind, int dtype, void* out, std::uint64_t n, std::uint64_t seed)
{
    switch (dtype) {
    case 0: return fill<std::uint32_t>(kind, out, n, seed);
    case 1: return fill<std::int32_t>(kind, out, n, seed);
    case 2: return fill<std::uint64_t>(kind, out, n, seed);
    case 3: return fill<std::int64_t>(kind, out, n, seed);
    default: return -1;
    }
}

/// RadixSortGPU<T>::Resize — next multiple of 1024.
std::uint32_t rsxh_resize(std::uint32_t nn)
{
    return RadixSortGPU<std::uint32_t>().Resize(nn);
}

std::uint64_t rsxh_default_uniform_seed(void)
{
    return dataset_detail::kDefaultUniformSeed;
}

}  // extern "C"
