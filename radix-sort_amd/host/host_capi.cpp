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

// Elements [offset, offset + n) of the dataset of `total` elements, without materialising the rest: how the ranks of a sharded
// sort each produce their contiguous shard of ONE dataset (BASELINE config 4: `Random` sharded contiguously over 8 GPUs).
// Random / RandomDistributed skip ahead in the generator's stream (std::mt19937::discard); both draw once per element.
template <typename T>
int fill_shard(int kind, void* out, std::uint64_t offset, std::uint64_t n, std::uint64_t total, std::uint64_t seed)
{
    if (offset > total || n > total - offset) return -3;
    T* dst = static_cast<T*>(out);
    switch (kind) {
    case 0:
        for (std::uint64_t i = 0; i < n; ++i) dst[i] = T{0};
        return 0;
    case 1:      // iota from min()  (Dataset.h:123-129 before the reverse)
        for (std::uint64_t i = 0; i < n; ++i) dst[i] = static_cast<T>(std::numeric_limits<T>::min() + static_cast<T>(offset + i));
        return 0;
    case 2:      // the same, reversed over the WHOLE dataset
        for (std::uint64_t i = 0; i < n; ++i) dst[i] = static_cast<T>(std::numeric_limits<T>::min() + static_cast<T>(total - 1 - (offset + i)));
        return 0;
    case 3: {
        const std::string text(dataset_detail::kRandomSeedText);
        std::seed_seq seq(text.begin(), text.end());
        std::mt19937 engine(seq);
        engine.discard(offset);
        for (std::uint64_t i = 0; i < n; ++i) dst[i] = static_cast<T>(engine());
        return 0;
    }
    case 4: {
        std::seed_seq seq({static_cast<std::uint32_t>(seed & 0xFFFFFFFFULL), static_cast<std::uint32_t>(seed >> 32)});
        if constexpr (sizeof(T) == 8) {
            std::mt19937_64 engine(seq);
            engine.discard(offset);
            for (std::uint64_t i = 0; i < n; ++i) dst[i] = static_cast<T>(engine());
        } else {
            std::mt19937 engine(seq);
            engine.discard(offset);
            for (std::uint64_t i = 0; i < n; ++i) dst[i] = static_cast<T>(engine());
        }
        if (n > 0 && offset == 0) dst[0] = std::numeric_limits<T>::max();
        if (n > 0 && offset + n == total) dst[n - 1] = std::numeric_limits<T>::min();
        return 0;
    }
    default: return -2;
    }
}

}  // namespace

extern "C" {

/// Elements [offset, offset + n) of the `total`-element dataset of that kind (same codes as rsxh_dataset_fill).
int rsxh_dataset_fill_shard(int kind, int dtype, void* out, std::uint64_t offset, std::uint64_t n, std::uint64_t total, std::uint64_t seed)
{
    switch (dtype) {
    case 0: return fill_shard<std::uint32_t>(kind, out, offset, n, total, seed);
    case 1: return fill_shard<std::int32_t>(kind, out, offset, n, total, seed);
    case 2: return fill_shard<std::uint64_t>(kind, out, offset, n, total, seed);
    case 3: return fill_shard<std::int64_t>(kind, out, offset, n, total, seed);
    default: return -1;
    }
}

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
