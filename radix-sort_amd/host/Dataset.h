// Dataset.h — input generators with the reference's names and contents
// (/root/reference/src/Dataset.h:22-137): Zeros, Range, InvertedRange, Random,
// RandomDistributed.  They define every benchmark input; the
// Performance/*.csv files of the reference hold timings, not data.
//
// Differences, both deliberate:
//   * RandomDistributed is clock-seeded in the reference and draws through libstdc++'s
//     uniform_int_distribution (Dataset.h:95-101), so no two runs — or standard
//     libraries — agree.  Here it takes an explicit seed (default below, recorded in
//     every report) and uses raw engine words of the key's width; the forced extremes at
//     both ends (Dataset.h:105-106) are kept.  `RandomDistributed(size)` therefore is
//     reproducible; pass `ClockSeed()` to get the reference's behaviour.
//   * default size follows the run-time capacity instead of the fixed 2^25.
#pragma once

#include "Parameters.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <string>
#include <vector>

template <typename T>
using Parameters = AlgorithmParameters<T>;

/// Owning input vector + a display name (CSV column 3 of the reference's reports).
template <typename T>
struct Dataset {
    using DataType = T;

    explicit Dataset(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems())) : dataset(size) {}
    virtual ~Dataset() = default;
    virtual const char* name() const { return "UNKNOWN"; }

    std::vector<DataType> dataset;
};

namespace dataset_detail {

inline constexpr std::uint64_t kDefaultUniformSeed = 0x5EEDCAFEF00DULL;
inline constexpr const char* kRandomSeedText = "Random Test Seed";   // Dataset.h:113

inline std::uint64_t ClockSeed()
{
    return static_cast<std::uint64_t>(std::chrono::high_resolution_clock::now().time_since_epoch().count());
}

template <typename T>
void fill_zeros(std::vector<T>& v)
{
    std::fill(v.begin(), v.end(), T{0});
}

template <typename T>
void fill_range(std::vector<T>& v)
{
    std::iota(v.begin(), v.end(), std::numeric_limits<T>::min());
}

template <typename T>
void fill_inverted_range(std::vector<T>& v)
{
    fill_range(v);
    std::reverse(v.begin(), v.end());
}

/// mt19937 seeded from the characters of "Random Test Seed"; every key type receives
/// the 32-bit draw converted to T (zero-extended for 64-bit keys) — Dataset.h:113-119.
template <typename T>
void fill_random(std::vector<T>& v)
{
    const std::string text(kRandomSeedText);
    std::seed_seq seq(text.begin(), text.end());
    std::mt19937 engine(seq);
    for (auto& x : v) {
        x = static_cast<T>(engine());
    }
}

template <typename T>
void fill_uniform(std::vector<T>& v, std::uint64_t seed)
{
    std::seed_seq seq({static_cast<std::uint32_t>(seed & 0xFFFFFFFFULL), static_cast<std::uint32_t>(seed >> 32)});
    if constexpr (sizeof(T) == 8) {
        std::mt19937_64 engine(seq);
        for (auto& x : v) x = static_cast<T>(engine());
    } else {
        std::mt19937 engine(seq);
        for (auto& x : v) x = static_cast<T>(engine());
    }
    if (!v.empty()) {
        v.front() = std::numeric_limits<T>::max();
        v.back() = std::numeric_limits<T>::min();
    }
}

}  // namespace dataset_detail

template <typename T>
struct Zeros : Dataset<T> {
    explicit Zeros(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems())) : Dataset<T>(size)
    {
        dataset_detail::fill_zeros(this->dataset);
    }
    const char* name() const override { return "Zeros"; }
};

template <typename T>
struct Range : Dataset<T> {
    explicit Range(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems())) : Dataset<T>(size)
    {
        dataset_detail::fill_range(this->dataset);
    }
    const char* name() const override { return "Range"; }
};

template <typename T>
struct InvertedRange : Dataset<T> {
    explicit InvertedRange(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems())) : Dataset<T>(size)
    {
        dataset_detail::fill_inverted_range(this->dataset);
    }
    const char* name() const override { return "Inverted Range"; }
};

template <typename T>
struct Random : Dataset<T> {
    explicit Random(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems())) : Dataset<T>(size)
    {
        dataset_detail::fill_random(this->dataset);
    }
    const char* name() const override { return "Random Random"; }
};

template <typename T>
struct RandomDistributed : Dataset<T> {
    explicit RandomDistributed(std::size_t size = static_cast<std::size_t>(Parameters<T>::MaxInputElems()),
                               std::uint64_t seed = dataset_detail::kDefaultUniformSeed)
        : Dataset<T>(size), seed_used(seed)
    {
        dataset_detail::fill_uniform(this->dataset, seed);
    }
    const char* name() const override { return "Random Uniform"; }
    std::uint64_t seed_used;
};
