// CRadixSortCPU.h — the harness's CPU referee `RadixSortCPU<T>::sort`, same name and call
// shape as the reference's (/root/reference/src/CRadixSortCPU.h:30-123): a
// single-threaded, stable LSD counting sort in base `_TOTALBITS / _NUM_BITS_PER_RADIX`
// whose round count comes from the raw maximum.  It is a REFEREE that CRadixSortTask
// times and compares against (src/CRadixSortTask.cpp:173-252); the GPU path never calls
// it.  Its known short-count cases (max an exact power of the base, max == 1, signed data
// with a small raw maximum) are reproduced, not fixed: std::sort is the ground truth.
#pragma once

#include "Parameters.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <span>
#include <type_traits>
#include <vector>

template <typename DataType>
class RadixSortCPU {
public:
    using Parameters = AlgorithmParameters<DataType>;
    static constexpr std::uint32_t NUM_BINS = Parameters::_TOTALBITS / Parameters::_NUM_BITS_PER_RADIX;

    template <typename ElemType>
    static void sort(std::span<ElemType>& arr)
    {
        if (arr.empty()) return;
        const std::uint64_t rounds = roundsFor(*std::max_element(arr.begin(), arr.end()));
        std::vector<ElemType> scratch(arr.size());
        for (std::uint64_t r = 0; r < rounds; ++r) {
            countSort(arr, scratch, static_cast<std::uint64_t>(std::pow(NUM_BINS, r)));
        }
    }

    /// Counting rounds the sort runs for a given raw maximum (src/CRadixSortCPU.h:62,67).
    template <typename ElemType>
    static std::uint64_t roundsFor(ElemType maxElem)
    {
        using U = std::make_unsigned_t<ElemType>;
        if (maxElem == 0) return 1;
        const U mag = maxElem < 0 ? static_cast<U>(U{0} - static_cast<U>(maxElem)) : static_cast<U>(maxElem);
        return static_cast<std::uint64_t>(std::ceil(std::log(mag) / std::log(NUM_BINS)));
    }

    /// True when the round count covers every significant bit of the biased keys, i.e. the
    /// input lies in the domain where this referee is a correct sort.  Outside it (raw
    /// maximum an exact power of the base, or small while negative keys are present — e.g.
    /// a signed Range padded with zeros) the reference's algorithm stops early.
    template <typename ElemType>
    static bool coversAllDigits(std::span<const ElemType> arr)
    {
        using U = std::make_unsigned_t<ElemType>;
        if (arr.empty()) return true;
        constexpr U bias = static_cast<U>(std::numeric_limits<ElemType>::min());
        U top = 0;
        ElemType rawMax = arr.front();
        for (const ElemType v : arr) {
            top = std::max<U>(top, static_cast<U>(static_cast<U>(v) - bias));
            rawMax = std::max(rawMax, v);
        }
        unsigned bits = 0;
        while (top) {
            ++bits;
            top >>= 1;
        }
        unsigned perRound = 0;
        for (std::uint32_t b = NUM_BINS; b > 1; b >>= 1) ++perRound;   // log2(NUM_BINS)
        return roundsFor(rawMax) * perRound >= bits;
    }

private:
    template <typename ElemType>
    static void countSort(std::span<ElemType>& arr, std::vector<ElemType>& scratch, std::uint64_t weight)
    {
        using U = std::make_unsigned_t<ElemType>;
        constexpr U bias = static_cast<U>(std::numeric_limits<ElemType>::min());
        std::size_t ends[NUM_BINS] = {};
        auto bin = [weight](ElemType v) { return static_cast<std::size_t>((static_cast<U>(static_cast<U>(v) - bias) / weight) % NUM_BINS); };
        for (const ElemType v : arr) ++ends[bin(v)];
        for (std::uint32_t b = 1; b < NUM_BINS; ++b) ends[b] += ends[b - 1];
        for (std::size_t i = arr.size(); i-- > 0;) scratch[--ends[bin(arr[i])]] = arr[i];
        std::copy(scratch.begin(), scratch.end(), arr.begin());
    }
};
