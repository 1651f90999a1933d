// HostData.h — the bundle of host memory the sort engine borrows for one run.
//
// Contract kept from the reference (/root/reference/src/HostData.h:20-64): five members with
// these exact names, in this order, because callers brace-initialise a HostSpans positionally
// (examples/basic_sort/basic_sort.cpp:55-61):
//     m_hKeys, m_hHistograms, m_hGlobsum, h_Permut, m_hResultFromGPU
// keys/result carry the key type, the three auxiliary arrays are uint32.  Everything else in
// this header (helpers, the run-time sized harness storage) is this repo's own.
#pragma once

#include "Dataset.h"
#include "Parameters.h"

#include <algorithm>
#include <cstdint>
#include <memory>
#include <numeric>
#include <span>
#include <vector>

template <typename KeyArray, typename WordArray>
struct HostBuffers {
    KeyArray m_hKeys;             // keys to sort; the engine uploads Resize(n) of them
    WordArray m_hHistograms;      // diagnostic: counter table of the last pass, _RADIX * _NUM_ITEMS words
    WordArray m_hGlobsum;         // diagnostic: scanned block sums of the last pass, _NUM_HISTOSPLIT words
    WordArray h_Permut;           // one uint32 per key: identity on upload; the argsort when payload mode is on
    KeyArray m_hResultFromGPU;    // sorted keys, Resize(n) of them
};

template <typename T>
using HostData = HostBuffers<std::vector<T>, std::vector<std::uint32_t>>;   // owning

template <typename T>
using HostSpans = HostBuffers<std::span<T>, std::span<std::uint32_t>>;      // borrowed views

/// Views over an owning bundle, member by member (what CRadixSortTask::InitResources assembles by
/// hand in the reference, src/CRadixSortTask.cpp:88-95 — there with every length set to the key count).
template <typename T>
HostSpans<T> MakeHostSpans(HostData<T>& owned)
{
    return HostSpans<T>{
        std::span<T>(owned.m_hKeys),
        std::span<std::uint32_t>(owned.m_hHistograms),
        std::span<std::uint32_t>(owned.m_hGlobsum),
        std::span<std::uint32_t>(owned.h_Permut),
        std::span<T>(owned.m_hResultFromGPU),
    };
}

/// Harness storage for one (type, dataset) task: the bundle above plus the outputs of the two
/// CPU referees.  The reference fixes every array at the compile-time 2^25 cap
/// (src/HostData.cpp:10-18); here the length follows the data: max(dataset, requested capacity),
/// rounded up to the 1024-key granule so the rounded tail exists and reads as zero.
template <typename T>
struct HostDataWithReference {
    using DataType = T;
    using Parameters = AlgorithmParameters<T>;
    using ResultBuffer = std::vector<T>;

    HostDataWithReference() = delete;
    explicit HostDataWithReference(std::shared_ptr<Dataset<T>> dataset, std::size_t capacity = 0)
    {
        const std::size_t granule = Parameters::_NUM_ITEMS;
        const std::size_t wanted = std::max(capacity, dataset->dataset.size());
        const std::size_t len = (wanted + granule - 1) / granule * granule;

        auto& b = mHostBuffers;
        b.m_hKeys.assign(len, T{0});
        std::copy(dataset->dataset.begin(), dataset->dataset.end(), b.m_hKeys.begin());
        b.h_Permut.resize(len);
        std::iota(b.h_Permut.begin(), b.h_Permut.end(), 0U);
        b.m_hHistograms.assign(static_cast<std::size_t>(Parameters::_RADIX) * Parameters::_NUM_ITEMS, 0U);
        b.m_hGlobsum.assign(Parameters::_NUM_HISTOSPLIT, 0U);
        m_resultSTLCPU.resize(len);
        m_resultRadixSortCPU.resize(len);
    }

    ResultBuffer m_resultSTLCPU;          // std::sort referee
    ResultBuffer m_resultRadixSortCPU;    // RadixSortCPU referee
    HostData<T> mHostBuffers;
};
