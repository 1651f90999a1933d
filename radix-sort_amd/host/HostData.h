// HostData.h — the host buffer bundle the engine borrows
// (/root/reference/src/HostData.h:20-64).  Field names, order and element types are the
// reference's: callers aggregate-initialise HostSpans in this order
// (examples/basic_sort/basic_sort.cpp:55-61).
#pragma once

#include "Parameters.h"

#include <cstdint>
#include <memory>
#include <span>
#include <vector>

template <typename T>
struct Dataset;

template <typename BufferData, typename BufferAux>
struct HostBuffers {
    BufferData m_hKeys;            ///< input keys (uploaded in full, rounded length)
    BufferAux m_hHistograms;       ///< read-back of the last pass's digit table
    BufferAux m_hGlobsum;          ///< read-back of the last pass's scanned block sums
    BufferAux h_Permut;            ///< uint32 payload / permutation, one per key
    BufferData m_hResultFromGPU;   ///< sorted keys
};

template <typename T>
using HostData = HostBuffers<std::vector<T>, std::vector<std::uint32_t>>;

template <typename T>
using HostSpans = HostBuffers<std::span<T>, std::span<std::uint32_t>>;

/// Harness-side storage: the buffers above plus the two CPU referees' outputs
/// (src/HostData.h:48-64).  The reference sizes everything at the compile-time 2^25 cap
/// (src/HostData.cpp:10-18); here the size is max(dataset length, requested capacity).
template <typename T>
struct HostDataWithReference {
    using DataType = T;
    using Parameters = AlgorithmParameters<DataType>;
    using ResultBuffer = std::vector<DataType>;

    explicit HostDataWithReference(std::shared_ptr<Dataset<DataType>> dataset, std::size_t capacity = 0);
    HostDataWithReference() = delete;

    ResultBuffer m_resultSTLCPU;
    ResultBuffer m_resultRadixSortCPU;
    HostData<DataType> mHostBuffers;
};
