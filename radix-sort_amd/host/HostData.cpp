#include "HostData.h"

#include "Dataset.h"

#include <algorithm>
#include <numeric>

template <typename T>
HostDataWithReference<T>::HostDataWithReference(std::shared_ptr<Dataset<T>> dataset, std::size_t capacity)
{
    const std::size_t granule = Parameters::_NUM_ITEMS;
    std::size_t len = std::max(capacity, dataset->dataset.size());
    len = (len + granule - 1) / granule * granule;          // room for the rounded length

    m_resultSTLCPU.resize(len);
    m_resultRadixSortCPU.resize(len);
    mHostBuffers.m_hKeys.assign(len, T{0});                 // tail past the dataset stays 0 (src/HostData.cpp:15)
    mHostBuffers.m_hHistograms.assign(Parameters::_RADIX * Parameters::_NUM_ITEMS, 0U);
    mHostBuffers.m_hGlobsum.assign(Parameters::_NUM_HISTOSPLIT, 0U);
    mHostBuffers.h_Permut.resize(len);
    std::iota(mHostBuffers.h_Permut.begin(), mHostBuffers.h_Permut.end(), 0U);   // identity (src/HostData.cpp:20)
    std::copy(dataset->dataset.begin(), dataset->dataset.end(), mHostBuffers.m_hKeys.begin());
}

template struct HostDataWithReference<std::int32_t>;
template struct HostDataWithReference<std::int64_t>;
template struct HostDataWithReference<std::uint32_t>;
template struct HostDataWithReference<std::uint64_t>;
