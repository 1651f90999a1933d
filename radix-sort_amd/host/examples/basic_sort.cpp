// basic_sort.cpp — smallest end-user program of the public API, the counterpart of the
// reference's examples/basic_sort/basic_sort.cpp:23-139: caller-owned vectors -> HostSpans
// -> initialize -> (padGPUData) -> uploadData -> calculate -> downloadData -> compare with
// std::sort -> getRuntimes -> release.  Returns non-zero on mismatch.
//
//   basic_sort [num_elements]        (default 2^20, like the reference)
#include "Common/ComputeState.h"
#include "Dataset.h"
#include "Parameters.h"
#include "RadixSortGPU.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <vector>

template <typename DataType>
static bool sortAndVerify(ComputeState& compute, std::uint32_t numElements)
{
    using Parameters = AlgorithmParameters<DataType>;
    RandomDistributed<DataType> dataset(numElements);

    RadixSortGPU<DataType> sorter;
    const std::uint32_t numRounded = sorter.Resize(numElements);

    std::vector<DataType> hKeys(numRounded), hResult(numRounded);
    std::vector<std::uint32_t> hHistograms(Parameters::_RADIX * Parameters::_NUM_ITEMS), hGlobsum(Parameters::_NUM_HISTOSPLIT), hPermut(numRounded);
    std::copy_n(dataset.dataset.begin(), numElements, hKeys.begin());
    std::iota(hPermut.begin(), hPermut.end(), 0U);

    HostSpans<DataType> spans{
        {hKeys.data(), hKeys.size()}, {hHistograms.data(), hHistograms.size()}, {hGlobsum.data(), hGlobsum.size()},
        {hPermut.data(), hPermut.size()}, {hResult.data(), hResult.size()},
    };
    if (sorter.initialize(compute.device(), compute.m_CLContext, numElements, spans) != OperationStatus::OK) {
        std::cerr << "Failed to initialize RadixSortGPU\n";
        return false;
    }
    auto& queue = compute.m_CLCommandQueue;
    if (numRounded != numElements) sorter.padGPUData(queue, sizeof(DataType) * numElements);
    if (sorter.uploadData(queue) != OperationStatus::OK) return std::cerr << "Upload failed\n", false;
    if (sorter.calculate(queue) != OperationStatus::OK) return std::cerr << "GPU sort failed\n", false;
    if (sorter.downloadData(queue) != OperationStatus::OK) return std::cerr << "Download failed\n", false;

    // the sort covers the rounded length, whose tail is the host buffer's zeros: compare like with like
    std::vector<DataType> reference(hKeys);
    std::sort(reference.begin(), reference.end());
    const bool correct = std::equal(reference.begin(), reference.end(), hResult.begin());

    const auto rt = sorter.getRuntimes();
    std::cout << "\n--- Timing (avg ms per launch) ---\n"
              << "  Histogram : " << rt.timeHisto.avg << "\n  Scan      : " << rt.timeScan.avg << "\n  Reorder   : " << rt.timeReorder.avg
              << "\n  Paste     : " << rt.timePaste.avg << "\n  Total     : " << rt.timeTotal.avg << "\n";
    sorter.release();
    return correct;
}

int main(int argc, char** argv)
{
    ComputeState compute;
    if (!compute.init()) return 1;
    const std::uint32_t N = argc > 1 ? static_cast<std::uint32_t>(std::strtoul(argv[1], nullptr, 0)) : (1U << 20U);
    std::cout << "Sorting " << N << " uint32_t values on the GPU...\n";
    const bool ok = sortAndVerify<std::uint32_t>(compute, N);
    std::cout << "\nResult: " << (ok ? "PASSED" : "FAILED") << "\n";
    return ok ? 0 : 1;
}
