// RadixSortGPU.h — the sort engine's public face, method for method the reference's
// RadixSortGPU<T> (/root/reference/src/RadixSortGPU.h:35-124):
//   initialize / uploadData / calculate / downloadData / release / setLogStream /
//   Resize / padGPUData / getRuntimes,  private Histogram / ScanHistogram / Reorder /
//   CopyDataToDevice / CopyDataFromDevice.
// cl::Device / cl::Context / cl::CommandQueue became the hipc:: stand-ins; everything
// below this class is the C ABI of include/radixsort_hip.h (one rsx_engine per object).
//
// Two execution modes of calculate():
//   * fused (default): the whole pass loop is enqueued with no host synchronisation;
//     per-launch times come from HIP events and fill the same RuntimesGPU fields.
//   * stepwise (setStepwise(true), or whenever a log stream is set): every launch is
//     followed by a stream sync and timed with the host stopwatch — the reference's own
//     accounting (src/RadixSortGPU.cpp:38-56,89-108,128-147,171-190,242-255).
#pragma once
#include <vector>

#include "Common/ComputeState.h"
#include "HostData.h"
#include "OperationStatus.h"
#include "Statistics.h"

#include <cstdint>
#include <iostream>
#include <string>

/// Runtime statistics of the GPU steps (src/RadixSortGPU.h:18-24).
struct RuntimesGPU {
    Statistics timeHisto{};
    Statistics timeScan{};
    Statistics timeReorder{};
    Statistics timePaste{};
    Statistics timeTotal{};
};

template <typename DataType>
class RadixSortGPU {
public:
    RadixSortGPU() = default;
    ~RadixSortGPU();
    RadixSortGPU(const RadixSortGPU&) = delete;
    RadixSortGPU& operator=(const RadixSortGPU&) = delete;

    OperationStatus initialize(hipc::Device Device, hipc::Context Context, std::uint32_t nn, const HostSpans<DataType>& hostSpans);
    OperationStatus uploadData(hipc::CommandQueue CommandQueue);
    OperationStatus calculate(hipc::CommandQueue CommandQueue);
    OperationStatus downloadData(hipc::CommandQueue CommandQueue);
    OperationStatus release();

    void setLogStream(std::ostream* out) noexcept;
    std::uint32_t Resize(std::uint32_t nn) const noexcept;
    void padGPUData(hipc::CommandQueue CommandQueue, std::size_t paddingOffset);
    RuntimesGPU getRuntimes() const;

    // -- additions ---------------------------------------------------------------
    /// Carry h_Permut through the sort as a uint32 payload (stable argsort).  Off by
    /// default: the reference's kernels accept the permutation buffers but never touch
    /// them (RadixSort.cl:79-80), so h_Permut comes back exactly as uploaded.
    /// Call before initialize().
    void enablePermutation(bool on) noexcept { mWithPermutation = on; }
    void setStepwise(bool on) noexcept { mStepwise = on; }
    /// Digit width of calculate()'s fused pass loop: 4 (the reference's _NUM_BITS_PER_RADIX, src/Parameters.h:25) or 8 (half the
    /// passes; same result).  The stepwise launchers and the diagnostic read-backs stay those of 4-bit passes.  Call before initialize().
    void setRadixBits(int bits) noexcept { mRadixBits = bits; }
    /// Page-lock the key / result (and permutation) spans for the engine's lifetime so that
    /// uploadData / downloadData DMA directly (the CL_MEM_USE_HOST_PTR idea the reference notes
    /// at src/ComputeDeviceData.cpp:26).  Call before initialize().
    void enablePinnedTransfers(bool on) noexcept { mPinHost = on; }
    std::uint32_t numberKeysRounded() const noexcept { return mNumberKeysRounded; }
    /// End to end, overlapped: one asynchronous uploadData -> calculate -> downloadData whose copies run on streams of
    /// their own, so that with two submissions in flight the upload of the next sort and the download of the previous
    /// one hide behind the current one (the reference runs the three strictly in sequence,
    /// src/CRadixSortTask.cpp:289-314).  The result lands in `resultOut` (rounded length; pinned memory or the copies
    /// serialise); waitOverlapped() returns when everything submitted has landed.  Needs enablePinnedTransfers(true).
    OperationStatus submitOverlapped(DataType* resultOut, std::uint32_t* permutationOut = nullptr);
    OperationStatus waitOverlapped();
    /// Page-locks an additional caller buffer for the engine's lifetime (a second result buffer for submitOverlapped).
    OperationStatus pinExtra(void* ptr, std::uint64_t bytes);
    /// Zero copy (the reference's visualizer sorts out of mapped host memory, examples/visualize/visualize.cpp:801-854):
    /// the first pass reads the pinned key span over PCIe, the last pass writes m_hResultFromGPU directly; no
    /// uploadData / downloadData around it.  Needs enablePinnedTransfers(true).
    OperationStatus calculateZeroCopy(hipc::CommandQueue CommandQueue, std::uint32_t* permutationOut = nullptr);   ///< permutationOut: pinned (pinExtra), required with enablePermutation

private:
    using Parameters = AlgorithmParameters<DataType>;

    void Histogram(hipc::CommandQueue CommandQueue, int pass);
    void ScanHistogram(hipc::CommandQueue CommandQueue);
    void Reorder(hipc::CommandQueue CommandQueue, int pass);
    void CopyDataToDevice(hipc::CommandQueue CommandQueue);
    void CopyDataFromDevice(hipc::CommandQueue CommandQueue);
    bool bindQueue(hipc::CommandQueue CommandQueue);
    void foldEventTimings();

    rsx_engine* mEngine{nullptr};            ///< the reference's shared_ptr<ComputeDeviceData>
    HostSpans<DataType> mHostSpans{};
    RuntimesGPU mRuntimesGPU{};
    std::uint32_t mNumberKeysRounded{0U};
    std::ostream* mOutStream{nullptr};
    void* mBoundStream{nullptr};
    bool mWithPermutation{false};
    bool mStepwise{false};
    int mRadixBits{4};
    bool mPinHost{false};
    bool mPinned{false};
    std::vector<void*> mExtraPinned{};
    int mLastStatus{RSX_OK};
};
