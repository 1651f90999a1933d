#include <algorithm>
#include "RadixSortGPU.h"

#include "Common/CTimer.h"

#include <limits>
#include <type_traits>

namespace {

template <typename T>
constexpr int key_is_signed = std::is_signed_v<T> ? 1 : 0;

void accumulate(Statistics& dst, const rsx_phase_stat& src)
{
    // rsx_timings was read with reset=1: src holds only samples not yet folded in
    dst.merge(static_cast<std::size_t>(src.n), src.sum_ms, src.min_ms, src.max_ms);
}

}  // namespace

template <typename DataType>
RadixSortGPU<DataType>::~RadixSortGPU()
{
    release();
}

template <typename DataType>
std::uint32_t RadixSortGPU<DataType>::Resize(std::uint32_t nn) const noexcept
{
    // next multiple of _NUM_GROUPS * _NUM_ITEMS_PER_GROUP = 1024 (src/RadixSortGPU.cpp:288-297)
    constexpr std::uint32_t granule = Parameters::_NUM_ITEMS;
    const std::uint32_t rest = nn % granule;
    return rest == 0 ? nn : nn + (granule - rest);
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::initialize(hipc::Device Device, hipc::Context Context, std::uint32_t nn,
                                                   const HostSpans<DataType>& hostSpans)
{
    using S = OperationStatus;
    if (!Context.valid() || Context.ordinal != Device.ordinal) return S::INITIALIZATION_FAILED;
    if (nn == 0 || nn > Parameters::_ENGINE_MAX_ELEMS) return S::RESIZE_FAILED;
    release();
    mNumberKeysRounded = Resize(nn);
    mHostSpans = hostSpans;                       // borrowed, caller keeps them alive (src/HostData.h:41-44)
    if (!mHostSpans.m_hKeys.data() || !mHostSpans.m_hResultFromGPU.data()) return S::HOST_BUFFERS_FAILED;
    if (mWithPermutation && !mHostSpans.h_Permut.data()) return S::HOST_BUFFERS_FAILED;

    const int rc = rsx_create(&mEngine, Device.ordinal, static_cast<int>(sizeof(DataType)), key_is_signed<DataType>,
                              mWithPermutation ? 1 : 0, mNumberKeysRounded);
    if (rc != RSX_OK) {
        mEngine = nullptr;
        return static_cast<S>(rc);
    }
    rsx_resize(mEngine, mNumberKeysRounded);
    rsx_set_option(mEngine, RSX_OPT_PROFILE, 1);   // RuntimesGPU is always filled, as in the reference
    rsx_set_option(mEngine, RSX_OPT_REF_DIAGNOSTICS, mRadixBits == 4 ? 1 : 0);   // m_hHistograms / m_hGlobsum in the reference's geometry (4-bit passes only)
    if (mRadixBits != 4 && rsx_set_option(mEngine, RSX_OPT_RADIX_BITS, mRadixBits) != RSX_OK) {
        release();
        return S::INITIALIZATION_FAILED;
    }
    mPinned = false;
    if (mPinHost) {
        const std::uint64_t keyBytes = static_cast<std::uint64_t>(mNumberKeysRounded) * sizeof(DataType);
        bool ok = rsx_pin_host(mEngine, mHostSpans.m_hKeys.data(), keyBytes) == RSX_OK;
        ok = ok && rsx_pin_host(mEngine, mHostSpans.m_hResultFromGPU.data(), keyBytes) == RSX_OK;
        if (ok && mWithPermutation) ok = rsx_pin_host(mEngine, mHostSpans.h_Permut.data(), static_cast<std::uint64_t>(mNumberKeysRounded) * 4U) == RSX_OK;
        if (!ok) {
            release();
            return S::HOST_BUFFERS_FAILED;
        }
        mPinned = true;
    }
    mRuntimesGPU = RuntimesGPU{};
    mBoundStream = nullptr;
    return S::OK;
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::release()
{
    if (!mEngine) return OperationStatus::OK;
    for (void* p : mExtraPinned) rsx_unpin_host(mEngine, p);
    mExtraPinned.clear();
    if (mPinned) {
        rsx_unpin_host(mEngine, mHostSpans.m_hKeys.data());
        rsx_unpin_host(mEngine, mHostSpans.m_hResultFromGPU.data());
        if (mWithPermutation) rsx_unpin_host(mEngine, mHostSpans.h_Permut.data());
        mPinned = false;
    }
    const int rc = rsx_destroy(mEngine);
    mEngine = nullptr;
    return static_cast<OperationStatus>(rc);
}

template <typename DataType>
bool RadixSortGPU<DataType>::bindQueue(hipc::CommandQueue CommandQueue)
{
    if (!mEngine) return false;
    if (CommandQueue.stream && CommandQueue.stream != mBoundStream) {
        if (rsx_set_stream(mEngine, CommandQueue.stream) != RSX_OK) return false;
        mBoundStream = CommandQueue.stream;
    }
    return true;
}

template <typename DataType>
void RadixSortGPU<DataType>::setLogStream(std::ostream* out) noexcept
{
    mOutStream = out;
}

template <typename DataType>
void RadixSortGPU<DataType>::padGPUData(hipc::CommandQueue CommandQueue, std::size_t paddingOffset)
{
    // fills inputKeys from paddingOffset (bytes) with max()-1 (src/RadixSortGPU.cpp:270-285).
    // Every caller of the reference pads BEFORE uploadData, which then overwrites the whole
    // rounded length from the host span — so the effective padding is whatever the host
    // buffer holds past nn.  Same here.
    if (!bindQueue(CommandQueue)) return;
    mLastStatus = rsx_fill_pad(mEngine, paddingOffset);
}

template <typename DataType>
void RadixSortGPU<DataType>::CopyDataToDevice(hipc::CommandQueue)
{
    mLastStatus = rsx_upload(mEngine, mHostSpans.m_hKeys.data(), mWithPermutation ? mHostSpans.h_Permut.data() : nullptr,
                             mNumberKeysRounded);
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::uploadData(hipc::CommandQueue CommandQueue)
{
    if (!bindQueue(CommandQueue)) return OperationStatus::DATA_UPLOAD_FAILED;
    CopyDataToDevice(CommandQueue);
    return mLastStatus == RSX_OK ? OperationStatus::OK : OperationStatus::DATA_UPLOAD_FAILED;
}

template <typename DataType>
void RadixSortGPU<DataType>::CopyDataFromDevice(hipc::CommandQueue)
{
    // keys, permutation, and the two diagnostic read-backs of the last pass: the pasted counter
    // table (_RADIX*_NUM_ITEMS words, [digit][group][item]) and the scanned block sums
    // (_NUM_HISTOSPLIT words) — src/RadixSortGPU.cpp:390-429.  The engine recomputes both in the
    // reference's geometry (RSX_OPT_REF_DIAGNOSTICS); nothing consumes them, they are for parity.
    // With 8-bit digits (setRadixBits(8)) no pass of the sort is the reference's last 4-bit pass, so its two tables do not exist:
    // they are not asked for (the engine refuses to hand out tables the last sort did not produce) and are left zeroed.
    const bool tables = mRadixBits == 4;
    if (!tables) {
        std::fill(mHostSpans.m_hHistograms.begin(), mHostSpans.m_hHistograms.end(), 0u);
        std::fill(mHostSpans.m_hGlobsum.begin(), mHostSpans.m_hGlobsum.end(), 0u);
    }
    mLastStatus = rsx_download(mEngine, mHostSpans.m_hResultFromGPU.data(), mWithPermutation ? mHostSpans.h_Permut.data() : nullptr,
                               mHostSpans.m_hHistograms.data(), (tables && mHostSpans.m_hHistograms.data()) ? Parameters::_RADIX * Parameters::_NUM_ITEMS : 0,
                               mHostSpans.m_hGlobsum.data(), (tables && mHostSpans.m_hGlobsum.data()) ? Parameters::_NUM_HISTOSPLIT : 0);
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::downloadData(hipc::CommandQueue CommandQueue)
{
    if (!bindQueue(CommandQueue)) return OperationStatus::DATA_DOWNLOAD_FAILED;
    CopyDataFromDevice(CommandQueue);
    return mLastStatus == RSX_OK ? OperationStatus::OK : OperationStatus::DATA_DOWNLOAD_FAILED;
}

// ---- stepwise launchers: enqueue, finish, host-time — the reference's accounting -------
template <typename DataType>
void RadixSortGPU<DataType>::Histogram(hipc::CommandQueue, int pass)
{
    CTimer timer;
    timer.Start();
    mLastStatus = rsx_histogram(mEngine, pass);
    if (mLastStatus == RSX_OK) mLastStatus = rsx_sync(mEngine);
    timer.Stop();
    mRuntimesGPU.timeHisto.update(timer.GetElapsedMilliseconds());
}

template <typename DataType>
void RadixSortGPU<DataType>::ScanHistogram(hipc::CommandQueue)
{
    {
        CTimer timer;
        timer.Start();
        mLastStatus = rsx_scan(mEngine);            // scan #1 + scan #2
        if (mLastStatus == RSX_OK) mLastStatus = rsx_sync(mEngine);
        timer.Stop();
        mRuntimesGPU.timeScan.update(timer.GetElapsedMilliseconds());
    }
    if (mLastStatus != RSX_OK) return;
    {
        CTimer timer;
        timer.Start();
        mLastStatus = rsx_paste(mEngine);
        if (mLastStatus == RSX_OK) mLastStatus = rsx_sync(mEngine);
        timer.Stop();
        mRuntimesGPU.timePaste.update(timer.GetElapsedMilliseconds());
    }
}

template <typename DataType>
void RadixSortGPU<DataType>::Reorder(hipc::CommandQueue, int pass)
{
    CTimer timer;
    timer.Start();
    mLastStatus = rsx_reorder(mEngine, pass);       // includes the input/output buffer-name swap
    if (mLastStatus == RSX_OK) mLastStatus = rsx_sync(mEngine);
    timer.Stop();
    mRuntimesGPU.timeReorder.update(timer.GetElapsedMilliseconds());
}

template <typename DataType>
void RadixSortGPU<DataType>::foldEventTimings()
{
    rsx_runtimes t;
    if (rsx_timings(mEngine, &t, /*reset=*/1) != RSX_OK) return;
    accumulate(mRuntimesGPU.timeHisto, t.histogram);
    accumulate(mRuntimesGPU.timeScan, t.scan);
    accumulate(mRuntimesGPU.timePaste, t.paste);
    accumulate(mRuntimesGPU.timeReorder, t.reorder);
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::calculate(hipc::CommandQueue CommandQueue)
{
    if (!bindQueue(CommandQueue)) return OperationStatus::CALCULATION_FAILED;
    mLastStatus = RSX_OK;
    if (mStepwise || mOutStream) {
        rsx_set_option(mEngine, RSX_OPT_PROFILE, 0);
        for (std::uint32_t pass = 0U; pass < Parameters::_NUM_PASSES && mLastStatus == RSX_OK; ++pass) {
            if (mOutStream) *mOutStream << "Pass " << pass << ":\nBuilding histograms" << std::endl;
            Histogram(CommandQueue, static_cast<int>(pass));
            if (mOutStream) *mOutStream << "Scanning histograms" << std::endl;
            if (mLastStatus == RSX_OK) ScanHistogram(CommandQueue);
            if (mOutStream) *mOutStream << "Reordering " << std::endl;
            if (mLastStatus == RSX_OK) Reorder(CommandQueue, static_cast<int>(pass));
            if (mOutStream) *mOutStream << "-------------------" << std::endl;
        }
        rsx_set_option(mEngine, RSX_OPT_PROFILE, 1);
    } else {
        mLastStatus = rsx_sort(mEngine);
        if (mLastStatus == RSX_OK) foldEventTimings();   // synchronises once, after the last pass
    }
    // timeTotal.avg = sum of the per-launch averages, n = histogram sample count
    // (src/RadixSortGPU.cpp:337-343) — kept, although it is not a per-sort time.
    mRuntimesGPU.timeTotal.avg = mRuntimesGPU.timeHisto.avg + mRuntimesGPU.timeScan.avg + mRuntimesGPU.timeReorder.avg + mRuntimesGPU.timePaste.avg;
    mRuntimesGPU.timeTotal.n = mRuntimesGPU.timeHisto.n;
    return mLastStatus == RSX_OK ? OperationStatus::OK : OperationStatus::CALCULATION_FAILED;
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::pinExtra(void* ptr, std::uint64_t bytes)
{
    if (!mEngine || rsx_pin_host(mEngine, ptr, bytes) != RSX_OK) return OperationStatus::HOST_BUFFERS_FAILED;
    mExtraPinned.push_back(ptr);
    return OperationStatus::OK;
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::submitOverlapped(DataType* resultOut, std::uint32_t* permutationOut)
{
    if (!mEngine || !mPinned || !resultOut) return OperationStatus::HOST_BUFFERS_FAILED;
    mLastStatus = rsx_pipeline_submit(mEngine, mHostSpans.m_hKeys.data(), mWithPermutation ? mHostSpans.h_Permut.data() : nullptr, mNumberKeysRounded,
                                      resultOut, mWithPermutation ? permutationOut : nullptr);
    return mLastStatus == RSX_OK ? OperationStatus::OK : static_cast<OperationStatus>(mLastStatus);
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::waitOverlapped()
{
    if (!mEngine) return OperationStatus::DATA_DOWNLOAD_FAILED;
    mLastStatus = rsx_pipeline_wait(mEngine);
    if (mLastStatus == RSX_OK) foldEventTimings();
    mRuntimesGPU.timeTotal.avg = mRuntimesGPU.timeHisto.avg + mRuntimesGPU.timeScan.avg + mRuntimesGPU.timeReorder.avg + mRuntimesGPU.timePaste.avg;
    mRuntimesGPU.timeTotal.n = mRuntimesGPU.timeHisto.n;
    return mLastStatus == RSX_OK ? OperationStatus::OK : OperationStatus::DATA_DOWNLOAD_FAILED;
}

template <typename DataType>
OperationStatus RadixSortGPU<DataType>::calculateZeroCopy(hipc::CommandQueue CommandQueue, std::uint32_t* permutationOut)
{
    if (!bindQueue(CommandQueue) || !mPinned || (mWithPermutation && !permutationOut)) return OperationStatus::HOST_BUFFERS_FAILED;
    void *dIn = nullptr, *dOut = nullptr, *dPin = nullptr, *dPout = nullptr;
    mLastStatus = rsx_host_device_pointer(mEngine, mHostSpans.m_hKeys.data(), &dIn);
    if (mLastStatus == RSX_OK) mLastStatus = rsx_host_device_pointer(mEngine, mHostSpans.m_hResultFromGPU.data(), &dOut);
    if (mLastStatus == RSX_OK && mWithPermutation) mLastStatus = rsx_host_device_pointer(mEngine, mHostSpans.h_Permut.data(), &dPin);
    if (mLastStatus == RSX_OK && mWithPermutation) mLastStatus = rsx_host_device_pointer(mEngine, permutationOut, &dPout);
    if (mLastStatus != RSX_OK) return OperationStatus::HOST_BUFFERS_FAILED;
    mLastStatus = rsx_sort_from_to(mEngine, dIn, static_cast<const std::uint32_t*>(dPin), mNumberKeysRounded, 0, static_cast<int>(Parameters::_NUM_PASSES), dOut,
                                   static_cast<std::uint32_t*>(dPout));
    if (mLastStatus == RSX_OK) mLastStatus = rsx_sync(mEngine);
    if (mLastStatus == RSX_OK) foldEventTimings();
    mRuntimesGPU.timeTotal.avg = mRuntimesGPU.timeHisto.avg + mRuntimesGPU.timeScan.avg + mRuntimesGPU.timeReorder.avg + mRuntimesGPU.timePaste.avg;
    mRuntimesGPU.timeTotal.n = mRuntimesGPU.timeHisto.n;
    return mLastStatus == RSX_OK ? OperationStatus::OK : OperationStatus::CALCULATION_FAILED;
}

template <typename DataType>
RuntimesGPU RadixSortGPU<DataType>::getRuntimes() const
{
    return mRuntimesGPU;
}

// the four key types of the reference (src/RadixSortGPU.cpp:598-601)
template class RadixSortGPU<std::int32_t>;
template class RadixSortGPU<std::int64_t>;
template class RadixSortGPU<std::uint32_t>;
template class RadixSortGPU<std::uint64_t>;
