// RadixSortMultiGPU.cpp — see RadixSortMultiGPU.h.
#include "RadixSortMultiGPU.h"

#include "Common/CTimer.h"

#include <dlfcn.h>

#include <algorithm>
#include <numeric>
#include <set>
#include <type_traits>

namespace {

constexpr int kCapsAt = 256;            // row[256] receive capacity, row[257] output capacity, row[258] status word
constexpr std::uint64_t kSamples = 1024;

int defaultPartitionBits(int world)
{
    int lg = 0;
    while ((1 << lg) < world) ++lg;
    return std::max(4, std::min(8, lg + 3));      // eight waves per rank, at least 4 bits (one pass unit saved), at most 8
}

#define RSX_STEP(expr, what)                         \
    do {                                             \
        const int rc_ = (expr);                      \
        if (rc_ != RSX_OK) return fail(r, rc_, what); \
    } while (0)

}  // namespace

// ---- RCCL through libradixsort_rccl.so -----------------------------------------------------------------------------------------
namespace shardcomm {

namespace {
struct RcclApi {
    void* lib{nullptr};
    decltype(&rsxc_rccl_create) create{nullptr};
    decltype(&rsxc_rccl_destroy) destroy{nullptr};
    decltype(&rsxc_rccl_all_gather) all_gather{nullptr};
    decltype(&rsxc_rccl_all_to_all_v) all_to_all_v{nullptr};
    decltype(&rsxc_rccl_fence) fence{nullptr};
    decltype(&rsxc_rccl_last_error) last_error{nullptr};
};

RcclApi& rcclApi()
{
    static RcclApi api = [] {
        RcclApi a;
        // next to libradixsort_host.so (which this code lives in)
        Dl_info info{};
        std::string dir = ".";
        if (dladdr(reinterpret_cast<void*>(&rcclApi), &info) && info.dli_fname) {
            const std::string path(info.dli_fname);
            const auto slash = path.rfind('/');
            if (slash != std::string::npos) dir = path.substr(0, slash);
        }
        a.lib = dlopen((dir + "/libradixsort_rccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) throw std::runtime_error(std::string("libradixsort_rccl.so: ") + dlerror());
        auto sym = [&](const char* name) {
            void* p = dlsym(a.lib, name);
            if (!p) throw std::runtime_error(std::string("libradixsort_rccl.so lacks ") + name);
            return p;
        };
        a.create = reinterpret_cast<decltype(a.create)>(sym("rsxc_rccl_create"));
        a.destroy = reinterpret_cast<decltype(a.destroy)>(sym("rsxc_rccl_destroy"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(sym("rsxc_rccl_all_gather"));
        a.all_to_all_v = reinterpret_cast<decltype(a.all_to_all_v)>(sym("rsxc_rccl_all_to_all_v"));
        a.fence = reinterpret_cast<decltype(a.fence)>(sym("rsxc_rccl_fence"));
        a.last_error = reinterpret_cast<decltype(a.last_error)>(sym("rsxc_rccl_last_error"));
        return a;
    }();
    return api;
}
}  // namespace

std::vector<std::unique_ptr<IShardComm>> RcclComm::create(const std::vector<int>& devices, const std::vector<void*>& streams)
{
    RcclApi& api = rcclApi();
    std::vector<void*> handles(devices.size(), nullptr);
    if (api.create(static_cast<int>(devices.size()), devices.data(), handles.data()) != 0)
        throw std::runtime_error(std::string("ncclCommInitAll failed: ") + api.last_error());
    std::vector<std::unique_ptr<IShardComm>> out;
    for (std::size_t r = 0; r < devices.size(); ++r) {
        std::unique_ptr<RcclComm> c(new RcclComm);
        c->comm_ = handles[r];
        c->stream_ = streams[r];
        c->world_ = static_cast<int>(devices.size());
        out.push_back(std::move(c));
    }
    return out;
}

RcclComm::~RcclComm()
{
    if (comm_) rcclApi().destroy(comm_);
}

int RcclComm::allGather(const void* d_send, void* d_recv, std::size_t bytes)
{
    return rcclApi().all_gather(comm_, d_send, d_recv, bytes, stream_);
}

int RcclComm::allToAllv(const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv, const std::uint64_t* recvOff,
                        const std::uint64_t* recvCnt, std::size_t elemBytes)
{
    return rcclApi().all_to_all_v(comm_, world_, d_send, sendOff, sendCnt, d_recv, recvOff, recvCnt, elemBytes, stream_);
}

int RcclComm::fence()
{
    return rcclApi().fence(comm_, stream_);
}

}  // namespace shardcomm

// ---- the rank threads ----------------------------------------------------------------------------------------------------------
template <typename T>
RadixSortMultiGPU<T>::~RadixSortMultiGPU()
{
    release();
}

template <typename T>
std::uint64_t RadixSortMultiGPU<T>::Resize(std::uint64_t nn) const noexcept
{
    constexpr std::uint64_t granule = AlgorithmParameters<T>::_NUM_ITEMS;
    return (nn + granule - 1) / granule * granule;
}

template <typename T>
const char* RadixSortMultiGPU<T>::communicator() const
{
    return (!mRanks.empty() && mRanks[0].comm) ? mRanks[0].comm->name() : "none";
}

template <typename T>
std::vector<std::uint64_t> RadixSortMultiGPU<T>::rankLoads() const
{
    std::vector<std::uint64_t> out;
    for (const Rank& r : mRanks) out.push_back(r.nOut);
    return out;
}

template <typename T>
void RadixSortMultiGPU<T>::worker(int rank)
{
    std::uint64_t seen = 0;
    for (;;) {
        const std::function<int(Rank&)>* job = nullptr;
        {
            std::unique_lock<std::mutex> lock(mMutex);
            mWake.wait(lock, [&] { return mStop || mJobId != seen; });
            if (mStop) return;
            seen = mJobId;
            job = mJob;
        }
        Rank& r = mRanks[static_cast<std::size_t>(rank)];
        try {
            r.rc = (*job)(r);
        } catch (const std::exception& exc) {
            r.rc = fail(r, RSX_CALCULATION_FAILED, exc.what());
        }
        {
            std::lock_guard<std::mutex> lock(mMutex);
            if (--mPending == 0) mDone.notify_all();
        }
    }
}

template <typename T>
OperationStatus RadixSortMultiGPU<T>::runOnAllRanks(const std::function<int(Rank&)>& job, OperationStatus onFailure)
{
    if (mRanks.empty() || mThreads.empty()) return OperationStatus::INITIALIZATION_FAILED;
    if (mHub->failed()) {
        mLastError = "an earlier step failed (" + mHub->why() + "): release() and initialize() again";
        return onFailure;
    }
    {
        std::unique_lock<std::mutex> lock(mMutex);
        mJob = &job;
        mPending = mWorld;
        ++mJobId;
        mWake.notify_all();
        mDone.wait(lock, [&] { return mPending == 0; });
        mJob = nullptr;
    }
    for (const Rank& r : mRanks) {
        if (r.rc != RSX_OK) {
            mLastError = "rank " + std::to_string(r.rank) + ": " + r.error;
            return static_cast<OperationStatus>(r.rc) == OperationStatus::OK ? onFailure : static_cast<OperationStatus>(r.rc);
        }
    }
    return OperationStatus::OK;
}

template <typename T>
void RadixSortMultiGPU<T>::stopWorkers()
{
    {
        std::lock_guard<std::mutex> lock(mMutex);
        mStop = true;
        mWake.notify_all();
    }
    for (std::thread& t : mThreads) {
        if (t.joinable()) t.join();
    }
    mThreads.clear();
    mStop = false;
}

template <typename T>
int RadixSortMultiGPU<T>::fail(Rank& r, int rc, const std::string& what)
{
    // rc == RSX_OK: a verdict of this driver (another rank failed, buffers too small, ...), no C-ABI call behind it whose text would explain anything
    r.error = rc == RSX_OK ? what : what + " (" + rsx_last_error() + ")";
    r.rc = rc == RSX_OK ? static_cast<int>(RSX_CALCULATION_FAILED) : rc;
    if (mHub) mHub->abort("rank " + std::to_string(r.rank) + ": " + r.error);      // the other ranks find out at their next rendezvous: nobody hangs
    return r.rc;
}

// ---- set-up --------------------------------------------------------------------------------------------------------------------
template <typename T>
OperationStatus RadixSortMultiGPU<T>::initialize(const ShardedSortOptions& options, std::uint64_t nn, const HostSpans<T>& hostSpans)
{
    using S = OperationStatus;
    release();
    mOpt = options;
    mWorld = static_cast<int>(options.devices.size());
    if (mWorld < 1 || mWorld > 16) return S::INITIALIZATION_FAILED;
    if (nn == 0) return S::RESIZE_FAILED;
    mTotal = Resize(nn);
    mHostSpans = hostSpans;
    if (!mHostSpans.m_hKeys.data() || !mHostSpans.m_hResultFromGPU.data() || mHostSpans.m_hKeys.size() < mTotal || mHostSpans.m_hResultFromGPU.size() < mTotal)
        return S::HOST_BUFFERS_FAILED;
    if (mOpt.withPermutation && (!mHostSpans.h_Permut.data() || mHostSpans.h_Permut.size() < mTotal)) return S::HOST_BUFFERS_FAILED;
    mCanWave = (mWorld & (mWorld - 1)) == 0;
    mBits = mOpt.partitionBits > 0 ? mOpt.partitionBits : defaultPartitionBits(mWorld);
    if (mBits < 1 || mBits > 8 || (1 << mBits) < mWorld) {
        mLastError = "partitionBits must be in 1..8 with 2^bits >= the number of ranks";
        return S::INITIALIZATION_FAILED;
    }
    mGrouping = mOpt.doublingGroups ? 1 : 0;
    const std::set<int> distinct(options.devices.begin(), options.devices.end());
    const bool oneEach = static_cast<int>(distinct.size()) == mWorld;
    mUseRccl = options.comm == ShardedSortOptions::Comm::Rccl || (options.comm == ShardedSortOptions::Comm::Auto && oneEach && mWorld > 1);
    if (mUseRccl && !oneEach) {
        mLastError = "RCCL needs a device of its own for every rank: use the loopback communicator for several ranks on one GPU";
        return S::INITIALIZATION_FAILED;
    }

    mHub = std::make_shared<shardcomm::HostHub>(mWorld);
    mRanks = std::vector<Rank>(static_cast<std::size_t>(mWorld));
    // contiguous shards: the first `extra` ranks hold one 16-byte granule more
    const std::uint64_t granule = 16 / sizeof(T) * 4;      // keeps every shard's first key 64-byte aligned in the host array
    const std::uint64_t granules = mTotal / granule, per = granules / static_cast<std::uint64_t>(mWorld), extra = granules % static_cast<std::uint64_t>(mWorld);
    std::uint64_t at = 0;
    const std::uint64_t share = (mTotal + static_cast<std::uint64_t>(mWorld) - 1) / static_cast<std::uint64_t>(mWorld);
    for (int i = 0; i < mWorld; ++i) {
        Rank& r = mRanks[static_cast<std::size_t>(i)];
        r.rank = i;
        r.device = options.devices[static_cast<std::size_t>(i)];
        r.n = (per + (static_cast<std::uint64_t>(i) < extra ? 1 : 0)) * granule;
        if (i == mWorld - 1) r.n = mTotal - at;
        r.first = at;
        at += r.n;
        r.cap = 2 * share + 4 * 256 + 1024;      // what a rank may receive: twice its share (maxImbalance is 1.25) + the waves' alignment gaps
        if (allocateRank(r) != RSX_OK) {
            mLastError = "rank " + std::to_string(i) + ": " + r.error;
            const int rc = r.rc;
            release();
            return static_cast<S>(rc);
        }
    }
    try {
        if (mUseRccl) {
            std::vector<void*> streams;
            for (Rank& r : mRanks) streams.push_back(r.cstream);
            auto comms = shardcomm::RcclComm::create(options.devices, streams);
            for (int i = 0; i < mWorld; ++i) mRanks[static_cast<std::size_t>(i)].comm = std::move(comms[static_cast<std::size_t>(i)]);
        } else {
            auto shared = std::make_shared<shardcomm::LoopbackComm::Shared>(mHub);
            for (Rank& r : mRanks) r.comm = std::make_unique<shardcomm::LoopbackComm>(shared, r.rank, r.C);
        }
    } catch (const std::exception& exc) {
        mLastError = exc.what();
        release();
        return S::INITIALIZATION_FAILED;
    }
    // peer stores: every rank addresses every receive buffer by its pointer (one process); another device needs peer access first
    if (mOpt.exchange == ShardedSortOptions::Exchange::PeerStores) {
        for (Rank& r : mRanks) {
            std::vector<std::uint64_t> keys, pays;
            for (const Rank& o : mRanks) {
                if (o.device != r.device && rsx_peer_enable(r.E, o.device) != RSX_OK) {
                    mLastError = "rank " + std::to_string(r.rank) + " cannot reach device " + std::to_string(o.device) + " (" + rsx_last_error() + ")";
                    release();
                    return S::INITIALIZATION_FAILED;
                }
                keys.push_back(reinterpret_cast<std::uint64_t>(o.recv));
                pays.push_back(reinterpret_cast<std::uint64_t>(o.rpay));
            }
            bool ok = rsx_copy_to_device(r.E, r.d_peerKeys, keys.data(), keys.size() * 8) == RSX_OK;
            ok = ok && rsx_copy_to_device(r.E, r.d_peerPays, pays.data(), pays.size() * 8) == RSX_OK;
            ok = ok && rsx_sync(r.E) == RSX_OK;
            if (!ok) {
                mLastError = std::string("peer address tables: ") + rsx_last_error();
                release();
                return S::INITIALIZATION_FAILED;
            }
        }
    }
    mRuntimes = RuntimesGPU{};
    for (int i = 0; i < mWorld; ++i) mThreads.emplace_back([this, i] { worker(i); });
    return S::OK;
}

template <typename T>
int RadixSortMultiGPU<T>::allocateRank(Rank& r)
{
    const int pay = mOpt.withPermutation ? 1 : 0;
    RSX_STEP(rsx_create(&r.E, r.device, static_cast<int>(sizeof(T)), std::is_signed_v<T> ? 1 : 0, pay, std::max<std::uint64_t>(r.cap, r.n)), "rsx_create (sort engine)");
    RSX_STEP(rsx_create(&r.C, r.device, 4, 0, 0, 1024), "rsx_create (communication stream)");
    RSX_STEP(rsx_get_stream(r.C, &r.cstream), "rsx_get_stream");
    if (mOpt.radixBits != 4) RSX_STEP(rsx_set_option(r.E, RSX_OPT_RADIX_BITS, mOpt.radixBits), "RSX_OPT_RADIX_BITS");
    if (r.rank == 0) RSX_STEP(rsx_set_option(r.E, RSX_OPT_PROFILE, 2), "RSX_OPT_PROFILE");      // rank 0's scatter launches fill RuntimesGPU::timeReorder
    auto alloc = [&](void** p, std::uint64_t bytes) { return rsx_peer_alloc(r.E, std::max<std::uint64_t>(bytes, 16), p, nullptr); };
    const std::uint64_t kb = sizeof(T);
    RSX_STEP(alloc(&r.keys, r.n * kb), "device buffer (shard)");
    RSX_STEP(alloc(&r.staging, r.n * kb), "device buffer (staging)");
    RSX_STEP(alloc(&r.recv, r.cap * kb), "device buffer (receive)");
    RSX_STEP(alloc(&r.out, r.cap * kb), "device buffer (output)");
    if (pay) {
        RSX_STEP(alloc(reinterpret_cast<void**>(&r.pay), r.n * 4), "device buffer (payload)");
        RSX_STEP(alloc(reinterpret_cast<void**>(&r.spay), r.n * 4), "device buffer (payload staging)");
        RSX_STEP(alloc(reinterpret_cast<void**>(&r.rpay), r.cap * 4), "device buffer (payload receive)");
        RSX_STEP(alloc(reinterpret_cast<void**>(&r.opay), r.cap * 4), "device buffer (payload output)");
    }
    RSX_STEP(alloc(reinterpret_cast<void**>(&r.d_row), kRowLen * 8), "device buffer (count row)");
    RSX_STEP(alloc(reinterpret_cast<void**>(&r.d_table), static_cast<std::uint64_t>(mWorld) * kRowLen * 8), "device buffer (count table)");
    RSX_STEP(alloc(reinterpret_cast<void**>(&r.d_peerKeys), 16 * 8), "device buffer (peer addresses)");
    RSX_STEP(alloc(reinterpret_cast<void**>(&r.d_peerPays), 16 * 8), "device buffer (peer addresses)");
    return RSX_OK;
}

template <typename T>
OperationStatus RadixSortMultiGPU<T>::release()
{
    stopWorkers();
    int status = RSX_OK;
    for (Rank& r : mRanks) {
        r.comm.reset();
        if (r.E) {
            (void)rsx_sync(r.E);
            for (void* p : {r.keys, r.staging, r.recv, r.out, static_cast<void*>(r.pay), static_cast<void*>(r.spay), static_cast<void*>(r.rpay), static_cast<void*>(r.opay),
                            static_cast<void*>(r.d_row), static_cast<void*>(r.d_table), static_cast<void*>(r.d_peerKeys), static_cast<void*>(r.d_peerPays)}) {
                if (p && rsx_peer_free(r.E, p) != RSX_OK) status = RSX_CLEANUP_FAILED;
            }
            if (rsx_destroy(r.E) != RSX_OK) status = RSX_CLEANUP_FAILED;
        }
        if (r.C && rsx_destroy(r.C) != RSX_OK) status = RSX_CLEANUP_FAILED;
    }
    mRanks.clear();
    mHub.reset();
    mWorld = 0;
    return static_cast<OperationStatus>(status);
}

// ---- transfers -----------------------------------------------------------------------------------------------------------------
template <typename T>
OperationStatus RadixSortMultiGPU<T>::uploadData()
{
    const std::function<int(Rank&)> job = [this](Rank& r) -> int {
        RSX_STEP(rsx_copy_to_device(r.E, r.keys, mHostSpans.m_hKeys.data() + r.first, r.n * sizeof(T)), "upload of the shard");
        if (mOpt.withPermutation) RSX_STEP(rsx_copy_to_device(r.E, r.pay, mHostSpans.h_Permut.data() + r.first, r.n * 4), "upload of the permutation");
        RSX_STEP(rsx_sync(r.E), "upload");
        return RSX_OK;
    };
    return runOnAllRanks(job, OperationStatus::DATA_UPLOAD_FAILED);
}

template <typename T>
OperationStatus RadixSortMultiGPU<T>::downloadData()
{
    const std::function<int(Rank&)> job = [this](Rank& r) -> int {
        const std::vector<std::uint64_t> loads = mHub->allGather(r.rank, r.nOut);
        if (mHub->failed()) return fail(r, RSX_OK, "another rank failed");
        const std::uint64_t at = std::accumulate(loads.begin(), loads.begin() + r.rank, std::uint64_t{0});
        if (std::accumulate(loads.begin(), loads.end(), std::uint64_t{0}) != mTotal) return fail(r, RSX_DATA_DOWNLOAD_FAILED, "the ranks' outputs do not add up to the input");
        RSX_STEP(rsx_copy_from_device(r.E, mHostSpans.m_hResultFromGPU.data() + at, r.out, r.nOut * sizeof(T)), "download of the rank's output");
        if (mOpt.withPermutation) RSX_STEP(rsx_copy_from_device(r.E, mHostSpans.h_Permut.data() + at, r.opay, r.nOut * 4), "download of the permutation");
        RSX_STEP(rsx_sync(r.E), "download");
        return RSX_OK;
    };
    return runOnAllRanks(job, OperationStatus::DATA_DOWNLOAD_FAILED);
}

template <typename T>
OperationStatus RadixSortMultiGPU<T>::calculate()
{
    const std::function<int(Rank&)> job = [this](Rank& r) -> int {
        const int rc = stepRank(r);
        if (rc != RSX_OK) return rc;
        RSX_STEP(rsx_sync(r.E), "sort step");       // also reports a table scan that timed out in this step
        return RSX_OK;
    };
    CTimer timer;
    timer.Start();
    const OperationStatus status = runOnAllRanks(job, OperationStatus::CALCULATION_FAILED);
    timer.Stop();
    if (status == OperationStatus::OK) {
        const double ms = timer.GetElapsedMilliseconds();
        mRuntimes.timeTotal.merge(1, ms, ms, ms);
        rsx_runtimes rt{};
        if (rsx_timings(mRanks[0].E, &rt, 1) == RSX_OK) {
            mRuntimes.timeReorder.merge(static_cast<std::size_t>(rt.reorder.n), rt.reorder.sum_ms, rt.reorder.min_ms, rt.reorder.max_ms);
        }
        mLastPath = mRanks[0].path;
    }
    return status;
}

// ---- one step of one rank --------------------------------------------------------------------------------------------------------
template <typename T>
int RadixSortMultiGPU<T>::stepRank(Rank& r)
{
    const std::uint64_t status = rsx_check_status(r.E) != RSX_OK ? 1 : 0;      // an error of an EARLIER step: travels with the counts, every rank stops together
    if (mWorld == 1 && !mOpt.forceExchange) {
        if (status) return fail(r, RSX_CALCULATION_FAILED, "the engine reported an error of an earlier step");
        RSX_STEP(rsx_sort_from(r.E, r.keys, r.pay, r.n), "rsx_sort_from");
        RSX_STEP(rsx_copy_result(r.E, r.out, r.opay), "rsx_copy_result");
        r.nOut = r.n;
        r.path = "local";
        return RSX_OK;
    }
    if (mCanWave) {
        const std::array<std::uint64_t, 3> tail{r.cap, r.cap, status};
        if (tail != r.tail) {
            RSX_STEP(rsx_copy_to_device(r.E, r.d_row + kCapsAt, tail.data(), sizeof tail), "count row tail");
            RSX_STEP(rsx_sync(r.E), "count row tail");        // (pageable source on this stack; only when the capacities or the status change)
            r.tail = tail;
        }
        RSX_STEP(rsx_msd_count(r.E, r.keys, r.n, mBits, mWorld, r.d_row), "rsx_msd_count");
        RSX_STEP(rsx_wait_for(r.C, r.E), "rsx_wait_for");
        const int rc = mOpt.exchange == ShardedSortOptions::Exchange::PeerStores ? pipelinedPeerStores(r) : pipelinedAllToAll(r);
        if (rc != -1) return rc;             // -1: the top bits do not balance (every rank found the same): the general path takes over
        return splitterPath(r, 0);           // (the status word has been seen by everybody in the row exchange)
    }
    return splitterPath(r, status);
}

template <typename T>
int RadixSortMultiGPU<T>::sortWaves(Rank& r, std::uint64_t start, std::uint64_t count, std::uint64_t done, int groupWaves)
{
    if (count == 0) return RSX_OK;
    const bool pay = mOpt.withPermutation;
    const int units = shardplan::group_pass_units(static_cast<int>(sizeof(T)) * 8, mBits, groupWaves);      // the group's keys share the top mBits - log2(groupWaves) bits
    return rsx_sort_from_to(r.E, static_cast<const char*>(r.recv) + start * sizeof(T), pay ? r.rpay + start : nullptr, count, 0, units,
                            static_cast<char*>(r.out) + done * sizeof(T), pay ? r.opay + done : nullptr);
}

template <typename T>
int RadixSortMultiGPU<T>::exchangeWave(Rank& r, int wave, const shardplan::Table& counts, const shardplan::WaveLayout& layout, std::uint64_t& sendAt)
{
    const int k = (1 << mBits) / mWorld;
    std::vector<std::uint64_t> sendOff(static_cast<std::size_t>(mWorld)), sendCnt(sendOff.size()), recvOff(sendOff.size()), recvCnt(sendOff.size());
    for (int p = 0; p < mWorld; ++p) {
        const auto pi = static_cast<std::size_t>(p);
        sendOff[pi] = sendAt;                                                   // staging is [wave][destination]: one running offset
        sendCnt[pi] = counts[static_cast<std::size_t>(r.rank)][static_cast<std::size_t>(p * k + wave)];
        sendAt += sendCnt[pi];
        recvOff[pi] = layout.offset[static_cast<std::size_t>(r.rank)][static_cast<std::size_t>(wave)][pi];
        recvCnt[pi] = counts[pi][static_cast<std::size_t>(r.rank * k + wave)];
    }
    int rc = r.comm->allToAllv(r.staging, sendOff.data(), sendCnt.data(), r.recv, recvOff.data(), recvCnt.data(), sizeof(T));
    if (rc == RSX_OK && mOpt.withPermutation) rc = r.comm->allToAllv(r.spay, sendOff.data(), sendCnt.data(), r.rpay, recvOff.data(), recvCnt.data(), 4);
    return rc;
}

template <typename T>
int RadixSortMultiGPU<T>::pipelinedAllToAll(Rank& r)
{
    const int nb = 1 << mBits, k = nb / mWorld;
    // the row to the host on the communication stream: behind the count only, the scatter runs beside it
    RSX_STEP(rsx_copy_from_device(r.C, r.hostRow.data(), r.d_row, sizeof(Row)), "count row to the host");
    RSX_STEP(rsx_msd_scatter(r.E, r.keys, r.pay, r.n, r.staging, r.spay), "rsx_msd_scatter");
    RSX_STEP(rsx_sync(r.C), "count row to the host");
    const std::vector<Row> rows = mHub->allGather(r.rank, r.hostRow);
    if (mHub->failed()) return fail(r, RSX_OK, "another rank failed");
    shardplan::Table counts(static_cast<std::size_t>(mWorld));
    std::vector<std::uint64_t> recvCaps, outCaps;
    bool anyStatus = false;
    for (int s = 0; s < mWorld; ++s) {
        const Row& row = rows[static_cast<std::size_t>(s)];
        counts[static_cast<std::size_t>(s)].assign(row.begin(), row.begin() + nb);
        recvCaps.push_back(row[kCapsAt]);
        outCaps.push_back(row[kCapsAt + 1]);
        anyStatus = anyStatus || row[kCapsAt + 2] != 0;
    }
    if (anyStatus) return fail(r, RSX_CALCULATION_FAILED, "a rank's engine reported an error of an earlier step (a table scan that timed out): every rank stops together");
    const shardplan::WaveLayout layout = shardplan::wave_layout(counts, mWorld, nb, 4, mGrouping);
    const std::uint64_t total = std::accumulate(layout.load.begin(), layout.load.end(), std::uint64_t{0});
    const double imbalance = static_cast<double>(*std::max_element(layout.load.begin(), layout.load.end())) / std::max(1.0, static_cast<double>(total) / mWorld);
    const bool fits = shardplan::check_capacity_extent(layout.extent, layout.load, recvCaps, outCaps) < 0;
    if (!fits || imbalance > mOpt.maxImbalance) return -1;      // same verdict on every rank: it only depends on the gathered table
    // Issue order: the first group's waves, the first group's sort, then EVERY remaining wave (the exchange runs back to back, as early as the links
    // allow, each wave closed by a mark on the communication stream), then the remaining sorts, each behind the mark of its group's last wave.
    const auto groups = shardplan::wave_groups(k, mGrouping);
    std::uint64_t sendAt = 0, done = 0;
    int issued = 0;
    auto issueUpTo = [&](int end) {
        for (; issued < end; ++issued) {
            int rc = exchangeWave(r, issued, counts, layout, sendAt);
            if (rc == RSX_OK) rc = rsx_record_mark(r.C, issued);
            if (rc != RSX_OK) return rc;
        }
        return static_cast<int>(RSX_OK);
    };
    RSX_STEP(rsx_wait_for(r.C, r.E), "rsx_wait_for");                                   // the staging buffer is complete
    RSX_STEP(issueUpTo(groups[0].first + groups[0].second), "exchange of a wave");
    for (std::size_t g = 0; g < groups.size(); ++g) {
        const int first = groups[g].first, waves = groups[g].second;
        RSX_STEP(rsx_wait_mark(r.E, r.C, first + waves - 1), "rsx_wait_mark");          // every wave of this group has landed
        std::uint64_t cnt = 0;
        for (int w = first; w < first + waves; ++w) {
            for (const auto& row : counts) cnt += row[static_cast<std::size_t>(r.rank * k + w)];
        }
        RSX_STEP(sortWaves(r, layout.start[static_cast<std::size_t>(r.rank)][static_cast<std::size_t>(first)], cnt, done, waves), "local sort of a group of waves");
        done += cnt;
        if (g == 0) RSX_STEP(issueUpTo(k), "exchange of a wave");
    }
    r.nOut = done;
    r.path = "waves";
    return RSX_OK;
}

template <typename T>
int RadixSortMultiGPU<T>::pipelinedPeerStores(Rank& r)
{
    const int k = (1 << mBits) / mWorld;
    const bool pay = mOpt.withPermutation;
    // (the all_gather is also the step's opening barrier: nobody pushes into a receive buffer whose owner still sorts out of it)
    RSX_STEP(r.comm->allGather(r.d_row, r.d_table, sizeof(Row)), "all_gather of the count rows");
    RSX_STEP(rsx_msd_scatter(r.E, r.keys, r.pay, r.n, r.staging, r.spay), "rsx_msd_scatter");
    RSX_STEP(rsx_msd_plan(r.E, r.d_table, kRowLen, kCapsAt, r.rank, mGrouping, r.cstream), "rsx_msd_plan");
    const auto groups = shardplan::wave_groups(k, mGrouping);
    std::vector<bool> closesGroup(static_cast<std::size_t>(k), false);
    for (const auto& g : groups) closesGroup[static_cast<std::size_t>(g.first + g.second - 1)] = true;
    auto push = [&](int w) {
        int rc = rsx_msd_push(r.E, w, r.staging, pay ? r.spay : nullptr, r.d_peerKeys, pay ? r.d_peerPays : nullptr, mOpt.pushParts, r.cstream);
        if (rc == RSX_OK && closesGroup[static_cast<std::size_t>(w)]) rc = r.comm->fence();      // one fence per GROUP: every rank's pushes of the group have finished
        return rc;
    };
    int issued = 0;
    auto issueUpTo = [&](int end) {
        for (; issued < end; ++issued) {
            int rc = push(issued);
            if (rc == RSX_OK) rc = rsx_record_mark(r.C, issued);
            if (rc != RSX_OK) return rc;
        }
        return static_cast<int>(RSX_OK);
    };
    RSX_STEP(issueUpTo(groups[0].first + groups[0].second), "push of a wave");
    std::vector<std::uint64_t> start(static_cast<std::size_t>(k)), count(start.size()), loads(static_cast<std::size_t>(mWorld));
    std::uint64_t verdict = 0;
    RSX_STEP(rsx_msd_plan_wait(r.E, start.data(), count.data(), loads.data(), &verdict), "rsx_msd_plan_wait");      // the host's one wait: its own wave sizes
    // every rank computed the same verdict from the same table, and with a non-zero verdict no push wrote anything
    if (verdict >> 32) return fail(r, RSX_CALCULATION_FAILED, "a rank's engine reported an error of an earlier step: every rank stops together");
    const std::uint64_t total = std::accumulate(loads.begin(), loads.end(), std::uint64_t{0});
    const double imbalance = static_cast<double>(*std::max_element(loads.begin(), loads.end())) / std::max(1.0, static_cast<double>(total) / mWorld);
    // the fixed bucket ownership overflows somebody's buffers or leaves the ranks uneven (keys that do not use their top bits): the general path takes over.
    // (The first wave may already have been pushed — into receive buffers the general path overwrites later on the same communication streams.)
    if (verdict || imbalance > mOpt.maxImbalance) return -1;
    std::uint64_t done = 0;
    for (std::size_t g = 0; g < groups.size(); ++g) {
        const int first = groups[g].first, waves = groups[g].second;
        RSX_STEP(rsx_wait_mark(r.E, r.C, first + waves - 1), "rsx_wait_mark");          // every wave of this group has landed here
        std::uint64_t cnt = 0;
        for (int w = first; w < first + waves; ++w) cnt += count[static_cast<std::size_t>(w)];
        RSX_STEP(sortWaves(r, start[static_cast<std::size_t>(first)], cnt, done, waves), "local sort of a group of waves");
        done += cnt;
        if (g == 0) RSX_STEP(issueUpTo(k), "push of a wave");          // every remaining wave now: the pushes run back to back beside the sorts
    }
    r.nOut = done;
    r.path = "waves-p2p";
    return RSX_OK;
}

namespace {
struct SampleSet {
    std::vector<std::uint64_t> values;
    std::uint64_t n;
    std::uint64_t status;
};
}  // namespace

template <typename T>
int RadixSortMultiGPU<T>::splitterPath(Rank& r, std::uint64_t status)
{
    if (mWorld > 8) return fail(r, RSX_CALCULATION_FAILED, "keys that do not balance on their top bits need the splitter path, which serves at most 8 ranks");
    SampleSet mine{{}, r.n, status};
    const std::uint64_t want = std::min(kSamples, r.n);
    if (want) {
        mine.values.resize(static_cast<std::size_t>(want));
        RSX_STEP(rsx_sample_keys(r.E, r.keys, r.n, static_cast<std::uint32_t>(want), mine.values.data()), "rsx_sample_keys");
    }
    const std::vector<SampleSet> all = mHub->allGather(r.rank, mine);
    if (mHub->failed()) return fail(r, RSX_OK, "another rank failed");
    std::vector<std::vector<std::uint64_t>> samples;
    std::vector<std::uint64_t> sizes;
    for (const SampleSet& s : all) {
        if (s.status) return fail(r, RSX_CALCULATION_FAILED, "a rank's engine reported an error of an earlier step: every rank stops together");
        samples.push_back(s.values);
        sizes.push_back(s.n);
    }
    const std::vector<std::uint64_t> splitters = shardplan::choose_splitters(samples, sizes, mWorld);
    if (splitters.empty()) {          // nobody has keys
        RSX_STEP(rsx_sort_from(r.E, r.keys, r.pay, r.n), "rsx_sort_from");
        RSX_STEP(rsx_copy_result(r.E, r.out, r.opay), "rsx_copy_result");
        r.nOut = r.n;
        r.path = "equal";
        return RSX_OK;
    }
    std::vector<std::uint64_t> counts(2 * splitters.size() + 1);
    RSX_STEP(rsx_partition_count_split(r.E, r.keys, r.n, splitters.data(), static_cast<int>(splitters.size()), counts.data()), "rsx_partition_count_split");
    const shardplan::Table table = mHub->allGather(r.rank, counts);
    if (mHub->failed()) return fail(r, RSX_OK, "another rank failed");
    const shardplan::ExchangePlan plan = shardplan::split_plan(table, r.rank, mWorld);
    std::vector<std::uint64_t> caps;
    for (const Rank& o : mRanks) caps.push_back(o.cap);
    if (shardplan::check_capacity(plan.loads, caps, caps, true, 0) >= 0) return fail(r, RSX_RESIZE_FAILED, "a rank's buffers are too small for its share");      // every rank alike
    RSX_STEP(rsx_partition_scatter_split(r.E, r.keys, r.pay, r.n, r.staging, r.spay), "rsx_partition_scatter_split");
    std::vector<std::uint64_t> sendOff(static_cast<std::size_t>(mWorld)), recvOff(sendOff.size());
    std::exclusive_scan(plan.send.begin(), plan.send.end(), sendOff.begin(), std::uint64_t{0});
    std::exclusive_scan(plan.recv.begin(), plan.recv.end(), recvOff.begin(), std::uint64_t{0});
    RSX_STEP(rsx_wait_for(r.C, r.E), "rsx_wait_for");
    RSX_STEP(r.comm->allToAllv(r.staging, sendOff.data(), plan.send.data(), r.recv, recvOff.data(), plan.recv.data(), sizeof(T)), "all-to-all of the keys");
    if (mOpt.withPermutation) RSX_STEP(r.comm->allToAllv(r.spay, sendOff.data(), plan.send.data(), r.rpay, recvOff.data(), plan.recv.data(), 4), "all-to-all of the permutation");
    RSX_STEP(rsx_wait_for(r.E, r.C), "rsx_wait_for");
    const std::uint64_t nrecv = plan.n_recv();
    RSX_STEP(rsx_sort_from(r.E, r.recv, r.rpay, nrecv), "rsx_sort_from");
    RSX_STEP(rsx_copy_result(r.E, r.out, r.opay), "rsx_copy_result");
    r.nOut = nrecv;
    r.path = "split";
    return RSX_OK;
}

template class RadixSortMultiGPU<std::int32_t>;
template class RadixSortMultiGPU<std::int64_t>;
template class RadixSortMultiGPU<std::uint32_t>;
template class RadixSortMultiGPU<std::uint64_t>;
