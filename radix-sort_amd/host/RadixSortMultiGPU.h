// RadixSortMultiGPU.h — the sharded sort behind the reference's engine interface: method for method RadixSortGPU<T>
// (/root/reference/src/RadixSortGPU.h:41-87: initialize / uploadData / calculate / downloadData / release / Resize / getRuntimes),
// over N ranks instead of one device.  The reference has nothing of the kind (one in-order queue on one device,
// /root/reference/Common/ComputeState.cpp:88-101; its harness drives exactly one engine, src/CRadixSortTask.cpp:289-314); this is
// SURVEY §8(e) for C++ hosts: ONE process, one host thread + one sort engine + one communication stream per rank (SURVEY §8b
// "threading"), the same C ABI (include/radixsort_hip.h: rsx_msd_* / rsx_sort_from_to / rsx_partition_*_split) and the same
// planner (ShardPlanner.h) the Python driver uses (radix-sort_amd/distributed.py).
//
// One step (`calculate`), per rank:
//   count    rsx_msd_count: bucket sizes of the top B bits into a device row (with this rank's capacities and status word)
//   scatter  rsx_msd_scatter into wave-major staging, beside the exchange of the rows
//   exchange AllToAll:   rows to the host (communication stream, behind the count only) and across the rank threads through shared
//                        memory; wave_layout; one grouped send/recv per wave on the communication stream
//            PeerStores: device all_gather of the rows, rsx_msd_plan on the device, one rsx_msd_push + fence per wave into the other
//                        ranks' receive buffers (same process: their pointers, after rsx_peer_enable for another device)
//   sort     the waves in doubling groups {0} {1} {2,3} {4..7}: wave 0 as soon as it has landed (rsx_wait_for on the communication stream), the
//            later, larger groups at the big-sort rate while the next group travels; up to 4 buckets cost no extra pass unit at B = 6
// Inputs that do not balance on their top bits take the splitter path (samples -> quantile splitters -> tie-splitting cut plan,
// one all-to-all, full local sort), up to 8 ranks.  Rank-order concatenation of the ranks' outputs is the sorted array
// (with `withPermutation`: the stable argsort), which downloadData assembles in m_hResultFromGPU.
#pragma once

#include "HostData.h"
#include "OperationStatus.h"
#include "RadixSortGPU.h"
#include "ShardComm.h"
#include "ShardPlanner.h"

#include <array>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ShardedSortOptions {
    enum class Exchange { AllToAll, PeerStores };
    enum class Comm { Auto, Loopback, Rccl };
    std::vector<int> devices{0};        ///< HIP device ordinal of every rank (repeats = several ranks on one GPU: loopback communicator only)
    Exchange exchange{Exchange::AllToAll};
    Comm comm{Comm::Auto};              ///< Auto: RCCL when every rank has a device of its own and there is more than one, else loopback
    int partitionBits{0};               ///< top key bits of the exchange partition = 2^bits / ranks waves per rank; 0: eight waves per rank, at least 4 bits
    int radixBits{4};                   ///< digit width of the local sorts (RSX_OPT_RADIX_BITS)
    bool withPermutation{false};        ///< carry h_Permut as a uint32 payload: the result is the stable argsort
    bool forceExchange{false};          ///< one rank: still partition and exchange (the communicator talks to itself)
    double maxImbalance{1.25};          ///< the fixed bucket ownership is used while no rank gets more than this x its share
    int pushParts{0};                   ///< workgroups per destination of a wave's push (0: the library's default)
    bool doublingGroups{true};          ///< the local sorts take the waves in groups {0} {1} {2,3} {4..7} ... (ShardPlanner.h) instead of one by one
};

template <typename DataType>
class RadixSortMultiGPU {
public:
    static constexpr int kRowSlots = 256;                ///< bucket slots of a count row (rsx_msd_count writes all of them)
    static constexpr int kRowLen = kRowSlots + 3;        ///< + receive capacity, output capacity, status word
    using Row = std::array<std::uint64_t, kRowLen>;

    RadixSortMultiGPU() = default;
    ~RadixSortMultiGPU();
    RadixSortMultiGPU(const RadixSortMultiGPU&) = delete;
    RadixSortMultiGPU& operator=(const RadixSortMultiGPU&) = delete;

    OperationStatus initialize(const ShardedSortOptions& options, std::uint64_t nn, const HostSpans<DataType>& hostSpans);
    OperationStatus uploadData();       ///< contiguous shards of m_hKeys (and h_Permut), rank by rank
    OperationStatus calculate();        ///< one sharded sort of what was uploaded
    OperationStatus downloadData();     ///< rank outputs, concatenated in rank order, into m_hResultFromGPU (and h_Permut)
    OperationStatus release();

    std::uint64_t Resize(std::uint64_t nn) const noexcept;      ///< next multiple of 1024, like RadixSortGPU::Resize (src/RadixSortGPU.cpp:288-297)
    RuntimesGPU getRuntimes() const { return mRuntimes; }        ///< timeTotal: wall clock of calculate(); the kernel phases: rank 0's engine
    int world() const { return mWorld; }
    int partitionBits() const { return mBits; }
    const std::string& lastPath() const { return mLastPath; }    ///< "local" | "waves" | "waves-p2p" | "split" | "equal"
    const std::string& lastError() const { return mLastError; }
    const char* communicator() const;
    std::vector<std::uint64_t> rankLoads() const;                ///< keys every rank ended up with in the last calculate()

private:
    struct Rank {
        int rank{0}, device{0};
        rsx_engine* E{nullptr};          ///< the sort engine (its stream: counts, scatters, local sorts)
        rsx_engine* C{nullptr};          ///< a second, tiny engine whose stream is this rank's communication stream
        void* cstream{nullptr};
        std::unique_ptr<shardcomm::IShardComm> comm;
        std::uint64_t n{0}, first{0}, cap{0}, nOut{0};
        void *keys{nullptr}, *staging{nullptr}, *recv{nullptr}, *out{nullptr};
        std::uint32_t *pay{nullptr}, *spay{nullptr}, *rpay{nullptr}, *opay{nullptr};
        std::uint64_t *d_row{nullptr}, *d_table{nullptr}, *d_peerKeys{nullptr}, *d_peerPays{nullptr};
        Row hostRow{};
        std::array<std::uint64_t, 3> tail{~0ULL, ~0ULL, ~0ULL};
        std::string path, error;
        int rc{0};
    };

    // rank threads: started by initialize(), each runs the jobs the public methods hand out, in lockstep
    void worker(int rank);
    OperationStatus runOnAllRanks(const std::function<int(Rank&)>& job, OperationStatus onFailure);
    void stopWorkers();

    int allocateRank(Rank& r);
    int stepRank(Rank& r);
    int pipelinedAllToAll(Rank& r);
    int pipelinedPeerStores(Rank& r);
    int splitterPath(Rank& r, std::uint64_t status);
    int exchangeWave(Rank& r, int wave, const shardplan::Table& counts, const shardplan::WaveLayout& layout, std::uint64_t& sendAt);
    int sortWaves(Rank& r, std::uint64_t start, std::uint64_t count, std::uint64_t done, int groupWaves);
    int fail(Rank& r, int rc, const std::string& what);

    ShardedSortOptions mOpt{};
    HostSpans<DataType> mHostSpans{};
    std::vector<Rank> mRanks;
    std::shared_ptr<shardcomm::HostHub> mHub;
    std::uint64_t mTotal{0};
    int mWorld{0}, mBits{0}, mGrouping{1};
    bool mCanWave{false}, mUseRccl{false};
    RuntimesGPU mRuntimes{};
    std::string mLastPath, mLastError;

    std::vector<std::thread> mThreads;
    std::mutex mMutex;
    std::condition_variable mWake, mDone;
    const std::function<int(Rank&)>* mJob{nullptr};
    std::uint64_t mJobId{0};
    int mPending{0};
    bool mStop{false};
};
