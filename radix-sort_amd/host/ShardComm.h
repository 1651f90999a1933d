// ShardComm.h — what the rank threads of RadixSortMultiGPU<T> talk through.
//
// The C++ sharded sort is ONE process with one host thread, one engine and one communication stream per rank (SURVEY §8b
// "threading": "one handle per device driven from one host thread (or one thread per device) — handles share nothing"; the
// reference itself is one in-order queue on one device, /root/reference/Common/ComputeState.cpp:88-101).  So everything the
// HOST needs from the other ranks (count rows, samples, verdicts) moves through shared memory — HostHub — and only DEVICE data
// moves through a communicator:
//   RcclComm      ranks on different GPUs: ncclCommInitAll + ncclAllGather / grouped ncclSend+ncclRecv / a one-word
//                 ncclAllReduce as the fence (libradixsort_rccl.so, compiled by hipcc, loaded on demand — a single-GPU harness
//                 never needs librccl)
//   LoopbackComm  rank threads on ANY devices, also several on one GPU (RCCL refuses two ranks on one device): pulls with
//                 device copies, ordered across the ranks' communication streams by events — RCCL's stream semantics, no
//                 device-wide synchronisation (the C++ twin of tests/test_gpu_sharded.py:14-95)
// Every device operation is enqueued on the rank's communication stream (the stream of its `comm engine`), never blocks the
// host beyond the rendezvous of the rank threads, and is ordered against the sort engine's stream by rsx_wait_for.
#pragma once

#include "radixsort_hip.h"

#include <atomic>
#include <barrier>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace shardcomm {

/// Shared by the rank threads of one sharded sort: the rendezvous and the host-side all-gather.  A rank that fails calls
/// abort(): it leaves the barrier for good and every other rank finds `failed()` after its next rendezvous.
class HostHub {
public:
    explicit HostHub(int world) : world_(world), barrier_(world), slots_(static_cast<std::size_t>(world), nullptr) {}
    int world() const { return world_; }
    void barrier() { barrier_.arrive_and_wait(); }
    void abort(const std::string& why)
    {
        if (!failed_.exchange(true)) why_ = why;
        barrier_.arrive_and_drop();
    }
    bool failed() const { return failed_.load(); }
    const std::string& why() const { return why_; }

    /// Everybody's `mine`, in rank order (two rendezvous: publish, then everybody has read).
    template <typename V>
    std::vector<V> allGather(int rank, const V& mine)
    {
        slots_[static_cast<std::size_t>(rank)] = &mine;
        barrier();
        std::vector<V> out;
        out.reserve(static_cast<std::size_t>(world_));
        if (!failed()) {
            for (int r = 0; r < world_; ++r) out.push_back(*static_cast<const V*>(slots_[static_cast<std::size_t>(r)]));
        }
        barrier();
        return out;
    }

private:
    int world_;
    std::barrier<> barrier_;
    std::vector<const void*> slots_;
    std::atomic<bool> failed_{false};
    std::string why_;
};

/// Device-side collectives of one rank.  Offsets and counts are in ELEMENTS of elemBytes bytes; all calls enqueue on the rank's
/// communication stream and return; 0 = RSX_OK.
class IShardComm {
public:
    virtual ~IShardComm() = default;
    virtual const char* name() const = 0;
    /// bytes from d_send of every rank into d_recv[rank * bytes] on every rank.
    virtual int allGather(const void* d_send, void* d_recv, std::size_t bytes) = 0;
    /// This rank sends sendCnt[d] elements from d_send + sendOff[d] to rank d and receives recvCnt[s] elements from rank s at d_recv + recvOff[s].
    virtual int allToAllv(const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv, const std::uint64_t* recvOff,
                          const std::uint64_t* recvCnt, std::size_t elemBytes) = 0;
    /// Stream-ordered barrier: what follows on this rank's communication stream runs after everything every rank enqueued on its own before its fence.
    virtual int fence() = 0;
};

/// Rank threads of one process, any devices (also several ranks on one GPU): every rank PULLS its part with device copies on its own
/// communication stream after waiting (event, no host block) for the source's stream, and a rank's part of a collective is over when
/// everybody has pulled from it, too — so a sender may overwrite its buffer as soon as its stream has passed the collective.
class LoopbackComm final : public IShardComm {
public:
    struct Shared {
        explicit Shared(std::shared_ptr<HostHub> h) : hub(std::move(h)), engine(static_cast<std::size_t>(hub->world()), nullptr), args(static_cast<std::size_t>(hub->world())) {}
        std::shared_ptr<HostHub> hub;          // the rank threads' one rendezvous (a rank that fails leaves it for good: nobody hangs)
        std::vector<rsx_engine*> engine;       // every rank's communication engine (its stream is the rank's communication stream)
        struct Args {
            const void* send;
            const std::uint64_t* off;
            const std::uint64_t* cnt;
        };
        std::vector<Args> args;
    };

    LoopbackComm(std::shared_ptr<Shared> shared, int rank, rsx_engine* commEngine) : s_(std::move(shared)), rank_(rank), mine_(commEngine)
    {
        s_->engine[static_cast<std::size_t>(rank)] = commEngine;
    }
    const char* name() const override { return "loopback (device copies between rank threads of one process)"; }

    int allGather(const void* d_send, void* d_recv, std::size_t bytes) override
    {
        return collective(d_send, nullptr, nullptr, [&](int src, const Shared::Args& a) {
            return rsx_copy_on_device(mine_, static_cast<char*>(d_recv) + static_cast<std::size_t>(src) * bytes, a.send, bytes);
        });
    }

    int allToAllv(const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv, const std::uint64_t* recvOff,
                  const std::uint64_t* recvCnt, std::size_t elemBytes) override
    {
        return collective(d_send, sendOff, sendCnt, [&](int src, const Shared::Args& a) {
            const std::uint64_t n = a.cnt[rank_];
            if (n != recvCnt[src]) return static_cast<int>(RSX_CALCULATION_FAILED);      // the two sides of the plan disagree: a bug, not an input
            return rsx_copy_on_device(mine_, static_cast<char*>(d_recv) + recvOff[src] * elemBytes, static_cast<const char*>(a.send) + a.off[rank_] * elemBytes,
                                      n * elemBytes);
        });
    }

    int fence() override
    {
        return collective(nullptr, nullptr, nullptr, [](int, const Shared::Args&) { return static_cast<int>(RSX_OK); });
    }

private:
    template <typename Pull>
    int collective(const void* send, const std::uint64_t* off, const std::uint64_t* cnt, Pull&& pull)
    {
        const int world = s_->hub->world();
        s_->args[static_cast<std::size_t>(rank_)] = Shared::Args{send, off, cnt};
        s_->hub->barrier();                           // everybody has enqueued what it sends and published where it is
        if (s_->hub->failed()) return RSX_CALCULATION_FAILED;
        int rc = RSX_OK;
        for (int src = 0; src < world && rc == RSX_OK; ++src) {
            rc = rsx_wait_for(mine_, s_->engine[static_cast<std::size_t>(src)]);
            if (rc == RSX_OK) rc = pull(src, s_->args[static_cast<std::size_t>(src)]);
        }
        s_->hub->barrier();                           // everybody has enqueued its pulls
        if (s_->hub->failed()) return RSX_CALCULATION_FAILED;
        for (int src = 0; src < world && rc == RSX_OK; ++src) {
            rc = rsx_wait_for(mine_, s_->engine[static_cast<std::size_t>(src)]);      // ... and this rank's stream has seen them finish
        }
        s_->hub->barrier();                           // the argument slots may be reused
        return rc;
    }

    std::shared_ptr<Shared> s_;
    int rank_;
    rsx_engine* mine_;
};

/// Real RCCL between the GPUs of this process (libradixsort_rccl.so: RcclComm.hip).  create() makes one communicator per device with
/// ncclCommInitAll (call it from ONE thread); each rank thread then uses its own.
class RcclComm final : public IShardComm {
public:
    /// devices: one distinct HIP device ordinal per rank.  streams: every rank's communication stream (hipStream_t).  Throws std::runtime_error
    /// when the library or RCCL is missing or refuses (e.g. two ranks on one device).
    static std::vector<std::unique_ptr<IShardComm>> create(const std::vector<int>& devices, const std::vector<void*>& streams);
    ~RcclComm() override;
    const char* name() const override { return "RCCL (ncclAllGather, grouped ncclSend/ncclRecv, ncclAllReduce fence)"; }
    int allGather(const void* d_send, void* d_recv, std::size_t bytes) override;
    int allToAllv(const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv, const std::uint64_t* recvOff,
                  const std::uint64_t* recvCnt, std::size_t elemBytes) override;
    int fence() override;

private:
    RcclComm() = default;
    void* lib_{nullptr};
    void* comm_{nullptr};
    void* stream_{nullptr};
    int world_{0};
};

}  // namespace shardcomm

// ---- C entry points of libradixsort_rccl.so (RcclComm.hip), resolved with dlsym ------------------------------------------------
extern "C" {
int rsxc_rccl_create(int ndev, const int* devices, void** comms_out);      // ncclCommInitAll; comms_out[ndev]
int rsxc_rccl_destroy(void* comm);
int rsxc_rccl_all_gather(void* comm, const void* d_send, void* d_recv, std::size_t bytes, void* hip_stream);
int rsxc_rccl_all_to_all_v(void* comm, int world, const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv,
                           const std::uint64_t* recvOff, const std::uint64_t* recvCnt, std::size_t elemBytes, void* hip_stream);
int rsxc_rccl_fence(void* comm, void* hip_stream);
const char* rsxc_rccl_last_error(void);
}
