// RcclComm.hip — libradixsort_rccl.so: the RCCL side of ShardComm.h as five C entry points (compiled by hipcc because <rccl/rccl.h> pulls in
// the HIP runtime headers; the rest of radix-sort_amd/host is plain g++ over C ABIs).  One process, one communicator per GPU
// (ncclCommInitAll), every rank thread drives its own; collectives run on the caller's stream.  Nothing in the reference corresponds
// (single device, /root/reference/Common/ComputeState.cpp:88-101); this is SURVEY §8(e)'s "histogram all-to-all" and "bucket exchange"
// for hosts that are not Python.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

struct Comm {
    ncclComm_t comm{};
    int device{0};
    int rank{0};
    int world{0};
    int* fence_word{nullptr};      // device: the one word the fence all-reduces
};

int fail(const char* what, const char* detail)
{
    g_error = std::string(what) + ": " + detail;
    std::fprintf(stderr, "radixsort_rccl: %s\n", g_error.c_str());
    return 4;      // RSX_CALCULATION_FAILED
}

#define NCCL_TRY(expr)                                                   \
    do {                                                                 \
        const ncclResult_t r_ = (expr);                                  \
        if (r_ != ncclSuccess) return fail(#expr, ncclGetErrorString(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                   \
    do {                                                                \
        const hipError_t r_ = (expr);                                   \
        if (r_ != hipSuccess) return fail(#expr, hipGetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

const char* rsxc_rccl_last_error(void)
{
    return g_error.c_str();
}

int rsxc_rccl_create(int ndev, const int* devices, void** comms_out)
{
    if (ndev < 1 || !devices || !comms_out) return fail("rsxc_rccl_create", "bad arguments");
    std::vector<ncclComm_t> comms(static_cast<std::size_t>(ndev));
    NCCL_TRY(ncclCommInitAll(comms.data(), ndev, devices));
    for (int r = 0; r < ndev; ++r) {
        Comm* c = new Comm;
        c->comm = comms[static_cast<std::size_t>(r)];
        c->device = devices[r];
        c->rank = r;
        c->world = ndev;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->fence_word), sizeof(int)));
        HIP_TRY(hipMemset(c->fence_word, 0, sizeof(int)));
        comms_out[r] = c;
    }
    return 0;
}

int rsxc_rccl_destroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->fence_word);
    (void)ncclCommDestroy(c->comm);
    delete c;
    return 0;
}

int rsxc_rccl_all_gather(void* comm, const void* d_send, void* d_recv, std::size_t bytes, void* hip_stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return fail("rsxc_rccl_all_gather", "null communicator");
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(ncclAllGather(d_send, d_recv, bytes, ncclInt8, c->comm, static_cast<hipStream_t>(hip_stream)));
    return 0;
}

// RCCL has no all-to-all-v: one group of point-to-point calls, every pair on its own xGMI link at once
int rsxc_rccl_all_to_all_v(void* comm, int world, const void* d_send, const std::uint64_t* sendOff, const std::uint64_t* sendCnt, void* d_recv,
                           const std::uint64_t* recvOff, const std::uint64_t* recvCnt, std::size_t elemBytes, void* hip_stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c || world != c->world) return fail("rsxc_rccl_all_to_all_v", "null communicator or wrong world size");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    NCCL_TRY(ncclGroupStart());
    ncclResult_t first = ncclSuccess;            // the group is closed whatever happens inside it
    for (int p = 0; p < world && first == ncclSuccess; ++p) {
        if (sendCnt[p]) first = ncclSend(static_cast<const char*>(d_send) + sendOff[p] * elemBytes, sendCnt[p] * elemBytes, ncclInt8, p, c->comm, s);
        if (first == ncclSuccess && recvCnt[p]) first = ncclRecv(static_cast<char*>(d_recv) + recvOff[p] * elemBytes, recvCnt[p] * elemBytes, ncclInt8, p, c->comm, s);
    }
    const ncclResult_t closed = ncclGroupEnd();
    if (first != ncclSuccess) return fail("ncclSend / ncclRecv", ncclGetErrorString(first));
    if (closed != ncclSuccess) return fail("ncclGroupEnd", ncclGetErrorString(closed));
    return 0;
}

int rsxc_rccl_fence(void* comm, void* hip_stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return fail("rsxc_rccl_fence", "null communicator");
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(ncclAllReduce(c->fence_word, c->fence_word, 1, ncclInt32, ncclSum, c->comm, static_cast<hipStream_t>(hip_stream)));
    return 0;
}

}  // extern "C"
