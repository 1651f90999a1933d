// basic_sort.cpp — the smallest complete program on the public API; plays the role of the
// reference's examples/basic_sort/basic_sort.cpp:23-139 (2^20 random uint32 keys, compare with
// std::sort, print the per-step timings, non-zero exit code on mismatch).
//
//   basic_sort [num_elements] [--int64] [--argsort] [--pinned] [--ranks R [--gpus G] [--peer-stores]]
//
// The call sequence is the API contract: caller-owned vectors -> HostSpans -> initialize ->
// (padGPUData) -> uploadData -> calculate -> downloadData -> getRuntimes -> release.
// --ranks R: the same sequence on RadixSortMultiGPU<T> — the array sharded over R ranks (rank r on device r % G; several ranks on one GPU
// talk through the loopback communicator, one rank per GPU through RCCL).
#include "Common/ComputeState.h"
#include "Dataset.h"
#include "HostData.h"
#include "Parameters.h"
#include "RadixSortGPU.h"
#include "RadixSortMultiGPU.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <numeric>
#include <string>

namespace {

struct Choices {
    std::uint32_t count = 1U << 20U;
    bool wide = false;       // int64 keys instead of uint32
    bool argsort = false;    // carry h_Permut through the sort
    bool pinned = false;
    int ranks = 0;           // > 0: RadixSortMultiGPU over this many ranks
    int gpus = 1;
    bool peer_stores = false;
};

Choices parse(int argc, char** argv)
{
    Choices c;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--int64") c.wide = true;
        else if (a == "--argsort") c.argsort = true;
        else if (a == "--pinned") c.pinned = true;
        else if (a == "--peer-stores") c.peer_stores = true;
        else if ((a == "--ranks" || a == "--gpus") && i + 1 < argc) (a == "--ranks" ? c.ranks : c.gpus) = std::atoi(argv[++i]);
        else c.count = static_cast<std::uint32_t>(std::strtoul(a.c_str(), nullptr, 0));
    }
    return c;
}

template <typename Key>
bool run(ComputeState& gpu, const Choices& opt)
{
    RadixSortGPU<Key> sorter;
    sorter.enablePermutation(opt.argsort);
    sorter.enablePinnedTransfers(opt.pinned);

    // the engine sorts Resize(count) elements: whatever the host buffer holds past `count`
    // (zeros here) is part of the sorted range, exactly as in the reference
    const std::uint32_t rounded = sorter.Resize(opt.count);
    HostData<Key> host;
    host.m_hKeys.assign(rounded, Key{0});
    host.m_hResultFromGPU.assign(rounded, Key{0});
    host.m_hHistograms.assign(AlgorithmParameters<Key>::_HISTOSIZE, 0U);
    host.m_hGlobsum.assign(AlgorithmParameters<Key>::_NUM_HISTOSPLIT, 0U);
    host.h_Permut.resize(rounded);
    std::iota(host.h_Permut.begin(), host.h_Permut.end(), 0U);
    {
        const RandomDistributed<Key> input(opt.count);
        std::copy(input.dataset.begin(), input.dataset.end(), host.m_hKeys.begin());
    }

    const auto fail = [&](const char* step, OperationStatus s) {
        std::cerr << step << " failed: " << to_string(s) << " (" << rsx_last_error() << ")\n";
        return false;
    };
    auto& queue = gpu.m_CLCommandQueue;
    OperationStatus s = sorter.initialize(gpu.device(), gpu.m_CLContext, opt.count, MakeHostSpans(host));
    if (s != OperationStatus::OK) return fail("initialize", s);
    if (rounded != opt.count) sorter.padGPUData(queue, sizeof(Key) * opt.count);   // overwritten by the upload, as in the reference
    if ((s = sorter.uploadData(queue)) != OperationStatus::OK) return fail("uploadData", s);
    if ((s = sorter.calculate(queue)) != OperationStatus::OK) return fail("calculate", s);
    if ((s = sorter.downloadData(queue)) != OperationStatus::OK) return fail("downloadData", s);

    std::vector<Key> expect(host.m_hKeys);
    std::sort(expect.begin(), expect.end());
    bool ok = expect == host.m_hResultFromGPU;
    if (opt.argsort) {
        for (std::uint32_t i = 0; i < rounded && ok; ++i) {
            const std::uint32_t from = host.h_Permut[i];
            ok = from < rounded && host.m_hKeys[from] == host.m_hResultFromGPU[i] &&
                 (i == 0 || host.m_hResultFromGPU[i - 1] != host.m_hResultFromGPU[i] || host.h_Permut[i - 1] < from);
        }
    }

    const RuntimesGPU t = sorter.getRuntimes();
    std::cout << "per-launch averages [ms]: histogram " << t.timeHisto.avg << "  scan " << t.timeScan.avg << "  paste " << t.timePaste.avg
              << "  reorder " << t.timeReorder.avg << "  (sum " << t.timeTotal.avg << ")\n";
    sorter.release();
    return ok;
}

// The sharded engine behind the same five calls.
template <typename Key>
bool runSharded(const Choices& opt)
{
    RadixSortMultiGPU<Key> sorter;
    const std::uint64_t rounded = sorter.Resize(opt.count);
    HostData<Key> host;
    host.m_hKeys.assign(rounded, Key{0});
    host.m_hResultFromGPU.assign(rounded, Key{0});
    host.h_Permut.resize(rounded);
    std::iota(host.h_Permut.begin(), host.h_Permut.end(), 0U);
    {
        const RandomDistributed<Key> input(opt.count);
        std::copy(input.dataset.begin(), input.dataset.end(), host.m_hKeys.begin());
    }
    ShardedSortOptions so;
    so.devices.clear();
    for (int r = 0; r < opt.ranks; ++r) so.devices.push_back(r % std::max(opt.gpus, 1));
    so.withPermutation = opt.argsort;
    so.exchange = opt.peer_stores ? ShardedSortOptions::Exchange::PeerStores : ShardedSortOptions::Exchange::AllToAll;
    const auto fail = [&](const char* step, OperationStatus s) {
        std::cerr << step << " failed: " << to_string(s) << " (" << sorter.lastError() << ")\n";
        return false;
    };
    OperationStatus s = sorter.initialize(so, opt.count, MakeHostSpans(host));
    if (s != OperationStatus::OK) return fail("initialize", s);
    if ((s = sorter.uploadData()) != OperationStatus::OK) return fail("uploadData", s);
    if ((s = sorter.calculate()) != OperationStatus::OK) return fail("calculate", s);
    if ((s = sorter.downloadData()) != OperationStatus::OK) return fail("downloadData", s);
    std::vector<Key> expect(host.m_hKeys);
    std::sort(expect.begin(), expect.end());
    bool ok = expect == host.m_hResultFromGPU;
    for (std::uint64_t i = 0; opt.argsort && i < rounded && ok; ++i) {
        const std::uint32_t from = host.h_Permut[i];
        ok = from < rounded && host.m_hKeys[from] == host.m_hResultFromGPU[i] &&
             (i == 0 || host.m_hResultFromGPU[i - 1] != host.m_hResultFromGPU[i] || host.h_Permut[i - 1] < from);
    }
    std::cout << sorter.world() << " ranks, path " << sorter.lastPath() << ", communicator: " << sorter.communicator() << "; keys per rank:";
    for (const std::uint64_t load : sorter.rankLoads()) std::cout << ' ' << load;
    std::cout << "\nstep (wall clock of calculate()): " << sorter.getRuntimes().timeTotal.avg << " ms\n";
    sorter.release();
    return ok;
}

}  // namespace

int main(int argc, char** argv)
{
    const Choices opt = parse(argc, argv);
    ComputeState gpu;
    if (!gpu.init()) return 1;
    std::cout << "Sorting " << opt.count << (opt.wide ? " int64_t" : " uint32_t") << " values on the GPU"
              << (opt.argsort ? " (with permutation)" : "") << "...\n";
    const bool ok = opt.ranks > 0 ? (opt.wide ? runSharded<std::int64_t>(opt) : runSharded<std::uint32_t>(opt))
                                  : (opt.wide ? run<std::int64_t>(gpu, opt) : run<std::uint32_t>(gpu, opt));
    std::cout << "Result: " << (ok ? "PASSED" : "FAILED") << "\n";
    return ok ? 0 : 1;
}
