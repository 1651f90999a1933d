// host_selftest.cpp — CPU-only checks of the host mirror (no GPU needed): constants,
// Resize, option parsing, Statistics, dataset generators, the CPU referee, the host
// buffer bundle, CSV schema, and that the GPU path refuses to start without a device.
// Exit code = number of failed checks.
#include "CRadixSortCPU.h"
#include "CRadixSortTask.h"
#include "Dataset.h"
#include "HostData.h"
#include "OperationStatus.h"
#include "Parameters.h"
#include "RadixSortGPU.h"
#include "RadixSortMultiGPU.h"
#include "ShardComm.h"
#include "ShardPlanner.h"
#include "RadixSortOptions.h"
#include "Statistics.h"

#include <algorithm>
#include <iostream>
#include <sstream>
#include <thread>

static int g_failed = 0, g_checked = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++g_checked;                                                                 \
        if (!(cond)) {                                                               \
            ++g_failed;                                                              \
            std::cerr << "CHECK failed: " #cond " (" << __FILE__ << ":" << __LINE__ << ")\n"; \
        }                                                                            \
    } while (0)

template <typename T>
static void check_type()
{
    using P = AlgorithmParameters<T>;
    CHECK(P::_RADIX == 16 && P::_NUM_BITS_PER_RADIX == 4);
    CHECK(P::_NUM_PASSES == sizeof(T) * 2);
    CHECK(P::_NUM_ITEMS == 1024 && P::_NUM_HISTOSPLIT == 512 && P::_HISTOSIZE == 16384);
    CHECK(P::_NUM_MAX_INPUT_ELEMS == (1U << 25));
    CHECK(RadixSortCPU<T>::NUM_BINS == sizeof(T) * 2);

    RadixSortGPU<T> gpu;
    CHECK(gpu.Resize(0) == 0 && gpu.Resize(1) == 1024 && gpu.Resize(1000) == 1024 && gpu.Resize(1024) == 1024 && gpu.Resize(1025) == 2048);
    CHECK(gpu.Resize(1U << 28) == (1U << 28));

    // datasets (Dataset.h:84-137)
    const std::size_t n = 5000;
    Zeros<T> z(n);
    Range<T> r(n);
    InvertedRange<T> ir(n);
    Random<T> rnd(n);
    RandomDistributed<T> uni(n);
    CHECK(z.dataset.size() == n && std::all_of(z.dataset.begin(), z.dataset.end(), [](T v) { return v == 0; }));
    CHECK(r.dataset.front() == std::numeric_limits<T>::min() && r.dataset[7] == static_cast<T>(std::numeric_limits<T>::min() + 7));
    CHECK(ir.dataset.back() == std::numeric_limits<T>::min() && std::is_sorted(ir.dataset.rbegin(), ir.dataset.rend()));
    CHECK(rnd.dataset[0] == static_cast<T>(2421477274U) && rnd.dataset[7] == static_cast<T>(942266821U));
    CHECK(uni.dataset.front() == std::numeric_limits<T>::max() && uni.dataset.back() == std::numeric_limits<T>::min());
    CHECK(RandomDistributed<T>(n).dataset == uni.dataset && RandomDistributed<T>(n, 7).dataset != uni.dataset);
    CHECK(std::string(z.name()) == "Zeros" && std::string(r.name()) == "Range" && std::string(ir.name()) == "Inverted Range");
    CHECK(std::string(rnd.name()) == "Random Random" && std::string(uni.name()) == "Random Uniform");

    // the CPU referee agrees with std::sort on every dataset family
    for (const std::vector<T>* src : {&z.dataset, &r.dataset, &ir.dataset, &rnd.dataset, &uni.dataset}) {
        std::vector<T> a(*src), b(*src);
        std::span<T> view(a);
        RadixSortCPU<T>::sort(view);
        std::sort(b.begin(), b.end());
        CHECK(a == b);
    }

    // host buffer bundle (src/HostData.cpp:8-28)
    auto ds = std::make_shared<Random<T>>(3000);
    HostDataWithReference<T> hd(ds, 3000);
    CHECK(hd.mHostBuffers.m_hKeys.size() == 3072 && hd.mHostBuffers.h_Permut.size() == 3072);
    CHECK(hd.mHostBuffers.m_hHistograms.size() == 16384 && hd.mHostBuffers.m_hGlobsum.size() == 512);
    CHECK(hd.mHostBuffers.h_Permut[0] == 0 && hd.mHostBuffers.h_Permut[3071] == 3071);
    CHECK(hd.mHostBuffers.m_hKeys[2999] == ds->dataset[2999] && hd.mHostBuffers.m_hKeys[3000] == 0);

    // no device here -> the engine must refuse, not fall back
    int devices = 0;
    if (rsx_device_count(&devices) != RSX_OK || devices == 0) {
        std::vector<T> k(1024), res(1024);
        std::vector<std::uint32_t> h(16384), g(512), p(1024);
        HostSpans<T> spans{{k.data(), k.size()}, {h.data(), h.size()}, {g.data(), g.size()}, {p.data(), p.size()}, {res.data(), res.size()}};
        CHECK(gpu.initialize(hipc::Device{0}, hipc::Context{0}, 1000, spans) == OperationStatus::INITIALIZATION_FAILED);
        CHECK(gpu.uploadData({}) == OperationStatus::DATA_UPLOAD_FAILED);
        CHECK(gpu.calculate({}) == OperationStatus::CALCULATION_FAILED);
        CHECK(gpu.downloadData({}) == OperationStatus::DATA_DOWNLOAD_FAILED);
        CHECK(gpu.release() == OperationStatus::OK);
    }
}

int main()
{
    check_type<std::uint32_t>();
    check_type<std::int32_t>();
    check_type<std::uint64_t>();
    check_type<std::int64_t>();

    // referee reproduces the reference's short-round-count outputs
    {
        std::vector<std::uint32_t> v{8, 1, 0, 7};
        std::span<std::uint32_t> s(v);
        RadixSortCPU<std::uint32_t>::sort(s);
        CHECK((v == std::vector<std::uint32_t>{8, 0, 1, 7}));
        std::vector<std::int32_t> w{5, -3, 100, -100, 0};
        std::span<std::int32_t> t(w);
        RadixSortCPU<std::int32_t>::sort(t);
        CHECK((w == std::vector<std::int32_t>{0, 5, 100, -100, -3}));
    }

    // domain test of the referee: full-range data is inside, zero-padded signed Range is not
    {
        std::vector<std::int32_t> in_dom{std::numeric_limits<std::int32_t>::max(), -5, 7};
        std::vector<std::int32_t> out_dom{std::numeric_limits<std::int32_t>::min(), std::numeric_limits<std::int32_t>::min() + 1, 0};
        std::vector<std::uint32_t> pow_of_base{8, 1, 0, 7};
        CHECK(RadixSortCPU<std::int32_t>::coversAllDigits(std::span<const std::int32_t>(in_dom)));
        CHECK(!RadixSortCPU<std::int32_t>::coversAllDigits(std::span<const std::int32_t>(out_dom)));
        CHECK(!RadixSortCPU<std::uint32_t>::coversAllDigits(std::span<const std::uint32_t>(pow_of_base)));
    }

    // OperationStatus values (src/OperationStatus.h:4-17) == rsx_status
    CHECK(static_cast<int>(OperationStatus::OK) == RSX_OK && static_cast<int>(OperationStatus::RESIZE_FAILED) == RSX_RESIZE_FAILED);
    CHECK(static_cast<int>(OperationStatus::LOADING_SOURCE_FAILED) == 11 && static_cast<int>(OperationStatus::DATA_DOWNLOAD_FAILED) == RSX_DATA_DOWNLOAD_FAILED);

    // options (src/RadixSortOptions.h:26-36)
    {
        RadixSortOptions d;
        CHECK(d.num_elements == (1U << 25) && !d.perf_to_stdout && !d.perf_to_csv && !d.perf_csv_to_stdout && !d.verbose);
        RadixSortOptions o({"--num-elements", "4294966272", "--perf-to-stdout", "--perf-csv-to-stdout", "-v", "--with-permutation"});
        CHECK(o.num_elements == 4294966272ULL && o.perf_to_stdout && o.perf_csv_to_stdout && o.verbose && o.with_permutation && !o.perf_to_csv);
        bool threw = false;
        try {
            RadixSortOptions bad({"--num-elements"});
        } catch (const std::invalid_argument&) {
            threw = true;
        }
        CHECK(threw);
    }

    // Statistics
    {
        Statistics s;
        s.update(3.0);
        CHECK(s.n == 1 && s.min == 3.0 && s.max == 3.0 && s.avg == 3.0);
        s.update(1.0);
        s.update(5.0);
        CHECK(s.n == 3 && s.min == 1.0 && s.max == 5.0 && s.sum == 9.0 && s.avg == 3.0);
    }

    // CSV schema: the reference's ten columns first (Performance/performance.csv:1)
    {
        std::ostringstream os;
        RuntimesGPU g;
        g.timeReorder.avg = 0.4;
        RuntimesCPU c;
        writePerformance(os, g, c, 1U << 28, "Random Random", "uint32_t", 4, 5.0);
        const std::string text = os.str();
        CHECK(text.rfind("NumElements,Datatype,Dataset,avgHistogram,avgScan,avgPaste,avgReorder,avgTotalGPU,avgTotalSTLCPU,avgTotalRDXCPU", 0) == 0);
        CHECK(text.find("268435456,uint32_t,Random Random,") != std::string::npos);
        CHECK(std::count(text.begin(), text.begin() + static_cast<std::ptrdiff_t>(text.find('\n')), ',') == 13);
    }

    // sharded sort, host side: options, the rank threads' rendezvous, the planner the C++ driver calls directly
    {
        RadixSortOptions o({"--gpus", "8", "--exchange", "peer-stores", "--partition-bits", "6", "--comm", "loopback"});
        CHECK(o.useSharded() && o.numRanks() == 8 && o.exchange == "peer-stores" && o.partition_bits == 6 && o.comm == "loopback");
        RadixSortOptions one({"--ranks", "4"});
        CHECK(one.useSharded() && one.gpus == 1 && one.numRanks() == 4 && !RadixSortOptions({"--num-elements", "5"}).useSharded());
        bool threw = false;
        try {
            RadixSortOptions bad({"--comm", "carrier-pigeon"});
        } catch (const std::invalid_argument&) {
            threw = true;
        }
        CHECK(threw);

        const int world = 6;
        auto hub = std::make_shared<shardcomm::HostHub>(world);
        std::vector<std::vector<std::uint64_t>> seen(world);
        std::vector<std::thread> threads;
        for (int r = 0; r < world; ++r) {
            threads.emplace_back([&, r] {
                for (std::uint64_t round = 0; round < 50; ++round) {
                    const std::vector<std::uint64_t> mine{static_cast<std::uint64_t>(r), round};
                    const auto all = hub->allGather(r, mine);
                    for (int s = 0; s < world; ++s) seen[r].push_back(all[s][0] * 1000 + all[s][1]);
                }
            });
        }
        for (auto& t : threads) t.join();
        bool same = true;
        for (int r = 1; r < world; ++r) same = same && seen[r] == seen[0];
        CHECK(same && seen[0].size() == 50u * world && seen[0][world + 2] == 2 * 1000 + 1);
        // a rank that fails leaves the rendezvous for good: the others find out and nobody hangs
        auto hub2 = std::make_shared<shardcomm::HostHub>(3);
        std::vector<int> outcome(3, -1);
        std::vector<std::thread> t2;
        for (int r = 0; r < 3; ++r) {
            t2.emplace_back([&, r] {
                if (r == 1) {
                    hub2->abort("rank 1 gave up");
                    outcome[r] = 1;
                    return;
                }
                const int v = r;
                const auto all = hub2->allGather(r, v);
                outcome[r] = hub2->failed() && all.empty() ? 2 : 3;
                if (hub2->failed()) hub2->abort("following rank 1");
            });
        }
        for (auto& t : t2) t.join();
        CHECK(outcome == (std::vector<int>{2, 1, 2}) && hub2->why() == "rank 1 gave up");

        const shardplan::Table table{{5, 0, 7, 1}, {2, 2, 2, 2}};
        const shardplan::WaveLayout l = shardplan::wave_layout(table, 2, 4, 4);
        CHECK(l.start[0] == (std::vector<std::uint64_t>{0, 8}) && l.offset[0][0] == (std::vector<std::uint64_t>{0, 5}) && l.load == (std::vector<std::uint64_t>{9, 12}));
        CHECK(l.start[1] == (std::vector<std::uint64_t>{0, 12}) && l.extent == (std::vector<std::uint64_t>{10, 15}));
        CHECK(shardplan::check_capacity_extent(l.extent, l.load, {10, 15}, {9, 12}) == -1 && shardplan::check_capacity_extent(l.extent, l.load, {10, 14}, {9, 12}) == 1);

        // without a device the sharded engine refuses like the single one
        RadixSortMultiGPU<std::uint32_t> multi;
        CHECK(multi.Resize(1) == 1024 && multi.Resize(1ULL << 32) == (1ULL << 32));
        int devices = 0;
        if (rsx_device_count(&devices) != RSX_OK || devices == 0) {
            std::vector<std::uint32_t> keys(2048, 1U), result(2048), words(2048);
            HostSpans<std::uint32_t> spans{std::span<std::uint32_t>(keys), std::span<std::uint32_t>(words), std::span<std::uint32_t>(words), std::span<std::uint32_t>(words),
                                           std::span<std::uint32_t>(result)};
            ShardedSortOptions so;
            so.devices = {0, 0};
            CHECK(multi.initialize(so, 2048, spans) != OperationStatus::OK && !multi.lastError().empty());
            CHECK(multi.calculate() == OperationStatus::INITIALIZATION_FAILED);
        }
    }

    std::cout << "host_selftest: " << (g_checked - g_failed) << "/" << g_checked << " checks passed" << std::endl;
    return g_failed;
}
