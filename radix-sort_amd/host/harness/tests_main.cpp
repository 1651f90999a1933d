// tests_main.cpp — the reference's one integration test (tests/tests.cpp:90-113) as a
// plain executable: {uint32,int32,uint64,int64} x {Zeros, Range, InvertedRange,
// RandomDistributed, Random}, each through CRadixSortTask.  Exit code 0 only when every
// combination validates (Catch2 is not available offline; `REQUIRE` became the exit code).
//
//   rsx_tests [--num-elements N] [--perf-to-stdout] [--perf-csv-to-stdout] [--perf-to-csv]
//             [--with-permutation] [--stepwise] [--skip-cpu] [-v]
#include "CRadixSortTask.h"
#include "CTestBase.h"
#include "Dataset.h"
#include "RadixSortOptions.h"

#include <array>
#include <exception>
#include <memory>

template <typename DataType>
static std::array<std::shared_ptr<Dataset<DataType>>, 5> DatasetCreator(std::size_t num_elements)
{
    return {
        std::make_shared<Zeros<DataType>>(num_elements),
        std::make_shared<Range<DataType>>(num_elements),
        std::make_shared<InvertedRange<DataType>>(num_elements),
        std::make_shared<RandomDistributed<DataType>>(num_elements),
        std::make_shared<Random<DataType>>(num_elements),
    };
}

class CRunner : public CTestBase {
public:
    using CTestBase::CTestBase;

    template <typename DataType>
    bool runTask(const RadixSortOptions& options, const LocalWorkSize& lws)
    {
        bool success = true;
        for (const auto& dataset : DatasetCreator<DataType>(options.num_elements)) {
            CRadixSortTask<DataType> radixSort(options, dataset);
            const bool ok = RunComputeTask(radixSort, lws);
            if (!ok) std::cerr << "FAILED: " << dataset->name() << std::endl;
            success = success && ok;
            ++m_ran;
            m_failed += ok ? 0 : 1;
        }
        return success;
    }

    bool DoCompute() override
    {
        const RadixSortOptions options(m_arguments);
        if (options.num_elements > AlgorithmParameters<std::uint32_t>::_ENGINE_MAX_ELEMS) {
            std::cerr << "--num-elements beyond the engine's 32-bit slot range" << std::endl;
            return false;
        }
        AlgorithmParameters<std::uint32_t>::MaxInputElems() = options.num_elements;
        const LocalWorkSize lws{1, 1, 1};   // meaningless for the sort (tests/tests.cpp:78-79)
        bool ok = runTask<std::uint32_t>(options, lws);
        ok = runTask<std::int32_t>(options, lws) && ok;
        ok = runTask<std::uint64_t>(options, lws) && ok;
        ok = runTask<std::int64_t>(options, lws) && ok;
        std::cout << "Main test: " << (m_ran - m_failed) << "/" << m_ran << " task runs validated" << std::endl;
        return ok;
    }

private:
    int m_ran{0}, m_failed{0};
};

int main(int argc, char** argv)
{
    std::vector<std::string> args(argv + 1, argv + argc);
    try {
        CRunner runner(args);
        if (!runner.InitCLContext()) {
            std::cerr << "No HIP device: this harness has no CPU fallback for the GPU path." << std::endl;
            return 2;
        }
        return runner.DoCompute() ? 0 : 1;
    } catch (const std::exception& exc) {
        std::cerr << "Unhandled: " << exc.what() << std::endl;
        return 3;
    }
}
