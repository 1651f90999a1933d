// CRadixSortTask.h — one (key type, dataset) experiment behind the IComputeTask interface.
//
// Mirrors the reference's orchestrator (/root/reference/src/CRadixSortTask.h:22-92,
// src/CRadixSortTask.cpp) in what the harness can observe:
//   InitResources   sizes the host bundle, builds the HostSpans, initialises RadixSortGPU    (cpp:75-110)
//   ComputeCPU      times the two referees, std::sort and RadixSortCPU, on the ROUNDED array (cpp:173-222)
//   ComputeGPU      optional pad, one validated run, then the timed repetitions + report      (cpp:120-170,357-437)
//   ValidateResults memcmp of both sorts against std::sort over the requested length          (cpp:225-252)
//   ReleaseResources
// and differs where the reference is loose: failures of upload/calculate/download are remembered
// instead of asserted away, the CPU radix referee is judged only inside its correct domain, and the
// GPU result is additionally compared with that referee directly and (optionally) as an argsort.
#pragma once

#include "Common/IComputeTask.h"
#include "HostData.h"
#include "Parameters.h"
#include "RadixSortGPU.h"
#include "RadixSortMultiGPU.h"
#include "RadixSortOptions.h"
#include "Statistics.h"

#include <cstddef>
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <string_view>

/// Timings of the CPU referees (src/CRadixSortTask.h:14-17); only `.avg` is filled, as there.
struct RuntimesCPU {
    Statistics timeRadix{};
    Statistics timeSTL{};
};

/// One CSV record: header line + value line.  Columns 1-10 are the reference's
/// (src/CRadixSortTask.cpp:327-352): NumElements, Datatype, Dataset, avgHistogram, avgScan, avgPaste,
/// avgReorder, avgTotalGPU, avgTotalSTLCPU, avgTotalRDXCPU; appended: MkeysPerSec (end to end),
/// scatterGBs, scatterPctOfPeak, nGPU.
void writePerformance(std::ostream& stream, const RuntimesGPU& runtimesGPU, const RuntimesCPU& runtimesCPU, std::size_t numberKeys,
                      const std::string& datasetName, std::string_view datatype, std::size_t keyBytes, double avgTotalGPU_ms, int numGPUs = 1);

template <typename T>
class CRadixSortTask : public IComputeTask {
public:
    using DataType = T;

    CRadixSortTask(const RadixSortOptions& options, std::shared_ptr<Dataset<DataType>> dataset);
    ~CRadixSortTask() override = default;

    // -- IComputeTask, in the order CTestBase::RunComputeTask calls them ------------------------
    bool InitResources(hipc::Device Device, hipc::Context Context) override;
    void ComputeCPU() override;
    void ComputeGPU(hipc::Context Context, hipc::CommandQueue CommandQueue, const LocalWorkSize& LocalWorkSize) override;
    bool ValidateResults() override;
    void ReleaseResources() override;

    // -- read-only views for callers that want the numbers without parsing stdout ----------------
    const RuntimesCPU& runtimesCPU() const { return mRuntimesCPU; }
    RuntimesGPU runtimesGPU() const { return mMulti ? mMulti->getRuntimes() : mRadixSortGPU.getRuntimes(); }
    double averageTotalGPUms() const { return mAvgTotalGPUms; }      ///< upload + sort + download, wall clock

protected:
    using Parameters = AlgorithmParameters<DataType>;

    std::uint32_t Resize(std::uint32_t nn);                           ///< remembers nn, returns the 1024-rounded length
    void ExecuteTask(hipc::Context Context, hipc::CommandQueue CommandQueue, const LocalWorkSize& LocalWorkSize);
    void TestPerformance(hipc::CommandQueue CommandQueue, std::size_t numIterations);

    RadixSortOptions mOptions;
    std::shared_ptr<Dataset<DataType>> m_selectedDataset;
    HostDataWithReference<DataType> mHostData;    ///< keys, diagnostics, permutation, GPU result + both referees' outputs
    RadixSortGPU<DataType> mRadixSortGPU;
    std::unique_ptr<RadixSortMultiGPU<DataType>> mMulti;      ///< --gpus N / --ranks R / --sharded: the sharded engine takes the place of mRadixSortGPU
    RuntimesCPU mRuntimesCPU{};
    std::uint32_t mNumberKeys{0U};                ///< what the caller asked for
    std::uint32_t mNumberKeysRounded{0U};         ///< what is uploaded, sorted and downloaded
    double mAvgTotalGPUms{0.0};
    bool mExecutionFailed{false};
    std::vector<DataType> mSecondResult{};        ///< --overlap: odd submissions download here
    std::vector<std::uint32_t> mFirstPermOut{}, mSecondPermOut{};
};
