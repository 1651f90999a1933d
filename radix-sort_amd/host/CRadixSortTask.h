// CRadixSortTask.h — the IComputeTask realisation that drives one (key type, dataset)
// experiment: CPU referees, GPU sort, validation, performance report.  Same public
// surface and call order as the reference (/root/reference/src/CRadixSortTask.h:22-92).
#pragma once

#include "Common/IComputeTask.h"
#include "HostData.h"
#include "Parameters.h"
#include "RadixSortGPU.h"
#include "RadixSortOptions.h"
#include "Statistics.h"

#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <string_view>

/// Runtime statistics of the CPU referees (src/CRadixSortTask.h:14-17).
struct RuntimesCPU {
    Statistics timeRadix{};
    Statistics timeSTL{};
};

/// CSV writer with the reference's 10-column schema (src/CRadixSortTask.cpp:318-353) plus
/// appended columns (Mkeys/s, scatter GB/s, % of HBM peak, nGPU) that old tooling ignores.
void writePerformance(std::ostream& stream, const RuntimesGPU& runtimesGPU, const RuntimesCPU& runtimesCPU, std::size_t numberKeys,
                      const std::string& datasetName, std::string_view datatype, std::size_t keyBytes, double avgTotalGPU_ms);

template <typename T>
class CRadixSortTask : public IComputeTask {
public:
    using DataType = T;

    CRadixSortTask(const RadixSortOptions& options, std::shared_ptr<Dataset<DataType>> dataset);
    ~CRadixSortTask() override = default;

    bool InitResources(hipc::Device Device, hipc::Context Context) override;
    void ReleaseResources() override;
    void ComputeGPU(hipc::Context Context, hipc::CommandQueue CommandQueue, const LocalWorkSize& LocalWorkSize) override;
    void ComputeCPU() override;
    bool ValidateResults() override;

    const RuntimesCPU& runtimesCPU() const { return mRuntimesCPU; }
    RuntimesGPU runtimesGPU() const { return mRadixSortGPU.getRuntimes(); }
    double averageTotalGPUms() const { return mAvgTotalGPUms; }

protected:
    using Parameters = AlgorithmParameters<DataType>;

    std::uint32_t Resize(std::uint32_t nn);
    void ExecuteTask(hipc::Context Context, hipc::CommandQueue CommandQueue, const LocalWorkSize& LocalWorkSize);
    void TestPerformance(hipc::CommandQueue CommandQueue, std::size_t numIterations);

    std::uint32_t mNumberKeys{0U};          // requested number of keys
    std::uint32_t mNumberKeysRounded{0U};   // next multiple of 1024
    HostDataWithReference<DataType> mHostData;
    std::shared_ptr<Dataset<DataType>> m_selectedDataset;
    RuntimesCPU mRuntimesCPU{};
    RadixSortGPU<DataType> mRadixSortGPU;
    RadixSortOptions mOptions;
    double mAvgTotalGPUms{0.0};
    bool mExecutionFailed{false};
};
