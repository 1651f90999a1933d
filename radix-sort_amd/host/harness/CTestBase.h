// CTestBase.h — drives one IComputeTask through its five calls in the order the reference's
// harness uses (tests/CTestBase.cpp:20-67): resources, CPU referees, GPU sort, validation,
// release.  One deliberate difference: the reference reports success even after it has printed
// "INVALID RESULTS!" (tests/CTestBase.cpp:56-66); here a failed validation fails the run.
#pragma once

#include "Common/CommonDefs.h"
#include "Common/ComputeState.h"
#include "Common/IComputeTask.h"

#include <iostream>
#include <string>
#include <utility>
#include <vector>

class CTestBase {
public:
    explicit CTestBase(std::vector<std::string> arguments = {}) : m_arguments(std::move(arguments)) {}
    virtual ~CTestBase() = default;

    virtual bool DoCompute() = 0;

    /// Name kept from the reference; finds the HIP device.
    virtual bool InitCLContext() { return m_computeState.init(); }

    virtual bool RunComputeTask(IComputeTask& Task, const LocalWorkSize& LocalWorkSize)
    {
        const hipc::Context ctx = m_computeState.m_CLContext;
        if (!ctx.valid()) {
            std::cerr << "RunComputeTask: no device context (InitCLContext() not called or failed)\n";
            return false;
        }
        struct Releaser {                      // ReleaseResources() on every exit path
            IComputeTask& task;
            ~Releaser() { task.ReleaseResources(); }
        } releaser{Task};

        if (!Task.InitResources(m_computeState.device(), ctx)) {
            std::cerr << "RunComputeTask: resource allocation failed, task skipped\n";
            return false;
        }
        std::cout << "[1/3] CPU referees (std::sort, RadixSortCPU) ... " << std::flush;
        Task.ComputeCPU();
        std::cout << "done\n[2/3] GPU sort" << std::endl;
        Task.ComputeGPU(ctx, m_computeState.m_CLCommandQueue, LocalWorkSize);
        std::cout << "[3/3] validation" << std::endl;
        const bool valid = Task.ValidateResults();
        std::cout << (valid ? "GOLD TEST PASSED!" : "INVALID RESULTS!") << std::endl;
        return valid;
    }

protected:
    ComputeState m_computeState;
    std::vector<std::string> m_arguments;
};
