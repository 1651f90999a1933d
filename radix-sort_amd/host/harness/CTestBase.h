// CTestBase.h — the run sequence of one task, after the reference's
// tests/CTestBase.{h,cpp}: InitResources -> ComputeCPU -> ComputeGPU -> ValidateResults ->
// ReleaseResources.  Unlike the reference (tests/CTestBase.cpp:56-66, which returns true
// even after printing "INVALID RESULTS!"), a failed validation fails the run.
#pragma once

#include "Common/CommonDefs.h"
#include "Common/ComputeState.h"
#include "Common/IComputeTask.h"

#include <iostream>
#include <string>
#include <vector>

class CTestBase {
public:
    explicit CTestBase(std::vector<std::string> arguments = {}) : m_arguments(std::move(arguments)) {}
    virtual ~CTestBase() = default;

    virtual bool DoCompute() = 0;

    virtual bool InitCLContext() { return m_computeState.init(); }   // name kept from the reference

    virtual bool RunComputeTask(IComputeTask& Task, const LocalWorkSize& LocalWorkSize)
    {
        if (!m_computeState.m_CLContext.valid()) {
            std::cerr << "Error: RunComputeTask() cannot execute because the device context is null.\n";
            return false;
        }
        if (!Task.InitResources(m_computeState.device(), m_computeState.m_CLContext)) {
            std::cerr << "Error during resource allocation. Aborting execution." << std::endl;
            Task.ReleaseResources();
            return false;
        }
        std::cout << "Computing CPU reference result...";
        Task.ComputeCPU();
        std::cout << "DONE" << std::endl;
        std::cout << "Computing GPU result..." << std::endl;
        Task.ComputeGPU(m_computeState.m_CLContext, m_computeState.m_CLCommandQueue, LocalWorkSize);
        std::cout << "DONE" << std::endl;
        const bool valid = Task.ValidateResults();
        std::cout << (valid ? "GOLD TEST PASSED!\n" : "INVALID RESULTS!\n");
        Task.ReleaseResources();
        return valid;
    }

protected:
    ComputeState m_computeState;
    std::vector<std::string> m_arguments;
};
