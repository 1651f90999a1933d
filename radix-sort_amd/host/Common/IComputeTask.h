// IComputeTask.h — the five-method task interface the harness drives
// (/root/reference/Common/IComputeTask.h:12-35), cl:: handles replaced by the hipc::
// stand-ins of ComputeState.h.  Call order is fixed by CTestBase::RunComputeTask:
// InitResources -> ComputeCPU -> ComputeGPU -> ValidateResults -> ReleaseResources.
#pragma once

#include "CommonDefs.h"
#include "ComputeState.h"

class IComputeTask {
public:
    virtual ~IComputeTask() = default;
    virtual bool InitResources(hipc::Device Device, hipc::Context Context) = 0;
    virtual void ReleaseResources() = 0;
    virtual void ComputeGPU(hipc::Context Context, hipc::CommandQueue CommandQueue, const LocalWorkSize& LocalWorkSize) = 0;
    virtual void ComputeCPU() = 0;
    virtual bool ValidateResults() = 0;
};
