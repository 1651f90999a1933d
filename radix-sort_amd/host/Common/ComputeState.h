// ComputeState.h — the HIP stand-in for the reference's OpenCL plumbing
// (Common/ComputeState.{h,cpp}: platform/device discovery, context, one in-order queue).
// A `Device`/`Context` is a HIP device ordinal, a `CommandQueue` names a HIP stream
// (nullptr = the sort engine's own stream).  They exist so that IComputeTask,
// CRadixSortTask and RadixSortGPU keep the reference's signatures; the device work
// itself goes through the C ABI in include/radixsort_hip.h.
#pragma once

#include "radixsort_hip.h"

#include <iostream>

namespace hipc {

struct Device {
    int ordinal{0};
};

struct Context {
    int ordinal{-1};
    bool valid() const { return ordinal >= 0; }
};

struct CommandQueue {
    void* stream{nullptr};   // hipStream_t, or nullptr for the engine-owned stream
};

}  // namespace hipc

class ComputeState {
public:
    /// Finds a GPU (first device, like the reference's `devices.front()`,
    /// Common/ComputeState.cpp:62), prints what it found.  false = no HIP device.
    bool init(int ordinal = 0)
    {
        int count = 0;
        if (rsx_device_count(&count) != RSX_OK || count <= ordinal) {
            std::cerr << "No suitable HIP GPU device found (" << rsx_last_error() << ")\n";
            return false;
        }
        char name[256];
        if (rsx_device_name(ordinal, name, sizeof name) == RSX_OK) {
            std::cout << "Using HIP device " << ordinal << ": " << name << "\n";
        }
        m_Device.ordinal = ordinal;
        m_CLContext.ordinal = ordinal;
        return true;
    }

    hipc::Device device() const { return m_Device; }

    hipc::Device m_Device{};
    hipc::Context m_CLContext{};          // reference member name kept (tests/CTestBase.cpp:22-31)
    hipc::CommandQueue m_CLCommandQueue{};
};
