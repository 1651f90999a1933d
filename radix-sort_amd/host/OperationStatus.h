// OperationStatus.h — return codes of the RadixSortGPU API, value-for-value the
// reference's enum (/root/reference/src/OperationStatus.h:4-17).  The C ABI
// (include/radixsort_hip.h, rsx_status) uses the same integers, so a status crosses the
// boundary with a static_cast.  The last four codes belong to OpenCL run-time
// compilation; the HIP kernels are compiled ahead of time, so they can no longer occur
// but keep their slots.
#pragma once

enum class OperationStatus : int {
    OK = 0,
    HOST_BUFFERS_FAILED = 1,
    INITIALIZATION_FAILED = 2,
    DATA_UPLOAD_FAILED = 3,
    CALCULATION_FAILED = 4,
    DATA_DOWNLOAD_FAILED = 5,
    CLEANUP_FAILED = 6,
    RESIZE_FAILED = 7,
    KERNEL_CREATION_FAILED = 8,
    PROGRAM_CREATION_FAILED = 9,
    NO_SOURCE_FOUND = 10,
    LOADING_SOURCE_FAILED = 11,
};

inline const char* to_string(OperationStatus s)
{
    switch (s) {
    case OperationStatus::OK: return "OK";
    case OperationStatus::HOST_BUFFERS_FAILED: return "HOST_BUFFERS_FAILED";
    case OperationStatus::INITIALIZATION_FAILED: return "INITIALIZATION_FAILED";
    case OperationStatus::DATA_UPLOAD_FAILED: return "DATA_UPLOAD_FAILED";
    case OperationStatus::CALCULATION_FAILED: return "CALCULATION_FAILED";
    case OperationStatus::DATA_DOWNLOAD_FAILED: return "DATA_DOWNLOAD_FAILED";
    case OperationStatus::CLEANUP_FAILED: return "CLEANUP_FAILED";
    case OperationStatus::RESIZE_FAILED: return "RESIZE_FAILED";
    case OperationStatus::KERNEL_CREATION_FAILED: return "KERNEL_CREATION_FAILED";
    case OperationStatus::PROGRAM_CREATION_FAILED: return "PROGRAM_CREATION_FAILED";
    case OperationStatus::NO_SOURCE_FOUND: return "NO_SOURCE_FOUND";
    case OperationStatus::LOADING_SOURCE_FAILED: return "LOADING_SOURCE_FAILED";
    }
    return "UNKNOWN";
}
