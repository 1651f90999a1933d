// Parameters.h — AlgorithmParameters<T>, the tuning/derived-constant bundle of the
// reference (/root/reference/src/Parameters.h:9-60), for the MI355X engine.
//
// Every name the reference exposes is kept, because callers size their host buffers
// with them (examples/basic_sort/basic_sort.cpp:45-49: `_RADIX * _NUM_ITEMS` words of
// histogram read-back, `_NUM_HISTOSPLIT` block sums).  Two things differ:
//   * `_NUM_MAX_INPUT_ELEMS` is no longer a hard cap.  The reference asserts n <= 2^25
//     and allocates four 2^25-element vectors per task (src/HostData.cpp:10-18); here
//     capacity is a run-time value (MaxInputElems(), default unchanged) so the 2^28 and
//     2^30 configurations fit.
//   * the device geometry (tile of 256 threads x 16 keys, [digit][tile] table) is the
//     engine's business and is reported at run time (rsx_get_geometry); the constants
//     `_NUM_ITEMS_PER_GROUP`, `_NUM_GROUPS`, `_NUM_HISTOSPLIT` keep their reference
//     values only as host-buffer sizing and as the 1024-element rounding granule.
#pragma once

#include <cstddef>
#include <cstdint>
#include <limits>

template <typename KeyType>
struct AlgorithmParameters {
    using DataType = KeyType;

    // -- reference-valued constants (src/Parameters.h:17-29) ---------------------
    static constexpr std::uint32_t _NUM_ITEMS_PER_GROUP = 64U;
    static constexpr std::uint32_t _NUM_GROUPS = 16U;
    static constexpr std::uint32_t _NUM_ITEMS = _NUM_ITEMS_PER_GROUP * _NUM_GROUPS;   // rounding granule of Resize()
    static constexpr std::uint32_t _NUM_HISTOSPLIT = 512U;
    static constexpr std::uint32_t _NUM_BITS_PER_RADIX = 4U;
    static constexpr std::uint32_t _NUM_MAX_INPUT_ELEMS = 1U << 25U;   // default capacity only

    // -- derived (src/Parameters.h:36-52) -----------------------------------------
    static constexpr std::uint32_t _TOTALBITS = static_cast<std::uint32_t>(sizeof(DataType)) << 3U;
    static constexpr DataType _MAXINT = std::numeric_limits<DataType>::max();
    static constexpr std::uint32_t _RADIX = 1U << _NUM_BITS_PER_RADIX;
    static constexpr std::uint32_t _NUM_PASSES = _TOTALBITS / _NUM_BITS_PER_RADIX;
    static constexpr std::uint32_t _HISTOSIZE = _NUM_ITEMS * _RADIX;
    static constexpr std::uint32_t _NUM_PERFORMANCE_ITERATIONS = 5U;

    // -- MI355X engine additions -----------------------------------------------------
    /// Largest length the 32-bit slot arithmetic and the two-level table scan admit.
    static constexpr std::uint64_t _ENGINE_MAX_ELEMS = 0xFFFFFC00ULL;   // 2^32 - 1024
    /// Run-time capacity hook: the default mirrors the reference, callers may raise it.
    static std::uint64_t& MaxInputElems()
    {
        static std::uint64_t value = _NUM_MAX_INPUT_ELEMS;
        return value;
    }

    static_assert(_TOTALBITS % _NUM_BITS_PER_RADIX == 0, "digit width must divide the key width");
    static_assert(_NUM_MAX_INPUT_ELEMS % _NUM_ITEMS == 0, "default capacity must be a multiple of the rounding granule");
    static_assert(_HISTOSIZE % _NUM_HISTOSPLIT == 0, "histogram read-back must split evenly");
};
