// RadixSortOptions.h — run-time flags of the harness, the reference's five
// (/root/reference/src/RadixSortOptions.h:26-36) with identical spellings:
//   --num-elements N   --perf-to-stdout   --perf-to-csv   --perf-csv-to-stdout   -v/--verbose
// Differences: N is parsed as 64-bit (the reference's std::stoi cannot express 2^31 and
// above) and a missing value after --num-elements is an error instead of reading past
// the vector.  Extra, engine-specific switches are appended at the end.
#pragma once

#include "Parameters.h"

#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

struct RadixSortOptions {
    std::size_t num_elements;
    bool perf_to_stdout{false};
    bool perf_to_csv{false};
    bool perf_csv_to_stdout{false};
    bool verbose{false};
    // -- additions ---------------------------------------------------------------
    bool with_permutation{false};   ///< --with-permutation: carry h_Permut through the sort (argsort)
    bool stepwise{false};           ///< --stepwise: sync + host-time every launch like the reference
    bool skip_cpu{false};           ///< --skip-cpu: no CPU referees (large sizes); validation uses sortedness
    bool pinned{false};             ///< --pinned: page-lock the host key/result buffers for the transfers
    bool overlap{false};            ///< --overlap: the timed loop keeps two sorts in flight (upload / sort / download on three streams); implies --pinned
    bool zero_copy{false};          ///< --zero-copy: the timed loop sorts straight out of / into mapped host memory; implies --pinned
    int radix_bits{4};              ///< --radix-bits 4|8: digit width of the fused sort (the reference's _NUM_BITS_PER_RADIX is a compile-time 4, src/Parameters.h:25)

    explicit RadixSortOptions(const std::vector<std::string>& args = {})
        : num_elements(AlgorithmParameters<float>::_NUM_MAX_INPUT_ELEMS)   // default 2^25 (src/RadixSortOptions.h:18)
    {
        // switch spelling -> the flag it raises; unknown words are ignored, as in the reference
        struct Switch {
            const char* spelling;
            bool RadixSortOptions::*flag;
        };
        static constexpr Switch kSwitches[] = {
            {"--perf-to-stdout", &RadixSortOptions::perf_to_stdout},
            {"--perf-to-csv", &RadixSortOptions::perf_to_csv},
            {"--perf-csv-to-stdout", &RadixSortOptions::perf_csv_to_stdout},
            {"-v", &RadixSortOptions::verbose},
            {"--verbose", &RadixSortOptions::verbose},
            {"--with-permutation", &RadixSortOptions::with_permutation},
            {"--stepwise", &RadixSortOptions::stepwise},
            {"--skip-cpu", &RadixSortOptions::skip_cpu},
            {"--pinned", &RadixSortOptions::pinned},
            {"--overlap", &RadixSortOptions::overlap},
            {"--zero-copy", &RadixSortOptions::zero_copy},
        };
        for (auto it = args.begin(); it != args.end(); ++it) {
            if (*it == "--num-elements") {
                if (++it == args.end()) throw std::invalid_argument("--num-elements needs a value");
                num_elements = static_cast<std::size_t>(std::stoull(*it));
                continue;
            }
            if (*it == "--radix-bits") {
                if (++it == args.end()) throw std::invalid_argument("--radix-bits needs a value");
                radix_bits = std::stoi(*it);
                if (radix_bits != 4 && radix_bits != 8) throw std::invalid_argument("--radix-bits must be 4 or 8");
                continue;
            }
            for (const Switch& sw : kSwitches) {
                if (*it == sw.spelling) this->*sw.flag = true;
            }
        }
        if (overlap || zero_copy) pinned = true;
    }
};
