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
    // -- sharded sort (RadixSortMultiGPU: one rank thread per GPU in this process) -----------------------------------------
    int gpus{1};                    ///< --gpus N: shard every task over the first N devices (N > 1, or --sharded, selects RadixSortMultiGPU)
    int ranks{0};                   ///< --ranks R: R rank threads dealt round-robin over the --gpus devices (default: one per GPU); more ranks than GPUs = loopback communicator
    bool sharded{false};            ///< --sharded: go through the partition + exchange even with one rank (the communicator talks to itself)
    std::string comm{"auto"};       ///< --comm auto|rccl|loopback
    std::string exchange{"all-to-all"};   ///< --exchange all-to-all|peer-stores
    int partition_bits{0};          ///< --partition-bits B: top key bits of the exchange partition (2^B / ranks waves per rank); 0 = default
    bool single_waves{false};       ///< --single-waves: sort every wave by itself instead of in doubling groups {0} {1} {2,3} {4..7}

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
            {"--sharded", &RadixSortOptions::sharded},
            {"--single-waves", &RadixSortOptions::single_waves},
        };
        for (auto it = args.begin(); it != args.end(); ++it) {
            if (*it == "--num-elements") {
                if (++it == args.end()) throw std::invalid_argument("--num-elements needs a value");
                num_elements = static_cast<std::size_t>(std::stoull(*it));
                continue;
            }
            if (*it == "--gpus" || *it == "--ranks" || *it == "--partition-bits") {
                const std::string flag = *it;
                if (++it == args.end()) throw std::invalid_argument(flag + " needs a value");
                const int v = std::stoi(*it);
                if (v < 0 || v > 16) throw std::invalid_argument(flag + " out of range");
                (flag == "--gpus" ? gpus : flag == "--ranks" ? ranks : partition_bits) = v;
                continue;
            }
            if (*it == "--comm" || *it == "--exchange") {
                const std::string flag = *it;
                if (++it == args.end()) throw std::invalid_argument(flag + " needs a value");
                (flag == "--comm" ? comm : exchange) = *it;
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
        if (gpus < 1) gpus = 1;
        if (comm != "auto" && comm != "rccl" && comm != "loopback") throw std::invalid_argument("--comm must be auto, rccl or loopback");
        if (exchange != "all-to-all" && exchange != "peer-stores") throw std::invalid_argument("--exchange must be all-to-all or peer-stores");
    }

    /// More than one rank, or asked for explicitly: the task runs on RadixSortMultiGPU.
    bool useSharded() const { return sharded || gpus > 1 || ranks > 1; }
    int numRanks() const { return ranks > 0 ? ranks : gpus; }
};
