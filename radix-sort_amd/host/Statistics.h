// Statistics.h — running summary of a stream of timing samples; the record type behind
// RuntimesGPU / RuntimesCPU.  Field names (min, max, avg, sum, n) and update() are the reference's
// (/root/reference/src/Statistics.h:5-32) because reports read them directly.
//
// One behavioural fix: the reference updates the minimum in the `else` branch of the maximum test,
// so a first sample can never become the minimum (min stays +infinity until a later, smaller one
// arrives).  Here both extremes are tracked independently.  merge()/reset() are additions used
// when per-launch HIP-event timings are folded in batches.
#pragma once

#include <algorithm>
#include <cstddef>
#include <limits>

struct Statistics {
    double min{std::numeric_limits<double>::infinity()};
    double max{-std::numeric_limits<double>::infinity()};
    double avg{0.0};
    double sum{0.0};
    std::size_t n{0U};

    /// Adds one sample.
    void update(double value)
    {
        sum += value;
        n += 1;
        avg = sum / static_cast<double>(n);
        min = std::min(min, value);
        max = std::max(max, value);
    }

    /// Adds a batch that was summarised elsewhere (count, total, extremes).
    void merge(std::size_t count, double total, double lowest, double highest)
    {
        if (count == 0) return;
        n += count;
        sum += total;
        avg = sum / static_cast<double>(n);
        min = std::min(min, lowest);
        max = std::max(max, highest);
    }

    void reset() { *this = Statistics{}; }
};
