// Statistics.h — min/max/avg/sum/n accumulator behind RuntimesGPU / RuntimesCPU
// (/root/reference/src/Statistics.h:5-32).  Same field names.  One behavioural fix: the
// reference tests `else if (value < min)` after the max test, so the first sample can
// never become the minimum (min stays +inf until a later, smaller sample); here min and
// max are updated independently.
#pragma once

#include <cstddef>
#include <limits>

struct Statistics {
    double min{std::numeric_limits<double>::infinity()};
    double max{-std::numeric_limits<double>::infinity()};
    double avg{0.0};
    double sum{0.0};
    std::size_t n{0U};

    void update(double value)
    {
        ++n;
        sum += value;
        avg = sum / static_cast<double>(n);
        if (value > max) max = value;
        if (value < min) min = value;
    }
};
