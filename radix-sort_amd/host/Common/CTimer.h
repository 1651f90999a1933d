// CTimer.h — host stopwatch with the reference's interface (Common/CTimer.h:14-38):
// Start / Stop / GetElapsedMilliseconds.  Header-only, steady clock.
#pragma once

#include <chrono>

class CTimer {
public:
    using Clock = std::chrono::steady_clock;
    void Start() { m_Start = Clock::now(); }
    void Stop() { m_End = Clock::now(); }
    double GetElapsedMilliseconds() const { return std::chrono::duration<double, std::milli>(m_End - m_Start).count(); }

private:
    Clock::time_point m_Start{}, m_End{};
};
