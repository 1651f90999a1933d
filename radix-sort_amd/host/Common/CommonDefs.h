// CommonDefs.h — LocalWorkSize as in the reference (Common/CommonDefs.h:22); the value is
// ignored by the sort (tests/tests.cpp:78-79) and kept for signature parity only.
#pragma once

#include <array>
#include <cstddef>

using LocalWorkSize = std::array<std::size_t, 3>;
