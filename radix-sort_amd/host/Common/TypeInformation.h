// TypeInformation.h — key type -> "uint32_t"-style name for reports, the stdint half of the
// reference's TypeNameString (Common/CLTypeInformation.h:8-46).  The OpenCL type names are
// gone with run-time compilation.
#pragma once

#include <cstdint>
#include <string_view>

template <typename T>
struct TypeNameString;

template <> struct TypeNameString<std::int32_t> { static constexpr std::string_view stdint_name = "int32_t"; };
template <> struct TypeNameString<std::uint32_t> { static constexpr std::string_view stdint_name = "uint32_t"; };
template <> struct TypeNameString<std::int64_t> { static constexpr std::string_view stdint_name = "int64_t"; };
template <> struct TypeNameString<std::uint64_t> { static constexpr std::string_view stdint_name = "uint64_t"; };
