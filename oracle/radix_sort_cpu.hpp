// oracle/radix_sort_cpu.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the reference's single-threaded LSD counting sort
// (`RadixSortCPU<T>::sort`, /root/reference/src/CRadixSortCPU.h:58-122) and of
// the reference's input generators (/root/reference/src/Dataset.h:84-137).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// call into this directory, and only as the checker / reported baseline.
// The shipped sort path (radix-sort_amd/) never links or loads it.
//
// Parity status: PINNED.  The restatement is checked byte-for-byte against the
// reference's own headers compiled from /root/reference (oracle/_ref, built by
// oracle/Makefile) and against the golden digests the survey recorded from those
// headers (tests/golden/oracle_golden.json, made by oracle/make_golden.py).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <string>
#include <type_traits>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------
// RadixSortCPU restatement
// ---------------------------------------------------------------------------

/// Base of the counting sort.  The reference derives it as
/// `_TOTALBITS / _NUM_BITS_PER_RADIX` (CRadixSortCPU.h:35 with Parameters.h:25,36),
/// i.e. the *pass count* of the GPU sort (8 for 32-bit, 16 for 64-bit keys),
/// not the GPU radix of 16.
template <typename T>
inline constexpr unsigned kBase = static_cast<unsigned>(sizeof(T) * 8U / 4U);

/// Unsigned magnitude as the reference's file-local `abs` computes it
/// (CRadixSortCPU.h:20-25): negate in the signed type, convert to unsigned.
template <typename T>
inline std::make_unsigned_t<T> magnitude(T v)
{
    using U = std::make_unsigned_t<T>;
    if constexpr (std::is_signed_v<T>) {
        return v < 0 ? static_cast<U>(U{0} - static_cast<U>(v)) : static_cast<U>(v);
    } else {
        return v;
    }
}

/// Number of counting-sort rounds the reference runs (CRadixSortCPU.h:62,67):
/// taken from the RAW (signed) maximum, `ceil(log(|max|) / log(base))`, and 1 when
/// the maximum is 0.  This under-counts for exact powers of the base, for
/// max == 1 and for signed inputs with a small raw maximum; the oracle keeps that
/// behaviour on purpose (see tests/golden known-answer cases).
template <typename T>
inline std::uint64_t round_count(const T* data, std::size_t n)
{
    const T max_elem = *std::max_element(data, data + n);
    if (!max_elem) {
        return 1;
    }
    const double num = std::log(magnitude(max_elem));
    const double den = std::log(kBase<T>);
    return static_cast<std::uint64_t>(std::ceil(num / den));
}

/// One stable counting-sort round on digit `(v / weight) % base`
/// (CRadixSortCPU.h:82-122).  `v` is the key shifted into the unsigned range by
/// subtracting numeric_limits<T>::min() (:93,97,109).  When `payload` is non-null
/// it is permuted together with the keys — an extension the reference does not
/// have (its permutation buffer is never written); semantics = stable argsort.
template <typename T>
inline void counting_round(T* keys, std::uint32_t* payload, std::size_t n, std::uint64_t weight)
{
    using U = std::make_unsigned_t<T>;
    constexpr unsigned base = kBase<T>;
    constexpr U bias = static_cast<U>(std::numeric_limits<T>::min());

    std::vector<T> placed(n, T{0});                 // reference allocates per round (:88)
    std::vector<std::uint32_t> placed_payload(payload ? n : 0);
    std::vector<std::size_t> fill(base, 0);         // (:90)

    auto digit_of = [weight](T k) {
        const U v = static_cast<U>(static_cast<U>(k) - bias);
        return static_cast<unsigned>((v / weight) % base);
    };

    for (std::size_t i = 0; i < n; ++i) {           // occurrences (:96-99)
        ++fill[digit_of(keys[i])];
    }
    for (unsigned b = 1; b < base; ++b) {           // inclusive prefix (:103-105)
        fill[b] += fill[b - 1];
    }
    for (std::int64_t i = static_cast<std::int64_t>(n) - 1; i >= 0; --i) {  // backwards, stable (:108-113)
        const unsigned b = digit_of(keys[i]);
        const std::size_t slot = --fill[b];
        placed[slot] = keys[i];
        if (payload) {
            placed_payload[slot] = payload[i];
        }
    }
    std::copy(placed.begin(), placed.end(), keys);  // copy back (:117-121)
    if (payload) {
        std::copy(placed_payload.begin(), placed_payload.end(), payload);
    }
}

/// RadixSortCPU<T>::sort restated (CRadixSortCPU.h:58-72).  Empty input is a
/// no-op here (the reference dereferences max_element of an empty span).
template <typename T>
inline void radix_sort(T* keys, std::size_t n, std::uint32_t* payload = nullptr)
{
    if (n == 0) {
        return;
    }
    const std::uint64_t rounds = round_count(keys, n);
    for (std::uint64_t r = 0; r < rounds; ++r) {
        const auto weight = static_cast<std::uint64_t>(std::pow(kBase<T>, r));   // (:70)
        counting_round(keys, payload, n, weight);
    }
}

// ---------------------------------------------------------------------------
// Second referee: std::sort (CRadixSortTask.cpp:32-43) and the stable argsort
// that defines payload semantics (SURVEY §8c "parity unpinned" for payloads).
// ---------------------------------------------------------------------------

template <typename T>
inline void std_sort(T* keys, std::size_t n)
{
    std::sort(keys, keys + n);
}

template <typename T>
inline void stable_argsort(const T* keys, std::uint32_t* perm_inout, std::size_t n)
{
    // perm_inout holds the payload of each input slot on entry; on exit it holds
    // the payloads in key order, equal keys keeping input order.
    std::vector<std::uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0U);
    std::stable_sort(idx.begin(), idx.end(), [keys](std::uint32_t a, std::uint32_t b) { return keys[a] < keys[b]; });
    std::vector<std::uint32_t> out(n);
    for (std::size_t i = 0; i < n; ++i) {
        out[i] = perm_inout[idx[i]];
    }
    std::copy(out.begin(), out.end(), perm_inout);
}

// ---------------------------------------------------------------------------
// Input generators restated (Dataset.h:84-137)
// ---------------------------------------------------------------------------

enum class DatasetKind : int {
    Zeros = 0,          // Dataset.h:84-89
    Range = 1,          // :132-137  iota from numeric_limits::min()
    InvertedRange = 2,  // :123-129  the same, reversed
    Random = 3,         // :110-120  mt19937 seeded from the string "Random Test Seed"
    SeededUniform = 4,  // stand-in for the clock-seeded RandomDistributed (:92-107)
};

inline constexpr const char* kReferenceSeedText = "Random Test Seed";   // Dataset.h:113

template <typename T>
inline void fill_dataset(DatasetKind kind, T* out, std::size_t n, std::uint64_t seed = 0x5EEDCAFEF00DULL)
{
    switch (kind) {
    case DatasetKind::Zeros:
        std::fill(out, out + n, T{0});
        break;
    case DatasetKind::Range:
        std::iota(out, out + n, std::numeric_limits<T>::min());
        break;
    case DatasetKind::InvertedRange:
        std::iota(out, out + n, std::numeric_limits<T>::min());
        std::reverse(out, out + n);
        break;
    case DatasetKind::Random: {
        // 32-bit engine for every key type: 64-bit keys receive zero-extended 32-bit
        // values, int32 the same bits reinterpreted (Dataset.h:115-119).
        const std::string text(kReferenceSeedText);
        std::seed_seq seq(text.begin(), text.end());
        std::mt19937 engine(seq);
        for (std::size_t i = 0; i < n; ++i) {
            out[i] = static_cast<T>(engine());
        }
        break;
    }
    case DatasetKind::SeededUniform: {
        // The reference seeds from the clock and draws through libstdc++'s
        // uniform_int_distribution — irreproducible.  Stand-in: raw engine output of
        // the key's width from a two-word seed (same shape as Dataset.h:96), then the
        // forced extremes of Dataset.h:105-106.
        std::seed_seq seq({static_cast<std::uint32_t>(seed & 0xFFFFFFFFULL), static_cast<std::uint32_t>(seed >> 32)});
        if constexpr (sizeof(T) == 8) {
            std::mt19937_64 engine(seq);
            for (std::size_t i = 0; i < n; ++i) {
                out[i] = static_cast<T>(engine());
            }
        } else {
            std::mt19937 engine(seq);
            for (std::size_t i = 0; i < n; ++i) {
                out[i] = static_cast<T>(engine());
            }
        }
        if (n > 0) {
            out[0] = std::numeric_limits<T>::max();
            out[n - 1] = std::numeric_limits<T>::min();
        }
        break;
    }
    }
}

// ---------------------------------------------------------------------------
// FNV-1a-64 over the little-endian bytes of an array (digest used by the golden
// vectors)
// ---------------------------------------------------------------------------
inline std::uint64_t fnv1a64(const void* bytes, std::size_t nbytes)
{
    const auto* p = static_cast<const unsigned char*>(bytes);
    std::uint64_t h = 0xcbf29ce484222325ULL;
    for (std::size_t i = 0; i < nbytes; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

// ---------------------------------------------------------------------------
// Emulation of the reference's GPU pass structure on the host: 1024 virtual
// processors with contiguous sub-lists, counter table in [digit][group][item]
// order, 512-block two-level exclusive scan (RadixSort.cl:16-197 driven by
// RadixSortGPU.cpp:16-267).  Produces the same m_hHistograms / m_hGlobsum
// diagnostics the reference downloads after its last pass (RadixSortGPU.cpp:412-428).
// ---------------------------------------------------------------------------
template <typename T>
inline void emulate_reference_gpu_sort(T* keys, std::size_t n, std::uint32_t* table_out /*16*1024*/,
                                       std::uint32_t* globsum_out /*512*/)
{
    using U = std::make_unsigned_t<T>;
    constexpr unsigned R = 16, G = 16, I = 64, VP = G * I, HS = 512, BITS = 4;
    constexpr unsigned passes = sizeof(T) * 8 / BITS;
    constexpr U bias = static_cast<U>(std::numeric_limits<T>::min());
    const std::size_t sub = n / VP;                       // RadixSort.cl:39
    std::vector<T> other(n);
    std::vector<std::uint32_t> table(R * VP), globsum(HS);
    T* in = keys;
    T* out = other.data();
    for (unsigned pass = 0; pass < passes; ++pass) {
        std::fill(table.begin(), table.end(), 0U);
        auto digit_of = [pass](T k) {
            const U v = static_cast<U>(static_cast<U>(k) - bias);      // `+ OFFSET` (RadixSort.cl:51)
            return static_cast<unsigned>((v >> (pass * BITS)) & (R - 1));
        };
        for (unsigned vp = 0; vp < VP; ++vp) {            // histogram (:48-61,68-70)
            const unsigned gr = vp / I, it = vp % I;
            for (std::size_t j = 0; j < sub; ++j) {
                ++table[I * (digit_of(in[vp * sub + j]) * G + gr) + it];
            }
        }
        const unsigned per_block = R * VP / HS;           // 32 entries per scan group (RadixSortGPU.cpp:70-72)
        for (unsigned b = 0; b < HS; ++b) {               // scan #1 (:125-181)
            std::uint32_t run = 0;
            for (unsigned e = 0; e < per_block; ++e) {
                const std::uint32_t c = table[b * per_block + e];
                table[b * per_block + e] = run;
                run += c;
            }
            globsum[b] = run;
        }
        std::uint32_t run = 0;                            // scan #2 over the block sums
        for (unsigned b = 0; b < HS; ++b) {
            const std::uint32_t c = globsum[b];
            globsum[b] = run;
            run += c;
        }
        for (unsigned b = 0; b < HS; ++b) {               // paste (:185-197)
            for (unsigned e = 0; e < per_block; ++e) {
                table[b * per_block + e] += globsum[b];
            }
        }
        if (pass + 1 == passes) {
            if (table_out) std::copy(table.begin(), table.end(), table_out);
            if (globsum_out) std::copy(globsum.begin(), globsum.end(), globsum_out);
        }
        std::vector<std::uint32_t> cursor(table);         // reorder (:96-118) advances a private copy
        for (unsigned vp = 0; vp < VP; ++vp) {
            const unsigned gr = vp / I, it = vp % I;
            for (std::size_t j = 0; j < sub; ++j) {
                const T k = in[vp * sub + j];
                out[cursor[I * (digit_of(k) * G + gr) + it]++] = k;
            }
        }
        std::swap(in, out);                               // RadixSortGPU.cpp:263-266
    }
    if (in != keys) {
        std::copy(in, in + n, keys);
    }
}

}  // namespace oracle
