// ShardPlanner.h — the host arithmetic of the sharded sort: from the gathered [source rank][bucket] count table to who
// sends what where.  ONE implementation: RadixSortMultiGPU<T> (C++) uses it directly, radix-sort_amd/planner.py (ShardedSorter) through the
// extern "C" entry points at the end; the device computes the same wave layout itself for the peer-store path (rsx_msd_plan) and the GPU
// tests compare the two.  The reference has nothing to mirror here (single device, /root/reference/Common/ComputeState.cpp:88-101).
//
// Everything is a pure function of gathered data, so that every rank reaches the same decision without another exchange — in particular
// the capacity verdict: a rank that found out alone that it overflows and raised would leave its peers hanging in a collective.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace shardplan {

using Table = std::vector<std::vector<std::uint64_t>>;      // [source rank][bucket]

struct ExchangePlan {
    std::vector<std::uint64_t> send, recv;      // keys this rank sends to / receives from every rank
    std::vector<std::uint64_t> loads;           // keys every rank ends up with (the same list on all ranks)
    double imbalance = 0.0;                     // largest load / ideal load
    std::uint64_t n_recv() const;
};

// Receive layout of the pipelined ("waves") paths.  nbuckets = 2^bits buckets in natural order, rank r owns buckets r*k .. r*k+k-1
// (k = nbuckets / world), wave w holds bucket r*k + w of every rank.  At destination d the waves follow each other and inside a wave the
// sources follow each other in rank order.  grouping 0: every wave starts on a multiple of `align` keys (each wave is sorted by itself).
// grouping 1 ("doubling groups"): only waves 0, 1, 2, 4, 8, ... do — the waves of a group {0} {1} {2,3} {4..7} {8..15} are contiguous
// without gaps and are sorted TOGETHER once the group's last wave has landed: a group of 2^j aligned buckets still shares the top
// bits - j key bits, so up to 4 buckets cost no extra pass unit at bits = 6, and the later, larger sorts run at the big-sort rate while
// the exposed first wave stays 1/k of the exchange.
struct WaveLayout {
    std::vector<std::vector<std::uint64_t>> start;                  // [destination][wave]  first slot of the wave
    std::vector<std::vector<std::vector<std::uint64_t>>> offset;    // [destination][wave][source]  first slot of that source's keys
    std::vector<std::uint64_t> load;                                // [destination]  keys it ends up with
    std::vector<std::uint64_t> extent;                              // [destination]  slots its receive buffer needs (alignment gaps included)
};
WaveLayout wave_layout(const Table& table, int world, int nbuckets, int align = 4, int grouping = 0);

/// (first wave, number of waves) of every group the local sorts take together: grouping 0 -> k groups of one wave; 1 -> {0} {1} {2,3} {4..7} ...
std::vector<std::pair<int, int>> wave_groups(int waves, int grouping);
/// 4-bit pass units the local sort of a group of `group_waves` waves needs (its keys share the top partition_bits - log2(group_waves) bits).
int group_pass_units(int key_bits, int partition_bits, int group_waves);

// Bucket -> rank as contiguous ranges cut where the running total crosses k/world of all keys.
std::vector<int> balanced_owner(const std::vector<std::uint64_t>& totals, int world);
// One all-to-all of whole buckets dealt out by balanced_owner.
ExchangePlan plan_from_table(const Table& table, int rank, int world);

// Splitter path: world-1 weighted quantiles of the gathered samples (rank r's samples stand for shard_sizes[r] / #samples keys each),
// deduplicated, increasing, at most 7.  Values in unsigned sort order.
std::vector<std::uint64_t> choose_splitters(const std::vector<std::vector<std::uint64_t>>& samples, const std::vector<std::uint64_t>& shard_sizes, int world);
// Global positions where one rank's share ends: ideal cuts kept inside odd ("equal to a splitter") buckets, snapped to the nearer end of even ones.
std::vector<std::uint64_t> split_cuts(const std::vector<std::uint64_t>& totals, int world);
ExchangePlan split_plan(const Table& table, int rank, int world);

// 16 equal-width buckets over [lo, hi] (rsx_partition_range): bucket(x) = mulhi(x - lo, mul); ranges of at most 16 values: bucket = x - lo (mul 0).
void range_buckets(std::uint64_t lo, std::uint64_t hi, int key_bits, int* shift, std::uint64_t* mul);

// -1 if every rank's buffers hold its load; otherwise the first rank that overflows.  need_out: the output buffer must hold the load too.
int check_capacity(const std::vector<std::uint64_t>& loads, const std::vector<std::uint64_t>& recv_caps, const std::vector<std::uint64_t>& out_caps,
                   bool need_out, std::uint64_t slack);
int check_capacity_extent(const std::vector<std::uint64_t>& extents, const std::vector<std::uint64_t>& loads, const std::vector<std::uint64_t>& recv_caps,
                          const std::vector<std::uint64_t>& out_caps);

// How a rank reaches every other rank's receive buffer in the peer-store exchange: decided from what every rank published about itself.
enum class PeerAccess : int { Self = 0, SamePointer = 1, EnablePeerThenPointer = 2, OpenIpcHandle = 3 };
struct PeerIdentity {
    std::uint64_t host_hash;      // hash of the host name (ranks on other hosts are not reachable by peer stores at all)
    std::uint64_t process_token;  // random per process, drawn once: equal pids in different namespaces / hosts do not collide
    std::int64_t pid;
    int device;                   // device ordinal inside that process
};
// Self: the rank itself.  SamePointer: a thread of this process on this device.  EnablePeerThenPointer: a thread of this process on ANOTHER device
// (rsx_peer_enable first).  OpenIpcHandle: another process on this host.  Throws std::runtime_error for a rank on another host.
std::vector<PeerAccess> peer_access_plan(const std::vector<PeerIdentity>& ranks, int my_rank);

}  // namespace shardplan

// ---- C entry points (ctypes; all tables row-major uint64) -------------------------------------------------------------------------
extern "C" {
int rsxh_plan_wave_layout(const std::uint64_t* table, int world, int nbuckets, int align, int grouping, std::uint64_t* start, std::uint64_t* offset,
                          std::uint64_t* load, std::uint64_t* extent);
// groups_out: 2 ints per group (first wave, waves), at most `waves` groups; returns the number of groups (or -1)
int rsxh_plan_wave_groups(int waves, int grouping, int* groups_out);
int rsxh_plan_group_pass_units(int key_bits, int partition_bits, int group_waves);
int rsxh_plan_balanced_owner(const std::uint64_t* totals, int nbuckets, int world, int* owner);
int rsxh_plan_from_table(const std::uint64_t* table, int world, int nbuckets, int rank, std::uint64_t* send, std::uint64_t* recv, std::uint64_t* loads,
                         double* imbalance);
// nrows sample rows (one per rank that published samples; nsamples[r] values each, concatenated), world - 1 quantiles wanted
int rsxh_plan_choose_splitters(const std::uint64_t* samples, const std::uint32_t* nsamples, const std::uint64_t* shard_sizes, int nrows, int world, std::uint64_t* out,
                               int* nout);
int rsxh_plan_split_cuts(const std::uint64_t* totals, int nbuckets, int world, std::uint64_t* cuts);
int rsxh_plan_split(const std::uint64_t* table, int world, int nbuckets, int rank, std::uint64_t* send, std::uint64_t* recv, std::uint64_t* loads,
                    double* imbalance);
int rsxh_plan_range_buckets(std::uint64_t lo, std::uint64_t hi, int key_bits, int* shift, std::uint64_t* mul);
int rsxh_plan_check_capacity(const std::uint64_t* loads, const std::uint64_t* recv_caps, const std::uint64_t* out_caps, int world, int need_out,
                             std::uint64_t slack);
// identities: 4 x int64 per rank {host hash, process token, pid, device}; access_out: one PeerAccess value per rank; -1 on a rank of another host
int rsxh_plan_peer_access(const std::int64_t* identities, int world, int my_rank, int* access_out);
}
