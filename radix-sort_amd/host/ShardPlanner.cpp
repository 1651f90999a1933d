// ShardPlanner.cpp — see ShardPlanner.h.
#include "ShardPlanner.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <utility>

namespace shardplan {

std::uint64_t ExchangePlan::n_recv() const
{
    return std::accumulate(recv.begin(), recv.end(), std::uint64_t{0});
}

std::vector<std::pair<int, int>> wave_groups(int waves, int grouping)
{
    std::vector<std::pair<int, int>> out;
    if (grouping == 0) {
        for (int w = 0; w < waves; ++w) out.emplace_back(w, 1);
    } else {
        for (int w = 0, g = 1; w < waves; w += g, g = (w <= 1 ? 1 : w)) out.emplace_back(w, std::min(g, waves - w));      // {0} {1} {2,3} {4..7} ...
    }
    return out;
}

int group_pass_units(int key_bits, int partition_bits, int group_waves)
{
    int lg = 0;
    while ((1 << lg) < group_waves) ++lg;
    return (key_bits - (partition_bits - lg) + 3) / 4;
}

WaveLayout wave_layout(const Table& table, int world, int nbuckets, int align, int grouping)
{
    if (world < 1 || nbuckets < world || nbuckets % world != 0 || static_cast<int>(table.size()) != world || align < 1)
        throw std::invalid_argument("wave_layout: the table must have one row of nbuckets counts per rank, nbuckets a multiple of world");
    const int waves = nbuckets / world;
    const std::uint64_t a = static_cast<std::uint64_t>(align);
    WaveLayout out;
    out.start.assign(world, std::vector<std::uint64_t>(waves, 0));
    out.offset.assign(world, std::vector<std::vector<std::uint64_t>>(waves, std::vector<std::uint64_t>(world, 0)));
    out.load.assign(world, 0);
    out.extent.assign(world, 0);
    for (int d = 0; d < world; ++d) {
        std::uint64_t at = 0, total = 0;
        for (int w = 0; w < waves; ++w) {
            if (grouping == 0 || (w & (w - 1)) == 0) at = (at + a - 1) / a * a;      // doubling groups: only waves 0, 1, 2, 4, 8, ... start aligned
            out.start[d][w] = at;
            const int b = d * waves + w;
            for (int src = 0; src < world; ++src) {
                out.offset[d][w][src] = at;
                at += table[src].at(b);
                total += table[src][b];
            }
        }
        out.load[d] = total;
        out.extent[d] = at;
    }
    return out;
}

std::vector<int> balanced_owner(const std::vector<std::uint64_t>& totals, int world)
{
    const std::uint64_t total = std::accumulate(totals.begin(), totals.end(), std::uint64_t{0});
    std::vector<int> owner;
    owner.reserve(totals.size());
    std::uint64_t run = 0;
    int rank = 0;
    for (const std::uint64_t c : totals) {
        // move on to the next rank once this one has its share, judged at the bucket's midpoint
        while (rank < world - 1 && total > 0 &&
               (static_cast<double>(run) + static_cast<double>(c) / 2.0) * static_cast<double>(world) >= static_cast<double>(static_cast<std::uint64_t>(rank + 1) * total)) {
            ++rank;
        }
        owner.push_back(rank);
        run += c;
    }
    return owner;
}

namespace {

std::vector<std::uint64_t> column_totals(const Table& table)
{
    std::vector<std::uint64_t> totals(table.empty() ? 0 : table[0].size(), 0);
    for (const auto& row : table) {
        for (std::size_t b = 0; b < totals.size(); ++b) totals[b] += row.at(b);
    }
    return totals;
}

ExchangePlan finish(const Table& sends, std::vector<std::uint64_t> loads, const std::vector<std::uint64_t>& totals, int rank, int world)
{
    ExchangePlan plan;
    plan.send = sends.at(rank);
    plan.recv.resize(world);
    for (int s = 0; s < world; ++s) plan.recv[s] = sends[s][rank];
    plan.loads = std::move(loads);
    const double sum = static_cast<double>(std::accumulate(totals.begin(), totals.end(), std::uint64_t{0}));
    const double ideal = std::max(1.0, sum / static_cast<double>(world));
    plan.imbalance = static_cast<double>(*std::max_element(plan.loads.begin(), plan.loads.end())) / ideal;
    return plan;
}

}  // namespace

ExchangePlan plan_from_table(const Table& table, int rank, int world)
{
    if (static_cast<int>(table.size()) != world || rank < 0 || rank >= world) throw std::invalid_argument("plan_from_table: one row per rank");
    const std::vector<std::uint64_t> totals = column_totals(table);
    const std::vector<int> owner = balanced_owner(totals, world);
    Table sends(world, std::vector<std::uint64_t>(world, 0));
    for (int s = 0; s < world; ++s) {
        for (std::size_t b = 0; b < totals.size(); ++b) sends[s][owner[b]] += table[s][b];
    }
    std::vector<std::uint64_t> loads(world, 0);
    for (int d = 0; d < world; ++d) {
        for (int s = 0; s < world; ++s) loads[d] += sends[s][d];
    }
    return finish(sends, std::move(loads), totals, rank, world);
}

std::vector<std::uint64_t> choose_splitters(const std::vector<std::vector<std::uint64_t>>& samples, const std::vector<std::uint64_t>& shard_sizes, int world)
{
    constexpr std::size_t kMaxSplitters = 7;
    std::vector<std::pair<std::uint64_t, double>> weighted;
    for (std::size_t r = 0; r < samples.size() && r < shard_sizes.size(); ++r) {
        if (shard_sizes[r] > 0 && !samples[r].empty()) {
            const double w = static_cast<double>(shard_sizes[r]) / static_cast<double>(samples[r].size());
            for (const std::uint64_t v : samples[r]) weighted.emplace_back(v, w);
        }
    }
    std::vector<std::uint64_t> out;
    if (weighted.empty()) return out;
    std::stable_sort(weighted.begin(), weighted.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    double total = 0.0;
    for (const auto& vw : weighted) total += vw.second;
    double run = 0.0;
    int k = 1;
    for (const auto& vw : weighted) {
        run += vw.second;
        while (k < world && run * static_cast<double>(world) >= static_cast<double>(k) * total) {
            if (out.empty() || out.back() != vw.first) out.push_back(vw.first);
            ++k;
        }
    }
    if (out.size() > kMaxSplitters) out.resize(kMaxSplitters);
    return out;
}

std::vector<std::uint64_t> split_cuts(const std::vector<std::uint64_t>& totals, int world)
{
    const std::uint64_t total = std::accumulate(totals.begin(), totals.end(), std::uint64_t{0});
    std::vector<std::uint64_t> starts{0};
    for (const std::uint64_t c : totals) starts.push_back(starts.back() + c);
    std::vector<std::uint64_t> cuts{0};
    for (int k = 1; k < world; ++k) {
        const std::uint64_t ideal = static_cast<std::uint64_t>(k) * total / static_cast<std::uint64_t>(world);
        std::uint64_t cut = ideal;
        for (std::size_t b = 0; b < totals.size(); ++b) {
            const std::uint64_t lo = starts[b], hi = starts[b + 1];
            if (lo < ideal && ideal < hi) {
                if (b % 2 == 0) cut = (ideal - lo <= hi - ideal) ? lo : hi;
                break;
            }
        }
        cuts.push_back(std::max(cut, cuts.back()));
    }
    cuts.push_back(total);
    return cuts;
}

ExchangePlan split_plan(const Table& table, int rank, int world)
{
    if (static_cast<int>(table.size()) != world || rank < 0 || rank >= world) throw std::invalid_argument("split_plan: one row per rank");
    const std::vector<std::uint64_t> totals = column_totals(table);
    const std::vector<std::uint64_t> cuts = split_cuts(totals, world);
    Table sends(world, std::vector<std::uint64_t>(world, 0));
    std::uint64_t pos = 0;
    for (std::size_t b = 0; b < totals.size(); ++b) {
        for (int r = 0; r < world; ++r) {
            const std::uint64_t lo = pos, hi = pos + table[r][b];
            for (int d = 0; d < world; ++d) {
                const std::uint64_t a = std::max(lo, cuts[d]), z = std::min(hi, cuts[d + 1]);
                if (z > a) sends[r][d] += z - a;
            }
            pos = hi;
        }
    }
    std::vector<std::uint64_t> loads(world);
    for (int d = 0; d < world; ++d) loads[d] = cuts[d + 1] - cuts[d];
    return finish(sends, std::move(loads), totals, rank, world);
}

void range_buckets(std::uint64_t lo, std::uint64_t hi, int key_bits, int* shift, std::uint64_t* mul)
{
    *shift = 0;
    *mul = 0;
    __extension__ typedef unsigned __int128 u128;
    const u128 span1 = static_cast<u128>(hi - lo) + 1;
    if (span1 <= 16) return;
    const u128 m = (static_cast<u128>(16) << key_bits) / span1;
    *mul = static_cast<std::uint64_t>(m);
}

int check_capacity(const std::vector<std::uint64_t>& loads, const std::vector<std::uint64_t>& recv_caps, const std::vector<std::uint64_t>& out_caps, bool need_out,
                   std::uint64_t slack)
{
    for (std::size_t r = 0; r < loads.size(); ++r) {
        if (loads[r] + slack > recv_caps.at(r) || (need_out && loads[r] > out_caps.at(r))) return static_cast<int>(r);
    }
    return -1;
}

int check_capacity_extent(const std::vector<std::uint64_t>& extents, const std::vector<std::uint64_t>& loads, const std::vector<std::uint64_t>& recv_caps,
                          const std::vector<std::uint64_t>& out_caps)
{
    for (std::size_t r = 0; r < loads.size(); ++r) {
        if (extents.at(r) > recv_caps.at(r) || loads[r] > out_caps.at(r)) return static_cast<int>(r);
    }
    return -1;
}

std::vector<PeerAccess> peer_access_plan(const std::vector<PeerIdentity>& ranks, int my_rank)
{
    const PeerIdentity& me = ranks.at(static_cast<std::size_t>(my_rank));
    std::vector<PeerAccess> out;
    out.reserve(ranks.size());
    for (std::size_t r = 0; r < ranks.size(); ++r) {
        const PeerIdentity& o = ranks[r];
        if (static_cast<int>(r) == my_rank) {
            out.push_back(PeerAccess::Self);
        } else if (o.host_hash != me.host_hash) {
            throw std::runtime_error("peer_access_plan: rank " + std::to_string(r) + " runs on another host: peer stores reach the GPUs of one node only");
        } else if (o.process_token == me.process_token && o.pid == me.pid) {
            out.push_back(o.device == me.device ? PeerAccess::SamePointer : PeerAccess::EnablePeerThenPointer);
        } else {
            out.push_back(PeerAccess::OpenIpcHandle);
        }
    }
    return out;
}

}  // namespace shardplan

// ---- C entry points ---------------------------------------------------------------------------------------------------------------
namespace {

shardplan::Table to_table(const std::uint64_t* flat, int world, int nbuckets)
{
    shardplan::Table t(static_cast<std::size_t>(world));
    for (int r = 0; r < world; ++r) t[r].assign(flat + static_cast<std::size_t>(r) * nbuckets, flat + static_cast<std::size_t>(r + 1) * nbuckets);
    return t;
}

int put_plan(const shardplan::ExchangePlan& p, int world, std::uint64_t* send, std::uint64_t* recv, std::uint64_t* loads, double* imbalance)
{
    for (int r = 0; r < world; ++r) {
        send[r] = p.send[r];
        recv[r] = p.recv[r];
        loads[r] = p.loads[r];
    }
    if (imbalance) *imbalance = p.imbalance;
    return 0;
}

}  // namespace

extern "C" {

int rsxh_plan_wave_groups(int waves, int grouping, int* groups_out)
{
    if (waves < 1 || !groups_out || (grouping != 0 && grouping != 1)) return -1;
    const auto groups = shardplan::wave_groups(waves, grouping);
    for (std::size_t i = 0; i < groups.size(); ++i) {
        groups_out[2 * i] = groups[i].first;
        groups_out[2 * i + 1] = groups[i].second;
    }
    return static_cast<int>(groups.size());
}

int rsxh_plan_group_pass_units(int key_bits, int partition_bits, int group_waves)
{
    return shardplan::group_pass_units(key_bits, partition_bits, group_waves);
}

int rsxh_plan_wave_layout(const std::uint64_t* table, int world, int nbuckets, int align, int grouping, std::uint64_t* start, std::uint64_t* offset,
                          std::uint64_t* load, std::uint64_t* extent)
{
    try {
        if (grouping != 0 && grouping != 1) return -1;
        const shardplan::WaveLayout l = shardplan::wave_layout(to_table(table, world, nbuckets), world, nbuckets, align, grouping);
        const int waves = nbuckets / world;
        for (int d = 0; d < world; ++d) {
            for (int w = 0; w < waves; ++w) {
                start[d * waves + w] = l.start[d][w];
                for (int s = 0; s < world; ++s) offset[(static_cast<std::size_t>(d) * waves + w) * world + s] = l.offset[d][w][s];
            }
            load[d] = l.load[d];
            if (extent) extent[d] = l.extent[d];
        }
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

int rsxh_plan_balanced_owner(const std::uint64_t* totals, int nbuckets, int world, int* owner)
{
    if (!totals || !owner || nbuckets < 1 || world < 1) return -1;
    const std::vector<int> o = shardplan::balanced_owner(std::vector<std::uint64_t>(totals, totals + nbuckets), world);
    std::copy(o.begin(), o.end(), owner);
    return 0;
}

int rsxh_plan_from_table(const std::uint64_t* table, int world, int nbuckets, int rank, std::uint64_t* send, std::uint64_t* recv, std::uint64_t* loads,
                         double* imbalance)
{
    try {
        return put_plan(shardplan::plan_from_table(to_table(table, world, nbuckets), rank, world), world, send, recv, loads, imbalance);
    } catch (const std::exception&) {
        return -1;
    }
}

int rsxh_plan_choose_splitters(const std::uint64_t* samples, const std::uint32_t* nsamples, const std::uint64_t* shard_sizes, int nrows, int world, std::uint64_t* out,
                               int* nout)
{
    if (!nsamples || !shard_sizes || !out || !nout || world < 1 || nrows < 0) return -1;
    std::vector<std::vector<std::uint64_t>> s(static_cast<std::size_t>(nrows));
    std::size_t at = 0;
    for (int r = 0; r < nrows; ++r) {
        s[r].assign(samples + at, samples + at + nsamples[r]);
        at += nsamples[r];
    }
    const std::vector<std::uint64_t> sp = shardplan::choose_splitters(s, std::vector<std::uint64_t>(shard_sizes, shard_sizes + nrows), world);
    std::copy(sp.begin(), sp.end(), out);
    *nout = static_cast<int>(sp.size());
    return 0;
}

int rsxh_plan_split_cuts(const std::uint64_t* totals, int nbuckets, int world, std::uint64_t* cuts)
{
    if (!totals || !cuts || nbuckets < 1 || world < 1) return -1;
    const std::vector<std::uint64_t> c = shardplan::split_cuts(std::vector<std::uint64_t>(totals, totals + nbuckets), world);
    std::copy(c.begin(), c.end(), cuts);
    return 0;
}

int rsxh_plan_split(const std::uint64_t* table, int world, int nbuckets, int rank, std::uint64_t* send, std::uint64_t* recv, std::uint64_t* loads, double* imbalance)
{
    try {
        return put_plan(shardplan::split_plan(to_table(table, world, nbuckets), rank, world), world, send, recv, loads, imbalance);
    } catch (const std::exception&) {
        return -1;
    }
}

int rsxh_plan_range_buckets(std::uint64_t lo, std::uint64_t hi, int key_bits, int* shift, std::uint64_t* mul)
{
    if (!shift || !mul || (key_bits != 32 && key_bits != 64) || hi < lo) return -1;
    shardplan::range_buckets(lo, hi, key_bits, shift, mul);
    return 0;
}

int rsxh_plan_check_capacity(const std::uint64_t* loads, const std::uint64_t* recv_caps, const std::uint64_t* out_caps, int world, int need_out, std::uint64_t slack)
{
    return shardplan::check_capacity(std::vector<std::uint64_t>(loads, loads + world), std::vector<std::uint64_t>(recv_caps, recv_caps + world),
                                     std::vector<std::uint64_t>(out_caps, out_caps + world), need_out != 0, slack);
}

int rsxh_plan_peer_access(const std::int64_t* identities, int world, int my_rank, int* access_out)
{
    if (!identities || !access_out || world < 1 || my_rank < 0 || my_rank >= world) return -1;
    std::vector<shardplan::PeerIdentity> ids(static_cast<std::size_t>(world));
    for (int r = 0; r < world; ++r) {
        ids[r].host_hash = static_cast<std::uint64_t>(identities[4 * r]);
        ids[r].process_token = static_cast<std::uint64_t>(identities[4 * r + 1]);
        ids[r].pid = identities[4 * r + 2];
        ids[r].device = static_cast<int>(identities[4 * r + 3]);
    }
    try {
        const std::vector<shardplan::PeerAccess> a = shardplan::peer_access_plan(ids, my_rank);
        for (int r = 0; r < world; ++r) access_out[r] = static_cast<int>(a[r]);
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

}  // extern "C"
