"""Independent pure-Python statement of the sharded sort's host arithmetic (TEST INFRASTRUCTURE: the checker of
radix-sort_amd/host/ShardPlanner.cpp, which is the ONE implementation the product's Python and C++ drivers share through
planner.py / ShardPlanner.h).  tests/test_planner.py feeds both the same tables and compares every field."""
from __future__ import annotations

from dataclasses import dataclass

RADIX = 16
MAX_SPLITTERS = 7


@dataclass
class ExchangePlan:
    send: list
    recv: list
    loads: list | None = None

    @property
    def n_recv(self) -> int:
        return sum(self.recv)


class CapacityError(RuntimeError):
    pass


def wave_groups(waves, grouping=0):
    if grouping == 0:
        return [(w, 1) for w in range(waves)]
    out, w = [], 0
    while w < waves:
        g = 1 if w < 2 else w
        out.append((w, min(g, waves - w)))
        w += g
    return out


def group_pass_units(key_bits, partition_bits, group_waves=1):
    return (key_bits - (partition_bits - (group_waves - 1).bit_length()) + 3) // 4


def wave_layout(table, world_size, nbuckets=16, align=4, grouping=0):
    """table[source][bucket] in natural bucket order; rank r owns buckets r*k .. r*k+k-1, wave w = bucket r*k+w of every rank."""
    k = nbuckets // world_size
    start, offset, load, extent = [], [], [], []
    for d in range(world_size):
        at, st, of, total = 0, [], [], 0
        for w in range(k):
            if grouping == 0 or w & (w - 1) == 0:
                at = (at + align - 1) // align * align
            st.append(at)
            row = []
            for src in range(world_size):
                row.append(at)
                at += table[src][d * k + w]
                total += table[src][d * k + w]
            of.append(row)
        start.append(st)
        offset.append(of)
        load.append(total)
        extent.append(at)
    return start, offset, load, extent


def peer_access(identities, my_rank):
    """0 self, 1 same pointer (thread of this process on this device), 2 enable peer access then pointer (thread of this process on
    another device), 3 open an IPC handle (another process of this host); a rank on another host is an error."""
    me = identities[my_rank]
    out = []
    for r, o in enumerate(identities):
        if r == my_rank:
            out.append(0)
        elif o[0] != me[0]:
            raise RuntimeError("another host")
        elif o[1] == me[1] and o[2] == me[2]:
            out.append(1 if o[3] == me[3] else 2)
        else:
            out.append(3)
    return out


def recv_splits(all_send_splits: list[list[int]], rank: int) -> list[int]:
    """Number of keys this rank receives from each source rank."""
    return [row[rank] for row in all_send_splits]


def balanced_owner(global_counts: list[int], world_size: int) -> list[int]:
    """Bucket -> rank as contiguous ranges cut where the running total crosses k/world of all
    keys (every rank computes the same map from the same gathered counts)."""
    total = sum(global_counts)
    owner, run, rank = [], 0, 0
    for c in global_counts:
        # move on to the next rank once this one has its share, judged at the bucket's midpoint
        while rank < world_size - 1 and (run + c / 2) * world_size >= (rank + 1) * total and total > 0:
            rank += 1
        owner.append(rank)
        run += c
    return owner


def check_capacity(loads: list[int], caps: list[tuple[int, int]], need_out: bool, slack: int = 0) -> None:
    for r, (load, (recv_cap, out_cap)) in enumerate(zip(loads, caps)):
        if load + slack > recv_cap or (need_out and load > out_cap):
            raise CapacityError(f"rank {r} would receive {load} keys but its buffers hold {recv_cap} (receive) / {out_cap} (output)")


def plan_from_table(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan from the gathered count table plus the resulting imbalance
    (largest rank load / ideal load)."""
    nb = len(table[0])
    totals = [sum(row[b] for row in table) for b in range(nb)]
    owner = balanced_owner(totals, world_size)
    sends = [[sum(row[b] for b in range(nb) if owner[b] == dst) for dst in range(world_size)] for row in table]
    loads = [sum(s[dst] for s in sends) for dst in range(world_size)]
    ideal = max(1.0, sum(totals) / world_size)
    return ExchangePlan(send=sends[rank], recv=recv_splits(sends, rank), loads=loads), max(loads) / ideal


def choose_splitters(samples: list[list[int]], shard_sizes: list[int], world_size: int) -> list[int]:
    """world_size-1 weighted quantiles of the gathered samples (each of rank r's samples stands
    for shard_sizes[r] / len(samples[r]) keys), deduplicated and increasing.  Values are in
    unsigned sort order.  May return fewer than world_size-1 (down to none, if no rank has keys)."""
    weighted = []
    for vals, n in zip(samples, shard_sizes):
        if n > 0 and vals:
            weighted.extend((v, n / len(vals)) for v in vals)
    if not weighted:
        return []
    weighted.sort(key=lambda t: t[0])
    total = sum(w for _, w in weighted)
    out, run, k = [], 0.0, 1
    for v, w in weighted:
        run += w
        while k < world_size and run * world_size >= k * total:
            if not out or out[-1] != v:
                out.append(v)
            k += 1
    return out[:MAX_SPLITTERS]


def split_cuts(totals: list[int], world_size: int) -> list[int]:
    """Global positions (in bucket-major, rank-major, index order) where one rank's share ends
    and the next begins: world_size+1 monotone values from 0 to the number of keys.  The ideal cut
    k*total/world is kept when it falls inside an odd ("equal to a splitter") bucket and moved to
    the nearer end of the bucket when it falls inside an even one, which cannot be cut."""
    total = sum(totals)
    starts = [0]
    for c in totals:
        starts.append(starts[-1] + c)
    cuts = [0]
    for k in range(1, world_size):
        ideal = k * total // world_size
        cut = ideal
        for b, c in enumerate(totals):
            lo, hi = starts[b], starts[b + 1]
            if lo < ideal < hi:
                if b % 2 == 0:
                    cut = lo if ideal - lo <= hi - ideal else hi
                break
        cuts.append(max(cut, cuts[-1]))
    cuts.append(total)
    return cuts


def split_plan(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan of the splitter path from the [source rank][bucket] count table.  Source r's
    keys of bucket b occupy global positions start_b + sum(table[r'][b] for r' < r) onwards; each
    rank sends to destination d the part of its keys inside [cut_d, cut_d+1) — contiguous in its
    bucket-grouped staging buffer and in destination order."""
    nb = len(table[0])
    totals = [sum(row[b] for row in table) for b in range(nb)]
    cuts = split_cuts(totals, world_size)
    sends = [[0] * world_size for _ in table]
    pos = 0
    for b in range(nb):
        for r, row in enumerate(table):
            lo, hi = pos, pos + row[b]
            for d in range(world_size):
                a, z = max(lo, cuts[d]), min(hi, cuts[d + 1])
                if z > a:
                    sends[r][d] += z - a
            pos = hi
    loads = [cuts[d + 1] - cuts[d] for d in range(world_size)]
    ideal = max(1.0, sum(totals) / world_size)
    return ExchangePlan(send=sends[rank], recv=recv_splits(sends, rank), loads=loads), max(loads) / ideal


def range_buckets(lo: int, hi: int, key_bits: int) -> tuple[int, int]:
    """(shift, mul) of the 16 equal-width buckets over [lo, hi] (C ABI rsx_partition_range):
    bucket(x) = mulhi(x, mul) with mul = floor(16 * 2^key_bits / (hi - lo + 1)); ranges of at
    most 16 values use bucket(x) = x (shift 0, mul 0)."""
    span1 = hi - lo + 1
    if span1 <= RADIX:
        return 0, 0
    mul = (RADIX << key_bits) // span1
    assert mul < (1 << key_bits) and ((hi - lo) * mul) >> key_bits < RADIX
    return 0, mul
