"""Host arithmetic of the sharded sort: ctypes binding of radix-sort_amd/host/ShardPlanner.{h,cpp} (libradixsort_host.so).

ONE implementation serves both drivers — `ShardedSorter` (Python, torch.distributed) through this module and
`RadixSortMultiGPU<T>` (C++) directly — so a plan is the same whichever host asks for it.  Everything here is a pure
function of gathered data: every rank reaches the same decision (and the same capacity verdict) without another exchange.
tests/_planner_ref.py keeps an independent pure-Python statement of the same functions as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "host", "libradixsort_host.so")
MAX_SPLITTERS = 7           # 2*7+1 = 15 buckets fit the 16-bucket kernels
_lib = None

# shardplan::PeerAccess
PEER_SELF, PEER_SAME_POINTER, PEER_ENABLE_THEN_POINTER, PEER_OPEN_IPC = 0, 1, 2, 3


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise FileNotFoundError(f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(HOST_LIB_PATH)
    U64P, I, U64 = C.POINTER(C.c_uint64), C.c_int, C.c_uint64
    sig = {
        "rsxh_plan_wave_layout": [U64P, I, I, I, I, U64P, U64P, U64P, U64P],
        "rsxh_plan_wave_groups": [I, I, C.POINTER(I)],
        "rsxh_plan_group_pass_units": [I, I, I],
        "rsxh_plan_balanced_owner": [U64P, I, I, C.POINTER(I)],
        "rsxh_plan_from_table": [U64P, I, I, I, U64P, U64P, U64P, C.POINTER(C.c_double)],
        "rsxh_plan_choose_splitters": [U64P, C.POINTER(C.c_uint32), U64P, I, I, U64P, C.POINTER(I)],
        "rsxh_plan_split_cuts": [U64P, I, I, U64P],
        "rsxh_plan_split": [U64P, I, I, I, U64P, U64P, U64P, C.POINTER(C.c_double)],
        "rsxh_plan_range_buckets": [U64, U64, I, C.POINTER(I), U64P],
        "rsxh_plan_check_capacity": [U64P, U64P, U64P, I, I, U64],
        "rsxh_plan_peer_access": [C.POINTER(C.c_int64), I, I, C.POINTER(I)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = I
    _lib = lib
    return lib


def _arr(values) -> C.Array:
    values = [int(v) for v in values]
    return (C.c_uint64 * max(len(values), 1))(*values)


def _flat(table) -> tuple[C.Array, int, int]:
    world, nb = len(table), len(table[0])
    if any(len(row) != nb for row in table):
        raise ValueError("ragged count table")
    return _arr([v for row in table for v in row]), world, nb


@dataclass
class ExchangePlan:
    send: list[int]
    recv: list[int]
    loads: list[int] | None = None      # keys every rank ends up with (same list on all ranks)

    @property
    def n_recv(self) -> int:
        return sum(self.recv)


class CapacityError(RuntimeError):
    """Some rank's buffers cannot hold what the exchange plan sends it.  Raised by EVERY rank, before
    any key moves (the verdict only depends on gathered data)."""


GROUP_SINGLE, GROUP_DOUBLING = 0, 1


def wave_groups(waves: int, grouping: int = GROUP_SINGLE) -> list[tuple[int, int]]:
    """(first wave, number of waves) of every group the local sorts take together: single -> one wave each; doubling -> {0} {1} {2,3} {4..7} ..."""
    out = (C.c_int * (2 * max(waves, 1)))()
    n = _load().rsxh_plan_wave_groups(waves, grouping, out)
    if n < 0:
        raise ValueError("wave_groups: bad arguments")
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def group_pass_units(key_bits: int, partition_bits: int, group_waves: int = 1) -> int:
    """4-bit pass units the local sort of a group of `group_waves` waves needs: its keys share the top partition_bits - log2(group_waves) bits."""
    return int(_load().rsxh_plan_group_pass_units(key_bits, partition_bits, group_waves))


def wave_layout(table: list[list[int]], world_size: int, nbuckets: int = 16, align: int = 4, grouping: int = GROUP_SINGLE):
    """Where everything lands in the receive buffers of the pipelined paths, from the gathered [source][bucket] count table
    (natural bucket order: rank r owns buckets r*k .. r*k+k-1, k = nbuckets / world; wave w = bucket r*k+w of every rank): at
    destination d the waves follow each other, inside a wave the sources follow each other in rank order; a wave (grouping
    single) or only the first wave of a doubling group starts on a multiple of `align` keys (16 bytes: the local sort loads 16
    bytes per lane).  Returns (start[d][w], offset[d][w][s], load[d])."""
    flat, world, nb = _flat([row[:nbuckets] for row in table])
    if world != world_size or nb != nbuckets:
        raise ValueError("wave_layout: one row of nbuckets counts per rank")
    waves = nbuckets // world
    start, offset, load, extent = (C.c_uint64 * (world * waves))(), (C.c_uint64 * (world * waves * world))(), (C.c_uint64 * world)(), (C.c_uint64 * world)()
    if _load().rsxh_plan_wave_layout(flat, world, nbuckets, align, grouping, start, offset, load, extent) != 0:
        raise ValueError("wave_layout: nbuckets must be a multiple of the world size")
    st = [[int(start[d * waves + w]) for w in range(waves)] for d in range(world)]
    of = [[[int(offset[(d * waves + w) * world + s]) for s in range(world)] for w in range(waves)] for d in range(world)]
    return st, of, [int(v) for v in load]


def wave_extents(table: list[list[int]], world_size: int, nbuckets: int, align: int = 4, grouping: int = GROUP_SINGLE) -> list[int]:
    """Slots every destination's receive buffer needs for wave_layout's placement (alignment gaps included)."""
    flat, world, nb = _flat([row[:nbuckets] for row in table])
    waves = nbuckets // world
    start, offset, load, extent = (C.c_uint64 * (world * waves))(), (C.c_uint64 * (world * waves * world))(), (C.c_uint64 * world)(), (C.c_uint64 * world)()
    if _load().rsxh_plan_wave_layout(flat, world, nbuckets, align, grouping, start, offset, load, extent) != 0:
        raise ValueError("wave_extents: nbuckets must be a multiple of the world size")
    return [int(v) for v in extent]


def balanced_owner(global_counts: list[int], world_size: int) -> list[int]:
    """Bucket -> rank as contiguous ranges cut where the running total crosses k/world of all keys."""
    out = (C.c_int * len(global_counts))()
    if _load().rsxh_plan_balanced_owner(_arr(global_counts), len(global_counts), world_size, out) != 0:
        raise ValueError("balanced_owner: bad arguments")
    return [int(v) for v in out]


def _plan(fn, table, rank, world_size):
    flat, world, nb = _flat(table)
    if world != world_size:
        raise ValueError("one row per rank")
    send, recv, loads, imb = (C.c_uint64 * world)(), (C.c_uint64 * world)(), (C.c_uint64 * world)(), C.c_double()
    if fn(flat, world, nb, rank, send, recv, loads, C.byref(imb)) != 0:
        raise ValueError("exchange plan: bad arguments")
    return ExchangePlan([int(v) for v in send], [int(v) for v in recv], [int(v) for v in loads]), float(imb.value)


def plan_from_table(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan (whole buckets dealt out by balanced_owner) plus the resulting imbalance (largest load / ideal load)."""
    return _plan(_load().rsxh_plan_from_table, table, rank, world_size)


def split_plan(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan of the splitter path: ideal cuts kept inside odd ("equal to a splitter") buckets — ties split by (rank, index) —
    and snapped to the nearer end of even ones."""
    return _plan(_load().rsxh_plan_split, table, rank, world_size)


def split_cuts(totals: list[int], world_size: int) -> list[int]:
    cuts = (C.c_uint64 * (world_size + 1))()
    if _load().rsxh_plan_split_cuts(_arr(totals), len(totals), world_size, cuts) != 0:
        raise ValueError("split_cuts: bad arguments")
    return [int(v) for v in cuts]


def choose_splitters(samples: list[list[int]], shard_sizes: list[int], world_size: int) -> list[int]:
    """world_size-1 weighted quantiles of the gathered samples, deduplicated and increasing, at most 7 (unsigned sort order)."""
    nrows = min(len(samples), len(shard_sizes))
    flat = _arr([v for row in samples[:nrows] for v in row])
    counts = (C.c_uint32 * max(nrows, 1))(*[len(row) for row in samples[:nrows]])
    out, nout = (C.c_uint64 * MAX_SPLITTERS)(), C.c_int(0)
    if _load().rsxh_plan_choose_splitters(flat, counts, _arr(shard_sizes[:nrows]), nrows, world_size, out, C.byref(nout)) != 0:
        raise ValueError("choose_splitters: bad arguments")
    return [int(out[i]) for i in range(nout.value)]


def range_buckets(lo: int, hi: int, key_bits: int) -> tuple[int, int]:
    """(shift, mul) of the 16 equal-width buckets over [lo, hi] (C ABI rsx_partition_range)."""
    shift, mul = C.c_int(0), C.c_uint64(0)
    if _load().rsxh_plan_range_buckets(lo, hi, key_bits, C.byref(shift), C.byref(mul)) != 0:
        raise ValueError("range_buckets: bad arguments")
    return int(shift.value), int(mul.value)


def check_capacity(loads: list[int], caps: list[tuple[int, int]], need_out: bool, slack: int = 0) -> None:
    bad = _load().rsxh_plan_check_capacity(_arr(loads), _arr([c[0] for c in caps]), _arr([c[1] for c in caps]), len(loads), int(need_out), slack)
    if bad >= 0:
        recv_cap, out_cap = caps[bad]
        raise CapacityError(f"rank {bad} would receive {loads[bad]} keys but its buffers hold {recv_cap} (receive) / {out_cap} (output)")


def peer_access(identities: list[tuple[int, int, int, int]], my_rank: int) -> list[int]:
    """How this rank reaches every rank's receive buffer (PEER_*), from every rank's (host hash, process token, pid, device)."""
    world = len(identities)
    flat = (C.c_int64 * (4 * world))(*[C.c_int64(int(v) & 0xFFFFFFFFFFFFFFFF).value for row in identities for v in row])
    out = (C.c_int * world)()
    if _load().rsxh_plan_peer_access(flat, world, my_rank, out) != 0:
        raise RuntimeError("peer_access: a rank runs on another host (peer stores reach the GPUs of one node only), or bad arguments")
    return [int(v) for v in out]
