"""The ONE planner (radix-sort_amd/host/ShardPlanner.cpp through planner.py — what ShardedSorter and the C++ RadixSortMultiGPU<T>
both run) against the independent pure-Python statement in tests/_planner_ref.py: the same tables into both, every field compared.
Tables come from random draws and from the count tables of the sharded-sort inputs the gloo tests use (uniform, ties, skew, ranges)."""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import _planner_ref as ref  # noqa: E402


def _planner():
    import __graft_entry__ as entry
    entry.load_package()
    from radix_sort_amd import planner
    return planner


def _random_table(rnd, world, nb):
    style = rnd.choice(["even", "sparse", "hot", "big"])
    if style == "even":
        return [[rnd.randint(900, 1100) for _ in range(nb)] for _ in range(world)]
    if style == "sparse":
        return [[rnd.choice([0, 0, 0, rnd.randint(0, 5000)]) for _ in range(nb)] for _ in range(world)]
    if style == "hot":
        hot = rnd.randrange(nb)
        return [[rnd.randint(0, 50) + (100000 if b == hot else 0) for b in range(nb)] for _ in range(world)]
    return [[rnd.randint(0, 1 << 27) for _ in range(nb)] for _ in range(world)]


def _dataset_tables():
    """[source][bucket] count tables of the gloo tests' inputs (tests/test_distributed_gloo.py: 14 input/strategy cases), for the
    top-bit buckets of both widths the drivers use."""
    from _oracle import Oracle
    from test_distributed_gloo import _make_full
    orc = Oracle()
    out = []
    for dtype, kind in [("uint32", "SeededUniform"), ("int32", "SeededUniform"), ("int64", "SeededUniform"), ("uint64", "SeededUniform"), ("int64", "Random"),
                        ("uint32", "Zeros"), ("int32", "Range"), ("uint64", "InvertedRange"), ("uint32", "HeavyTies"), ("int64", "FewValues"),
                        ("uint32", "Skewed"), ("int64", "Skewed"), ("int64", "HeavyTies"), ("uint32", "Random")]:
        for world, bits in [(2, 4), (4, 5), (8, 6)]:
            n = 3000
            full = _make_full(kind, dtype, n * world, orc)
            kb = full.dtype.itemsize * 8
            u = full.view(np.uint32 if kb == 32 else np.uint64)
            if full.dtype.kind == "i":
                u = u ^ u.dtype.type(1 << (kb - 1))
            b = (u >> u.dtype.type(kb - bits)).astype(np.int64)
            out.append((world, 1 << bits, [[int(v) for v in np.bincount(b[r * n:(r + 1) * n], minlength=1 << bits)] for r in range(world)]))
    return out


def test_wave_layout_matches_the_reference_statement():
    pl = _planner()
    rnd = random.Random(11)
    cases = [(w, nb, _random_table(rnd, w, nb)) for _ in range(60) for w in (1, 2, 4, 8, 16) for nb in (16, 64, 256) if nb >= w]
    cases += _dataset_tables()
    for world, nb, table in cases:
        for align, grouping in ((1, 0), (4, 0), (4, 1)):
            start, offset, load = pl.wave_layout(table, world, nb, align, grouping)
            extent = pl.wave_extents(table, world, nb, align, grouping)
            want = ref.wave_layout(table, world, nb, align, grouping)
            assert (start, offset, load, extent) == want
            if grouping:            # inside a group the waves lie gap-free; a group starts on a multiple of `align`
                for first, count in pl.wave_groups(nb // world, grouping):
                    for d in range(world):
                        assert start[d][first] % align == 0
                        for w in range(first, first + count - 1):
                            assert start[d][w + 1] == start[d][w] + sum(table[s][d * (nb // world) + w] for s in range(world))
    with pytest.raises(ValueError):
        pl.wave_layout([[1] * 16] * 3, 3, 16)          # 16 buckets do not divide over 3 ranks
    with pytest.raises(ValueError):
        pl.wave_layout([[1] * 16, [1] * 15], 2, 16)    # ragged


def test_wave_groups_and_pass_units():
    pl = _planner()
    assert pl.wave_groups(8, pl.GROUP_DOUBLING) == [(0, 1), (1, 1), (2, 2), (4, 4)] and pl.wave_groups(1, 1) == [(0, 1)] and pl.wave_groups(2, 1) == [(0, 1), (1, 1)]
    assert pl.wave_groups(16, 1)[-1] == (8, 8) and pl.wave_groups(4, pl.GROUP_SINGLE) == [(0, 1), (1, 1), (2, 1), (3, 1)]
    for waves in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        for grouping in (0, 1):
            groups = pl.wave_groups(waves, grouping)
            assert groups == ref.wave_groups(waves, grouping)
            assert [f for f, _ in groups] == [sum(c for _, c in groups[:i]) for i in range(len(groups))] and sum(c for _, c in groups) == waves
            for first, count in groups:
                assert first % count == 0                    # an aligned block of buckets: its keys share the top bits - log2(count) bits
    # 8 ranks on the top 6 bits: groups of up to 4 buckets still need 7 pass units of a 32-bit key, like a single bucket
    assert [pl.group_pass_units(32, 6, g) for g in (1, 2, 4)] == [7, 7, 7] and pl.group_pass_units(32, 6, 8) == 8
    for kb in (32, 64):
        for bits in range(1, 9):
            for g in (1, 2, 4, 8, 16):
                assert pl.group_pass_units(kb, bits, g) == ref.group_pass_units(kb, bits, g) == -(-(kb - bits + (g - 1).bit_length()) // 4)


def test_exchange_plans_match_the_reference_statement():
    pl = _planner()
    rnd = random.Random(12)
    tables = [(w, nb, _random_table(rnd, w, nb)) for _ in range(40) for w in (1, 2, 3, 4, 5, 8) for nb in (3, 7, 15, 16)]
    tables += [(w, nb, t) for w, nb, t in _dataset_tables() if nb == 16]
    for world, nb, table in tables:
        totals = [sum(row[b] for row in table) for b in range(nb)]
        assert pl.balanced_owner(totals, world) == ref.balanced_owner(totals, world)
        assert pl.split_cuts(totals, world) == ref.split_cuts(totals, world)
        for rank in range(world):
            for mine, theirs in ((pl.plan_from_table, ref.plan_from_table), (pl.split_plan, ref.split_plan)):
                got, gi = mine(table, rank, world)
                want, wi = theirs(table, rank, world)
                assert (got.send, got.recv, got.loads) == (want.send, want.recv, want.loads)
                assert gi == pytest.approx(wi, rel=1e-12)


def test_splitters_range_buckets_and_capacity_match():
    pl = _planner()
    rnd = random.Random(13)
    for _ in range(200):
        world = rnd.randint(1, 8)
        samples = [sorted(rnd.choice([rnd.randrange(1 << 64), rnd.randrange(100), 7]) for _ in range(rnd.choice([0, 1, 5, 64]))) for _ in range(world)]
        sizes = [rnd.choice([0, 10, 1000, 1 << 27]) for _ in range(world)]
        assert pl.choose_splitters(samples, sizes, world) == ref.choose_splitters(samples, sizes, world)
    for lo, hi, bits in [(0, 15, 32), (7, 7, 64), (0, 999, 32), (123, 123 + 2**31, 32), (0, 2**32 - 1, 32), (0, 2**64 - 1, 64), (5, 5 + 2**40, 64), (0, 16, 32)]:
        assert pl.range_buckets(lo, hi, bits) == ref.range_buckets(lo, hi, bits)
    for loads, caps, need_out, slack in [([5, 5], [(5, 5), (5, 5)], True, 0), ([5, 6], [(5, 5), (5, 9)], False, 0), ([5, 5], [(9, 4), (9, 9)], True, 0),
                                         ([5, 5], [(9, 4), (9, 9)], False, 0), ([5, 5], [(8, 9), (9, 9)], True, 4)]:
        outcomes = []
        for mod in (pl, ref):
            try:
                mod.check_capacity(loads, caps, need_out, slack)
                outcomes.append(None)
            except (pl.CapacityError, ref.CapacityError) as exc:
                outcomes.append(str(exc))
        assert outcomes[0] == outcomes[1]


def test_peer_access_decision():
    """How a rank reaches every other rank's receive buffer, from what everybody published about itself (host, process token, pid,
    device): threads of one process on one device share pointers, threads on ANOTHER device need rsx_peer_enable first, other
    processes of the host open an IPC handle, equal pids with different tokens are different processes, another host is refused."""
    pl = _planner()
    ids = [(77, 1000, 4242, 0), (77, 1000, 4242, 0), (77, 1000, 4242, 3), (77, 2000, 4242, 0), (77, 3000, 5151, 1)]
    assert pl.peer_access(ids, 0) == [pl.PEER_SELF, pl.PEER_SAME_POINTER, pl.PEER_ENABLE_THEN_POINTER, pl.PEER_OPEN_IPC, pl.PEER_OPEN_IPC]
    assert pl.peer_access(ids, 2) == [pl.PEER_ENABLE_THEN_POINTER, pl.PEER_ENABLE_THEN_POINTER, pl.PEER_SELF, pl.PEER_OPEN_IPC, pl.PEER_OPEN_IPC]
    assert pl.peer_access(ids, 3) == [pl.PEER_OPEN_IPC, pl.PEER_OPEN_IPC, pl.PEER_OPEN_IPC, pl.PEER_SELF, pl.PEER_OPEN_IPC]
    rnd = random.Random(14)
    for _ in range(100):
        world = rnd.randint(1, 8)
        ids = [(5, rnd.choice([1, 2]), rnd.choice([10, 11]), rnd.choice([0, 1])) for _ in range(world)]
        for r in range(world):
            assert pl.peer_access(ids, r) == ref.peer_access(ids, r)
    with pytest.raises(RuntimeError):
        pl.peer_access([(1, 1, 1, 0), (2, 1, 1, 0)], 0)            # a rank on another host
    # huge unsigned values survive the int64 transport
    assert pl.peer_access([((1 << 62) - 1, (1 << 62) - 3, 1, 0), ((1 << 62) - 1, (1 << 62) - 3, 1, 1)], 1) == [pl.PEER_ENABLE_THEN_POINTER, pl.PEER_SELF]


def test_process_identity_is_stable_and_fits_int64():
    import __graft_entry__ as entry
    entry.load_package()
    from radix_sort_amd.distributed import default_partition_bits, group_pass_units, process_identity
    a, b = process_identity(0), process_identity(3)
    assert a[:3] == b[:3] and (a[3], b[3]) == (0, 3) and a[2] == os.getpid()
    assert all(0 <= v < (1 << 63) for v in a)
    assert [default_partition_bits(w) for w in (1, 2, 4, 8, 16)] == [4, 4, 5, 6, 7]
    assert [group_pass_units(32, b, 1) for b in (1, 4, 5, 6, 8)] == [8, 7, 7, 7, 6] and group_pass_units(64, 6, 1) == 15
