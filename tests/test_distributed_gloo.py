"""world_size-2 tests of the sharded-sort host logic on CPU (gloo).

The device work is done by a TEST DOUBLE defined here (numpy + the oracle as checker's
stand-in); the product's default engine is the HIP one and has no CPU path.  What is
under test is radix-sort_amd/distributed.py: bucket ownership, split sizes, the
all_gather of counts, the all_to_all plan, and that rank-order concatenation is sorted
(and, with payloads, the stable argsort)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _dist_module():
    import __graft_entry__ as entry
    entry.load_package()
    from radix_sort_amd import distributed
    return distributed


def test_bucket_owner_and_splits():
    d = _dist_module()
    assert d.bucket_owner(8) == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    assert d.bucket_owner(2) == [0] * 8 + [1] * 8
    assert d.bucket_owner(1) == [0] * 16
    for w in range(1, 17):
        own = d.bucket_owner(w)
        assert own == sorted(own) and set(own) == set(range(w))      # monotone, every rank owns something
    offs = list(range(0, 17 * 10, 10))                                # 10 keys per bucket
    assert d.send_splits(offs, 4) == [40, 40, 40, 40]
    assert d.send_splits(offs, 3) == [60, 50, 50]
    assert d.recv_splits([[1, 2], [3, 4]], 0) == [1, 3] and d.recv_splits([[1, 2], [3, 4]], 1) == [2, 4]
    assert d.range_buckets(0, 15, 32) == (0, 0) and d.range_buckets(7, 7, 64) == (0, 0)
    for lo, hi, bits in [(0, 999, 32), (123, 123 + 2**31, 32), (0, 2**32 - 1, 32), (0, 2**64 - 1, 64), (5, 5 + 2**40, 64), (0, 16, 32)]:
        shift, mul = d.range_buckets(lo, hi, bits)
        assert shift == 0 and 0 < mul < 2**bits
        assert ((hi - lo) * mul) >> bits == 15 or hi - lo < 32          # the top key lands in the last bucket
        assert [((v - lo) * mul) >> bits for v in (lo, hi)] == sorted([((v - lo) * mul) >> bits for v in (lo, hi)])
    assert d.balanced_owner([10] * 16, 8) == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    own = d.balanced_owner([100] + [0] * 15, 4)
    assert own == sorted(own) and 0 <= own[0] < 4               # one hot bucket: any single owner, still monotone
    own = d.balanced_owner([5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 0, 0, 0, 0], 2)        # 12 used buckets
    assert own == sorted(own) and own.count(0) == 6
    own = d.balanced_owner([0] * 16, 3)
    assert own == sorted(own)
    with pytest.raises(ValueError):
        d.bucket_owner(17)


class _CpuEngineDouble:
    """Test double with the three engine methods ShardedSorter calls; operates on CPU torch
    tensors through their data_ptr()."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        self.result = None
        self.result_payload = None

    def _view(self, ptr, n, dtype):
        import ctypes as C
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n)

    def partition(self, d_keys, n, shift, bits, d_keys_out, d_payload=None, d_payload_out=None):
        keys = self._view(d_keys, n, self.dtype)
        u = keys.view(np.uint32 if self.dtype.itemsize == 4 else np.uint64)
        if self.dtype.kind == "i":
            u = u ^ u.dtype.type(1 << (self.dtype.itemsize * 8 - 1))
        d = ((u >> u.dtype.type(shift)) & u.dtype.type((1 << bits) - 1)).astype(np.int64)
        order = np.argsort(d, kind="stable")
        self._view(d_keys_out, n, self.dtype)[:] = keys[order]
        if d_payload:
            self._view(d_payload_out, n, np.uint32)[:] = self._view(d_payload, n, np.uint32)[order]
        counts = np.bincount(d, minlength=1 << bits)
        return [0] + [int(v) for v in np.cumsum(counts)]

    def _biased(self, keys):
        u = keys.view(np.uint32 if self.dtype.itemsize == 4 else np.uint64)
        if self.dtype.kind == "i":
            u = u ^ u.dtype.type(1 << (self.dtype.itemsize * 8 - 1))
        return u

    def partition_count(self, d_keys, n, shift, bits):
        u = self._biased(self._view(d_keys, n, self.dtype))
        d = ((u >> u.dtype.type(shift)) & u.dtype.type((1 << bits) - 1)).astype(np.int64)
        self._counted = (d_keys, n, shift, bits)
        return [int(v) for v in np.bincount(d, minlength=1 << bits)]

    def partition_scatter(self, d_keys, n, shift, bits, d_keys_out, d_payload=None, d_payload_out=None):
        assert self._counted == (d_keys, n, shift, bits)
        self.partition(d_keys, n, shift, bits, d_keys_out, d_payload, d_payload_out)

    def key_range(self, d_keys, n):
        if n == 0:
            return (1 << 64) - 1, 0
        u = self._biased(self._view(d_keys, n, self.dtype))
        return int(u.min()), int(u.max())

    def partition_range(self, d_keys, n, lo, shift, mul, d_keys_out, d_payload=None, d_payload_out=None):
        keys = self._view(d_keys, n, self.dtype)
        bits = self.dtype.itemsize * 8
        x = [int(v) - lo for v in self._biased(keys)]
        d = np.array([min(((v * mul) >> bits) if mul else (v >> shift), 15) for v in x], dtype=np.int64)
        order = np.argsort(d, kind="stable")
        self._view(d_keys_out, n, self.dtype)[:] = keys[order]
        if d_payload:
            self._view(d_payload_out, n, np.uint32)[:] = self._view(d_payload, n, np.uint32)[order]
        return [0] + [int(v) for v in np.cumsum(np.bincount(d, minlength=16))]

    def msd_count(self, d_keys, n, bits, world, d_counts):
        """rsx_msd_count: the 2^bits bucket sizes of the top `bits` bits into the caller's row (256 slots, natural order)."""
        u = self._biased(self._view(d_keys, n, self.dtype))
        kb = self.dtype.itemsize * 8
        top = (u >> u.dtype.type(kb - 8)).astype(np.int64)
        sub_shift = 8 - bits
        c, sub = top >> sub_shift, top & ((1 << sub_shift) - 1)
        k = (1 << bits) // world
        self._msd = (d_keys, n, (((c % k) * world + c // k) << sub_shift) | sub)          # wave-major position of every key
        self._view(d_counts, 256, np.int64)[:] = np.bincount(c, minlength=256)

    def msd_scatter(self, d_keys, n, d_staging, d_payload=None, d_staging_payload=None):
        assert self._msd[:2] == (d_keys, n)
        keys = self._view(d_keys, n, self.dtype)
        order = np.argsort(self._msd[2], kind="stable")
        self._view(d_staging, n, self.dtype)[:] = keys[order]
        if d_payload:
            self._view(d_staging_payload, n, np.uint32)[:] = self._view(d_payload, n, np.uint32)[order]

    def sort_from_to(self, d_keys, n, first_pass, last_pass, d_keys_out, d_payload=None, d_payload_out=None):
        keys = self._view(d_keys, n, self.dtype).copy()
        u = self._biased(keys)
        low = u & u.dtype.type((1 << (4 * last_pass)) - 1)          # only the passes asked for
        order = np.argsort(low, kind="stable")
        self._view(d_keys_out, n, self.dtype)[:] = keys[order]
        if d_payload:
            self._view(d_payload_out, n, np.uint32)[:] = self._view(d_payload, n, np.uint32).copy()[order]

    def sample_keys(self, d_keys, n, count):
        u = self._biased(self._view(d_keys, n, self.dtype))
        return [int(u[(2 * i + 1) * n // (2 * count)]) for i in range(count)]

    @staticmethod
    def _split_buckets(u, splitters):
        sp = np.array(splitters, dtype=u.dtype)
        return (np.searchsorted(sp, u, side="left") + np.searchsorted(sp, u, side="right")).astype(np.int64)

    def partition_count_split(self, d_keys, n, splitters):
        assert 1 <= len(splitters) <= 7 and splitters == sorted(set(splitters))
        self._split = (d_keys, n, list(splitters))
        d = self._split_buckets(self._biased(self._view(d_keys, n, self.dtype)), splitters)
        return [int(v) for v in np.bincount(d, minlength=2 * len(splitters) + 1)]

    def partition_scatter_split(self, d_keys, n, d_keys_out, d_payload=None, d_payload_out=None):
        assert self._split[:2] == (d_keys, n)
        keys = self._view(d_keys, n, self.dtype)
        order = np.argsort(self._split_buckets(self._biased(keys), self._split[2]), kind="stable")
        self._view(d_keys_out, n, self.dtype)[:] = keys[order]
        if d_payload:
            self._view(d_payload_out, n, np.uint32)[:] = self._view(d_payload, n, np.uint32)[order]

    def sort_from(self, d_keys, n, d_payload=None):
        keys = self._view(d_keys, n, self.dtype).copy()
        order = np.argsort(keys, kind="stable")
        self.result = keys[order]
        if d_payload:
            self.result_payload = self._view(d_payload, n, np.uint32).copy()[order]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_full(kind, dtype, n, orc):
    """Reference dataset kinds plus the distributions that stress the exchange plan."""
    if kind == "HeavyTies":            # 80 % one value in the middle of the key range, rest uniform
        x = orc.dataset("SeededUniform", dtype, n, seed=77)
        rng = np.random.default_rng(5)
        x[rng.random(n) < 0.8] = x.dtype.type(12345)
        return x
    if kind == "FewValues":            # three distinct keys
        rng = np.random.default_rng(6)
        return np.array([3, 70000, 9], dtype=dtype)[rng.integers(0, 3, n)]
    if kind == "Skewed":               # exponential magnitudes: equal-width buckets put ~all keys in bucket 0
        rng = np.random.default_rng(8)
        bits = np.dtype(dtype).itemsize * 8 - 1
        return (2.0 ** (rng.random(n) * bits)).astype(np.uint64).astype(dtype)
    return orc.dataset(kind, dtype, n, seed=77)


def _worker(rank, world, port, dtype, kind, with_payload, n_per_rank, q, strategy="auto", partition_bits=None, grouping="doubling"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = _dist_module()
        from _oracle import Oracle
        orc = Oracle()
        full = _make_full(kind, dtype, n_per_rank * world, orc)
        shard = full[rank * n_per_rank:(rank + 1) * n_per_rank].copy()
        signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dtype).name)
        t_keys = torch.from_numpy(shard.view(signed) if signed else shard)
        staging = torch.empty_like(t_keys)
        recv = torch.empty(n_per_rank * world, dtype=t_keys.dtype)
        pay = spay = rpay = None
        if with_payload:
            pay = torch.arange(rank * n_per_rank, (rank + 1) * n_per_rank, dtype=torch.int32)
            spay = torch.empty_like(pay)
            rpay = torch.empty(n_per_rank * world, dtype=torch.int32)
        eng = _CpuEngineDouble(dtype)
        sorter = d.ShardedSorter(eng, rank, world, np.dtype(dtype).itemsize * 8, dist, strategy=strategy, partition_bits=partition_bits, wave_grouping=grouping)
        out = opay = None
        if strategy in ("auto", "waves"):
            out = torch.empty_like(recv)
            opay = torch.empty_like(rpay) if with_payload else None
        n_local = sorter.sort(t_keys, staging, recv, pay, spay, rpay, out, opay)
        if sorter.result_in_out:
            res = out[:n_local].numpy().view(np.dtype(dtype)).copy()
            res_p = opay[:n_local].numpy().view(np.uint32).copy() if with_payload else None
        else:
            res, res_p = eng.result, eng.result_payload
        q.put((rank, n_local, res, res_p, sorter.last_path))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,kind,with_payload,strategy", [
    ("uint32", "SeededUniform", False, "auto"),
    ("int32", "SeededUniform", True, "auto"),
    ("int64", "SeededUniform", True, "waves"),
    ("uint32", "SeededUniform", True, "top"),
    ("uint64", "SeededUniform", True, "split"),
    ("int64", "Random", False, "auto"),       # all keys are small non-negative: top bits put them on one rank
    ("uint32", "Zeros", True, "auto"),        # every key equal: the tie bucket is cut at the shard boundary
    ("uint32", "Zeros", False, "range"),      # ... and the range path's all-equal shortcut
    ("int32", "Range", True, "auto"),         # small range at the bottom of the key space
    ("int32", "Range", True, "range"),
    ("uint64", "InvertedRange", False, "auto"),
    ("uint64", "InvertedRange", False, "range"),
    ("uint32", "HeavyTies", True, "auto"),    # 80 % one value: only cutting the tie bucket balances this
    ("int64", "FewValues", True, "auto"),
    ("uint32", "Skewed", False, "auto"),
    ("int64", "Skewed", True, "auto"),
])
def test_sharded_sort_world2(dtype, kind, with_payload, strategy):
    _run_world(2, dtype, kind, with_payload, strategy)


@pytest.mark.parametrize("dtype,kind,with_payload,strategy", [
    ("uint32", "SeededUniform", True, "auto"),     # four waves of one bucket per rank, pipelined
    ("int64", "HeavyTies", True, "auto"),          # three splitters, the tie bucket cut three times
])
def test_sharded_sort_world4(dtype, kind, with_payload, strategy):
    _run_world(4, dtype, kind, with_payload, strategy)


@pytest.mark.parametrize("world,dtype,with_payload,bits,grouping", [(2, "uint32", True, 1, "doubling"), (2, "int64", False, 6, "doubling"), (2, "uint64", True, 8, "single"),
                                                                    (4, "int32", True, 2, "single"), (4, "uint32", False, 8, "doubling"), (2, "int32", True, 5, "single")])
def test_sharded_sort_partition_bits(world, dtype, with_payload, bits, grouping):
    """The pipeline depth is a parameter: 2^bits / world waves per rank, sorted one by one or in doubling groups {0} {1} {2,3} {4..7} over the
    bits the group's keys do not share."""
    _run_world(world, dtype, "SeededUniform", with_payload, "waves", partition_bits=bits, grouping=grouping)


def _run_world(world, dtype, kind, with_payload, strategy, partition_bits=None, grouping="doubling"):
    import torch.multiprocessing as mp
    from _oracle import Oracle
    n_per_rank = 3000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype, kind, with_payload, n_per_rank, q, strategy, partition_bits, grouping)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc = Oracle()
    full = _make_full(kind, dtype, n_per_rank * world, orc)
    got = np.concatenate([o[2] for o in outs])
    assert sum(o[1] for o in outs) == full.size
    assert np.array_equal(got, np.sort(full, kind="stable"))
    assert len({o[4] for o in outs}) == 1                          # all ranks took the same path
    if strategy == "auto":
        assert outs[0][4] == ("waves" if kind == "SeededUniform" else "split")
    if kind in ("Range", "InvertedRange", "SeededUniform") or strategy != "range":
        assert max(o[1] for o in outs) <= 1.2 * full.size / world        # the ranks stay balanced
    if with_payload:
        got_p = np.concatenate([o[3] for o in outs])
        assert np.array_equal(got_p, np.argsort(full, kind="stable").astype(np.uint32))   # global stable argsort


def test_choose_splitters_and_split_plan():
    d = _dist_module()
    # quantiles of two equally weighted sample sets
    sp = d.choose_splitters([[1, 2, 3, 4], [5, 6, 7, 8]], [100, 100], 4)
    assert sp == [2, 4, 6]
    # weights: rank 1 holds 9x the keys, so the median sits inside its samples
    assert d.choose_splitters([[1, 2, 3, 4], [5, 6, 7, 8]], [10, 90], 2) == [6]
    # a dominant value collapses the quantiles into one splitter; empty ranks are skipped
    assert d.choose_splitters([[7] * 8, []], [50, 0], 8) == [7]
    assert d.choose_splitters([[], []], [0, 0], 2) == []
    assert len(d.choose_splitters([list(range(1000))], [1000], 8)) == 7

    # cuts: inside an odd bucket exact, inside an even bucket snapped to the nearer end
    assert d.split_cuts([0, 100, 0], 4) == [0, 25, 50, 75, 100]
    assert d.split_cuts([40, 0, 60], 2) == [0, 40, 100]
    assert d.split_cuts([10, 80, 10], 2) == [0, 50, 100]
    assert d.split_cuts([0, 0, 0], 3) == [0, 0, 0, 0]

    # all keys equal to the splitter, two ranks with 60 / 40 keys: rank 0 keeps 50, sends 10
    plan0, imb = d.split_plan([[0, 60, 0], [0, 40, 0]], 0, 2)
    plan1, _ = d.split_plan([[0, 60, 0], [0, 40, 0]], 1, 2)
    assert plan0.send == [50, 10] and plan0.recv == [50, 0]
    assert plan1.send == [0, 40] and plan1.recv == [10, 40] and imb == 1.0

    # every plan conserves keys and is consistent between senders and receivers
    import random
    rnd = random.Random(3)
    for _ in range(200):
        world = rnd.randint(1, 8)
        m = rnd.randint(1, 7)
        table = [[rnd.choice([0, 0, rnd.randint(0, 50), rnd.randint(0, 5000)]) for _ in range(2 * m + 1)] for _ in range(world)]
        plans = [d.split_plan(table, r, world)[0] for r in range(world)]
        for r in range(world):
            assert sum(plans[r].send) == sum(table[r])
            assert plans[r].recv == [plans[s].send[r] for s in range(world)]
        cuts = d.split_cuts([sum(row[b] for row in table) for b in range(2 * m + 1)], world)
        assert cuts == sorted(cuts) and [p.n_recv for p in plans] == [cuts[i + 1] - cuts[i] for i in range(world)]


def _overflow_worker(rank, world, port, strategy, q, short_payload=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = _dist_module()
        from _oracle import Oracle
        n = 4000
        full = Oracle().dataset("SeededUniform", "uint32", n * world, seed=3)
        keys = torch.from_numpy(full[rank * n:(rank + 1) * n].copy().view(np.int32))
        staging = torch.empty_like(keys)
        # rank 1's receive buffer is far too small; rank 0's is generous.  short_payload: the KEY buffers are generous everywhere
        # and only rank 1's payload receive buffer is short — the capacity that travels in the all_gather is the smaller of the two
        small = rank == 1
        recv = torch.empty(n // 4 if (small and not short_payload) else n * world, dtype=keys.dtype)
        out = torch.empty(n * world, dtype=keys.dtype)
        pay = spay = rpay = opay = None
        if short_payload:
            pay = torch.arange(n, dtype=torch.int32)
            spay = torch.empty_like(pay)
            rpay = torch.empty(n // 4 if small else n * world, dtype=torch.int32)
            opay = torch.empty(n * world, dtype=torch.int32)
        sorter = d.ShardedSorter(_CpuEngineDouble("uint32"), rank, world, 32, dist, strategy=strategy)
        try:
            sorter.sort(keys, staging, recv, pay, spay, rpay, out, opay)
            q.put((rank, "no error"))
        except d.CapacityError as exc:
            q.put((rank, "capacity:" + str(exc)))
        dist.barrier()          # both ranks are still in step: nobody is stuck inside an all_to_all
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("strategy,short_payload", [("auto", False), ("waves", False), ("top", False), ("split", False), ("range", False),
                                                    ("auto", True), ("top", True), ("split", True)])
def test_too_small_receive_buffer_raises_on_every_rank(strategy, short_payload):
    """One rank's receive buffer (or, with a payload, only its PAYLOAD receive buffer) cannot hold its share: EVERY rank
    must raise before any all_to_all is issued (a lone raise would leave the peers hanging in the collective)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, strategy, q, short_payload)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1].startswith("capacity:rank 1 would receive") for o in outs), outs


class _FlaggedEngine(_CpuEngineDouble):
    """The double with rsx_check_status: reports (once) that a fused table scan of an earlier step timed out."""

    def __init__(self, dtype, flagged):
        super().__init__(dtype)
        self.flagged = flagged

    def check_status(self):
        if self.flagged:
            self.flagged = False
            raise RuntimeError("rsx_check_status: OperationStatus::CALCULATION_FAILED (a fused table scan timed out)")


def _status_worker(rank, world, port, strategy, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = _dist_module()
        from _oracle import Oracle
        n = 4000
        kind = "SeededUniform" if strategy != "range" else "Range"
        full = Oracle().dataset(kind, "uint32", n * world, seed=3)
        keys = torch.from_numpy(full[rank * n:(rank + 1) * n].copy().view(np.int32))
        staging, recv, out = torch.empty_like(keys), torch.empty(n * world, dtype=keys.dtype), torch.empty(n * world, dtype=keys.dtype)
        eng = _FlaggedEngine("uint32", flagged=(rank == 1))          # only rank 1's engine has something to report
        sorter = d.ShardedSorter(eng, rank, world, 32, dist, strategy=strategy)
        outcomes = []
        for _ in range(2):                                           # the second step runs: the flag was reported once and is gone
            try:
                sorter.sort(keys, staging, recv, None, None, None, out, None)
                outcomes.append("sorted")
            except d.EngineStatusError as exc:
                outcomes.append("status:" + str(exc))
            dist.barrier()      # both ranks are still in step
        q.put((rank, outcomes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("strategy", ["auto", "waves", "top", "split", "range"])
def test_engine_status_of_one_rank_raises_on_every_rank(strategy):
    """One rank's engine reports a timed-out table scan of an earlier step: the flag rides in the row every path gathers anyway, and
    EVERY rank raises in that step — a lone raise would leave the peers hanging in the next collective."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_status_worker, args=(r, 2, port, strategy, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, outcomes in outs:
        assert outcomes[0].startswith("status:rank(s) [1] reported an engine error") and outcomes[1] == "sorted", outs


def test_sorter_constructor_validates():
    d = _dist_module()
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 0, 2, 32, dist=None)
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 2, 2, 32, dist=object())
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 0, 1, 32, strategy="nope")
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 0, 8, 32, dist=object(), partition_bits=2)      # 4 buckets do not feed 8 ranks
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 0, 2, 32, dist=object(), partition_bits=9)
    with pytest.raises(ValueError):
        d.ShardedSorter(_CpuEngineDouble("uint32"), 0, 2, 32, dist=object(), wave_grouping="triples")
