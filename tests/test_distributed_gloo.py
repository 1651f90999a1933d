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


def _worker(rank, world, port, dtype, kind, with_payload, n_per_rank, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = _dist_module()
        from _oracle import Oracle
        orc = Oracle()
        full = orc.dataset(kind, dtype, n_per_rank * world, seed=77)
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
        sorter = d.ShardedSorter(eng, rank, world, np.dtype(dtype).itemsize * 8, dist)
        n_local = sorter.sort(t_keys, staging, recv, pay, spay, rpay)
        q.put((rank, n_local, eng.result, eng.result_payload))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,kind,with_payload", [
    ("uint32", "SeededUniform", False),
    ("int32", "SeededUniform", True),
    ("uint64", "SeededUniform", True),
    ("int64", "Random", False),       # all keys are small non-negative: lands on few ranks
    ("uint32", "Zeros", True),        # every key equal: nothing moves
    ("int32", "Range", True),         # small range at the bottom of the key space: ranged buckets balance it
    ("uint64", "InvertedRange", False),
])
def test_sharded_sort_world2(dtype, kind, with_payload):
    import torch.multiprocessing as mp
    from _oracle import Oracle
    world, n_per_rank = 2, 3000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype, kind, with_payload, n_per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc = Oracle()
    full = orc.dataset(kind, dtype, n_per_rank * world, seed=77)
    got = np.concatenate([o[2] for o in outs])
    assert sum(o[1] for o in outs) == full.size
    assert np.array_equal(got, np.sort(full, kind="stable"))
    if kind in ("Range", "InvertedRange", "SeededUniform"):
        assert max(o[1] for o in outs) <= 0.6 * full.size        # ranged buckets keep the ranks balanced
    if with_payload:
        got_p = np.concatenate([o[3] for o in outs])
        assert np.array_equal(got_p, np.argsort(full, kind="stable").astype(np.uint32))   # global stable argsort
