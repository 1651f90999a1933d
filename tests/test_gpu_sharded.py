"""The sharded-sort driver with the REAL HIP engine, two ranks emulated as two threads on
one GPU (the test box has a single MI355X; RCCL refuses two ranks on one device).  The
collectives are a loopback object with torch.distributed's call signatures; everything
else — rsx_partition, split planning, rsx_sort_from on the received keys — is the product
path.  The 8-GPU RCCL run itself is the driver's scaling bench."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Loopback:
    """torch.distributed's call signatures for N ranks = N threads on ONE GPU, with RCCL's stream
    semantics and NO device-wide synchronisation: a collective runs on the rank's own communication
    stream, starts after the `ready` events every participant recorded on its CURRENT stream at call
    time, and a rank's current stream learns of its completion only through `work.wait()` (or, for a
    blocking call, before the call returns).  A sorter that forgot a wait, or whose engine ran on a
    stream other than torch's current one, therefore races here exactly as it would on 8 GPUs."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.done = [None] * world

    def view(self, rank):
        return _RankDist(self, rank)


class _Work:
    def __init__(self, events):
        self.events = events

    def wait(self):
        import torch
        cur = torch.cuda.current_stream()
        for ev in self.events:
            cur.wait_event(ev)          # stream-ordered, the host does not block
        return True


class _RankDist:
    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank
        self._comm = None

    def _comm_stream(self):
        import torch
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        return self._comm

    def _collective(self, payload, body, async_op):
        import torch
        hub = self.hub
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        hub.slots[self.rank] = (payload, ready)
        hub.barrier.wait()
        comm = self._comm_stream()
        with torch.cuda.stream(comm):
            for src in range(hub.world):
                comm.wait_event(hub.slots[src][1])
            body([hub.slots[src][0] for src in range(hub.world)])
            done = torch.cuda.Event()
            done.record(comm)
        hub.done[self.rank] = done
        hub.barrier.wait()
        # a rank's part of the collective is over when everybody has pulled its data, too
        work = _Work(list(hub.done))
        hub.barrier.wait()              # slots may be reused by the next call
        if async_op:
            return work
        work.wait()
        return None

    def all_gather_into_tensor(self, out, t, async_op=False):
        import torch
        return self._collective(t, lambda parts: out.copy_(torch.cat(parts)), async_op)

    def all_reduce(self, t, op=None, async_op=False):
        # used as a stream-ordered barrier only (the fence that closes a wave of the peer-store exchange): the value is not looked at
        return self._collective(t, lambda parts: None, async_op)

    def barrier(self):
        self.hub.barrier.wait()

    def all_to_all_single(self, out, inp, out_splits, in_splits, async_op=False):
        def body(parts):
            pos = 0
            for src, (s_inp, s_splits) in enumerate(parts):
                off, cnt = sum(s_splits[:self.rank]), s_splits[self.rank]
                assert cnt == out_splits[src]
                out[pos:pos + cnt].copy_(s_inp[off:off + cnt])
                pos += cnt
        return self._collective((inp, in_splits), body, async_op)


def _make_full(kind, dtype, n, oracle):
    if kind == "HeavyTies":            # 80 % one value, the rest uniform: only cutting the tie bucket balances it
        x = oracle.dataset("SeededUniform", dtype, n, seed=31)
        x[np.random.default_rng(5).random(n) < 0.8] = x.dtype.type(12345)
        return x
    if kind == "Skewed":               # exponential magnitudes
        bits = np.dtype(dtype).itemsize * 8 - 1
        return (2.0 ** (np.random.default_rng(8).random(n) * bits)).astype(np.uint64).astype(dtype)
    return oracle.dataset(kind, dtype, n, seed=31)


@pytest.mark.parametrize("dtype,kind,with_payload,world,strategy", [
    ("uint32", "SeededUniform", False, 2, "auto"), ("int64", "SeededUniform", True, 2, "auto"), ("uint32", "Zeros", True, 2, "auto"),
    ("int32", "Range", True, 2, "auto"), ("uint64", "InvertedRange", False, 2, "auto"),
    ("int32", "Range", True, 2, "range"), ("uint64", "InvertedRange", False, 2, "range"), ("uint32", "Zeros", True, 2, "range"),
    ("uint32", "HeavyTies", True, 4, "auto"), ("int64", "HeavyTies", True, 3, "auto"), ("uint64", "Skewed", True, 4, "auto"),
    ("int32", "Skewed", False, 4, "auto"), ("uint32", "SeededUniform", True, 4, "split"), ("uint32", "Random", False, 4, "auto"),
    ("uint64", "SeededUniform", True, 4, "waves"), ("int32", "SeededUniform", True, 2, "top"), ("uint32", "SeededUniform", False, 3, "auto"),
    # the driver's scaling run is 8 ranks: two pipelined waves; seven splitters when the top bits do not balance
    ("uint32", "SeededUniform", False, 8, "auto"), ("int64", "SeededUniform", True, 8, "auto"), ("uint32", "HeavyTies", True, 8, "auto"),
    ("uint64", "Skewed", False, 8, "auto"), ("int32", "Range", True, 8, "auto"), ("uint32", "Zeros", True, 8, "auto"),
    # the pipeline depth is a parameter: 2^bits / world waves per rank ("waves:bits")
    ("uint32", "SeededUniform", True, 8, "waves:3"), ("uint32", "SeededUniform", False, 8, "waves:4"), ("int64", "SeededUniform", True, 8, "waves:8"),
    ("uint64", "SeededUniform", True, 2, "waves:1"), ("int32", "SeededUniform", False, 16, "waves:8"), ("uint32", "Random", True, 4, "waves:7"),
    ("uint32", "SeededUniform", True, 8, "waves:6:single"), ("int64", "SeededUniform", False, 4, "waves:8:single")])
def test_ranks_on_one_gpu(rsx, oracle, dtype, kind, with_payload, world, strategy):
    import torch
    from radix_sort_amd.distributed import ShardedSorter
    strategy, bits, grouping = (strategy.split(":") + ["", "doubling"])[:3] if ":" in strategy else (strategy, "", "doubling")
    bits = int(bits) if bits else None
    grouping = grouping or "doubling"
    n = 100003
    full = _make_full(kind, dtype, n * world, oracle)
    hub = _Loopback(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            shard = full[rank * n:(rank + 1) * n].copy()
            signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dtype).name)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(shard.view(signed) if signed else shard).cuda()
                staging = torch.empty_like(keys)
                recv = torch.empty(n * world, dtype=keys.dtype, device="cuda")
                pay = spay = rpay = None
                if with_payload:
                    pay = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int32, device="cuda")
                    spay = torch.empty_like(pay)
                    rpay = torch.empty(n * world, dtype=torch.int32, device="cuda")
                with rsx.Engine(dtype, n * world, payload=with_payload) as eng:
                    if rank % 2 == 0:
                        eng.set_stream(stream.cuda_stream)      # odd ranks leave it to the sorter, which must bind the engine to torch's current stream itself
                    sorter = ShardedSorter(eng, rank, world, np.dtype(dtype).itemsize * 8, hub.view(rank), strategy=strategy, partition_bits=bits, wave_grouping=grouping)
                    obuf = torch.empty_like(recv)
                    opay = torch.empty_like(rpay) if with_payload else None
                    n_local = sorter.sort(keys, staging, recv, pay, spay, rpay, obuf, opay)
                    assert eng.get_stream() == stream.cuda_stream
                    if sorter.result_in_out:
                        torch.cuda.synchronize()
                        out = (obuf[:n_local].cpu().numpy().view(np.dtype(dtype)), opay[:n_local].cpu().numpy().view(np.uint32) if with_payload else None)
                    else:
                        out = eng.download(want_perm=True) if with_payload else (eng.download(), None)
                    results[rank] = (n_local, out[0], out[1], sorter.last_path)
        except Exception as exc:   # noqa: BLE001 - surface in the main thread
            errors.append(exc)
            hub.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    got = np.concatenate([r[1] for r in results])
    assert sum(r[0] for r in results) == full.size
    assert np.array_equal(got, np.sort(full, kind="stable"))
    if with_payload:
        assert np.array_equal(np.concatenate([r[2] for r in results]), np.argsort(full, kind="stable").astype(np.uint32))
    assert len({r[3] for r in results}) == 1
    if strategy == "auto":
        even = kind in ("SeededUniform", "Random")
        assert results[0][3] == (("waves" if world in (1, 2, 4, 8, 16) else "top") if even else "split")
    if strategy != "range":
        assert max(r[0] for r in results) <= 1.25 * n            # balanced whatever the distribution


@pytest.mark.parametrize("dt", ["uint32", "int32", "uint64", "int64"])
def test_sample_and_split_partition(rsx, oracle, dt):
    """rsx_sample_keys / rsx_partition_count_split / rsx_partition_scatter_split against numpy."""
    import torch
    n = 250007
    keys = oracle.dataset("SeededUniform", dt, n, seed=9)
    keys[::3] = keys[5]                                  # a heavy tie that will become a splitter
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    tk = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    out, pout = torch.empty_like(tk), torch.empty_like(pay)
    u = keys.view(np.uint32 if keys.dtype.itemsize == 4 else np.uint64)
    if keys.dtype.kind == "i":
        u = u ^ u.dtype.type(1 << (keys.dtype.itemsize * 8 - 1))
    with rsx.Engine(dt, n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        samples = e.sample_keys(tk.data_ptr(), n, 1024)
        assert len(samples) == 1024 and set(samples) <= set(int(v) for v in u)
        for i in (0, 1, 500, 1023):                          # one per stratum of n/1024 consecutive keys
            assert samples[i] in set(int(v) for v in u[i * n // 1024:(i + 1) * n // 1024])
        ordered = sorted(samples)
        for m in (1, 3, 7):
            sp = sorted({ordered[(k + 1) * 1024 // (m + 1)] for k in range(m)} | {int(u[5])})[:7]   # the tie value is a splitter
            if int(u[5]) not in sp:
                sp[-1] = int(u[5])
                sp = sorted(set(sp))
            spa = np.array(sp, dtype=u.dtype)
            d = (np.searchsorted(spa, u, side="left") + np.searchsorted(spa, u, side="right")).astype(np.int64)
            counts = e.partition_count_split(tk.data_ptr(), n, sp)
            assert counts == [int(v) for v in np.bincount(d, minlength=2 * len(sp) + 1)]
            assert counts[2 * sp.index(int(u[5])) + 1] >= n // 3
            e.partition_scatter_split(tk.data_ptr(), n, out.data_ptr(), pay.data_ptr(), pout.data_ptr())
            torch.cuda.synchronize()
            order = np.argsort(d, kind="stable")
            assert np.array_equal(out.cpu().numpy().view(keys.dtype), keys[order])
            assert np.array_equal(pout.cpu().numpy().view(np.uint32), order.astype(np.uint32))
        with pytest.raises(rsx.RadixSortError):
            e.partition_count_split(tk.data_ptr(), n, [5, 5])           # not strictly increasing
        with pytest.raises(rsx.RadixSortError):
            e.partition_count_split(tk.data_ptr(), n, list(range(8)))   # too many
        with pytest.raises(rsx.RadixSortError):
            e.partition_scatter_split(tk.data_ptr(), n, out.data_ptr(), pay.data_ptr(), pout.data_ptr())   # no count before it


@pytest.mark.parametrize("dt", ["uint32", "int32", "uint64", "int64"])
@pytest.mark.parametrize("units_short,radix_bits", [(1, 4), (2, 4), (1, 8), (2, 8)])
def test_partial_sort_into_caller_buffer(rsx, oracle, dt, units_short, radix_bits):
    """rsx_sort_from_to against numpy: the passes a wave's local sort needs (all but the top one or two 4-bit units), 4-bit and 8-bit
    chains, the last pass writing to an odd offset of a caller buffer."""
    import torch
    n = 150001
    keys = oracle.dataset("SeededUniform", dt, n, seed=units_short)
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    tk = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    bits = keys.dtype.itemsize * 8
    u = keys.view(np.uint32 if bits == 32 else np.uint64)
    if keys.dtype.kind == "i":
        u = u ^ u.dtype.type(1 << (bits - 1))
    with rsx.Engine(dt, n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        if radix_bits != 4:
            e.set_option(rsx.OPT_RADIX_BITS, radix_bits)
        units = bits // 4 - units_short
        dst, pdst = torch.zeros(n + 7, dtype=tk.dtype, device="cuda"), torch.zeros(n + 7, dtype=torch.int32, device="cuda")
        e.sort_from_to(tk.data_ptr(), n, 0, units, dst[5:].data_ptr(), pay.data_ptr(), pdst[5:].data_ptr())
        torch.cuda.synchronize()
        low = u & u.dtype.type((1 << (4 * units)) - 1)
        order = np.argsort(low, kind="stable")
        got = dst.cpu().numpy().view(keys.dtype)
        assert np.array_equal(got[5:5 + n], keys[order]) and not got[:5].any() and not got[5 + n:].any()
        assert np.array_equal(pdst.cpu().numpy().view(np.uint32)[5:5 + n], order.astype(np.uint32))
        # the options of the engine are as before: a plain sort still runs every pass
        e.sort_from(tk.data_ptr(), n, pay.data_ptr())
        assert np.array_equal(e.download(), np.sort(keys))
        with pytest.raises(rsx.RadixSortError):
            e.sort_from_to(tk.data_ptr(), n, 3, 3, dst.data_ptr(), pay.data_ptr(), pdst.data_ptr())


def test_world_size_one_is_plain_sort(rsx, oracle):
    import torch
    from radix_sort_amd.distributed import ShardedSorter
    keys_np = oracle.dataset("Random", "uint32", 50000)
    keys = torch.from_numpy(keys_np.view(np.int32)).cuda()
    with rsx.Engine("uint32", keys_np.size) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        assert ShardedSorter(eng, 0, 1, 32).sort(keys, None, None) == keys_np.size
        assert np.array_equal(eng.download(), np.sort(keys_np))


def _bench(args, env_extra=None):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    return json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_through_rccl_single_rank():
    """`python bench.py --gpus 1` with RSX_FORCE_EXCHANGE=1: the whole multi-GPU step (wave-major
    partition, all_gather of bucket counts and buffer capacities, asynchronous all_to_all_single
    with split sizes, partial local sorts) and the verification gather run through the real
    nccl/RCCL backend, one rank talking to itself."""
    line = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--log2-keys", "22", "--cpu-sample-log2", "20"], {"RSX_FORCE_EXCHANGE": "1"})
    assert line["n_gpus"] == 1 and line["value"] > 0 and "rehearsal" not in line
    assert line["config"]["verified"].startswith("bit-exact vs a host sort of all 4194304 keys")
    assert "all_to_all" in line["config"]["parallelism"] and "waves" in line["config"]["parallelism"]
    assert line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0
    assert {"count", "scatter", "plan", "wait", "local_sort"} <= set(line["sharded_phases_ms"])


@pytest.mark.parametrize("strategy,dtype,payload", [("top", "uint32", False), ("split", "int64", True), ("range", "uint64", False)])
def test_bench_through_rccl_other_exchange_paths(strategy, dtype, payload):
    line = _bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--log2-keys", "20", "--dtype", dtype, "--no-cpu-baseline"] + (["--payload"] if payload else []),
                  {"RSX_FORCE_EXCHANGE": "1", "RSX_STRATEGY": strategy})
    # one rank has no splitters to choose: the splitter path degenerates to its "all on one rank" shortcut
    path = {"top": "top", "split": "equal", "range": "range"}[strategy]
    assert f"[{path}]" in line["config"]["parallelism"] and line["config"]["verified"].startswith("bit-exact")


def test_bench_default_line_is_bit_exact_vs_the_cpu_baseline():
    """The N=1 line at a reduced size: the array the cpu_baseline leg sorted with RadixSortCPU is compared
    with the GPU output, and the line says so."""
    line = _bench(["--steps", "3", "--warmup", "1", "--log2-keys", "22", "--cpu-sample-log2", "22"])
    assert line["config"]["verified"].startswith("bit-exact vs RadixSortCPU") and "4194304 keys" in line["config"]["verified"]
    assert line["scaling"] == "weak" and line["config"]["workload"].startswith("2^22 uint32 Random")


@pytest.mark.parametrize("dt", ["uint32", "int32", "uint64", "int64"])
def test_key_range_and_ranged_partition(rsx, oracle, dt):
    import torch
    rng = np.random.default_rng(3)
    info = np.iinfo(dt)
    n = 77777
    base = int(info.min) + (int(info.max) - int(info.min)) // 3
    keys = (base + rng.integers(0, 1 << 20, size=n)).astype(dt)
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    tk = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
    out = torch.empty_like(tk)
    u = keys.view(np.uint32 if keys.dtype.itemsize == 4 else np.uint64).astype(np.uint64)
    if keys.dtype.kind == "i":
        u = u ^ np.uint64(1 << (keys.dtype.itemsize * 8 - 1))
    with rsx.Engine(dt, n) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        lo, hi = e.key_range(tk.data_ptr(), n)
        assert (lo, hi) == (int(u.min()), int(u.max()))
        from radix_sort_amd.distributed import range_buckets
        bits = keys.dtype.itemsize * 8
        shift, mul = range_buckets(lo, hi, bits)
        offs = e.partition_range(tk.data_ptr(), n, lo, shift, mul, out.data_ptr())
        torch.cuda.synchronize()
    d = np.array([min(((int(v) - lo) * mul) >> bits, 15) for v in u], dtype=np.int64)
    got = out.cpu().numpy().view(keys.dtype)
    assert np.array_equal(got, keys[np.argsort(d, kind="stable")])
    assert offs == [0] + [int(v) for v in np.cumsum(np.bincount(d, minlength=16))]
    assert min(np.diff(offs)) > 0                      # all 16 buckets used: the range is covered evenly


def test_partition_count_then_scatter(rsx, oracle):
    """The two-call bit-field partition: counts first (host decides), scatter afterwards."""
    import torch
    n = 123457
    keys = oracle.dataset("SeededUniform", "int32", n)
    tk = torch.from_numpy(keys).cuda()
    out = torch.empty_like(tk)
    u = keys.view(np.uint32) ^ np.uint32(1 << 31)
    d = (u >> np.uint32(28)).astype(np.int64)
    with rsx.Engine("int32", n) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        assert e.partition_count(tk.data_ptr(), n, 28, 4) == [int(v) for v in np.bincount(d, minlength=16)]
        e.partition_scatter(tk.data_ptr(), n, 28, 4, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), keys[np.argsort(d, kind="stable")])
        with pytest.raises(rsx.RadixSortError):          # the table was consumed
            e.partition_scatter(tk.data_ptr(), n, 28, 4, out.data_ptr())


def test_partition_refuses_a_misaligned_payload_input(rsx, oracle):
    """The partition kernels read keys AND payload 16 bytes per lane: an input payload pointer that is not 16-byte
    aligned is refused by every partition entry point instead of faulting on the device."""
    import torch
    n = 70000
    keys = oracle.dataset("SeededUniform", "uint32", n)
    tk = torch.from_numpy(keys.view(np.int32)).cuda()
    pay = torch.arange(n + 4, dtype=torch.int32, device="cuda")
    out, pout = torch.empty_like(tk), torch.empty(n, dtype=torch.int32, device="cuda")
    bad = pay[1:].data_ptr()                                  # 4 bytes past a 16-byte boundary
    with rsx.Engine("uint32", n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        with pytest.raises(rsx.RadixSortError):
            e.partition(tk.data_ptr(), n, 28, 4, out.data_ptr(), bad, pout.data_ptr())
        e.partition_count(tk.data_ptr(), n, 28, 4)
        with pytest.raises(rsx.RadixSortError):
            e.partition_scatter(tk.data_ptr(), n, 28, 4, out.data_ptr(), bad, pout.data_ptr())
        row = torch.zeros(259, dtype=torch.int64, device="cuda")
        e.msd_count(tk.data_ptr(), n, 6, 4, row.data_ptr())
        with pytest.raises(rsx.RadixSortError):
            e.msd_scatter(tk.data_ptr(), n, out.data_ptr(), bad, pout.data_ptr())
        # and the aligned call still works afterwards
        offs = e.partition(tk.data_ptr(), n, 28, 4, out.data_ptr(), pay.data_ptr(), pout.data_ptr())
        torch.cuda.synchronize()
        d = (keys >> np.uint32(28)).astype(np.int64)
        order = np.argsort(d, kind="stable")
        assert offs[-1] == n and np.array_equal(out.cpu().numpy().view(np.uint32), keys[order])
        assert np.array_equal(pout.cpu().numpy(), order.astype(np.int32))


# --------------------------------------------------------------------------- real processes, one rank each, sharing the one GPU
@pytest.mark.parametrize("ranks,extra,env", [
    (2, ["--total-log2-keys", "23"], {}),
    (4, ["--total-log2-keys", "24", "--cpu-sample-log2", "20"], {}),
    (4, ["--total-log2-keys", "22", "--dtype", "int64", "--payload", "--dataset", "Zeros", "--no-cpu-baseline"], {}),
    (3, ["--log2-keys", "20", "--dtype", "uint64", "--no-cpu-baseline"], {"RSX_STRATEGY": "split"}),
    (2, ["--total-log2-keys", "23", "--radix-bits", "8", "--payload", "--no-cpu-baseline"], {}),      # waves: 7 local passes = 3 bytes + a nibble
    (4, ["--total-log2-keys", "22", "--dtype", "int32", "--dataset", "InvertedRange", "--no-cpu-baseline", "--radix-bits", "8"], {"RSX_STRATEGY": "range"}),
])
def test_bench_ranks_as_processes_on_the_shared_gpu(ranks, extra, env):
    """`python bench.py --gpus N` as the driver runs it — bench.py starts N rank PROCESSES through torch.distributed.run,
    each with its own HIP engine, streams and ShardedSorter — except that all ranks sit on cuda:0 and the collectives
    are gloo staged through the host (tests/_host_staged_dist.py): RCCL will not put two ranks on one device.  Rank 0
    gathers every rank's output and compares the concatenation with a host sort of all the inputs."""
    line = _bench(["--gpus", str(ranks), "--steps", "2", "--warmup", "1"] + extra, dict(env, RSX_BENCH_SHARED_GPU="1"))
    assert line["n_gpus"] == ranks and line["rehearsal"] is True and line["value"] is None
    assert line["config"]["verified"].startswith("bit-exact vs a host sort of all")
    assert f"x{ranks}" in line["config"]["parallelism"]
    want_path = {"split": "split", "range": "range"}.get(env.get("RSX_STRATEGY"), "split" if "Zeros" in extra else "waves")
    assert f"[{want_path}]" in line["config"]["parallelism"], line["config"]["parallelism"]
    assert line["roofline"]["avg_launch_ms"] > 0


# --------------------------------------------------------------------------- BASELINE config 4 at its real size
def _shard(kind, dtype, offset, n, total):
    """A rank's contiguous shard of the ONE dataset (the product's own generator, host C ABI)."""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "radix-sort_amd", "host", "libradixsort_host.so"))
    lib.rsxh_dataset_fill_shard.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.rsxh_dataset_fill_shard.restype = C.c_int
    out = np.empty(n, dtype=dtype)
    kinds = {"Zeros": 0, "Range": 1, "InvertedRange": 2, "Random": 3}
    dts = {"uint32": 0, "int32": 1, "uint64": 2, "int64": 3}
    assert lib.rsxh_dataset_fill_shard(kinds[kind], dts[dtype], out.ctypes.data, offset, n, total, 0) == 0
    return out


@pytest.mark.parametrize("strategy,partition_bits,radix_bits,grouping", [("waves", 4, 4, "doubling"), ("waves", 6, 4, "doubling"), ("waves", 6, 8, "single"), ("waves-p2p", 4, 4, "single"),
                                                                          ("waves-p2p", 6, 4, "doubling"), ("waves-p2p", 7, 8, "doubling")])
def test_config4_2pow30_uint32_over_eight_ranks_bit_exact(rsx, strategy, partition_bits, radix_bits, grouping):
    """BASELINE config 4 at its real size and decomposition: 2^30 uint32 `Random` keys as eight contiguous shards of 2^27
    (rank r = draws r*2^27.. of the generator's stream), eight ranks with their own engine, streams and ShardedSorter — as
    eight THREADS of this one process on the box's one GPU (the pool's process guard admits at most 6 processes on a card,
    so eight rank processes cannot run here; four do: test_bench_config4_input_and_size_as_four_rank_processes), collectives
    = the loopback above with RCCL's stream semantics.  Both exchanges — all_to_all per wave, and peer stores (one push + fence per
    wave into the owners' receive buffers, the plan computed on the device) — at pipeline depths 2 (top 4 bits) and 8 (top 6 bits)
    waves per rank (and 16, top 7 bits, on the peer path), sorted one by one or in doubling groups {0} {1} {2,3} {4..7}, 4-bit and 8-bit local passes; the
    concatenation of the ranks' outputs must equal a host sort of all 2^30 keys, key for key."""
    import torch
    from radix_sort_amd.distributed import ShardedSorter
    world, n = 8, 1 << 27
    total = world * n
    hub = _Loopback(world)
    shards, results, errors = [None] * world, [None] * world, []
    p2p = strategy == "waves-p2p"

    def run(rank):
        try:
            shards[rank] = _shard("Random", "uint32", rank * n, n, total)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(shards[rank].view(np.int32)).cuda()
                staging = torch.empty_like(keys)
                recv = None if p2p else torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                obuf = torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                with rsx.Engine("uint32", 2 * n) as eng:
                    eng.set_stream(stream.cuda_stream)
                    if radix_bits != 4:
                        eng.set_option(rsx.OPT_RADIX_BITS, radix_bits)
                    sorter = ShardedSorter(eng, rank, world, 32, hub.view(rank), strategy=strategy, partition_bits=partition_bits, wave_grouping=grouping)
                    if p2p:
                        sorter.setup_peer_exchange(2 * n, keys.device)
                    try:
                        for _ in range(2):                       # twice: buffers, epochs and count rows are reused
                            n_local = sorter.sort(keys, staging, recv, None, None, None, obuf, None)
                        eng.sync()                               # reports a timed-out table scan, if any
                        assert sorter.result_in_out and sorter.last_path == strategy
                        results[rank] = obuf[:n_local].cpu().numpy().view(np.uint32)
                    finally:
                        if p2p:
                            sorter.close_peer_exchange()         # (barrier inside: nobody frees a buffer a peer still has mapped)
        except Exception as exc:   # noqa: BLE001 - surface in the main thread
            errors.append(exc)
            hub.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    sizes = [r.size for r in results]
    assert sum(sizes) == total and max(sizes) <= 1.25 * n
    full = np.concatenate(shards)
    del shards
    assert [int(v) for v in full[:4]] == [2421477274, 811668573, 145020712, 106868501]      # the reference's Random stream
    full.sort()
    at = 0
    for r in results:                                        # rank-order concatenation == the host sort, piece by piece
        assert np.array_equal(r, full[at:at + r.size])
        at += r.size


@pytest.mark.parametrize("extra", [[]])
def test_bench_config4_input_and_size_as_four_rank_processes(extra):
    """`python bench.py --gpus 4` with its default workload for N > 1 — BASELINE config 4's input and size: 2^30 uint32 keys,
    contiguous shards of the one `Random` stream — as the driver runs it: bench.py starts the rank PROCESSES itself (four of
    them, 2^28 keys each: the box admits at most 6 processes on its one GPU, so config 4's eight ranks run as threads in the
    test above), every rank on cuda:0 with its own engine, gloo collectives staged through the host.  Rank 0 gathers all
    outputs and compares them with a host sort of all 2^30 keys."""
    line = _bench(["--gpus", "4", "--steps", "1", "--warmup", "1"] + extra, {"RSX_BENCH_SHARED_GPU": "1"})
    assert line["n_gpus"] == 4 and line["rehearsal"] is True and line["value"] is None and line["scaling"] == "strong"
    assert line["config"]["total_keys"] == 1 << 30 and line["config"]["keys_per_gpu"] == 1 << 28
    assert line["config"]["verified"] == "bit-exact vs a host sort of all 1073741824 keys, gathered on rank 0"
    assert "[waves]" in line["config"]["parallelism"] and "x4" in line["config"]["parallelism"]
    assert "contiguous shards of one Random dataset" in line["config"]["workload"]
    if not extra:
        assert "BASELINE config 4's input and size over 4 ranks" in line["config"]["workload"]
    assert {"count", "scatter", "plan", "wait", "local_sort"} <= set(line["sharded_phases_ms"])


# --------------------------------------------------------------------------- peer-store exchange
@pytest.mark.parametrize("dtype,with_payload,world,radix_bits,bits", [("uint32", False, 8, 4, None), ("int64", True, 4, 4, 8), ("uint64", True, 2, 8, 3), ("uint32", True, 16, 4, 4),
                                                                     ("int32", True, 8, 8, 6), ("uint64", False, 1, 4, 5)])
def test_peer_store_exchange_thread_ranks(rsx, oracle, dtype, with_payload, world, radix_bits, bits):
    """strategy="waves-p2p" with ranks as threads of one process (the receive buffers are addressed by their pointers: the planner
    says "same pointer" for every peer): the plan is computed on the device, one push per wave copies the rank's segments into the
    owners' receive buffers on a second stream, one all_reduce per wave fences it, the local sorts follow wave by wave; three steps in
    a row reuse the buffers.  Rank-order concatenation = the stable sort of everything."""
    import torch
    from radix_sort_amd import planner
    from radix_sort_amd.distributed import ShardedSorter
    n = 150001
    full = oracle.dataset("SeededUniform", dtype, n * world, seed=41)
    hub = _Loopback(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            shard = full[rank * n:(rank + 1) * n].copy()
            signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dtype).name)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(shard.view(signed) if signed else shard).cuda()
                pay = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int32, device="cuda") if with_payload else None
                staging = torch.empty_like(keys)
                spay = torch.empty_like(pay) if with_payload else None
                obuf = torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                opay = torch.empty(2 * n, dtype=torch.int32, device="cuda") if with_payload else None
                with rsx.Engine(dtype, 2 * n, payload=with_payload) as eng:
                    eng.set_stream(stream.cuda_stream)
                    if radix_bits != 4:
                        eng.set_option(rsx.OPT_RADIX_BITS, radix_bits)
                    sorter = ShardedSorter(eng, rank, world, np.dtype(dtype).itemsize * 8, hub.view(rank), strategy="waves-p2p",
                                           partition_bits=bits, force_exchange=True, wave_grouping="single" if world == 4 else "doubling")
                    sorter.setup_peer_exchange(2 * n, keys.device, with_payload)
                    try:
                        assert sorter._peer["access"] == [planner.PEER_SELF if r == rank else planner.PEER_SAME_POINTER for r in range(world)]
                        sorter.record_timeline = True
                        for _ in range(3):
                            n_local = sorter.sort(keys, staging, None, pay, spay, None, obuf, opay)
                        eng.sync()
                        assert sorter.result_in_out and sorter.last_path == "waves-p2p"
                        assert set(sorter.timeline_ms()) <= {"count", "scatter", "plan", "fence", "local_sort"}
                        results[rank] = (obuf[:n_local].cpu().numpy().view(np.dtype(dtype)), opay[:n_local].cpu().numpy().view(np.uint32) if with_payload else None)
                    finally:
                        sorter.close_peer_exchange()         # (barrier inside: nobody frees a buffer a peer may still be writing into)
        except Exception as exc:   # noqa: BLE001 - surface in the main thread
            errors.append(exc)
            hub.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert np.array_equal(np.concatenate([r[0] for r in results]), np.sort(full, kind="stable"))
    if with_payload:
        assert np.array_equal(np.concatenate([r[1] for r in results]), np.argsort(full, kind="stable").astype(np.uint32))


def test_peer_store_capacity_verdict_raises_on_every_rank(rsx, oracle):
    """One rank's peer-visible receive buffer is too small: the device-side plan carries the verdict, no push writes anything, and EVERY
    rank raises CapacityError in the same step."""
    import torch
    from radix_sort_amd.distributed import CapacityError, ShardedSorter
    world, n = 4, 60000
    full = oracle.dataset("SeededUniform", "uint32", n * world, seed=5)
    hub = _Loopback(world)
    outcomes, errors = [None] * world, []

    def run(rank):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(full[rank * n:(rank + 1) * n].copy().view(np.int32)).cuda()
                staging, obuf = torch.empty_like(keys), torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                with rsx.Engine("uint32", 2 * n) as eng:
                    eng.set_stream(stream.cuda_stream)
                    sorter = ShardedSorter(eng, rank, world, 32, hub.view(rank), strategy="waves-p2p")
                    sorter.setup_peer_exchange(2 * n if rank != 2 else n // 2, keys.device)
                    try:
                        try:
                            sorter.sort(keys, staging, None, None, None, None, obuf, None)
                            outcomes[rank] = "sorted"
                        except CapacityError as exc:
                            outcomes[rank] = str(exc)
                        torch.cuda.synchronize()
                    finally:
                        sorter.close_peer_exchange()
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)
            hub.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(o is not None and o.startswith("rank 2 would receive") for o in outcomes), outcomes


@pytest.mark.parametrize("ranks,extra", [(2, ["--total-log2-keys", "23"]), (4, ["--total-log2-keys", "24", "--dtype", "uint64", "--payload", "--dataset", "RandomDistributed"])])
def test_bench_peer_store_exchange_between_rank_processes(ranks, extra):
    """RSX_STRATEGY=waves-p2p with every rank a PROCESS of its own (bench.py's launcher, all ranks on the box's one GPU): the
    receive buffers travel as IPC handles (rsx_peer_alloc / rsx_peer_open) and the scatter kernel of one process stores into
    memory another process allocated; everything gathered on rank 0 and compared with a host sort."""
    line = _bench(["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra, {"RSX_BENCH_SHARED_GPU": "1", "RSX_STRATEGY": "waves-p2p"})
    assert line["n_gpus"] == ranks and line["rehearsal"] is True
    assert line["config"]["verified"].startswith("bit-exact vs a host sort of all")
    assert "[waves-p2p]" in line["config"]["parallelism"] and "peer stores" in line["config"]["parallelism"]
    assert set(line["sharded_phases_ms"]) - {"note"} <= {"count", "scatter", "plan", "fence", "local_sort"}
