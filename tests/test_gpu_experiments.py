"""The EXPERIMENTS build of the engine (-DRSX_EXPERIMENTS, tools/_variants/libradixsort_hip_experiments.so, built by
__graft_entry__.build()): the measured-and-rejected alternatives the product library no longer contains — the 8-bit scatter
kernels 2 and 3, the staying grid, the table scan inside the reorder launch — and the test hook that raises the fused scan's
time-out word.  Same results as the product in every mode; skipped where the variant has not been built."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod(rsx):
    if not os.path.exists(rsx.EXPERIMENTS_LIB_PATH):
        pytest.skip("the experiments variant is not built (python -c 'import __graft_entry__ as g; g.build_hip_experiments()')")
    return rsx.experiments()


def test_the_product_library_does_not_know_the_experimental_options(rsx, mod):
    with rsx.Engine("uint32", 4096) as e:
        for opt in (mod.XOPT_DEBUG_RAISE_SCAN_TIMEOUT, mod.XOPT_INLINE_SCAN, mod.XOPT_INLINE_SCAN_MAX_GROUPS, mod.XOPT_REORDER8_KERNEL, mod.XOPT_REORDER8_STAY):
            with pytest.raises(rsx.RadixSortError):
                e.set_option(opt, 1)
    with mod.Engine("uint32", 4096) as e:
        e.set_option(mod.XOPT_REORDER8_KERNEL, 2)
    assert mod.load_library() is not rsx.load_library()


def test_scan_timeout_is_reported_once_and_the_engine_stays_usable(mod, oracle):
    """The word a timed-out fused scan stores to (mapped host memory) is reported by the next synchronising call — and by
    rsx_copy_result / rsx_check_status for work that has finished — exactly once; afterwards the engine sorts again."""
    import torch
    n = 100000
    keys = oracle.dataset("SeededUniform", "uint32", n, seed=4)
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    with mod.Engine("uint32", n) as e:
        e.upload(keys)
        e.sort()
        e.sync()
        for report in ("sync", "download", "copy_result", "check_status"):
            e.set_option(mod.XOPT_DEBUG_RAISE_SCAN_TIMEOUT, 1)          # the store a starved workgroup makes, on the engine's stream
            torch.cuda.synchronize()
            with pytest.raises(mod.RadixSortError) as err:
                {"sync": e.sync, "download": e.download, "copy_result": lambda: e.copy_result(dst.data_ptr()), "check_status": e.check_status}[report]()
            assert "timed out" in str(err.value)
            e.sync()                                                       # cleared: reported once
            e.check_status()
            e.upload(keys)
            e.sort()
            assert np.array_equal(e.download(), np.sort(keys))


@pytest.mark.parametrize("dt,payload,n", [("uint32", False, (1 << 22) + 4097), ("uint32", False, (1 << 24) + 4097), ("int32", True, (1 << 23) + 11),
                                          ("uint64", True, (1 << 22) + 8192), ("int64", False, 1 << 25), ("uint32", False, 1 << 26)])
def test_inline_scan_equals_the_scan_launches(mod, oracle, dt, payload, n):
    """Mid-size tables: the first workgroups of every reorder launch scan the pass's table themselves (RSX_XOPT_INLINE_SCAN, one dependent
    launch per pass) — keys, payload, final table and group sums must equal what the chain with scan launches leaves; repeated sorts
    advance the epochs and alternate the two count buffers through both roles."""
    keys = oracle.dataset("SeededUniform", dt, n, seed=n % 1000 + 7)
    keys[::5] = keys[3]
    perm = np.arange(n, dtype=np.uint32) if payload else None
    seen = []
    for inline in (1, 0):
        with mod.Engine(dt, n, payload=payload) as e:
            e.set_option(mod.XOPT_INLINE_SCAN, inline)
            e.set_option(mod.OPT_PROFILE, 2)                   # reorder launches only: how many launches a sort takes is visible in the timings
            e.upload(keys, perm)
            for _ in range(3):
                e.sort()
            e.sync()
            rt = e.timings()
            passes = keys.dtype.itemsize * 2
            assert rt.reorder.n == 3 * passes
            g = e.geometry()
            out = e.download(want_perm=payload, hist_cap=int(g.table_len), globsum_cap=int(g.num_scan_blocks))
            seen.append(out)
    assert np.array_equal(seen[0][0], np.sort(keys))
    if payload:
        assert np.array_equal(seen[0][1], np.argsort(keys, kind="stable").astype(np.uint32))
    for a, b in zip(seen[0], seen[1]):
        assert np.array_equal(a, b)


def test_inline_scan_border_and_external_buffers(mod, oracle):
    """RSX_XOPT_INLINE_SCAN_MAX_GROUPS moves the border to the chain with scan launches; the inline chain behind rsx_sort_from_to (external input
    left untouched, partial pass range, output into a caller buffer) and on ragged sizes whose XCD mapping has surplus workgroups."""
    import torch
    for n in ((1 << 22) + 1, (1 << 23) + 4095, 5 * (1 << 20) + 123):
        keys = oracle.dataset("SeededUniform", "uint32", n, seed=n % 97)
        t = torch.from_numpy(keys.view(np.int32)).cuda()
        dst = torch.zeros(n + 3, dtype=torch.int32, device="cuda")
        outs = []
        for limit in (64, 0):
            with mod.Engine("uint32", n) as e:
                e.set_option(mod.XOPT_INLINE_SCAN, 1)
                e.set_option(mod.XOPT_INLINE_SCAN_MAX_GROUPS, limit)
                e.set_stream(torch.cuda.current_stream().cuda_stream)
                e.sort_from_to(t.data_ptr(), n, 0, 7, dst[3:].data_ptr())
                e.sync()
                outs.append(dst.cpu().numpy().view(np.uint32).copy())
                e.sort_from(t.data_ptr(), n)
                assert np.array_equal(e.download(), np.sort(keys))
        assert np.array_equal(t.cpu().numpy().view(np.uint32), keys)
        low = keys & np.uint32((1 << 28) - 1)
        assert np.array_equal(outs[0][3:], keys[np.argsort(low, kind="stable")]) and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("kernel", [1, 2, 3])
@pytest.mark.parametrize("dt,n", [("uint32", 70001), ("int32", (1 << 20) + 4099), ("uint64", 300007), ("int64", 1 << 21)])
def test_8bit_scatter_kernel_variants_are_stable_and_exact(mod, oracle, kernel, dt, n):
    """RSX_XOPT_REORDER8_KERNEL: the three 8-bit scatter kernels (two ranking rounds / one trip through LDS / ranks from returning LDS
    atomics) give the same keys and the same STABLE payload order — ties everywhere (a third of the keys equal), ragged sizes, all
    key types, and constant data (the wave-uniform path of kernel 3)."""
    import os
    keys = oracle.dataset("SeededUniform", dt, n, seed=kernel + n % 13)
    keys[::3] = keys[7]
    # (kernel 1 carries a uint32 key and its payload as ONE 64-bit element by default; RSX_R8_PACKED=0, read at rsx_create, keeps them apart)
    cases = [(keys, "1"), (np.full(n, keys[11], dtype=dt), "1")] + ([(keys, "0")] if kernel == 1 and np.dtype(dt).itemsize == 4 else [])
    for data, packed in cases:
        os.environ["RSX_R8_PACKED"] = packed
        try:
            e = mod.Engine(dt, n, payload=True)
        finally:
            del os.environ["RSX_R8_PACKED"]
        with e:
            e.set_option(mod.OPT_RADIX_BITS, 8)
            e.set_option(mod.XOPT_REORDER8_KERNEL, kernel)
            e.upload(data, np.arange(n, dtype=np.uint32))
            e.sort()
            ks, ps = e.download(want_perm=True)
        assert np.array_equal(ks, np.sort(data))
        assert np.array_equal(ps, np.argsort(data, kind="stable").astype(np.uint32))
    with mod.Engine(dt, 4096) as e:
        with pytest.raises(mod.RadixSortError):
            e.set_option(mod.XOPT_REORDER8_KERNEL, 4)


@pytest.mark.parametrize("dt,n,stay", [("uint32", (1 << 22) + 4099, 2), ("int32", (1 << 21) + 1, 1), ("uint64", (1 << 22) - 4097, 2), ("int64", 3 * (1 << 20) + 77, 1),
                                       ("uint32", 9 * 4096 * 8 + 5, 1)])
def test_8bit_scatter_as_a_staying_grid_is_exact(mod, oracle, dt, n, stay):
    """RSX_XOPT_REORDER8_STAY: N workgroups per CU walk the tiles of their XCD's range and prefetch the next tile while they rank the
    current one — same keys, same stable payload order as the one-workgroup-per-tile launch; ragged last tile, XCD ranges that do not
    divide (the last range reaches past the last tile), keys only / payload / uint32 key and payload kept apart (RSX_R8_PACKED=0), and a
    size whose grid is smaller than the staying one (falls back to one workgroup per tile)."""
    import os
    keys = oracle.dataset("SeededUniform", dt, n, seed=stay + n % 17)
    keys[::5] = keys[3]
    want_k = np.sort(keys)
    want_p = np.argsort(keys, kind="stable").astype(np.uint32)
    for payload, packed in [(False, "1"), (True, "1")] + ([(True, "0")] if np.dtype(dt).itemsize == 4 else []):
        os.environ["RSX_R8_PACKED"] = packed
        try:
            e = mod.Engine(dt, n, payload=payload)
        finally:
            del os.environ["RSX_R8_PACKED"]
        with e:
            e.set_option(mod.OPT_RADIX_BITS, 8)
            e.set_option(mod.XOPT_REORDER8_STAY, stay)
            if payload:
                e.upload(keys, np.arange(n, dtype=np.uint32))
                e.sort()
                ks, ps = e.download(want_perm=True)
                assert np.array_equal(ps, want_p)
            else:
                e.upload(keys)
                e.sort()
                ks = e.download()
            assert np.array_equal(ks, want_k)
            for bad in (-2, 9):
                with pytest.raises(mod.RadixSortError):
                    e.set_option(mod.XOPT_REORDER8_STAY, bad)


@pytest.mark.parametrize("strategy", ["waves", "waves-p2p"])
def test_engine_status_of_one_rank_stops_every_rank_of_the_sharded_sort(mod, oracle, strategy):
    """The fused scan's time-out word of ONE rank's engine (raised through the experiments build's test hook) travels in the count row the
    sharded sort gathers anyway — on the peer-store path it is folded into the device-side plan's verdict (bit 32 + rank), no host round trip —
    and EVERY rank raises EngineStatusError in that step; nobody pushes, nobody hangs, and the next step sorts (the word is reported once)."""
    import threading

    import torch
    from radix_sort_amd.distributed import EngineStatusError, ShardedSorter
    from test_gpu_sharded import _Loopback
    world, n = 4, 60000
    full = oracle.dataset("SeededUniform", "uint32", n * world, seed=13)
    hub = _Loopback(world)
    outcomes, results, errors = [None] * world, [None] * world, []
    p2p = strategy == "waves-p2p"

    def run(rank):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(full[rank * n:(rank + 1) * n].copy().view(np.int32)).cuda()
                staging = torch.empty_like(keys)
                recv = None if p2p else torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                obuf = torch.empty(2 * n, dtype=keys.dtype, device="cuda")
                with mod.Engine("uint32", 2 * n) as eng:
                    eng.set_stream(stream.cuda_stream)
                    sorter = ShardedSorter(eng, rank, world, 32, hub.view(rank), strategy=strategy)
                    if p2p:
                        sorter.setup_peer_exchange(2 * n, keys.device)
                    try:
                        if rank == 2:
                            eng.set_option(mod.XOPT_DEBUG_RAISE_SCAN_TIMEOUT, 1)      # the store a starved scan workgroup makes, on the engine's stream
                            torch.cuda.synchronize()
                        try:
                            sorter.sort(keys, staging, recv, None, None, None, obuf, None)
                            outcomes[rank] = "sorted"
                        except EngineStatusError as exc:
                            outcomes[rank] = str(exc)
                        n_local = sorter.sort(keys, staging, recv, None, None, None, obuf, None)      # reported once: the next step runs
                        eng.sync()
                        results[rank] = obuf[:n_local].cpu().numpy().view(np.uint32)
                    finally:
                        if p2p:
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
    assert all(o is not None and o.startswith("rank(s) [2] reported an engine error") for o in outcomes), outcomes
    assert np.array_equal(np.concatenate(results), np.sort(full))
