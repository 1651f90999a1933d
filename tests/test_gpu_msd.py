"""The exchange step of the sharded sort on the top B <= 8 key bits (rsx_msd_count / _scatter / _plan / _push) against numpy
and against the host planner (radix-sort_amd/host/ShardPlanner, the ONE implementation the Python and C++ drivers share):
counts, the wave-major staging order, the device-side plan, and the pushed receive buffers of all ranks of a world emulated on
the one GPU (every "rank" is an engine of this process; the receive buffers are plain device tensors)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unsigned(keys):
    bits = keys.dtype.itemsize * 8
    u = keys.view(np.uint32 if bits == 32 else np.uint64)
    if keys.dtype.kind == "i":
        u = u ^ u.dtype.type(1 << (bits - 1))
    return u, bits


def _position(top_byte, bits, world):
    """wave-major position of a fine bucket (the key's top byte): [wave][rank][low 8 - bits bits]"""
    sub_shift = 8 - bits
    c, sub = top_byte >> sub_shift, top_byte & ((1 << sub_shift) - 1)
    k = (1 << bits) // world
    return (((c % k) * world + c // k) << sub_shift) | sub


@pytest.mark.parametrize("dt,bits,world,payload", [("uint32", 4, 8, False), ("uint32", 6, 8, False), ("int32", 8, 8, True), ("uint64", 5, 4, True), ("int64", 8, 16, False),
                                                    ("uint32", 3, 8, True), ("uint64", 1, 2, False), ("uint32", 8, 1, False), ("int64", 6, 2, True)])
def test_count_and_wave_major_scatter(rsx, oracle, dt, bits, world, payload):
    import torch
    n = 300007
    keys = oracle.dataset("SeededUniform", dt, n, seed=bits * 17 + world)
    keys[::7] = keys[3]
    u, kb = _unsigned(keys)
    top = (u >> u.dtype.type(kb - 8)).astype(np.int64)
    coarse = top >> (8 - bits)
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    tk = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
    pay = torch.arange(n, dtype=torch.int32, device="cuda") if payload else None
    staging = torch.empty_like(tk)
    spay = torch.empty_like(pay) if payload else None
    row = torch.full((259,), -1, dtype=torch.int64, device="cuda")
    with rsx.Engine(dt, n, payload=payload) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):                                   # twice: the tables are reused
            e.msd_count(tk.data_ptr(), n, bits, world, row.data_ptr())
            e.msd_scatter(tk.data_ptr(), n, staging.data_ptr(), pay.data_ptr() if payload else None, spay.data_ptr() if payload else None)
            torch.cuda.synchronize()
            want = np.bincount(coarse, minlength=256)
            assert row[:256].cpu().tolist() == [int(v) for v in want] and row[256:].cpu().tolist() == [-1, -1, -1]
            order = np.argsort(_position(top, bits, world), kind="stable")
            assert np.array_equal(staging.cpu().numpy().view(keys.dtype), keys[order])
            if payload:
                assert np.array_equal(spay.cpu().numpy().view(np.uint32), order.astype(np.uint32))
        with pytest.raises(rsx.RadixSortError):              # the count was consumed
            e.msd_scatter(tk.data_ptr(), n, staging.data_ptr(), pay.data_ptr() if payload else None, spay.data_ptr() if payload else None)
        for bad_bits, bad_world in ((0, 1), (9, 8), (4, 3), (2, 8), (4, 32)):
            with pytest.raises(rsx.RadixSortError):
                e.msd_count(tk.data_ptr(), n, bad_bits, bad_world, row.data_ptr())
        e.msd_count(tk.data_ptr(), 0, bits, world, row.data_ptr())
        torch.cuda.synchronize()
        assert row[:256].cpu().tolist() == [0] * 256
        # the engine still sorts (the 8-bit tables were borrowed, not broken)
        e.sort_from(tk.data_ptr(), n, pay.data_ptr() if payload else None)
        assert np.array_equal(e.download(), np.sort(keys))


@pytest.mark.parametrize("dt,bits,world,payload,grouping", [("uint32", 6, 8, False, 1), ("uint32", 4, 8, True, 0), ("int64", 8, 4, True, 1), ("uint64", 5, 2, False, 1), ("int32", 8, 16, False, 0),
                                                             ("uint32", 6, 8, True, 0)])
def test_device_plan_and_push_fill_every_receive_buffer(rsx, oracle, dt, bits, world, payload, grouping):
    """A whole world on the one GPU: every rank counts and scatters its shard, the [rank][bucket] table is assembled as the all_gather
    would, every rank plans on the device and pushes wave by wave into the receive buffers of all ranks.  Afterwards rank d's buffer holds,
    wave by wave (16-byte aligned), the keys of bucket d * k + w of rank 0, 1, ... in order — checked against numpy — and the plan the
    host reads back equals the host planner's for the same table."""
    import torch
    from radix_sort_amd import planner
    n = 120011
    full = oracle.dataset("SeededUniform", dt, n * world, seed=bits + 100 * world)
    full[::5] = full[2]
    u_all, kb = _unsigned(full)
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    tdt = torch.int32 if kb == 32 else torch.int64
    cap = 4 * n + 4 * 256          # the tie value (a fifth of all keys) lands on one rank
    waves = (1 << bits) // world
    stride = 259
    table = torch.zeros(world * stride, dtype=torch.int64, device="cuda")
    engines = [rsx.Engine(dt, n, payload=payload) for _ in range(world)]
    try:
        shards, stagings, spays, recvs, rpays = [], [], [], [], []
        for r, e in enumerate(engines):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            shard = full[r * n:(r + 1) * n]
            tk = torch.from_numpy((shard.view(signed) if signed else shard).copy()).cuda()
            pay = torch.arange(r * n, (r + 1) * n, dtype=torch.int32, device="cuda") if payload else None
            shards.append((tk, pay))
            stagings.append(torch.empty_like(tk))
            spays.append(torch.empty_like(pay) if payload else None)
            recvs.append(torch.full((cap,), -7, dtype=tdt, device="cuda"))
            rpays.append(torch.full((cap,), -7, dtype=torch.int32, device="cuda") if payload else None)
            e.msd_count(tk.data_ptr(), n, bits, world, table[r * stride:].data_ptr())
            e.msd_scatter(tk.data_ptr(), n, stagings[r].data_ptr(), pay.data_ptr() if payload else None, spays[r].data_ptr() if payload else None)
            table[r * stride + 256] = cap          # receive capacity
            table[r * stride + 257] = cap          # output capacity
        peer_k = torch.tensor([t.data_ptr() for t in recvs], dtype=torch.int64, device="cuda")
        peer_p = torch.tensor([t.data_ptr() for t in rpays], dtype=torch.int64, device="cuda") if payload else None
        torch.cuda.synchronize()
        host_table = table.cpu().view(world, stride).numpy()
        side = torch.cuda.Stream()
        plans = []
        for r, e in enumerate(engines):
            side.wait_stream(torch.cuda.current_stream())
            e.msd_plan(table.data_ptr(), stride, 256, r, side.cuda_stream if r % 2 else 0, grouping)      # odd ranks plan on a side stream
            plans.append(e.msd_plan_wait(waves, world))
            for w in range(waves):
                e.msd_push(w, stagings[r].data_ptr(), peer_k.data_ptr(), spays[r].data_ptr() if payload else None, peer_p.data_ptr() if payload else None, parts=(r % 3) * 7)
        torch.cuda.synchronize()
        counts = [[int(v) for v in host_table[r, :1 << bits]] for r in range(world)]
        start, offset, loads = planner.wave_layout(counts, world, 1 << bits, 4, grouping)
        top_all = (u_all >> u_all.dtype.type(kb - bits)).astype(np.int64)
        for d in range(world):
            ws, wc, ld, verdict = plans[d]
            assert verdict == 0 and ld == loads and ws == start[d]
            got = recvs[d].cpu().numpy().view(full.dtype)
            gotp = rpays[d].cpu().numpy().view(np.uint32) if payload else None
            used = np.zeros(cap, dtype=bool)
            for w in range(waves):
                b = d * waves + w
                assert wc[w] == sum(counts[s][b] for s in range(world))
                for s in range(world):
                    idx = np.flatnonzero(top_all[s * n:(s + 1) * n] == b) + s * n
                    # inside a segment the keys are grouped (stably) by the remaining bits of the top byte
                    sub = (u_all[idx] >> u_all.dtype.type(kb - 8)).astype(np.int64)
                    idx = idx[np.argsort(sub, kind="stable")]
                    at = offset[d][w][s]
                    assert np.array_equal(got[at:at + idx.size], full[idx]), (d, w, s)
                    if payload:
                        assert np.array_equal(gotp[at:at + idx.size], idx.astype(np.uint32)), (d, w, s)
                    used[at:at + idx.size] = True
            raw = recvs[d].cpu().numpy()
            assert (raw[~used] == -7).all()                 # nothing outside the segments
    finally:
        for e in engines:
            e.close()


def test_plan_verdict_when_a_rank_is_too_small(rsx, oracle):
    """One rank's receive buffer is too small: every rank's plan carries the same verdict and no push writes anything."""
    import torch
    world, bits, n = 4, 4, 50000
    full = oracle.dataset("SeededUniform", "uint32", n * world, seed=3)
    stride = 259
    table = torch.zeros(world * stride, dtype=torch.int64, device="cuda")
    engines = [rsx.Engine("uint32", n) for _ in range(world)]
    try:
        stagings, recvs = [], []
        for r, e in enumerate(engines):
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            tk = torch.from_numpy(full[r * n:(r + 1) * n].view(np.int32).copy()).cuda()
            stagings.append(torch.empty_like(tk))
            recvs.append(torch.full((2 * n,), -7, dtype=torch.int32, device="cuda"))
            e.msd_count(tk.data_ptr(), n, bits, world, table[r * stride:].data_ptr())
            e.msd_scatter(tk.data_ptr(), n, stagings[r].data_ptr())
            table[r * stride + 256] = 2 * n if r != 2 else n // 2
            table[r * stride + 257] = 2 * n
        peer_k = torch.tensor([t.data_ptr() for t in recvs], dtype=torch.int64, device="cuda")
        for r, e in enumerate(engines):
            e.msd_plan(table.data_ptr(), stride, 256, r)
            assert e.msd_plan_wait(4, world)[3] == 1 << 2
            for w in range(4):
                e.msd_push(w, stagings[r].data_ptr(), peer_k.data_ptr())
        torch.cuda.synchronize()
        for t in recvs:
            assert (t == -7).all()
        with pytest.raises(rsx.RadixSortError):
            engines[0].msd_push(4, stagings[0].data_ptr(), peer_k.data_ptr())
    finally:
        for e in engines:
            e.close()
