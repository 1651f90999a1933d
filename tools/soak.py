"""Soak run: many sorts of random shape on one engine per (dtype, payload), every result checked
against numpy's stable sort.  python tools/soak.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

EXPERIMENTS = os.environ.get("SOAK_EXPERIMENTS") == "1"
m = entry.load_package().experiments() if EXPERIMENTS else entry.load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
cap = int(os.environ.get("RSX_SOAK_CAP", 9_000_000))      # 2,197 tiles: both sides of the self-scan / fused-scan border
engines = {(dt, p): m.Engine(dt, cap, payload=p) for dt in ("uint32", "int32", "uint64", "int64") for p in (False, True)}
t0 = time.time()
for it in range(iters):
    dt = ("uint32", "int32", "uint64", "int64")[rng.integers(0, 4)]
    payload = bool(rng.integers(0, 2))
    n = int(rng.integers(1, cap)) if rng.integers(0, 4) else int(rng.integers(1, 9000))
    info = np.iinfo(dt)
    shape = rng.integers(0, 5)
    if shape == 0:
        keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    elif shape == 1:
        keys = rng.integers(0, 100, size=n).astype(dt)
    elif shape == 2:
        keys = np.sort(rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True))
    elif shape == 3:
        keys = np.full(n, int(rng.integers(0, 1000)), dtype=dt)
    else:
        keys = (rng.integers(0, 1 << 16, size=n).astype(np.int64) * 65537 % 1000003).astype(dt)
    e = engines[(dt, payload)]
    e.set_option(m.OPT_LOOKAHEAD, int(rng.integers(0, 2)))
    e.set_option(m.OPT_SMALL_SCAN, int(rng.integers(0, 2)))
    # every path of the chain dispatch, in any combination (all of them must give the same array)
    e.set_option(m.OPT_RADIX_BITS, (4, 4, 8)[rng.integers(0, 3)])
    e.set_option(m.OPT_SELF_SCAN, int(rng.integers(0, 4) != 0))
    e.set_option(m.OPT_FUSED_SCAN, int(rng.integers(0, 4) != 0))
    e.set_option(m.OPT_TILE_SORT, int(rng.integers(0, 4) != 0))
    e.set_option(m.OPT_XCD_PHASE, (-1, 0, int(rng.integers(1, 40)))[rng.integers(0, 3)])
    e.set_option(m.OPT_SMALL_TILE_MAX_KEYS, (1 << 19, 0, 1 << 20)[rng.integers(0, 3)])
    e.set_option(m.OPT_SELF_SCAN_MAX_TILES, (1024, 1024, 100)[rng.integers(0, 3)])
    e.set_option(m.OPT_FUSED_SCAN_MAX_GROUPS, (-1, 1, 0)[rng.integers(0, 3)])
    if EXPERIMENTS:      # SOAK_EXPERIMENTS=1: the experiments build — the scan inside the reorder launch, the three 8-bit scatter kernels, the staying grid
        e.set_option(m.XOPT_INLINE_SCAN, int(rng.integers(0, 2)))
        e.set_option(m.XOPT_INLINE_SCAN_MAX_GROUPS, (64, 2, 512)[rng.integers(0, 3)])
        e.set_option(m.XOPT_REORDER8_KERNEL, int(rng.integers(1, 4)))
        e.set_option(m.XOPT_REORDER8_STAY, (0, 0, 1, 2)[rng.integers(0, 4)])
    perm = np.arange(n, dtype=np.uint32) if payload else None
    e.upload(keys, perm)
    e.sort()
    if payload:
        ks, ps = e.download(want_perm=True)
        ok = np.array_equal(ks, np.sort(keys, kind="stable")) and np.array_equal(ps, np.argsort(keys, kind="stable").astype(np.uint32))
    else:
        ok = np.array_equal(e.download(), np.sort(keys, kind="stable"))
    e.sync()             # a fused / inline scan that timed out would be reported here
    if ok and it % 3 == 0:
        # the multi-GPU partitions on the same keys: sampled splitters, then the top bits
        import torch
        signed = {"uint32": np.int32, "uint64": np.int64}.get(dt)
        tk = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
        tp = torch.arange(n, dtype=torch.int32, device="cuda") if payload else None
        out, pout = torch.empty_like(tk), (torch.empty_like(tp) if payload else None)
        u = keys.view(np.uint32 if keys.dtype.itemsize == 4 else np.uint64)
        if keys.dtype.kind == "i":
            u = u ^ u.dtype.type(1 << (keys.dtype.itemsize * 8 - 1))
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        samples = e.sample_keys(tk.data_ptr(), n, min(1024, n))
        sp = sorted(set(rng.choice(np.array(samples, dtype=np.uint64), size=int(rng.integers(1, 8))).tolist()))
        spa = np.array(sp, dtype=u.dtype)
        d = (np.searchsorted(spa, u, side="left") + np.searchsorted(spa, u, side="right")).astype(np.int64)
        ok = e.partition_count_split(tk.data_ptr(), n, [int(v) for v in sp]) == [int(v) for v in np.bincount(d, minlength=2 * len(sp) + 1)]
        e.partition_scatter_split(tk.data_ptr(), n, out.data_ptr(), tp.data_ptr() if payload else None, pout.data_ptr() if payload else None)
        torch.cuda.synchronize()
        order = np.argsort(d, kind="stable")
        ok = ok and np.array_equal(out.cpu().numpy().view(keys.dtype), keys[order])
        if payload:
            ok = ok and np.array_equal(pout.cpu().numpy().view(np.uint32), order.astype(np.uint32))
        bits = keys.dtype.itemsize * 8
        dtop = (u >> u.dtype.type(bits - 4)).astype(np.int64)
        ok = ok and e.partition_count(tk.data_ptr(), n, bits - 4, 4) == [int(v) for v in np.bincount(dtop, minlength=16)]
        e.partition_scatter(tk.data_ptr(), n, bits - 4, 4, out.data_ptr(), tp.data_ptr() if payload else None, pout.data_ptr() if payload else None)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(out.cpu().numpy().view(keys.dtype), keys[np.argsort(dtop, kind="stable")])
        e.set_stream(0)
    if not ok:
        print("MISMATCH", it, dt, payload, n, shape, flush=True)
        sys.exit(1)
    if it % 50 == 0:
        print(f"iter {it} ok ({time.time() - t0:.0f} s)", flush=True)
print(f"soak ok: {iters} sorts in {time.time() - t0:.0f} s")
