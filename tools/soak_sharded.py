#!/usr/bin/env python3
"""Soak of the sharded sort on ONE GPU: random world size, key type, payload, pipeline depth, wave grouping, exchange and input size, rank
threads with the loopback collectives of tests/test_gpu_sharded.py (RCCL's stream semantics: a missing wait races here as it would on 8 GPUs),
several steps per engine set, every result against numpy's stable sort.      python tools/soak_sharded.py [iterations=60] [seed=1]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def one(rsx, Loopback, ShardedSorter, rng, it):
    import torch
    world = int(rng.choice([1, 2, 4, 8, 16, 3]))
    dtype = str(rng.choice(["uint32", "int32", "uint64", "int64"]))
    payload = bool(rng.integers(0, 2))
    strategy = str(rng.choice(["waves", "waves-p2p", "auto"])) if world != 3 else "auto"
    lo = max(world.bit_length() - 1, 1)
    bits = int(rng.integers(lo, 9)) if world != 3 else None
    grouping = str(rng.choice(["doubling", "single"]))
    radix_bits = int(rng.choice([4, 8]))
    n = int(rng.choice([1, 17, 1000, 4097, 70001, 300007, 1 << 20]))
    kind = str(rng.choice(["uniform", "uniform", "uniform", "ties", "small"])) if strategy == "auto" else "uniform"
    info = np.iinfo(dtype)
    total = n * world
    if kind == "uniform":
        full = rng.integers(info.min, info.max, size=total, dtype=dtype, endpoint=True)
    elif kind == "ties":
        full = rng.integers(info.min, info.max, size=total, dtype=dtype, endpoint=True)
        full[rng.random(total) < 0.7] = full[0]
    else:
        full = rng.integers(0, 1000, size=total).astype(dtype)
    hub = Loopback(world)
    results, errors = [None] * world, []
    steps = int(rng.integers(1, 4))
    desc = f"it {it}: world {world} {dtype}{'+pay' if payload else ''} {strategy} bits {bits} {grouping} r{radix_bits} n {n} {kind} x{steps}"

    def run(rank):
        try:
            shard = full[rank * n:(rank + 1) * n].copy()
            signed = {"uint32": np.int32, "uint64": np.int64}.get(dtype)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                keys = torch.from_numpy(shard.view(signed) if signed else shard).cuda()
                pay = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int32, device="cuda") if payload else None
                staging, spay = torch.empty_like(keys), (torch.empty_like(pay) if payload else None)
                cap = max(2 * n + 2048, total if strategy == "auto" else 0)
                recv = torch.empty(cap, dtype=keys.dtype, device="cuda")
                rpay = torch.empty(cap, dtype=torch.int32, device="cuda") if payload else None
                obuf = torch.empty(cap, dtype=keys.dtype, device="cuda")
                opay = torch.empty(cap, dtype=torch.int32, device="cuda") if payload else None
                with rsx.Engine(dtype, cap, payload=payload) as eng:
                    eng.set_stream(stream.cuda_stream)
                    if radix_bits != 4:
                        eng.set_option(rsx.OPT_RADIX_BITS, radix_bits)
                    sorter = ShardedSorter(eng, rank, world, np.dtype(dtype).itemsize * 8, hub.view(rank), strategy=strategy, partition_bits=bits, wave_grouping=grouping,
                                           force_exchange=True)
                    if strategy == "waves-p2p":
                        sorter.setup_peer_exchange(cap, keys.device, payload)
                    try:
                        for _ in range(steps):
                            n_local = sorter.sort(keys, staging, recv, pay, spay, rpay, obuf, opay)
                        eng.sync()
                        if sorter.result_in_out:
                            torch.cuda.synchronize()
                            results[rank] = (obuf[:n_local].cpu().numpy().view(np.dtype(dtype)), opay[:n_local].cpu().numpy().view(np.uint32) if payload else None)
                        else:
                            out = eng.download(want_perm=True) if payload else (eng.download(), None)
                            results[rank] = (out[0], out[1])
                    finally:
                        if strategy == "waves-p2p":
                            sorter.close_peer_exchange()
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)
            hub.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    if errors or any(t.is_alive() for t in threads):
        raise SystemExit(f"FAILED ({desc}): {errors or 'a rank thread hangs'}")
    got = np.concatenate([r[0] for r in results])
    if not np.array_equal(got, np.sort(full, kind="stable")):
        raise SystemExit(f"FAILED ({desc}): keys differ from numpy's sort")
    if payload and not np.array_equal(np.concatenate([r[1] for r in results]), np.argsort(full, kind="stable").astype(np.uint32)):
        raise SystemExit(f"FAILED ({desc}): payload is not the stable argsort")
    return desc


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rsx = entry.load_package()
    from radix_sort_amd.distributed import ShardedSorter
    from test_gpu_sharded import _Loopback
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for it in range(iters):
        desc = one(rsx, _Loopback, ShardedSorter, rng, it)
        if it % 10 == 0:
            print(f"{desc}  ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"sharded soak ok: {iters} configurations in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
