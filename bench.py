#!/usr/bin/env python3
"""bench.py — Mkeys/s of the whole LSD radix sort and scatter-pass HBM roofline fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one complete sort of the resident input (all 8 passes for uint32: per pass
histogram -> scan -> paste -> reorder).  At N=1 the workload is BASELINE.json configs[1]:
2^28 uint32 keys from the reference's `Random` generator (Dataset.h:110-120), 4-bit
digits.  Inputs are resident in HBM before the timed region; nothing crosses PCIe inside
it.  For N>1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank
holds 2^28 keys (weak scaling) and a step is: partition by top bits -> bucket-count
all_gather -> all_to_all of keys over xGMI -> local sort (radix-sort_amd/distributed.py).

Rank 0 prints ONE JSON line.  `roofline` is the reorder (scatter) kernel: algorithmic
bytes 2*n*(K+V) per launch over its mean launch time, measured live with HIP events on the
launch stream inside the timed region.  `cpu_baseline` is the reference's own
RadixSortCPU (oracle/_ref, kind "reference") or this repo's restatement (kind "port")
timed single-threaded on a bounded sample on this host.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)
KIND_CODES = {"Zeros": 0, "Range": 1, "InvertedRange": 2, "Random": 3, "RandomDistributed": 4}
DTYPE_CODES = {"uint32": 0, "int32": 1, "uint64": 2, "int64": 3}


def make_input(kind: str, dtype: str, n: int, seed: int) -> np.ndarray:
    """Synthetic input from the product's own Dataset.h generators (host C ABI)."""
    lib = C.CDLL(os.path.join(ROOT, "radix-sort_amd", "host", "libradixsort_host.so"))
    lib.rsxh_dataset_fill.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
    lib.rsxh_dataset_fill.restype = C.c_int
    out = np.empty(n, dtype=dtype)
    rc = lib.rsxh_dataset_fill(KIND_CODES[kind], DTYPE_CODES[dtype], out.ctypes.data, n, seed)
    if rc != 0:
        raise RuntimeError(f"rsxh_dataset_fill failed: {rc}")
    return out


def torch_view(t_np: np.ndarray):
    import torch
    signed = {"uint32": np.int32, "uint64": np.int64}.get(t_np.dtype.name)
    return torch.from_numpy(t_np.view(signed) if signed else t_np)


def cpu_baseline(sample: np.ndarray, whole: bool) -> dict:
    """Reference oracle timed like SortDataRadix (copy-in + sort), 1 thread (checker only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, RefOracle
    if RefOracle.available():
        ms, kind = RefOracle().time_radix_sort(sample, iters=1), "reference"
    else:
        ms, kind = Oracle().time_radix_sort(sample, iters=1), "port"
    return {
        "value": round(sample.size / ms * 1e-3, 3), "unit": "Mkeys/s", "cores": 1, "kind": kind,
        "sample": f"first 2^{int(np.log2(sample.size))} keys of the same input ({'the whole workload' if whole else 'a prefix'}), 1 iteration of copy-in + RadixSortCPU "
                  f"({ms:.0f} ms); host has {os.cpu_count()} logical cores",
    }


def load_traffic(workload: str):
    """HBM bytes per reorder launch from rocprofv3 PMC passes, if a summary was committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("reorder_hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-keys", type=int, default=28, help="keys per GPU = 2^this")
    ap.add_argument("--dtype", default="uint32", choices=list(DTYPE_CODES))
    ap.add_argument("--payload", action="store_true", help="carry a uint32 payload (h_Permut)")
    ap.add_argument("--dataset", default="Random", choices=list(KIND_CODES))
    ap.add_argument("--total-log2-keys", type=int, default=None, help="total keys over all GPUs = 2^this (e.g. 30 for BASELINE config 4); overrides --log2-keys")
    ap.add_argument("--cpu-sample-log2", type=int, default=28, help="CPU baseline sample = first 2^this keys of the input (2^28 ~ 13 s of RadixSortCPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="no HIP events inside the timed region (roofline then comes from the instrumented steps after it)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    rsx = entry.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dist = None
    force_exchange = os.environ.get("RSX_FORCE_EXCHANGE", "0") == "1"   # 1 rank, but through RCCL
    sharded = world > 1 or force_exchange
    if world > 1 or force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if args.total_log2_keys is not None:
        if (1 << args.total_log2_keys) % world:
            raise SystemExit("--total-log2-keys: total must divide evenly over the ranks")
        n = (1 << args.total_log2_keys) // world
    else:
        n = 1 << args.log2_keys
    key_bytes = np.dtype(args.dtype).itemsize
    pay_bytes = 4 if args.payload else 0
    # N=1: the reference's Random generator.  N>1: independent per-rank streams of the
    # seeded uniform generator (Random's fixed seed would give every rank the same shard).
    kind = args.dataset if not sharded else ("RandomDistributed" if args.dataset == "Random" else args.dataset)
    seed = 0x5EEDCAFEF00D + rank
    host_keys = make_input(kind, args.dtype, n, seed)
    keys = torch_view(host_keys).to(device)
    payload = torch.arange(n, dtype=torch.int32, device=device) if args.payload else None

    from radix_sort_amd.distributed import ShardedSorter
    capacity = 2 * n if sharded else n
    eng = rsx.Engine(args.dtype, capacity, payload=args.payload, device=local_rank)
    # a real (non-null) stream: sorts, RCCL collectives and the engine's HIP events all live on it
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    # timed region: HIP events bracket only the graded reorder launches (8 pairs per sort);
    # the per-phase table below comes from a fully instrumented step after it
    eng.set_option(rsx.OPT_PROFILE, 0 if args.no_events else 2)
    sorter = ShardedSorter(eng, rank, world, key_bytes * 8, dist, force_exchange=force_exchange, strategy=os.environ.get("RSX_STRATEGY", "auto"))
    staging = recv = spay = rpay = obuf = opay = None
    if sharded:
        staging = torch.empty_like(keys)
        recv = torch.empty(capacity, dtype=keys.dtype, device=device)
        obuf = torch.empty(capacity, dtype=keys.dtype, device=device)       # the pipelined path sorts into this
        if args.payload:
            spay = torch.empty_like(payload)
            rpay = torch.empty(capacity, dtype=payload.dtype, device=device)
            opay = torch.empty(capacity, dtype=payload.dtype, device=device)

    def step() -> int:
        return sorter.sort(keys, staging, recv, payload, spay, rpay, obuf, opay)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.timings(reset=True)

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_local = n
    for _ in range(args.steps):
        n_local = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rt = eng.timings(reset=True)
    eng.set_option(rsx.OPT_PROFILE, 1)
    sorter.record_timeline = sharded
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rt_all = eng.timings(reset=True)
    exchange_ms = {k: round(v, 4) for k, v in sorter.timeline_ms().items()} if sharded else None
    sorter.record_timeline = False
    if args.no_events:
        rt = rt_all

    ok = True
    if not args.no_verify:
        if sharded and sorter.result_in_out:
            torch.cuda.synchronize()
            got = obuf[:n_local].cpu().numpy().view(np.dtype(args.dtype))
        else:
            got = eng.download()
        ok = bool(np.all(got[:-1] <= got[1:])) and got.size == n_local
        if world == 1:
            u = host_keys.view(np.uint32 if key_bytes == 4 else np.uint64)
            v = got.view(u.dtype)
            ok = ok and int(np.bitwise_xor.reduce(u)) == int(np.bitwise_xor.reduce(v)) and int(u.sum(dtype=np.uint64)) == int(v.sum(dtype=np.uint64))
        if sharded:
            # across ranks: rank-order concatenation is sorted (boundary keys), nothing was lost or
            # invented (count, xor and sum of all keys before == after)
            ui = host_keys.view(np.uint32 if key_bytes == 4 else np.uint64)
            uo = got.view(ui.dtype)
            m63 = (1 << 63) - 1
            mine = [int(got.size), int(np.bitwise_xor.reduce(uo)) & m63 if got.size else 0, int(uo.sum(dtype=np.uint64)) & m63 if got.size else 0,
                    int(host_keys.size), int(np.bitwise_xor.reduce(ui)) & m63, int(ui.sum(dtype=np.uint64)) & m63,
                    int(got[0]) if got.size else 0, int(got[-1]) if got.size else 0]
            if np.dtype(args.dtype).kind == "u":
                mine[6], mine[7] = mine[6] - (1 << (key_bytes * 8 - 1)), mine[7] - (1 << (key_bytes * 8 - 1))   # fit int64, order kept
            t = torch.tensor(mine, dtype=torch.int64, device=device)
            rows = torch.empty(world * 8, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(rows, t)
            rows = rows.cpu().view(world, 8).tolist()
            full = [r for r in rows if r[0] > 0]
            ok = ok and sum(r[0] for r in rows) == sum(r[3] for r in rows)
            ok = ok and (np.bitwise_xor.reduce(np.array([r[1] for r in rows], dtype=np.int64)) == np.bitwise_xor.reduce(np.array([r[4] for r in rows], dtype=np.int64)))
            ok = ok and sum(r[2] for r in rows) & m63 == sum(r[5] for r in rows) & m63
            ok = ok and all(a[7] <= b[6] for a, b in zip(full, full[1:]))
        if not ok:
            raise SystemExit("bench: result is not a sorted permutation of the input — refusing to report a number")

    total_keys = n * world
    ms_per_step = elapsed / args.steps * 1e3
    reorder_ms = rt.reorder.avg_ms
    launches_per_step = rt.reorder.n / max(args.steps, 1)
    # algorithmic bytes of one reorder launch: n*(K+V) read + n*(K+V) written.
    # N>1: launches differ in size (the partition pass sees n keys, the local passes what arrived,
    # wave by wave on the pipelined path), so the figure is bytes of all launches / time of all launches
    passes = key_bytes * 2
    if sharded:
        local_passes = passes - 1 if sorter.last_path == "waves" else passes
        scatter_bytes_per_step = 2.0 * (key_bytes + pay_bytes) * (n + local_passes * n_local)
        scatter_bytes = scatter_bytes_per_step / launches_per_step if launches_per_step else 0.0
    else:
        scatter_bytes = 2.0 * n_local * (key_bytes + pay_bytes)
    achieved = scatter_bytes / (reorder_ms * 1e-3) * 1e-9 if reorder_ms > 0 else 0.0
    lg = n.bit_length() - 1 if n & (n - 1) == 0 else None
    workload = f"{'2^%d' % lg if lg is not None else n} {args.dtype}{'+u32 payload' if args.payload else ''} {kind}, 4-bit digits, {key_bytes * 2} passes"
    line = {
        "metric": "Mkeys/s + scatter-pass HBM GB/s (% of peak), 2^28 uint32 keys",
        "value": round(total_keys / (elapsed / args.steps) * 1e-6, 1),
        "unit": "Mkeys/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"uint32": "u32", "int32": "i32", "uint64": "u64", "int64": "i64"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": workload, "keys_per_gpu": n, "total_keys": total_keys,
                   "parallelism": "single GPU" if not sharded else f"msd-partition[{sorter.last_path}] x{world} + all_to_all (RCCL) + local LSD sort",
                   "verified": ok},
        "roofline": {
            "bound": "hbm", "kernel": "reorder_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(workload),
            "algorithmic_bytes_per_launch": scatter_bytes, "avg_launch_ms": round(reorder_ms, 5),
            "launches_per_step": launches_per_step,
        },
        "phases_ms_per_launch": {"histogram": round(rt_all.histogram.avg_ms, 5), "scan": round(rt_all.scan.avg_ms, 5),
                                 "paste": round(rt_all.paste.avg_ms, 5), "reorder": round(rt_all.reorder.avg_ms, 5),
                                 "note": "fully instrumented steps after the timed region"},
    }
    if exchange_ms is not None:
        line["sharded_phases_ms"] = dict(exchange_ms, note="rank 0, device time between marks of one instrumented step after the timed region")
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        m = min(1 << args.cpu_sample_log2, n)
        line["cpu_baseline"] = cpu_baseline(host_keys[:m], whole=(m == n))
    if rank == 0:
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
