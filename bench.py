#!/usr/bin/env python3
"""bench.py — Mkeys/s of the whole LSD radix sort and scatter-pass HBM roofline fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one complete sort of the resident input (all 8 passes for uint32: per pass
histogram -> scan -> paste -> reorder).  At N=1 the workload is BASELINE.json configs[1]:
2^28 uint32 keys from the reference's `Random` generator (Dataset.h:110-120), 4-bit
digits.  Inputs are resident in HBM before the timed region; nothing crosses PCIe inside
it.  For N>1 the default workload is BASELINE.json configs[3]: 2^30 uint32 keys in total,
2^30/N per GPU (`--log2-keys K` instead fixes 2^K keys PER GPU: weak scaling), and a step
is: count by the top `--partition-bits` bits -> bucket-count all_gather ("histogram
all-to-all") beside the scatter into wave-major staging -> the keys over xGMI wave by wave
(RCCL all_to_all, or RSX_STRATEGY=waves-p2p: peer stores) -> local sorts of the waves in
doubling groups (radix-sort_amd/distributed.py), one rank per GPU over RCCL.

`python bench.py --gpus N` from a plain shell starts the N ranks itself: the parent makes
no GPU call, launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` as
a CHILD process on a free port, relays rank 0's JSON line and exits with the child's
return code.  Under an external torch.distributed.run (WORLD_SIZE already set) it is a rank.

Rank 0 prints ONE JSON line.  `roofline` is the reorder (scatter) kernel: algorithmic
bytes 2*n*(K+V) per launch over its mean launch time, measured live with HIP events on the
launch stream inside the timed region.  `cpu_baseline` is the reference's own
RadixSortCPU (oracle/_ref, kind "reference") or this repo's restatement (kind "port")
timed single-threaded on a bounded sample on this host (rank 0).  At N=1 the sample is the
whole workload and the array it sorted is then compared, key for key, with the GPU's
output (`config.verified`).  At N>1 the ranks' outputs are gathered on rank 0 and compared
with a host sort of all the inputs.

RSX_BENCH_REHEARSAL=1 (tests only) swaps the device side for the CPU test double of
tests/_bench_rehearsal.py under gloo, so that the launcher and every line of the N>1 rank
logic can be run on a machine without GPUs; its line says so and carries no value.
RSX_BENCH_SHARED_GPU=1 (tests only) keeps the real engine but puts every rank process on
cuda:0 with gloo collectives staged through the host (tests/_host_staged_dist.py): the
one-GPU test box's rehearsal of N rank processes; its line says so and carries no value too.
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
METRIC = "Mkeys/s + scatter-pass HBM GB/s (% of peak), 2^28 uint32 keys"
BASE_SEED = 0x5EEDCAFEF00D


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


def make_shard(kind: str, dtype: str, offset: int, n: int, total: int, seed: int) -> np.ndarray:
    """Elements [offset, offset + n) of the `total`-element dataset (host C ABI rsxh_dataset_fill_shard): a rank's contiguous
    shard of ONE dataset — BASELINE config 4 is `Random` (Dataset.h:110-120) sharded contiguously, so rank r takes the
    generator's stream from draw r*n on (std::mt19937::discard)."""
    lib = C.CDLL(os.path.join(ROOT, "radix-sort_amd", "host", "libradixsort_host.so"))
    lib.rsxh_dataset_fill_shard.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.rsxh_dataset_fill_shard.restype = C.c_int
    out = np.empty(n, dtype=dtype)
    rc = lib.rsxh_dataset_fill_shard(KIND_CODES[kind], DTYPE_CODES[dtype], out.ctypes.data, offset, n, total, seed)
    if rc != 0:
        raise RuntimeError(f"rsxh_dataset_fill_shard failed: {rc}")
    return out


def torch_view(t_np: np.ndarray):
    import torch
    signed = {"uint32": np.int32, "uint64": np.int64}.get(t_np.dtype.name)
    return torch.from_numpy(t_np.view(signed) if signed else t_np)


def cpu_baseline(sample: np.ndarray, whole: bool):
    """Reference oracle timed like SortDataRadix (copy-in + sort), 1 thread (checker only).
    Returns (json entry, the sorted sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, RefOracle
    if RefOracle.available():
        (ms, ordered), kind = RefOracle().time_radix_sort(sample, iters=1, return_sorted=True), "reference"
    else:
        (ms, ordered), kind = Oracle().time_radix_sort(sample, iters=1, return_sorted=True), "port"
    lg = sample.size.bit_length() - 1
    entry = {
        "value": round(sample.size / ms * 1e-3, 3), "unit": "Mkeys/s", "cores": 1, "kind": kind,
        "sample": f"first {'2^%d' % lg if sample.size == 1 << lg else sample.size} keys of rank 0's input ({'the whole single-GPU workload' if whole else 'a prefix'}), "
                  f"1 iteration of copy-in + RadixSortCPU ({ms:.0f} ms); host has {os.cpu_count()} logical cores",
    }
    return entry, ordered


def kernel_source_digest() -> str:
    """sha256[:16] over the kernel sources (radix-sort_amd/csrc, sorted by name) — the same function as tools/pmc_summarize.py, which
    stamps every PMC summary with it."""
    import hashlib
    csrc = os.path.join(ROOT, "radix-sort_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode() + b"\0" + open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload: str):
    """(HBM bytes per reorder launch from the committed rocprofv3 PMC passes, note).  The figure is a replay of profiles/pmc_traffic.json,
    not a counter read in this run (PMC collection needs rocprofv3 around the process), so it is printed only while the kernel sources
    still hash to what the passes were taken on; otherwise traffic is null and the note says why."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
    except (OSError, ValueError):
        entry = None
    if not entry:
        return None, "no PMC pass committed for this workload"
    now = kernel_source_digest()
    if entry.get("kernel_source_digest") != now:
        return None, f"the committed PMC pass was taken on other kernel sources ({entry.get('kernel_source_digest', 'unstamped')} != {now}): re-run tools/final_profiles.sh"
    return entry.get("reorder_hbm_bytes_per_launch"), f"replayed from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on kernel sources {now})"


def free_port() -> int:
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n_gpus: int, script: str | None = None, argv: list[str] | None = None) -> int:
    """Start the N ranks as children (one torch.distributed.run process, which forks the ranks) and
    relay their output: rank 0's JSON line to stdout, everything else to stderr.  This process never
    touches HIP — the children are ordinary subprocesses, not an exec of a process that has
    initialised the GPU.  Returns the children's exit code (non-zero if any rank failed)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench: the ranks exited cleanly but rank 0 printed no result line\n")
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-keys", type=int, default=None, help="keys PER GPU = 2^this (default 28 at N=1; at N>1 the default is --total-log2-keys 30)")
    ap.add_argument("--total-log2-keys", type=int, default=None, help="total keys over all GPUs = 2^this (default 30 = BASELINE config 4 at N>1)")
    ap.add_argument("--dtype", default="uint32", choices=list(DTYPE_CODES))
    ap.add_argument("--payload", action="store_true", help="carry a uint32 payload (h_Permut)")
    ap.add_argument("--dataset", default="Random", choices=list(KIND_CODES))
    ap.add_argument("--cpu-sample-log2", type=int, default=None, help="CPU baseline sample = first 2^this keys of rank 0's input (default: 28 at N=1 = the whole workload, ~13 s; 26 at N>1)")
    ap.add_argument("--radix-bits", type=int, default=4, choices=[4, 8], help="digit width: 4 = the reference's configuration and the headline; 8 = half the passes, reported as a separate row")
    ap.add_argument("--partition-bits", type=int, default=int(os.environ.get("RSX_PARTITION_BITS", "0")) or None,
                    help="N>1: top key bits of the exchange partition = pipeline depth, 2^bits / N waves per rank (default: 8 waves per rank, at least 4 bits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-verify-full", action="store_true", help="N>1: skip the gather of all outputs on rank 0 and keep only the checksum / boundary checks")
    ap.add_argument("--no-events", action="store_true", help="no HIP events inside the timed region (roofline then comes from the instrumented steps after it)")
    return ap.parse_args(argv)


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))     # parent: no GPU call has been made in this process

    import torch
    import __graft_entry__ as entry
    rsx = entry.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or let bench.py start the ranks itself")
    rehearsal = os.environ.get("RSX_BENCH_REHEARSAL", "0") == "1"       # tests only: CPU test double under gloo
    shared_gpu = os.environ.get("RSX_BENCH_SHARED_GPU", "0") == "1"     # tests only: every rank a process of its own on cuda:0, gloo staged through the host
    force_exchange = os.environ.get("RSX_FORCE_EXCHANGE", "0") == "1"   # 1 rank, but through RCCL
    sharded = world > 1 or force_exchange
    dist = None
    if rehearsal:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _bench_rehearsal as reh
        device = torch.device("cpu")
        sync = lambda: None     # noqa: E731
    else:
        if shared_gpu:
            local_rank = 0
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        sync = torch.cuda.synchronize
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _host_staged_dist import HostStagedDist
            dist = HostStagedDist()
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # N=1: BASELINE config 2 (2^28 keys).  N>1: BASELINE config 4 (2^30 keys over all GPUs) unless a
    # per-GPU size is asked for, which makes it a weak-scaling run.
    if args.total_log2_keys is None and args.log2_keys is None:
        if world > 1:
            args.total_log2_keys = 30
        else:
            args.log2_keys = 28
    if args.total_log2_keys is not None:
        if (1 << args.total_log2_keys) % world:
            raise SystemExit("--total-log2-keys: total must divide evenly over the ranks")
        n = (1 << args.total_log2_keys) // world
        scaling = "strong"
    else:
        n = 1 << args.log2_keys
        scaling = "weak"
    key_bytes = np.dtype(args.dtype).itemsize
    pay_bytes = 4 if args.payload else 0
    # N=1: the reference's Random generator.  N>1 with a fixed TOTAL (strong scaling, BASELINE config 4): every rank holds its
    # contiguous n-key shard of the ONE dataset of n*world keys — for Random, draws rank*n .. rank*n+n-1 of the generator's
    # stream.  N>1 with a fixed size PER GPU (weak scaling): independent per-rank streams of the seeded uniform generator
    # (there is no "one dataset" whose size grows with the rank count in the reference).
    one_dataset = sharded and scaling == "strong"
    if one_dataset:
        kind = args.dataset
        host_keys = make_shard(kind, args.dtype, rank * n, n, n * world, BASE_SEED)
    else:
        kind = args.dataset if not sharded else ("RandomDistributed" if args.dataset == "Random" else args.dataset)
        host_keys = make_input(kind, args.dtype, n, BASE_SEED + rank)
    keys = torch_view(host_keys).to(device)
    payload = torch.arange(n, dtype=torch.int32, device=device) if args.payload else None

    from radix_sort_amd.distributed import ShardedSorter
    capacity = 2 * n if sharded else n
    if rehearsal:
        eng = reh.RehearsalEngine(args.dtype)
    else:
        eng = rsx.Engine(args.dtype, capacity, payload=args.payload, device=local_rank)
        # a real (non-null) stream: sorts, RCCL collectives and the engine's HIP events all live on it
        stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(stream)
        eng.set_stream(stream.cuda_stream)
    # timed region: HIP events bracket only the graded reorder launches (8 pairs per sort);
    # the per-phase table below comes from a fully instrumented step after it
    eng.set_option(rsx.OPT_PROFILE, 0 if args.no_events else 2)
    if args.radix_bits != 4:
        eng.set_option(rsx.OPT_RADIX_BITS, args.radix_bits)
    sorter = ShardedSorter(eng, rank, world, key_bytes * 8, dist, force_exchange=force_exchange, strategy=os.environ.get("RSX_STRATEGY", "auto"),
                           partition_bits=args.partition_bits, wave_grouping=os.environ.get("RSX_WAVE_GROUPING", "doubling"))
    if os.environ.get("RSX_PUSH_PARTS"):
        sorter.push_parts = int(os.environ["RSX_PUSH_PARTS"])
    if sharded and sorter.strategy == "waves-p2p":
        # peer-store exchange: the receive buffers are peer-visible device memory that the other ranks' scatter kernels write into
        sorter.setup_peer_exchange(capacity, device, args.payload)
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
    sync()
    eng.timings(reset=True)

    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    n_local = n
    for _ in range(args.steps):
        n_local = step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rt = eng.timings(reset=True)
    if hasattr(eng, "sync"):
        eng.sync()              # also reports a fused table scan that timed out in the last step (earlier ones: the sorter's check)
    eng.set_option(rsx.OPT_PROFILE, 1)
    sorter.record_timeline = sharded and not rehearsal
    for _ in range(2):
        step()
    sync()
    rt_all = eng.timings(reset=True)
    exchange_ms = {k: round(v, 4) for k, v in sorter.timeline_ms().items()} if sorter.record_timeline else ({} if sharded else None)
    sorter.record_timeline = False
    if args.no_events:
        rt = rt_all

    # ---- CPU baseline (rank 0; the other ranks wait for it in the collectives below) -------------------
    base_entry = base_sorted = None
    if rank == 0 and not args.no_cpu_baseline:
        lg = args.cpu_sample_log2 if args.cpu_sample_log2 is not None else (28 if world == 1 else 26)
        m = min(1 << lg, n)
        base_entry, base_sorted = cpu_baseline(host_keys[:m], whole=(m == n and world == 1))

    # ---- verification --------------------------------------------------------------------------------
    verified = False
    if not args.no_verify:
        if sharded and sorter.result_in_out:
            sync()
            got = obuf[:n_local].cpu().numpy().view(np.dtype(args.dtype))
        else:
            got = eng.download()
        ok = bool(np.all(got[:-1] <= got[1:])) and got.size == n_local
        verified = "ascending permutation of the input (xor and sum checksums)"
        if world == 1:
            u = host_keys.view(np.uint32 if key_bytes == 4 else np.uint64)
            v = got.view(u.dtype)
            ok = ok and int(np.bitwise_xor.reduce(u)) == int(np.bitwise_xor.reduce(v)) and int(u.sum(dtype=np.uint64)) == int(v.sum(dtype=np.uint64))
            if base_sorted is not None and base_sorted.size == got.size:
                # the array the CPU baseline just sorted with RadixSortCPU, key for key (src/CRadixSortTask.cpp:225-252)
                ok = ok and bool(np.array_equal(got, base_sorted))
                verified = f"bit-exact vs RadixSortCPU ({base_entry['kind']}), all {got.size} keys"
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
            verified = "ascending across ranks, permutation of the input (count, xor and sum checksums)"
            if not args.no_verify_full and n * world <= (1 << 31):
                # every rank's output to rank 0 (padded to the common capacity), which compares the
                # concatenation with a host sort of ALL the inputs, key for key
                src = obuf if sorter.result_in_out else torch_view(np.concatenate([got, np.zeros(capacity - got.size, dtype=got.dtype)])).to(device)
                sync()
                parts = [torch.empty(capacity, dtype=keys.dtype, device=device) for _ in range(world)] if rank == 0 else None
                if world > 1:
                    dist.gather(src[:capacity].contiguous(), parts, dst=0)
                else:
                    parts = [src]
                flag = torch.ones(1, dtype=torch.int64, device=device)
                if rank == 0:
                    sizes = [r[0] for r in rows]
                    cat = np.concatenate([p[:sz].cpu().numpy() for p, sz in zip(parts, sizes)]).view(np.dtype(args.dtype))
                    del parts
                    if one_dataset:
                        everything = make_shard(kind, args.dtype, 0, n * world, n * world, BASE_SEED)
                        assert np.array_equal(everything[:n], host_keys)
                    else:
                        everything = np.concatenate([host_keys] + [make_input(kind, args.dtype, n, BASE_SEED + r) for r in range(1, world)])
                    everything.sort()          # keys only: any correct sort gives the same array (numpy's default is the fast one: ~10 s for 2^30 uint32)
                    flag[0] = int(np.array_equal(cat, everything))
                    del cat, everything
                dist.broadcast(flag, src=0)
                ok = ok and bool(flag.item())
                verified = f"bit-exact vs a host sort of all {n * world} keys, gathered on rank 0"
        if not ok:
            raise SystemExit("bench: the result is not the sorted input — refusing to report a number")

    total_keys = n * world
    ms_per_step = elapsed / args.steps * 1e3
    reorder_ms = rt.reorder.avg_ms
    launches_per_step = rt.reorder.n / max(args.steps, 1)
    # algorithmic bytes of one reorder launch: n*(K+V) read + n*(K+V) written.
    # N>1: launches differ in size (the partition pass sees n keys, the local passes what arrived,
    # wave by wave on the pipelined path), so the figure is bytes of all launches / time of all launches
    passes = key_bytes * 8 // args.radix_bits
    waves = sorter.last_path in ("waves", "waves-p2p")
    if sharded:
        from radix_sort_amd.distributed import wave_groups, group_pass_units
        def launches(units):          # scatter launches of a local sort over `units` 4-bit pass units (8-bit digits: whole bytes + a nibble)
            return units if args.radix_bits == 4 else units // 2 + units % 2
        if waves:
            # the waves are sorted in groups (one by one, or doubling {0} {1} {2,3} {4..7}); a group of g of the k waves holds ~ g/k of what arrived
            k = (1 << sorter.partition_bits) // world
            local_passes = sum(launches(group_pass_units(key_bytes * 8, sorter.partition_bits, g)) * g / k for _, g in wave_groups(k, sorter.grouping))
        else:
            local_passes = launches(key_bytes * 2)
        scatter_bytes_per_step = 2.0 * (key_bytes + pay_bytes) * (n + local_passes * n_local)
        scatter_bytes = scatter_bytes_per_step / launches_per_step if launches_per_step else 0.0
    else:
        scatter_bytes = 2.0 * n_local * (key_bytes + pay_bytes)
    achieved = scatter_bytes / (reorder_ms * 1e-3) * 1e-9 if reorder_ms > 0 else 0.0

    def pow2(v):
        return f"2^{v.bit_length() - 1}" if v & (v - 1) == 0 else str(v)
    workload = f"{pow2(n)} {args.dtype}{'+u32 payload' if args.payload else ''} {kind}, {args.radix_bits}-bit digits, {passes} passes"
    if world > 1:
        shards = f"contiguous shards of one {kind} dataset" if one_dataset else f"{kind} per rank"
        workload = f"{pow2(total_keys)} {args.dtype}{'+u32 payload' if args.payload else ''} keys sharded {world}x ({pow2(n)} per GPU, {shards}), RCCL histogram all-gather + key all-to-all over xGMI, {args.radix_bits}-bit digits"
        if total_keys == 1 << 30 and args.dtype == "uint32" and not args.payload and args.radix_bits == 4 and one_dataset and kind == "Random":
            workload += " [BASELINE config 4]" if world == 8 else f" [BASELINE config 4's input and size over {world} ranks]"
    traffic, traffic_note = load_traffic(workload)
    line = {
        "metric": METRIC,
        "value": round(total_keys / (elapsed / args.steps) * 1e-6, 1),
        "unit": "Mkeys/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": {"uint32": "u32", "int32": "i32", "uint64": "u64", "int64": "i64"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": workload, "keys_per_gpu": n, "total_keys": total_keys,
                   "parallelism": "single GPU" if not sharded else (
                       f"msd-partition[{sorter.last_path}] x{world}" + (f", top {sorter.partition_bits} bits = {(1 << sorter.partition_bits) // world} waves per rank sorted {'in doubling groups' if sorter.grouping else 'one by one'}" if waves else "")
                       + (" + peer stores into the owners' receive buffers (one push + fence per wave)" if sorter.last_path == "waves-p2p" else " + all_to_all (RCCL)")
                       + " + local LSD sort"),
                   "verified": verified},
        "roofline": {
            "bound": "hbm", "kernel": "reorder_kernel" if args.radix_bits == 4 else "reorder8_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "algorithmic_bytes_per_launch": scatter_bytes, "avg_launch_ms": round(reorder_ms, 5),
            "launches_per_step": launches_per_step,
        },
        "phases_ms_per_launch": {"histogram": round(rt_all.histogram.avg_ms, 5), "scan": round(rt_all.scan.avg_ms, 5),
                                 "paste": round(rt_all.paste.avg_ms, 5), "reorder": round(rt_all.reorder.avg_ms, 5),
                                 "note": "fully instrumented steps after the timed region"},
    }
    if exchange_ms is not None:
        line["sharded_phases_ms"] = dict(exchange_ms, note="rank 0, device time between marks of one instrumented step after the timed region")
    if args.radix_bits != 4:
        line["config"]["note"] = "8-bit digits: a separately reported variant; BASELINE's configuration is 4-bit digits"
    if base_entry is not None:
        line["cpu_baseline"] = base_entry
    if rehearsal:
        line["metric"] = "REHEARSAL on a CPU test double (tests only) — not a measurement; " + METRIC
        line["value"] = None
        line["rehearsal"] = True
    elif shared_gpu:
        line["metric"] = "REHEARSAL: all ranks share ONE GPU, collectives staged through the host over gloo (tests only) — not a measurement; " + METRIC
        line["value"] = None
        line["rehearsal"] = True
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()          # nobody unmaps or frees a receive buffer a peer may still be writing into
    if getattr(sorter, "_peer", None) is not None:
        sorter.close_peer_exchange()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
