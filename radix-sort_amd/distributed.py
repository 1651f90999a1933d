"""Sharded sort across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The reference has no multi-device path ("Parallelism strategies: none"; one in-order queue on one device,
/root/reference/Common/ComputeState.cpp:88-101); this is the capability the north-star adds.  LSD passes are
not independent across shards, but a most-significant-bits partition is, so the data path is one exchange + local sorts.
The host ARITHMETIC of every path (wave layout, bucket dealing, splitters, cuts, capacity verdicts, peer access) is
radix-sort_amd/host/ShardPlanner.cpp through planner.py — the same code the C++ driver (RadixSortMultiGPU<T>) runs.

  1. Pipelined paths ("waves", tried first: 1, 2, 4, 8 or 16 ranks, the caller passes an output buffer, keys that use their
  whole bit range).  The shard is partitioned on its top B = `partition_bits` bits (1..8; default: 8 waves per rank, at least
  4 bits).  Rank r owns the k = 2^B / world consecutive buckets r*k .. r*k+k-1 and bucket r*k + w travels in WAVE w:
      count    `rsx_msd_count`: one read of the shard leaves the 2^B bucket sizes in a device row that also carries this
               rank's buffer capacities and status word; the row goes into the `all_gather` as it is (asynchronously)
      scatter  `rsx_msd_scatter` groups the shard into the staging buffer in wave-major order [wave][destination] — it needs
               only this rank's counts, so it runs WHILE the all_gather is in flight
      exchange strategy "waves": k asynchronous `all_to_all_single` calls (RCCL wants the split sizes on the host: the gathered
               table is copied there on a side stream, behind the all_gather only).
               strategy "waves-p2p": the receive buffers are peer-visible device memory; `rsx_msd_plan` computes every segment's
               place in its destination ON THE DEVICE from the gathered table (no host round trip in front of the data), and one
               `rsx_msd_push` per wave copies this rank's segments straight into the owners' buffers over xGMI on a second
               stream, each GROUP of waves (below) closed by a one-word all_reduce (its fence)
      sort     the waves are sorted in DOUBLING GROUPS {0} {1} {2,3} {4..7} ...: wave 0 as soon as it has landed (the exposed part
               of the exchange is 1/k of it), the later groups — gap-free in the receive buffer — together, at the big-sort rate, while
               the next group is on the links.  A group of 2^j aligned buckets shares the top B - j bits, so its sort
               (`rsx_sort_from_to`, straight into its place in the output) needs ceil((keybits - B + j) / 4) pass units: at B = 6 up
               to four buckets cost no more than one.  (`wave_grouping="single"` sorts every wave by itself.)

  2. Plain top-bit path (other world sizes, no output buffer, or strategy="top"): 16 buckets on the top nibble in key order
  (`rsx_partition_count` / `rsx_partition_scatter`), dealt to the ranks as contiguous ranges balanced on the global counts,
  ONE all-to-all, full local sort.  Also taken when dealing the buckets out unevenly balances what the fixed ownership of
  path 1 does not.

  3. Splitter path (when the top bits do not balance; up to 8 ranks): every rank samples 1024 of its keys
  (`rsx_sample_keys`), the samples are gathered, and world-1 quantile SPLITTERS are chosen.  Keys are bucketed as
  2 * #{splitters < key} + [key equals a splitter] (`rsx_partition_count_split` / `_scatter_split`): even buckets are the
  open intervals between splitters and move whole; odd buckets hold only keys EQUAL to a splitter, so they may be cut
  anywhere — ties are split by (rank, index), which keeps the ranks balanced (and the argsort stable) even when one key
  value is most of the input (`split_plan`).

  4. Range path (more than 8 ranks, or strategy="range"): global min/max (`rsx_key_range` + all_gather), 16 equal-width
  buckets over [lo, hi] (`rsx_partition_range`), balanced contiguous dealing, one all-to-all, full local sort.

Concatenating the ranks' outputs in rank order gives the globally sorted array; with payloads the result is the stable
argsort (chunks arrive in source-rank order and all local steps are stable).  Whether a plan fits the buffers — and whether
any rank's engine reported an error of an earlier step — is decided from GATHERED data, so every rank raises together or
goes on together: nobody is left hanging in a collective.

Nothing here touches the data on the host.  `engine` is the object that does the device work (radix_sort_amd.Engine in
production); tests inject a CPU test double through the same methods so the split/offset logic runs under gloo without a GPU.
"""
from __future__ import annotations

from .planner import (GROUP_DOUBLING, GROUP_SINGLE, MAX_SPLITTERS, PEER_ENABLE_THEN_POINTER, PEER_OPEN_IPC, CapacityError, ExchangePlan,  # noqa: F401  (re-exported)
                      balanced_owner, check_capacity, choose_splitters, group_pass_units, peer_access, plan_from_table, range_buckets, split_cuts, split_plan,
                      wave_extents, wave_groups, wave_layout)

RADIX = 16                  # buckets of the one-shot paths (top / split / range): the 16-bucket partition kernels
TOP_BITS = 4
MSD_SLOTS = 256             # bucket slots of a count row of the pipelined paths (rsx_msd_count writes all 256)
ROW_CAPS = MSD_SLOTS        # [ROW_CAPS], [ROW_CAPS + 1]: this rank's receive / output capacity in keys
ROW_STATUS = MSD_SLOTS + 2  # non-zero: this rank's engine reported an error of an earlier step
ROW_LEN = MSD_SLOTS + 3
SAMPLES_PER_RANK = 1024


class EngineStatusError(RuntimeError):
    """Some rank's engine reported an error of an earlier step (a fused table scan that timed out).  Raised by EVERY rank in
    the same step: the flag travels with the gathered counts."""


def default_partition_bits(world_size: int) -> int:
    """Eight waves per rank (the exposed first wave is 1/8 of the exchange), at least 4 bits (one LSD pass saved), at most 8."""
    return max(4, min(8, (world_size - 1).bit_length() + 3))


def bucket_owner(world_size: int) -> list[int]:
    """Bucket b (0..15, ascending key order) -> owning rank; contiguous, monotone ranges."""
    if not 1 <= world_size <= RADIX:
        raise ValueError(f"world_size must be in [1, {RADIX}], got {world_size}")
    return [b * world_size // RADIX for b in range(RADIX)]


def send_splits(bucket_offsets: list[int], world_size: int, owner: list[int] | None = None) -> list[int]:
    """Number of local keys going to each rank, from the 17 exclusive bucket offsets of a partition pass."""
    if len(bucket_offsets) != RADIX + 1:
        raise ValueError("expected 17 bucket offsets")
    owner = owner or bucket_owner(world_size)
    out = [0] * world_size
    for b in range(RADIX):
        out[owner[b]] += bucket_offsets[b + 1] - bucket_offsets[b]
    return out


def recv_splits(all_send_splits: list[list[int]], rank: int) -> list[int]:
    """Number of keys this rank receives from each source rank."""
    return [row[rank] for row in all_send_splits]


def _raise_together(statuses: list[int]) -> None:
    bad = [r for r, s in enumerate(statuses) if s]
    if bad:
        raise EngineStatusError(f"rank(s) {bad} reported an engine error of an earlier step (a fused table scan that timed out: that step's result is "
                                "undefined); every rank stops here together")


def global_key_range(local_lo: int, local_hi: int, world_size: int, dist, device, status: int = 0) -> tuple[int, int]:
    """Global [lo, hi] in unsigned sort order.  64-bit values travel as two 32-bit halves in an int64 tensor; a rank
    without keys contributes (UINT64_MAX, 0).  The rank's status word rides along (every rank raises together)."""
    if dist is None:
        _raise_together([status])
        return local_lo, local_hi
    import torch

    m = 0xFFFFFFFF
    t = torch.tensor([local_lo >> 32, local_lo & m, local_hi >> 32, local_hi & m, int(status)], dtype=torch.int64, device=device)
    gathered = torch.empty(5 * world_size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, 5).tolist()
    _raise_together([r[4] for r in rows])
    los = [(r[0] << 32) | r[1] for r in rows]
    his = [(r[2] << 32) | r[3] for r in rows]
    return min(los), max(his)


def gather_counts(counts: list[int], world_size: int, dist, device, caps: tuple[int, int] = (0, 0), status: int = 0):
    """[source rank][bucket] table of everybody's 16 bucket counts (one all_gather) and, riding in the same message, every
    rank's (receive-buffer, output-buffer) capacity in keys and status word — so that whether a plan fits, and whether
    anybody's engine is in trouble, is decided from the same data on every rank (a rank that found out alone and raised
    would leave its peers hanging in the all-to-all)."""
    if dist is None:
        _raise_together([status])
        return [list(counts)], [tuple(caps)]
    import torch

    t = torch.tensor(list(counts) + [int(caps[0]), int(caps[1]), int(status)], dtype=torch.int64, device=device)
    gathered = torch.empty(world_size * (RADIX + 3), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, RADIX + 3).tolist()
    _raise_together([r[RADIX + 2] for r in rows])
    return [r[:RADIX] for r in rows], [(r[RADIX], r[RADIX + 1]) for r in rows]


def plan_exchange(bucket_offsets: list[int], rank: int, world_size: int, dist, device, caps: tuple[int, int] = (0, 0), status: int = 0):
    """"Histogram all-to-all" of the one-shot paths: all ranks learn every rank's 16 bucket counts, deal the buckets to ranks
    in balanced contiguous ranges, and derive their send and receive split sizes.  Returns (plan, capacities of all ranks)."""
    counts = [bucket_offsets[b + 1] - bucket_offsets[b] for b in range(RADIX)]
    if dist is None:
        mine = send_splits(bucket_offsets, world_size)
        return ExchangePlan(send=mine, recv=mine, loads=[sum(mine)]), [tuple(caps)]
    table, all_caps = gather_counts(counts, world_size, dist, device, caps, status)   # [source rank][bucket]
    plan, _ = plan_from_table(table, rank, world_size)
    return plan, all_caps


def gather_samples(samples: list[int], n_local: int, world_size: int, dist, device, status: int = 0) -> tuple[list[list[int]], list[int]]:
    """Everybody's samples and shard sizes (one all_gather; 64-bit values as two int64 halves; the status word rides along)."""
    if dist is None:
        _raise_together([status])
        return [samples], [n_local]
    import torch

    m = 0xFFFFFFFF
    k = SAMPLES_PER_RANK
    padded = list(samples) + [0] * (k - len(samples))
    t = torch.tensor([n_local, len(samples), int(status)] + [v >> 32 for v in padded] + [v & m for v in padded], dtype=torch.int64, device=device)
    gathered = torch.empty(world_size * (3 + 2 * k), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, 3 + 2 * k).tolist()
    _raise_together([r[2] for r in rows])
    sizes = [r[0] for r in rows]
    out = [[(r[3 + i] << 32) | r[3 + k + i] for i in range(r[1])] for r in rows]
    return out, sizes


_PROCESS_TOKEN = None


def process_identity(device: int) -> tuple[int, int, int, int]:
    """(host hash, process token, pid, device ordinal): what a rank publishes about itself so that the others can decide how to
    reach its receive buffer (planner.peer_access).  The token is random and drawn once per process: equal pids in different
    pid namespaces or on different hosts do not pass for one process."""
    global _PROCESS_TOKEN
    import hashlib
    import os
    import secrets
    import socket
    if _PROCESS_TOKEN is None:
        _PROCESS_TOKEN = secrets.randbits(62)
    boot = ""
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            boot = f.read().strip()
    except OSError:
        pass
    host = int.from_bytes(hashlib.sha256((socket.gethostname() + "|" + boot).encode()).digest()[:8], "little") >> 2
    return host, _PROCESS_TOKEN, os.getpid(), int(device)


class ShardedSorter:
    """Per-rank driver.  Buffers are torch tensors (device memory + RCCL plumbing)."""

    STRATEGIES = ("auto", "waves", "waves-p2p", "split", "range", "top")

    def __init__(self, engine, rank: int, world_size: int, key_bits: int, dist=None, force_exchange: bool = False, strategy: str = "auto",
                 partition_bits: int | None = None, wave_grouping: str = "doubling"):
        if world_size > 1 and dist is None:
            raise ValueError("a torch.distributed module (or a stand-in with its calls) is required for world_size > 1")
        if not 0 <= rank < world_size:
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.engine = engine
        self.rank = rank
        self.world = world_size
        self.key_bits = key_bits
        self.dist = dist
        # force_exchange: run partition -> all_gather -> exchange even with one rank (the collectives then talk to self); lets a
        # 1-GPU box exercise the real RCCL call path
        self.force_exchange = force_exchange and dist is not None
        self.max_imbalance = 1.25      # the fixed bucket ownership is used when no rank would get more than this x its share
        if strategy not in self.STRATEGIES:
            raise ValueError(f"unknown strategy {strategy!r}")
        if strategy == "split" and world_size > MAX_SPLITTERS + 1:
            raise ValueError(f"the splitter path serves at most {MAX_SPLITTERS + 1} ranks")
        self.strategy = strategy
        self.can_wave = world_size in (1, 2, 4, 8, 16)
        bits = default_partition_bits(world_size) if partition_bits is None else int(partition_bits)
        if self.can_wave and not (1 <= bits <= 8 and (1 << bits) >= world_size):
            raise ValueError(f"partition_bits must be in 1..8 with 2^bits >= the world size, got {bits}")
        self.partition_bits = bits
        if wave_grouping not in ("doubling", "single"):
            raise ValueError(f"wave_grouping must be 'doubling' or 'single', got {wave_grouping!r}")
        self.grouping = GROUP_DOUBLING if wave_grouping == "doubling" else GROUP_SINGLE
        self.push_parts = 0            # workgroups per destination of a wave's push (0: the library's default)
        self.last_path = None          # "local" | "waves" | "waves-p2p" | "top" | "split" | "range" | "equal" (for tests and logs)
        self.last_imbalance = None
        # record_timeline: device-time marks around count / scatter / exchange / local sort (torch events on the current stream,
        # which must be the engine's stream); read with timeline_ms()
        self.record_timeline = False
        self._marks = []
        self._row = self._table = self._row_tail = self._table_host = self._side = self._push = None
        self._peer = None              # peer-store exchange: receive buffers of every rank as this rank addresses them (setup_peer_exchange)
        self._plain_table = None
        self._status_text = ""

    # -- peer-store exchange ------------------------------------------------------------------------------------------------
    def setup_peer_exchange(self, capacity: int, device, with_payload: bool = False) -> None:
        """Collective, once: every rank allocates a peer-visible receive buffer of `capacity` keys (and payloads), publishes who it
        is (host, process, device) and learns how to address everybody else's buffer (planner.peer_access — the decision is host
        arithmetic shared with the C++ driver): the pointer itself for a rank that is a thread of this process on this device,
        the pointer after `rsx_peer_enable` for a thread of this process on ANOTHER device, an opened IPC handle (peer access over
        xGMI) for a rank in another process of this host.  Needed by strategy "waves-p2p"."""
        import numpy as np
        import torch
        if self._peer is not None:
            self.close_peer_exchange()
        itemsize = self.key_bits // 8
        kaddr, khandle = self.engine.peer_alloc(capacity * itemsize)
        paddr, phandle = self.engine.peer_alloc(capacity * 4) if with_payload else (0, bytes(64))
        ident = process_identity(getattr(self.engine, "device", 0))
        row = list(ident) + [kaddr, paddr] + [int(v) for v in np.frombuffer(khandle, dtype=np.int64)] + [int(v) for v in np.frombuffer(phandle, dtype=np.int64)]
        rows = [row]
        if self.dist is not None:
            t = torch.tensor(row, dtype=torch.int64, device=device)
            gathered = torch.empty(self.world * len(row), dtype=torch.int64, device=device)
            self.dist.all_gather_into_tensor(gathered, t)
            rows = gathered.cpu().view(self.world, len(row)).tolist()
        access = peer_access([tuple(r[:4]) for r in rows], self.rank)
        keys, pays, opened = [], [], []
        for other, how in zip(rows, access):
            if how == PEER_OPEN_IPC:
                keys.append(self.engine.peer_open(np.array(other[6:14], dtype=np.int64).tobytes()))
                opened.append(keys[-1])
                pays.append(0)
                if with_payload:
                    pays[-1] = self.engine.peer_open(np.array(other[14:22], dtype=np.int64).tobytes())
                    opened.append(pays[-1])
            else:
                if how == PEER_ENABLE_THEN_POINTER:
                    self.engine.peer_enable(int(other[3]))
                keys.append(other[4])
                pays.append(other[5])
        self._peer = {"capacity": capacity, "keys": keys, "pays": pays, "opened": opened, "mine": (kaddr, paddr), "payload": with_payload, "access": access,
                      "keys_dev": torch.tensor(keys, dtype=torch.int64, device=device), "pays_dev": torch.tensor(pays, dtype=torch.int64, device=device),
                      "fence": torch.zeros(1, dtype=torch.int32, device=device)}

    def close_peer_exchange(self) -> None:
        """Collective: unmaps the other ranks' buffers, waits until every rank has done so, then frees this rank's."""
        if self._peer is None:
            return
        for p in self._peer["opened"]:
            self.engine.peer_close(p)
        barrier = getattr(self.dist, "barrier", None)
        if barrier is not None:
            barrier()                      # nobody still has this rank's buffer mapped (or is writing into it) when it is freed
        for p in self._peer["mine"]:
            if p:
                self.engine.peer_free(p)
        self._peer = None

    # -- plumbing -----------------------------------------------------------------------------------------------------------
    def _mark(self, label):
        if self.record_timeline:
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((label, ev))

    def timeline_ms(self) -> dict:
        """Milliseconds between consecutive marks of the last sort (synchronises)."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for (_, a), (label, b) in zip(self._marks, self._marks[1:]):
            out[label] = out.get(label, 0.0) + a.elapsed_time(b)
        return out

    def _bind_stream(self, keys) -> None:
        """The collectives and `work.wait()` order against torch's CURRENT stream only; every device step of the engine is
        asynchronous on the engine's stream.  The two must be the same stream or the all-to-all reads `staging` before the
        scatter has written it.  A real engine is re-bound to the current stream when it sits on another one (test doubles
        without streams are left alone)."""
        get = getattr(self.engine, "get_stream", None)
        if get is None or not getattr(keys, "is_cuda", False):
            return
        import torch
        cur = torch.cuda.current_stream(keys.device).cuda_stream
        if get() != cur:
            self.engine.set_stream(cur)

    def _engine_status(self) -> int:
        """1 if the engine reports an error of a step that has already finished (read without synchronising and cleared:
        rsx_check_status); never raises — the flag travels in the gathered row and every rank raises together."""
        check = getattr(self.engine, "check_status", None)
        if check is None:
            return 0
        try:
            check()
            return 0
        except RuntimeError as exc:
            self._status_text = str(exc)
            return 1

    def sort(self, keys, staging, recv, payload=None, staging_payload=None, recv_payload=None, out=None, out_payload=None):
        """keys: this rank's shard (device tensor, left untouched).  staging: same length as keys (bucket-grouped copy).
        recv: capacity for the incoming keys (not used by "waves-p2p", which receives into its peer-visible buffer).
        out (optional, same capacity as recv): enables the pipelined paths, whose result lands there.  Returns the number of
        keys this rank ends up with; `result_in_out` says whether they are in `out` or inside the engine (engine.download /
        copy_result)."""
        n = keys.numel()
        self.result_in_out = False
        self._marks = []
        self._bind_stream(keys)
        # what this rank can receive / hold at the end, in keys: with a payload the smaller of the key and the payload buffer
        # (a short payload buffer would otherwise fail in the all-to-all on ONE rank and leave the others hanging)
        def cap(a, b):
            if a is None or (payload is not None and b is None):
                return 0
            return a.numel() if payload is None else min(a.numel(), b.numel())
        self._caps = (cap(recv, recv_payload), cap(out, out_payload))
        if self.strategy == "waves-p2p" and self._peer is not None:
            self._caps = (self._peer["capacity"], self._caps[1])      # what this rank receives into is its peer-visible buffer
        self._status = self._engine_status()
        self._mark("start")
        if self.world == 1 and not self.force_exchange:
            _raise_together([self._status])
            self.engine.sort_from(keys.data_ptr(), n, payload.data_ptr() if payload is not None else None)
            self.last_path = "local"
            return n
        pay_in = payload.data_ptr() if payload is not None else None
        pay_st = staging_payload.data_ptr() if staging_payload is not None else None
        can_pipeline = out is not None and self.can_wave and (payload is None or out_payload is not None)
        self._plain_table = None
        if self.strategy == "waves-p2p":
            if not can_pipeline or self._peer is None or (payload is not None and not self._peer["payload"]) or staging is None:
                raise ValueError("strategy 'waves-p2p' needs a staging and an output buffer, 1, 2, 4, 8 or 16 ranks and setup_peer_exchange() "
                                 "(with a payload buffer if a payload is carried)")
            return self._sort_in_waves(keys, n, staging, None, payload, staging_payload, None, out, out_payload, pay_in, pay_st, p2p=True)
        if self.strategy == "waves" and not can_pipeline:
            raise ValueError("strategy 'waves' needs an output buffer and 1, 2, 4, 8 or 16 ranks")
        if self.strategy == "waves" or (self.strategy == "auto" and can_pipeline):
            done = self._sort_in_waves(keys, n, staging, recv, payload, staging_payload, recv_payload, out, out_payload, pay_in, pay_st, p2p=False)
            if done is not None:
                return done
            self._status = 0               # the wave attempt's gathered row has already carried (and cleared) it
        # the counts of the wave attempt say whether dealing the top-nibble buckets out unevenly could work
        top_worth_a_try = self._plain_table is None or plan_from_table(self._plain_table, self.rank, self.world)[1] <= self.max_imbalance
        if self.strategy == "top" or (self.strategy == "auto" and top_worth_a_try):
            # buckets on the top 4 key bits, if they deal out evenly
            top_shift = self.key_bits - TOP_BITS
            table, caps = gather_counts(self.engine.partition_count(keys.data_ptr(), n, top_shift, TOP_BITS), self.world, self.dist, keys.device, self._caps, self._status)
            self._status = 0
            plan, imbalance = plan_from_table(table, self.rank, self.world)
            self._mark("count+plan")
            if imbalance <= self.max_imbalance or self.strategy == "top":
                check_capacity(plan.loads, caps, need_out=False)
                self.engine.partition_scatter(keys.data_ptr(), n, top_shift, TOP_BITS, staging.data_ptr(), pay_in, pay_st)
                self._mark("scatter")
                self.last_path, self.last_imbalance = "top", imbalance
                return self._exchange_and_sort(plan, n, staging, recv, payload, staging_payload, recv_payload)
        if self.strategy == "split" or (self.strategy == "auto" and self.world <= MAX_SPLITTERS + 1):
            return self._sort_by_splitters(keys, n, staging, recv, payload, staging_payload, recv_payload, pay_in, pay_st)
        lo, hi = self.engine.key_range(keys.data_ptr(), n)
        lo, hi = global_key_range(lo, hi, self.world, self.dist, keys.device, self._status)
        self._status = 0
        if lo >= hi:
            # every key everywhere is the same value (or there are no keys): rank-order concatenation is already sorted and
            # stable, nothing has to move
            self.engine.sort_from(keys.data_ptr(), n, payload.data_ptr() if payload is not None else None)
            self.last_path = "equal"
            return n
        self.last_path = "range"
        shift, mul = range_buckets(lo, hi, self.key_bits)
        self._mark("count+plan")
        offs = self.engine.partition_range(
            keys.data_ptr(), n, lo, shift, mul, staging.data_ptr(),
            payload.data_ptr() if payload is not None else None,
            staging_payload.data_ptr() if staging_payload is not None else None)
        self._mark("scatter")
        plan, caps = plan_exchange(offs, self.rank, self.world, self.dist, keys.device, self._caps, self._status)
        check_capacity(plan.loads, caps, need_out=False)
        self._mark("count+plan")
        return self._exchange_and_sort(plan, n, staging, recv, payload, staging_payload, recv_payload)

    # -- pipelined paths ----------------------------------------------------------------------------------------------------
    def _rows(self, device):
        """This rank's count row (device; its tail = capacities + status word, rewritten only when they change) and the gathered table."""
        import torch
        if self._row is None or self._row.device != device:
            self._row = torch.zeros(ROW_LEN, dtype=torch.int64, device=device)
            self._table = torch.zeros(self.world * ROW_LEN, dtype=torch.int64, device=device)
            self._row_tail = None
            self._table_host = None
        tail = (int(self._caps[0]), int(self._caps[1]), int(self._status))
        if self._row_tail != tail:
            self._row[ROW_CAPS:] = torch.tensor(tail, dtype=torch.int64)
            self._row_tail = tail
        return self._row, self._table

    def _table_to_host(self, work, table) -> list[list[int]]:
        """The gathered table on the host, behind the all_gather ONLY: the copy runs on a side stream, so the scatter that is in
        flight on the engine's stream is not waited for."""
        if not getattr(table, "is_cuda", False):
            if work is not None:
                work.wait()
            return table.view(self.world, ROW_LEN).tolist()
        import torch
        if self._side is None:
            self._side = torch.cuda.Stream(device=table.device)
        if self._table_host is None:
            self._table_host = torch.empty(table.numel(), dtype=table.dtype, pin_memory=True)
        with torch.cuda.stream(self._side):
            if work is not None:
                work.wait()
            self._table_host.copy_(table, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        done.synchronize()
        return self._table_host.view(self.world, ROW_LEN).tolist()

    def _sort_in_waves(self, keys, n, staging, recv, payload, staging_payload, recv_payload, out, out_payload, pay_in, pay_st, p2p):
        """Pipelined paths.  Returns None (nothing has moved between ranks yet) when the fixed bucket ownership would leave a rank
        with more than max_imbalance x its share or somebody's buffers are too small, and the strategy allows another path."""
        world, bits = self.world, self.partition_bits
        nb = 1 << bits
        k = nb // world
        itemsize = self.key_bits // 8
        groups = wave_groups(k, self.grouping) if self.can_wave else []
        row, table = self._rows(keys.device)
        self.engine.msd_count(keys.data_ptr(), n, bits, world, row.data_ptr())
        self._mark("count")
        # (the all_gather is also the step's opening barrier: it completes only once every rank has enqueued its own, behind the local
        # sorts of its previous step — nobody is still reading a receive buffer this step is about to write into)
        work = self.dist.all_gather_into_tensor(table, row, async_op=True)
        self.engine.msd_scatter(keys.data_ptr(), n, staging.data_ptr(), pay_in, pay_st)          # needs only this rank's counts: runs beside the all_gather
        self._mark("scatter")
        if p2p:
            return self._finish_waves_p2p(work, table, keys, staging, staging_payload, payload, out, out_payload, k, groups, itemsize)
        rows = self._table_to_host(work, table)
        _raise_together([r[ROW_STATUS] for r in rows])
        counts = [r[:nb] for r in rows]
        caps = [(r[ROW_CAPS], r[ROW_CAPS + 1]) for r in rows]
        start, _, loads = wave_layout(counts, world, nb, 4, self.grouping)
        imbalance = max(loads) / max(1.0, sum(loads) / world)
        fits = True
        try:
            check_capacity(loads, caps, need_out=True, slack=4 * k)     # each wave starts 16-byte aligned in recv
        except CapacityError:
            if self.strategy == "waves":
                raise                                                   # on every rank alike
            fits = False
        if (imbalance > self.max_imbalance and self.strategy != "waves") or not fits:
            # same decision on every rank: it only depends on the gathered table.  Leave the top-nibble counts for the caller's next decision
            if bits >= TOP_BITS:
                g = 1 << (bits - TOP_BITS)
                self._plain_table = [[sum(c[b * g:(b + 1) * g]) for b in range(RADIX)] for c in counts]
            self._mark("plan")
            return None
        # The waves are handed to the collective stream from a SIDE stream that waits for the scatter only (a collective orders itself behind
        # torch's current stream: issued from the engine's stream, wave w + 1 would wait for the local sort of wave w - 1): the host never stands
        # between the device and its next kernel, the links and the CUs work side by side.
        mine = counts[self.rank]
        on_device = getattr(keys, "is_cuda", False)
        if on_device:
            import torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=keys.device)
            self._side.wait_stream(torch.cuda.current_stream(keys.device))          # the staging buffer is complete
        state = {"send_at": 0, "next": 0}
        pending = []

        def issue():
            w = state["next"]
            state["next"] += 1
            send = [mine[d * k + w] for d in range(world)]
            rcv = [counts[s][self.rank * k + w] for s in range(world)]
            n_send, n_recv, at, send_at = sum(send), sum(rcv), start[self.rank][w], state["send_at"]
            state["send_at"] += n_send

            def call():
                works = [self.dist.all_to_all_single(recv[at:at + n_recv], staging[send_at:send_at + n_send], rcv, send, async_op=True)]
                if payload is not None:
                    works.append(self.dist.all_to_all_single(recv_payload[at:at + n_recv], staging_payload[send_at:send_at + n_send], rcv, send, async_op=True))
                return works
            if on_device:
                with torch.cuda.stream(self._side):
                    works = call()
            else:
                works = call()
            pending.append((works, at, n_recv))

        # issue order: the first group's waves, (below) the first group's sort, then EVERY remaining wave — the exchange runs back to back, as
        # early as the links allow — then the remaining sorts
        while state["next"] < groups[0][0] + groups[0][1]:
            issue()
        self._mark("plan")
        done = 0
        for g, (first, nwaves) in enumerate(groups):
            if g == 1:
                while state["next"] < k:
                    issue()
            n_group = 0
            for w in range(first, first + nwaves):
                works, _, n_recv = pending[w]
                for work in works:
                    if work is not None:
                        work.wait()                      # the engine's stream waits for the waves of this group only
                n_group += n_recv
            self._mark("wait")
            if n_group:
                at = pending[first][1]
                self.engine.sort_from_to(
                    recv[at:].data_ptr(), n_group, 0, group_pass_units(self.key_bits, bits, nwaves), out[done:].data_ptr(),
                    recv_payload[at:].data_ptr() if payload is not None else None,
                    out_payload[done:].data_ptr() if payload is not None else None)
            self._mark("local_sort")
            done += n_group
        self.last_path, self.last_imbalance, self.result_in_out = "waves", imbalance, True
        return done

    def _finish_waves_p2p(self, work, table, keys, staging, staging_payload, payload, out, out_payload, k, groups, itemsize):
        """The exchange by peer stores: plan on the device, one push + fence per wave on a second stream, the local sort of wave w on
        the engine's stream as soon as its fence has passed.  The host blocks once, for the plan's copy (its own wave sizes: the sort
        launches need them), while the pushes are already queued."""
        import torch
        world = self.world
        peer = self._peer
        if self._push is None:
            self._push = torch.cuda.Stream(device=keys.device)
        fences = []
        closes_group = {first + nwaves - 1 for first, nwaves in groups}

        def push(w):
            with torch.cuda.stream(self._push):
                self.engine.msd_push(w, staging.data_ptr(), peer["keys_dev"].data_ptr(), staging_payload.data_ptr() if payload is not None else None,
                                     peer["pays_dev"].data_ptr() if payload is not None else None, self.push_parts, self._push.cuda_stream)
                # one fence per GROUP, behind its last wave: every rank's pushes of the group have finished, what this rank received of it is complete
                fences.append(self.dist.all_reduce(peer["fence"], async_op=True) if w in closes_group else None)

        with torch.cuda.stream(self._push):
            work.wait()
            self.engine.msd_plan(table.data_ptr(), ROW_LEN, ROW_CAPS, self.rank, self._push.cuda_stream, self.grouping)
        for w in range(groups[0][0] + groups[0][1]):      # the first group now; the rest right after the first group's sort has been enqueued
            push(w)
        wave_start, wave_count, loads, verdict = self.engine.msd_plan_wait(k, world)
        self.last_imbalance = max(loads) / max(1.0, sum(loads) / world)
        if verdict:
            # every rank computed the same verdict from the same table; the pushes wrote nothing
            _raise_together([(verdict >> (32 + r)) & 1 for r in range(world)])
            bad = [r for r in range(world) if (verdict >> r) & 1]
            raise CapacityError(f"rank {bad[0]} would receive {loads[bad[0]]} keys but its buffers are too small (ranks {bad})")
        self._mark("plan")
        mine_k, mine_p = peer["mine"]
        done = 0
        bits = self.partition_bits
        for g, (first, nwaves) in enumerate(groups):
            if g == 1:
                for w in range(len(fences), k):          # every remaining wave now: the pushes run back to back beside the sorts
                    push(w)
            for w in range(first, first + nwaves):
                if fences[w] is not None:
                    fences[w].wait()                     # the engine's stream waits for the waves of this group only
            self._mark("fence")
            n_group = sum(wave_count[first:first + nwaves])
            if n_group:
                at = wave_start[first]
                self.engine.sort_from_to(
                    mine_k + at * itemsize, n_group, 0, group_pass_units(self.key_bits, bits, nwaves), out[done:].data_ptr(),
                    mine_p + at * 4 if payload is not None else None,
                    out_payload[done:].data_ptr() if payload is not None else None)
            self._mark("local_sort")
            done += n_group
        self.last_path, self.result_in_out = "waves-p2p", True
        return done

    # -- one-shot paths -----------------------------------------------------------------------------------------------------
    def _sort_by_splitters(self, keys, n, staging, recv, payload, staging_payload, recv_payload, pay_in, pay_st):
        count = min(SAMPLES_PER_RANK, n)
        mine = self.engine.sample_keys(keys.data_ptr(), n, count) if count else []
        samples, sizes = gather_samples(mine, n, self.world, self.dist, keys.device, self._status)
        self._status = 0
        splitters = choose_splitters(samples, sizes, self.world)
        if not splitters:              # nobody has keys
            self.engine.sort_from(keys.data_ptr(), n, pay_in)
            self.last_path = "equal"
            return n
        counts = self.engine.partition_count_split(keys.data_ptr(), n, splitters)
        counts = counts + [0] * (RADIX - len(counts))
        table, caps = gather_counts(counts, self.world, self.dist, keys.device, self._caps, self._status)
        self._status = 0
        table = [row[:2 * len(splitters) + 1] for row in table]
        plan, imbalance = split_plan(table, self.rank, self.world)
        check_capacity(plan.loads, caps, need_out=False)
        self._mark("count+plan")
        self.engine.partition_scatter_split(keys.data_ptr(), n, staging.data_ptr(), pay_in, pay_st)
        self._mark("scatter")
        self.last_path, self.last_imbalance = "split", imbalance
        return self._exchange_and_sort(plan, n, staging, recv, payload, staging_payload, recv_payload)

    def _exchange_and_sort(self, plan, n, staging, recv, payload, staging_payload, recv_payload):
        if plan.n_recv > recv.numel() or (payload is not None and plan.n_recv > recv_payload.numel()):
            # unreachable after check_capacity (which ran on every rank, on gathered data, before anything moved): a bug, not an input
            raise RuntimeError(f"rank {self.rank}: the exchange plan delivers {plan.n_recv} keys into a receive buffer of {recv.numel()}")
        self.dist.all_to_all_single(recv[:plan.n_recv], staging[:n], plan.recv, plan.send)
        if payload is not None:
            self.dist.all_to_all_single(recv_payload[:plan.n_recv], staging_payload[:n], plan.recv, plan.send)
        self._mark("all_to_all")
        self.engine.sort_from(recv.data_ptr(), plan.n_recv, recv_payload.data_ptr() if payload is not None else None)
        self._mark("local_sort")
        return plan.n_recv
