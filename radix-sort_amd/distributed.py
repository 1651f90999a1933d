"""Sharded sort across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The reference has no multi-device path ("Parallelism strategies: none"); this
is the capability the north-star adds.  LSD passes are not independent
across shards, but a most-significant-bits partition is, so the data path is:

  1. Pipelined path, tried first (1, 2, 4, 8 or 16 ranks, caller passes an output buffer; keys that use
  their whole bit range, e.g. uniform random): every rank counts its keys by the top 4 key bits in
  WAVE-MAJOR order (`rsx_partition_count_waves`): rank r owns buckets r*k .. r*k+k-1 (k = 16/world) and
  wave w holds bucket r*k+w of every rank r.  The counts are exchanged (`all_gather`), and if no rank
  would receive more than 1.25x its share the keys are grouped in that order (`_scatter_waves`), so each
  wave is contiguous and in rank order.  The k waves leave as k asynchronous all-to-alls; while wave
  w+1 is on the links, wave w — whose keys at a rank all share the top nibble — is sorted with one LSD
  pass fewer (`rsx_sort_from_to`) straight into its place in the output.  All but the first wave of
  the exchange hides behind the local sorts, and partition + 7 passes is the single-GPU pass count.

  1b. Peer-store variant of 1 (strategy="waves-p2p", selectable; RCCL stays the default until a scaling curve exists): the
  receive buffers are peer-visible device memory (`rsx_peer_alloc`, mapped by the other ranks once through IPC handles), and the
  wave-major scatter writes every bucket STRAIGHT to its place in the owner's receive buffer (`rsx_partition_scatter_waves_peer`;
  the places follow from the gathered count table, `wave_layout`).  No staging write, no re-read, no all-to-all launch: the
  exchange is the scatter kernel's stores over xGMI, closed by one tiny all_reduce on the stream (every rank's scatter has
  finished before anybody sorts what it received).

  2. Plain top-bit path (other world sizes, no output buffer, or strategy="top"): the same 16 buckets
  in key order (`rsx_partition_count` / `rsx_partition_scatter`), dealt to the ranks as contiguous
  ranges balanced on the global counts, ONE all-to-all, full local sort.  Also taken when dealing the
  buckets out unevenly balances what the fixed ownership of path 1 does not.

  3. Splitter path (when the top bits do not balance; up to 8 ranks): every rank samples 1024 of its keys
  (`rsx_sample_keys`), the samples are gathered, and world-1 quantile SPLITTERS are chosen.  Keys
  are bucketed as 2 * #{splitters < key} + [key equals a splitter] (`rsx_partition_count_split` /
  `rsx_partition_scatter_split`): even buckets are the open intervals between splitters and move
  whole; odd buckets hold only keys EQUAL to a splitter, so they may be cut anywhere — ties are
  split by (rank, index), which keeps the ranks balanced (and the argsort stable) even when one
  key value is most of the input (`split_plan`).

  4. Range path (more than 8 ranks, or strategy="range"):

  0. every rank finds the min and max of its keys (`rsx_key_range`, one read) and the ranks
     agree on the global range [lo, hi] (`all_gather` of 4 words).  If lo == hi all keys are
     equal and nothing needs to move;
  1. every rank groups its shard into 16 equal-width buckets over [lo, hi] —
     bucket = ((key ^ sign) - lo) >> shift, a monotone function of the key — with ONE stable
     pass of the same histogram/scan/reorder kernels (C ABI `rsx_partition_range`), which
     also yields the 16 bucket sizes.  (Buckets on the top 4 key BITS would put small-range
     or sorted inputs, e.g. `Range`, on a single rank);
  2. "histogram all-to-all": all ranks exchange their 16 bucket counts
     (`all_gather`, 16 x int64 per rank — latency-bound, KBs);
  3. buckets are dealt to ranks as contiguous ranges balanced on the global counts
     (`balanced_owner`), so each rank's
     outgoing data is already contiguous per destination; `all_to_all_single` with
     split sizes moves the keys (and payloads) — every GPU talks to every peer over its
     own xGMI link at once, which suits the point-to-point fabric (a ring would be
     per-link bound);
  4. every rank runs the ordinary single-GPU LSD sort on what it received.
Concatenating the ranks' outputs in rank order gives the globally sorted array; with
payloads the result is the stable argsort (chunks arrive in source-rank order and both
local steps are stable).

Nothing here touches the data on the host.  `engine` is the object that does the device
work (radix_sort_amd.Engine in production); tests inject a CPU test double through the
same methods so the split/offset logic runs under gloo without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

RADIX = 16          # buckets of the partition pass = top 4 bits
PARTITION_BITS = 4


def bucket_owner(world_size: int) -> list[int]:
    """Bucket b (0..15, ascending key order) -> owning rank; contiguous, monotone ranges.
    8 ranks: two buckets each; 2 ranks: eight each; any world_size <= 16 works."""
    if not 1 <= world_size <= RADIX:
        raise ValueError(f"world_size must be in [1, {RADIX}], got {world_size}")
    return [b * world_size // RADIX for b in range(RADIX)]


def send_splits(bucket_offsets: list[int], world_size: int, owner: list[int] | None = None) -> list[int]:
    """Number of local keys going to each rank, from the 17 exclusive bucket offsets of the
    partition pass."""
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


@dataclass
class ExchangePlan:
    send: list[int]
    recv: list[int]
    loads: list[int] | None = None      # keys every rank ends up with (same list on all ranks)

    @property
    def n_recv(self) -> int:
        return sum(self.recv)


def global_key_range(local_lo: int, local_hi: int, world_size: int, dist, device) -> tuple[int, int]:
    """Step 0: global [lo, hi] in unsigned sort order.  64-bit values travel as two 32-bit
    halves in an int64 tensor; a rank without keys contributes (UINT64_MAX, 0)."""
    if dist is None:
        return local_lo, local_hi
    import torch

    m = 0xFFFFFFFF
    t = torch.tensor([local_lo >> 32, local_lo & m, local_hi >> 32, local_hi & m], dtype=torch.int64, device=device)
    gathered = torch.empty(4 * world_size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, 4).tolist()
    los = [(r[0] << 32) | r[1] for r in rows]
    his = [(r[2] << 32) | r[3] for r in rows]
    return min(los), max(his)


def range_buckets(lo: int, hi: int, key_bits: int) -> tuple[int, int]:
    """(shift, mul) of the 16 equal-width buckets over [lo, hi] (C ABI rsx_partition_range):
    bucket(x) = mulhi(x, mul) with mul = floor(16 * 2^key_bits / (hi - lo + 1)); ranges of at
    most 16 values use bucket(x) = x (shift 0, mul 0)."""
    span1 = hi - lo + 1
    if span1 <= RADIX:
        return 0, 0
    mul = (RADIX << key_bits) // span1
    assert mul < (1 << key_bits) and ((hi - lo) * mul) >> key_bits < RADIX
    return 0, mul


def balanced_owner(global_counts: list[int], world_size: int) -> list[int]:
    """Bucket -> rank as contiguous ranges cut where the running total crosses k/world of all
    keys (every rank computes the same map from the same gathered counts)."""
    total = sum(global_counts)
    owner, run, rank = [], 0, 0
    for c in global_counts:
        # move on to the next rank once this one has its share, judged at the bucket's midpoint
        while rank < world_size - 1 and (run + c / 2) * world_size >= (rank + 1) * total and total > 0:
            rank += 1
        owner.append(rank)
        run += c
    return owner


def gather_counts(counts: list[int], world_size: int, dist, device, caps: tuple[int, int] = (0, 0)):
    """[source rank][bucket] table of everybody's 16 bucket counts (one all_gather) and, riding in the
    same message, every rank's (receive-buffer, output-buffer) capacity in keys — so that whether a
    plan fits is decided from the same data on every rank (a rank that found out alone and raised
    would leave its peers hanging in the all-to-all)."""
    if dist is None:
        return [list(counts)], [tuple(caps)]
    import torch

    t = torch.tensor(list(counts) + [int(caps[0]), int(caps[1])], dtype=torch.int64, device=device)
    gathered = torch.empty(world_size * (RADIX + 2), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, RADIX + 2).tolist()
    return [r[:RADIX] for r in rows], [(r[RADIX], r[RADIX + 1]) for r in rows]


class CapacityError(RuntimeError):
    """Some rank's buffers cannot hold what the exchange plan sends it.  Raised by EVERY rank, before
    any key moves (the verdict only depends on gathered data)."""


def check_capacity(loads: list[int], caps: list[tuple[int, int]], need_out: bool, slack: int = 0) -> None:
    for r, (load, (recv_cap, out_cap)) in enumerate(zip(loads, caps)):
        if load + slack > recv_cap or (need_out and load > out_cap):
            raise CapacityError(f"rank {r} would receive {load} keys but its buffers hold {recv_cap} (receive) / {out_cap} (output)")


def plan_from_table(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan from the gathered count table plus the resulting imbalance
    (largest rank load / ideal load)."""
    totals = [sum(row[b] for row in table) for b in range(RADIX)]
    owner = balanced_owner(totals, world_size)
    sends = [[sum(row[b] for b in range(RADIX) if owner[b] == dst) for dst in range(world_size)] for row in table]
    loads = [sum(s[dst] for s in sends) for dst in range(world_size)]
    ideal = max(1.0, sum(totals) / world_size)
    return ExchangePlan(send=sends[rank], recv=recv_splits(sends, rank), loads=loads), max(loads) / ideal


def plan_exchange(bucket_offsets: list[int], rank: int, world_size: int, dist, device, caps: tuple[int, int] = (0, 0)):
    """Step 2 of the module docstring ("histogram all-to-all"): all ranks learn every rank's 16
    bucket counts, deal the buckets to ranks in balanced contiguous ranges, and derive their
    send and receive split sizes.  Returns (plan, capacities of all ranks)."""
    counts = [bucket_offsets[b + 1] - bucket_offsets[b] for b in range(RADIX)]
    if dist is None:
        mine = send_splits(bucket_offsets, world_size)
        return ExchangePlan(send=mine, recv=mine, loads=[sum(mine)]), [tuple(caps)]
    table, all_caps = gather_counts(counts, world_size, dist, device, caps)   # [source rank][bucket]
    plan, _ = plan_from_table(table, rank, world_size)
    return plan, all_caps


def wave_layout(table: list[list[int]], world_size: int) -> tuple[list[list[int]], list[list[list[int]]], list[int]]:
    """Where everything lands in the receive buffers of the pipelined paths, from the gathered [source][wave * world + dest]
    count table: at destination d the waves follow each other, each starting on a 16-byte boundary (4 keys: the local sort
    loads 16 bytes per lane), and inside a wave the sources follow each other in rank order.  Returns
    (start[d][w] — first slot of wave w at destination d, offset[d][w][s] — first slot of source s's keys in it,
    load[d] — keys destination d ends up with)."""
    k = RADIX // world_size
    start, offset, load = [], [], []
    for d in range(world_size):
        at, st, of, total = 0, [], [], 0
        for w in range(k):
            at = (at + 3) & ~3
            st.append(at)
            row = []
            for src in range(world_size):
                row.append(at)
                at += table[src][w * world_size + d]
                total += table[src][w * world_size + d]
            of.append(row)
        start.append(st)
        offset.append(of)
        load.append(total)
    return start, offset, load


SAMPLES_PER_RANK = 1024
MAX_SPLITTERS = 7           # 2*7+1 = 15 buckets fit the 16-bucket kernels


def choose_splitters(samples: list[list[int]], shard_sizes: list[int], world_size: int) -> list[int]:
    """world_size-1 weighted quantiles of the gathered samples (each of rank r's samples stands
    for shard_sizes[r] / len(samples[r]) keys), deduplicated and increasing.  Values are in
    unsigned sort order.  May return fewer than world_size-1 (down to none, if no rank has keys)."""
    weighted = []
    for vals, n in zip(samples, shard_sizes):
        if n > 0 and vals:
            weighted.extend((v, n / len(vals)) for v in vals)
    if not weighted:
        return []
    weighted.sort(key=lambda t: t[0])
    total = sum(w for _, w in weighted)
    out, run, k = [], 0.0, 1
    for v, w in weighted:
        run += w
        while k < world_size and run * world_size >= k * total:
            if not out or out[-1] != v:
                out.append(v)
            k += 1
    return out[:MAX_SPLITTERS]


def split_cuts(totals: list[int], world_size: int) -> list[int]:
    """Global positions (in bucket-major, rank-major, index order) where one rank's share ends
    and the next begins: world_size+1 monotone values from 0 to the number of keys.  The ideal cut
    k*total/world is kept when it falls inside an odd ("equal to a splitter") bucket and moved to
    the nearer end of the bucket when it falls inside an even one, which cannot be cut."""
    total = sum(totals)
    starts = [0]
    for c in totals:
        starts.append(starts[-1] + c)
    cuts = [0]
    for k in range(1, world_size):
        ideal = k * total // world_size
        cut = ideal
        for b, c in enumerate(totals):
            lo, hi = starts[b], starts[b + 1]
            if lo < ideal < hi:
                if b % 2 == 0:
                    cut = lo if ideal - lo <= hi - ideal else hi
                break
        cuts.append(max(cut, cuts[-1]))
    cuts.append(total)
    return cuts


def split_plan(table: list[list[int]], rank: int, world_size: int) -> tuple[ExchangePlan, float]:
    """Exchange plan of the splitter path from the [source rank][bucket] count table.  Source r's
    keys of bucket b occupy global positions start_b + sum(table[r'][b] for r' < r) onwards; each
    rank sends to destination d the part of its keys inside [cut_d, cut_d+1) — contiguous in its
    bucket-grouped staging buffer and in destination order."""
    nb = len(table[0])
    totals = [sum(row[b] for row in table) for b in range(nb)]
    cuts = split_cuts(totals, world_size)
    sends = [[0] * world_size for _ in table]
    pos = 0
    for b in range(nb):
        for r, row in enumerate(table):
            lo, hi = pos, pos + row[b]
            for d in range(world_size):
                a, z = max(lo, cuts[d]), min(hi, cuts[d + 1])
                if z > a:
                    sends[r][d] += z - a
            pos = hi
    loads = [cuts[d + 1] - cuts[d] for d in range(world_size)]
    ideal = max(1.0, sum(totals) / world_size)
    return ExchangePlan(send=sends[rank], recv=recv_splits(sends, rank), loads=loads), max(loads) / ideal


def gather_samples(samples: list[int], n_local: int, world_size: int, dist, device) -> tuple[list[list[int]], list[int]]:
    """Everybody's samples and shard sizes (one all_gather; 64-bit values as two int64 halves)."""
    if dist is None:
        return [samples], [n_local]
    import torch

    m = 0xFFFFFFFF
    k = SAMPLES_PER_RANK
    padded = list(samples) + [0] * (k - len(samples))
    t = torch.tensor([n_local, len(samples)] + [v >> 32 for v in padded] + [v & m for v in padded], dtype=torch.int64, device=device)
    gathered = torch.empty(world_size * (2 + 2 * k), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    rows = gathered.cpu().view(world_size, 2 + 2 * k).tolist()
    sizes = [r[0] for r in rows]
    out = [[(r[2 + i] << 32) | r[2 + k + i] for i in range(r[1])] for r in rows]
    return out, sizes


class ShardedSorter:
    """Per-rank driver.  Buffers are torch tensors (device memory + RCCL plumbing)."""

    def __init__(self, engine, rank: int, world_size: int, key_bits: int, dist=None, force_exchange: bool = False, strategy: str = "auto"):
        if world_size > 1 and dist is None:
            raise ValueError("a torch.distributed module (or a stand-in with its calls) is required for world_size > 1")
        if not 0 <= rank < world_size:
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.engine = engine
        self.rank = rank
        self.world = world_size
        self.key_bits = key_bits
        self.dist = dist
        # force_exchange: run partition -> all_gather -> all_to_all even with one rank (the
        # collectives then talk to self); lets a 1-GPU box exercise the real RCCL call path
        self.force_exchange = force_exchange and dist is not None
        self.max_imbalance = 1.25      # top-bit buckets are used when no rank would get more than this x its share
        # general path when the top bits do not balance: "split" (sampled splitters, <= 8 ranks),
        # "range" (equal-width buckets over the global key range), "auto" = split where possible
        if strategy not in ("auto", "waves", "waves-p2p", "split", "range", "top"):
            raise ValueError(f"unknown strategy {strategy!r}")
        if strategy == "split" and world_size > MAX_SPLITTERS + 1:
            raise ValueError(f"the splitter path serves at most {MAX_SPLITTERS + 1} ranks")
        self.strategy = strategy
        self.last_path = None          # "local" | "top" | "split" | "range" | "equal" (for tests and logs)
        self.last_imbalance = None
        # record_timeline: device-time marks around plan / scatter / exchange / local sort (torch
        # events on the current stream, which must be the engine's stream); read with timeline_ms()
        self.record_timeline = False
        self._marks = []
        self._count_row = self._count_table = self._count_row_caps = None      # device row of the pipelined path's counts (+ capacities)
        self._peer = None              # peer-store exchange: receive buffers of every rank as this rank addresses them (setup_peer_exchange)

    # -- peer-store exchange ------------------------------------------------------------------------------------------------
    def setup_peer_exchange(self, capacity: int, device, with_payload: bool = False) -> None:
        """Collective, once: every rank allocates a peer-visible receive buffer of `capacity` keys (and payloads) and learns
        how to address everybody else's — the pointer itself for ranks that are threads of this process, an opened IPC handle
        (lazy peer access over xGMI) for ranks in other processes.  Needed by strategy "waves-p2p"."""
        import os
        import numpy as np
        import torch
        if self._peer is not None:
            self.close_peer_exchange()
        itemsize = self.key_bits // 8
        kaddr, khandle = self.engine.peer_alloc(capacity * itemsize)
        paddr, phandle = self.engine.peer_alloc(capacity * 4) if with_payload else (0, bytes(64))
        row = [os.getpid(), kaddr, paddr] + [int(v) for v in np.frombuffer(khandle, dtype=np.int64)] + [int(v) for v in np.frombuffer(phandle, dtype=np.int64)]
        rows = [row]
        if self.dist is not None:
            t = torch.tensor(row, dtype=torch.int64, device=device)
            gathered = torch.empty(self.world * len(row), dtype=torch.int64, device=device)
            self.dist.all_gather_into_tensor(gathered, t)
            rows = gathered.cpu().view(self.world, len(row)).tolist()
        keys, pays, opened = [], [], []
        for r, other in enumerate(rows):
            if r == self.rank or other[0] == os.getpid():
                keys.append(other[1])
                pays.append(other[2])
            else:
                keys.append(self.engine.peer_open(np.array(other[3:11], dtype=np.int64).tobytes()))
                opened.append(keys[-1])
                if with_payload:
                    pays.append(self.engine.peer_open(np.array(other[11:19], dtype=np.int64).tobytes()))
                    opened.append(pays[-1])
                else:
                    pays.append(0)
        self._peer = {"capacity": capacity, "keys": keys, "pays": pays, "opened": opened, "mine": (kaddr, paddr), "payload": with_payload,
                      "fence": torch.zeros(1, dtype=torch.int32, device=device)}

    def close_peer_exchange(self) -> None:
        """Unmaps the other ranks' buffers and frees this rank's (collective in effect: nobody may still be writing)."""
        if self._peer is None:
            return
        for p in self._peer["opened"]:
            self.engine.peer_close(p)
        for p in self._peer["mine"]:
            if p:
                self.engine.peer_free(p)
        self._peer = None

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
        """The collectives and `work.wait()` order against torch's CURRENT stream only; every device step of
        the engine is asynchronous on the engine's stream.  The two must be the same stream or the
        all-to-all reads `staging` before the scatter has written it.  A real engine is re-bound to the
        current stream when it sits on another one (test doubles without streams are left alone)."""
        get = getattr(self.engine, "get_stream", None)
        if get is None or not getattr(keys, "is_cuda", False):
            return
        import torch
        cur = torch.cuda.current_stream(keys.device).cuda_stream
        if get() != cur:
            self.engine.set_stream(cur)

    def sort(self, keys, staging, recv, payload=None, staging_payload=None, recv_payload=None, out=None, out_payload=None):
        """keys: this rank's shard (device tensor, left untouched).
        staging: same length as keys (bucket-grouped copy).  recv: capacity for the
        incoming keys.  out (optional, same capacity as recv): enables the pipelined path, whose
        result lands there.  Returns the number of keys this rank ends up with; `result_in_out`
        says whether they are in `out` or inside the engine (engine.download / copy_result)."""
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
        # a fused table scan of an EARLIER step that timed out is reported here (no synchronisation; see rsx_check_status) —
        # callers end a batch of steps with engine.sync(), which reports the last one's
        check = getattr(self.engine, "check_status", None)
        if check is not None:
            check()
        self._mark("start")
        if self.world == 1 and not self.force_exchange:
            self.engine.sort_from(keys.data_ptr(), n, payload.data_ptr() if payload is not None else None)
            self.last_path = "local"
            return n
        pay_in = payload.data_ptr() if payload is not None else None
        pay_st = staging_payload.data_ptr() if staging_payload is not None else None
        can_pipeline = out is not None and self.world in (1, 2, 4, 8, 16) and (payload is None or out_payload is not None)
        if self.strategy == "waves" and not can_pipeline:
            raise ValueError("strategy 'waves' needs an output buffer and 1, 2, 4, 8 or 16 ranks")
        self._plain_table = None
        if self.strategy == "waves-p2p":
            if not can_pipeline or self._peer is None or (payload is not None and not self._peer["payload"]):
                raise ValueError("strategy 'waves-p2p' needs an output buffer, 1, 2, 4, 8 or 16 ranks and setup_peer_exchange() (with a payload buffer if a payload is carried)")
            return self._sort_in_waves_p2p(keys, n, payload, out, out_payload, pay_in)
        if self.strategy == "waves" or (self.strategy == "auto" and can_pipeline):
            done = self._sort_in_waves(keys, n, staging, recv, payload, staging_payload, recv_payload, out, out_payload, pay_in, pay_st)
            if done is not None:
                return done
        # the counts of the wave attempt say whether dealing the top-bit buckets out unevenly could work
        top_worth_a_try = self._plain_table is None or plan_from_table(self._plain_table, self.rank, self.world)[1] <= self.max_imbalance
        if self.strategy == "top" or (self.strategy == "auto" and top_worth_a_try):
            # fast path: buckets on the top 4 key bits, if they deal out evenly
            top_shift = self.key_bits - PARTITION_BITS
            table, caps = gather_counts(self.engine.partition_count(keys.data_ptr(), n, top_shift, PARTITION_BITS), self.world, self.dist, keys.device, self._caps)
            plan, imbalance = plan_from_table(table, self.rank, self.world)
            self._mark("count+plan")
            if imbalance <= self.max_imbalance or self.strategy == "top":
                check_capacity(plan.loads, caps, need_out=False)
                self.engine.partition_scatter(keys.data_ptr(), n, top_shift, PARTITION_BITS, staging.data_ptr(), pay_in, pay_st)
                self._mark("scatter")
                self.last_path, self.last_imbalance = "top", imbalance
                return self._exchange_and_sort(plan, n, staging, recv, payload, staging_payload, recv_payload)
        if self.strategy == "split" or (self.strategy == "auto" and self.world <= MAX_SPLITTERS + 1):
            return self._sort_by_splitters(keys, n, staging, recv, payload, staging_payload, recv_payload, pay_in, pay_st)
        lo, hi = self.engine.key_range(keys.data_ptr(), n)
        lo, hi = global_key_range(lo, hi, self.world, self.dist, keys.device)
        if lo >= hi:
            # every key everywhere is the same value (or there are no keys): rank-order
            # concatenation is already sorted and stable, nothing has to move
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
        plan, caps = plan_exchange(offs, self.rank, self.world, self.dist, keys.device, self._caps)
        check_capacity(plan.loads, caps, need_out=False)
        self._mark("count+plan")
        return self._exchange_and_sort(plan, n, staging, recv, payload, staging_payload, recv_payload)

    def _sort_in_waves(self, keys, n, staging, recv, payload, staging_payload, recv_payload, out, out_payload, pay_in, pay_st):
        """Pipelined fast path; returns None (nothing moved yet) when the fixed bucket ownership
        would leave a rank with more than max_imbalance x its share."""
        world, k = self.world, RADIX // self.world
        if self.dist is not None and getattr(keys, "is_cuda", False) and hasattr(self.engine, "partition_count_waves_device"):
            # the counts never visit the host on their way into the all_gather: one host round trip (the gathered table) per step
            table, caps = self._gather_wave_counts_on_device(keys, n)
            counts = table[self.rank]
        else:
            counts = self.engine.partition_count_waves(keys.data_ptr(), n, world)          # [wave * world + rank]
            table, caps = gather_counts(counts, world, self.dist, keys.device, self._caps)   # [source][wave * world + rank]
        loads = [sum(row[w * world + d] for row in table for w in range(k)) for d in range(world)]
        total = sum(loads)
        imbalance = max(loads) / max(1.0, total / world)
        self._mark("count+plan")
        fits = True
        try:
            check_capacity(loads, caps, need_out=True, slack=4 * k)     # each wave starts 16-byte aligned in recv
        except CapacityError:
            if self.strategy == "waves":
                raise                                                   # on every rank alike
            fits = False
        if (imbalance > self.max_imbalance and self.strategy != "waves") or not fits:
            # same decision on every rank: it only depends on the gathered table.  Leave the counts in
            # plain bucket order (b = rank * k + wave) for the caller's next decision
            self._plain_table = [[row[(b % k) * world + b // k] for b in range(RADIX)] for row in table]
            return None
        self.engine.partition_scatter_waves(keys.data_ptr(), n, staging.data_ptr(), pay_in, pay_st)
        self._mark("scatter")
        # all waves are queued on the collective stream at once; they run in order behind each other
        pending, send_at, recv_at = [], 0, 0
        for w in range(k):
            send = [counts[w * world + d] for d in range(world)]
            rcv = [table[s][w * world + self.rank] for s in range(world)]
            n_send, n_recv = sum(send), sum(rcv)
            recv_at = (recv_at + 3) & ~3                 # 16-byte aligned start: the sort loads 16 bytes per lane
            works = [self.dist.all_to_all_single(recv[recv_at:recv_at + n_recv], staging[send_at:send_at + n_send], rcv, send, async_op=True)]
            if payload is not None:
                works.append(self.dist.all_to_all_single(recv_payload[recv_at:recv_at + n_recv], staging_payload[send_at:send_at + n_send],
                                                         rcv, send, async_op=True))
            pending.append((works, recv_at, n_recv))
            send_at += n_send
            recv_at += n_recv
        # a wave's keys share the top nibble at this rank: the most significant LSD pass is not needed
        passes = self.key_bits // PARTITION_BITS - 1
        done = 0
        for works, at, n_recv in pending:
            for work in works:
                if work is not None:
                    work.wait()                          # the engine's stream waits for this wave only
            self._mark("wait")
            if n_recv:
                self.engine.sort_from_to(
                    recv[at:].data_ptr(), n_recv, 0, passes, out[done:].data_ptr(),
                    recv_payload[at:].data_ptr() if payload is not None else None,
                    out_payload[done:].data_ptr() if payload is not None else None)
            self._mark("local_sort")
            done += n_recv
        self.last_path, self.last_imbalance, self.result_in_out = "waves", imbalance, True
        return done

    def _sort_in_waves_p2p(self, keys, n, payload, out, out_payload, pay_in):
        """Pipelined path with the exchange done by the scatter kernel's own stores into the owners' receive buffers."""
        world, k = self.world, RADIX // self.world
        itemsize = self.key_bits // 8
        if self.dist is not None and getattr(keys, "is_cuda", False) and hasattr(self.engine, "partition_count_waves_device"):
            table, caps = self._gather_wave_counts_on_device(keys, n)
        else:
            counts = self.engine.partition_count_waves(keys.data_ptr(), n, world)
            table, caps = gather_counts(counts, world, self.dist, keys.device, self._caps)
        # (the all_gather above is also the step's opening barrier: it completes only once every rank has enqueued its own, behind
        # the local sorts of its previous step — nobody is still reading the receive buffer this step is about to write into)
        start, offset, loads = wave_layout(table, world)
        self.last_imbalance = max(loads) / max(1.0, sum(loads) / world)
        self._mark("count+plan")
        check_capacity(loads, caps, need_out=True, slack=4 * k)          # every rank alike: gathered data only
        peer_keys = [self._peer["keys"][p % world] + offset[p % world][p // world][self.rank] * itemsize for p in range(RADIX)]
        peer_pays = [self._peer["pays"][p % world] + offset[p % world][p // world][self.rank] * 4 for p in range(RADIX)] if payload is not None else None
        self.engine.partition_scatter_waves_peer(keys.data_ptr(), n, peer_keys, pay_in, peer_pays)
        self._mark("scatter")
        if self.dist is not None:
            self.dist.all_reduce(self._peer["fence"])                    # every rank's scatter has finished: what this rank received is complete
        self._mark("fence")
        passes = self.key_bits // PARTITION_BITS - 1
        mine_k, mine_p = self._peer["mine"]
        done = 0
        for w in range(k):
            n_recv = sum(table[src][w * world + self.rank] for src in range(world))
            if n_recv:
                at = start[self.rank][w]
                self.engine.sort_from_to(
                    mine_k + at * itemsize, n_recv, 0, passes, out[done:].data_ptr(),
                    mine_p + at * 4 if payload is not None else None,
                    out_payload[done:].data_ptr() if payload is not None else None)
            done += n_recv
        self._mark("local_sort")
        self.last_path, self.result_in_out = "waves-p2p", True
        return done

    def _gather_wave_counts_on_device(self, keys, n):
        """gather_counts for the pipelined path without the host in the middle: the engine leaves its 16 counts in a device
        row that also carries this rank's two buffer capacities, the row goes into the all_gather as it is, and only the
        gathered table is copied to the host."""
        import torch
        if self._count_row is None or self._count_row.device != keys.device:
            self._count_row = torch.zeros(RADIX + 2, dtype=torch.int64, device=keys.device)
            self._count_table = torch.empty(self.world * (RADIX + 2), dtype=torch.int64, device=keys.device)
            self._count_row_caps = None
        if self._count_row_caps != self._caps:
            self._count_row[RADIX:] = torch.tensor([int(self._caps[0]), int(self._caps[1])], dtype=torch.int64)
            self._count_row_caps = self._caps
        self.engine.partition_count_waves_device(keys.data_ptr(), n, self.world, self._count_row.data_ptr())
        self.dist.all_gather_into_tensor(self._count_table, self._count_row)
        rows = self._count_table.cpu().view(self.world, RADIX + 2).tolist()
        return [r[:RADIX] for r in rows], [(r[RADIX], r[RADIX + 1]) for r in rows]

    def _sort_by_splitters(self, keys, n, staging, recv, payload, staging_payload, recv_payload, pay_in, pay_st):
        count = min(SAMPLES_PER_RANK, n)
        mine = self.engine.sample_keys(keys.data_ptr(), n, count) if count else []
        samples, sizes = gather_samples(mine, n, self.world, self.dist, keys.device)
        splitters = choose_splitters(samples, sizes, self.world)
        if not splitters:              # nobody has keys
            self.engine.sort_from(keys.data_ptr(), n, pay_in)
            self.last_path = "equal"
            return n
        counts = self.engine.partition_count_split(keys.data_ptr(), n, splitters)
        counts = counts + [0] * (RADIX - len(counts))
        table, caps = gather_counts(counts, self.world, self.dist, keys.device, self._caps)
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
