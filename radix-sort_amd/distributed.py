"""Sharded sort across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The reference has no multi-device path ("Parallelism strategies: none"); this
is the capability the north-star adds.  LSD passes are not independent
across shards, but a most-significant-bits partition is, so the data path is:

  1. every rank groups its shard by the top 4 key bits — ONE stable radix pass of the
     same histogram/scan/reorder kernels (C ABI `rsx_partition`), which also yields the
     16 bucket sizes;
  2. "histogram all-to-all": all ranks exchange their 16 bucket counts
     (`all_gather`, 16 x int64 per rank — latency-bound, KBs);
  3. buckets are dealt to ranks as contiguous ranges (`bucket_owner`), so each rank's
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
same three methods so the split/offset logic runs under gloo without a GPU.
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


def send_splits(bucket_offsets: list[int], world_size: int) -> list[int]:
    """Number of local keys going to each rank, from the 17 exclusive bucket offsets of the
    partition pass."""
    if len(bucket_offsets) != RADIX + 1:
        raise ValueError("expected 17 bucket offsets")
    owner = bucket_owner(world_size)
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

    @property
    def n_recv(self) -> int:
        return sum(self.recv)


def plan_exchange(bucket_offsets: list[int], rank: int, world_size: int, dist, device) -> ExchangePlan:
    """Steps 2 of the module docstring: all_gather of the per-rank send splits."""
    import torch

    mine = send_splits(bucket_offsets, world_size)
    if world_size == 1 and dist is None:
        return ExchangePlan(send=mine, recv=mine)
    t = torch.tensor(mine, dtype=torch.int64, device=device)
    gathered = torch.empty(world_size * world_size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered, t)
    table = gathered.cpu().view(world_size, world_size).tolist()
    return ExchangePlan(send=mine, recv=recv_splits(table, rank))


class ShardedSorter:
    """Per-rank driver.  Buffers are torch tensors (device memory + RCCL plumbing)."""

    def __init__(self, engine, rank: int, world_size: int, key_bits: int, dist=None, force_exchange: bool = False):
        self.engine = engine
        self.rank = rank
        self.world = world_size
        self.key_bits = key_bits
        self.dist = dist
        # force_exchange: run partition -> all_gather -> all_to_all even with one rank (the
        # collectives then talk to self); lets a 1-GPU box exercise the real RCCL call path
        self.force_exchange = force_exchange and dist is not None
        if world_size > 1 and dist is None:
            raise ValueError("a torch.distributed module is required for world_size > 1")

    def sort(self, keys, staging, recv, payload=None, staging_payload=None, recv_payload=None):
        """keys: this rank's shard (device tensor, left untouched).
        staging: same length as keys (bucket-grouped copy).  recv: capacity for the
        incoming keys.  Returns the number of keys this rank ends up with; the sorted
        keys stay inside the engine (engine.copy_result / result_device)."""
        n = keys.numel()
        if self.world == 1 and not self.force_exchange:
            self.engine.sort_from(keys.data_ptr(), n, payload.data_ptr() if payload is not None else None)
            return n
        offs = self.engine.partition(
            keys.data_ptr(), n, self.key_bits - PARTITION_BITS, PARTITION_BITS, staging.data_ptr(),
            payload.data_ptr() if payload is not None else None,
            staging_payload.data_ptr() if staging_payload is not None else None)
        plan = plan_exchange(offs, self.rank, self.world, self.dist, keys.device)
        if plan.n_recv > recv.numel():
            raise RuntimeError(f"rank {self.rank}: receives {plan.n_recv} keys but the receive buffer holds {recv.numel()}")
        self.dist.all_to_all_single(recv[:plan.n_recv], staging[:n], plan.recv, plan.send)
        if payload is not None:
            self.dist.all_to_all_single(recv_payload[:plan.n_recv], staging_payload[:n], plan.recv, plan.send)
        self.engine.sort_from(recv.data_ptr(), plan.n_recv, recv_payload.data_ptr() if payload is not None else None)
        return plan.n_recv
