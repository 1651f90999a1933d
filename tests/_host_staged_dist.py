"""torch.distributed's call signatures over gloo for DEVICE tensors, staged through host memory (TEST INFRASTRUCTURE).

The GPU test box has one MI355X and RCCL refuses two ranks on one device, so real multi-process runs of the sharded sort
(`bench.py --gpus N` under RSX_BENCH_SHARED_GPU=1, tests/test_gpu_sharded.py) put every rank on cuda:0 — each its own
process, its own HIP engine, its own streams — and swap only the transport: a collective waits for the caller's current
stream, moves the tensors to the host, runs the gloo collective there and copies the result back.  `async_op=True`
returns a finished work object (stream ordering of real asynchronous collectives is what the thread loopback of
tests/test_gpu_sharded.py checks)."""
from __future__ import annotations

import torch
import torch.distributed as td


class _Done:
    def wait(self):
        return True


class HostStagedDist:
    ReduceOp = td.ReduceOp

    def __init__(self):
        assert td.is_initialized() and td.get_backend() == "gloo"

    @staticmethod
    def _host(t):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        return t.detach().cpu().contiguous()

    def barrier(self):
        td.barrier()

    def destroy_process_group(self):
        td.destroy_process_group()

    def all_reduce(self, t, op=td.ReduceOp.SUM, async_op=False):
        h = self._host(t)
        td.all_reduce(h, op=op)
        t.copy_(h)
        return _Done() if async_op else None

    def broadcast(self, t, src=0):
        h = self._host(t)
        td.broadcast(h, src=src)
        t.copy_(h)

    def all_gather_into_tensor(self, out, t, async_op=False):
        h = self._host(t)
        parts = [torch.empty_like(h) for _ in range(td.get_world_size())]
        td.all_gather(parts, h)
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.shape))
        return _Done() if async_op else None

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        h_in = self._host(inp)
        h_out = torch.empty(out.shape, dtype=out.dtype)
        td.all_to_all_single(h_out, h_in, output_split_sizes, input_split_sizes)
        out.copy_(h_out)
        return _Done() if async_op else None

    def gather(self, t, gather_list=None, dst=0):
        h = self._host(t)
        parts = [torch.empty_like(h) for _ in range(td.get_world_size())] if td.get_rank() == dst else None
        td.gather(h, parts, dst=dst)
        if parts is not None:
            for g, p in zip(gather_list, parts):
                g.copy_(p)
