"""Times the one-pass partitions of the multi-GPU exchange on one GPU: top-4-bits
(rsx_partition_count/scatter), equal-width range (rsx_key_range + rsx_partition_range) and
sampled splitters (rsx_sample_keys + rsx_partition_count_split/scatter_split).
usage: python tools/partition_bench.py [log2_keys] [dtype]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

rsx = entry.load_package()
import torch  # noqa: E402
from radix_sort_amd.distributed import choose_splitters, range_buckets  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dt = sys.argv[2] if len(sys.argv) > 2 else "uint32"
n = 1 << log2n
bits = np.dtype(dt).itemsize * 8
tdt = torch.int32 if bits == 32 else torch.int64
keys = torch.randint(-(1 << (bits - 1)), (1 << (bits - 1)) - 1, (n,), dtype=tdt, device="cuda")
out = torch.empty_like(keys)
with rsx.Engine(dt, n) as e:
    e.set_stream(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def top():
        e.partition_count(keys.data_ptr(), n, bits - 4, 4)
        e.partition_scatter(keys.data_ptr(), n, bits - 4, 4, out.data_ptr())

    def ranged():
        lo, hi = e.key_range(keys.data_ptr(), n)
        shift, mul = range_buckets(lo, hi, bits)
        e.partition_range(keys.data_ptr(), n, lo, shift, mul, out.data_ptr())

    def split(world=8):
        s = e.sample_keys(keys.data_ptr(), n, 1024)
        sp = choose_splitters([s], [n], world)
        e.partition_count_split(keys.data_ptr(), n, sp)
        e.partition_scatter_split(keys.data_ptr(), n, out.data_ptr())

    print(f"n=2^{log2n} {dt}: top-bits {timed(top):.3f} ms, range {timed(ranged):.3f} ms, "
          f"splitters(7) {timed(split):.3f} ms, splitters(1) {timed(lambda: split(2)):.3f} ms")
