"""Runs of one build on one box come in two modes 2-3 % apart.  Is the mode a property of the process, or of where the buffers sit?
Several engines alive at once in ONE process (different allocations), each timed on the same device-resident input."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

m = entry.load_package()
n = int(os.environ["MODE_N"]) if os.environ.get("MODE_N") else 1 << int(os.environ.get("MODE_LOG2", "28"))
dtype = os.environ.get("MODE_DTYPE", "uint32")           # uint32 | uint64
bits = int(os.environ.get("MODE_BITS", "4"))             # digit width
sorts = int(os.environ.get("MODE_SORTS", "40"))
if dtype == "uint64":
    keys = torch.from_numpy(np.random.default_rng(1).integers(0, 2**63, size=n, dtype=np.int64)).cuda()
else:
    keys = torch.from_numpy(np.random.default_rng(1).integers(0, 2**32, size=n, dtype=np.uint32).view(np.int32)).cuda()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
payload = os.environ.get("MODE_PAYLOAD", "0") == "1"
perm = torch.arange(n, dtype=torch.int32, device="cuda") if payload else None
engines = []
for i in range(int(os.environ.get("MODE_ENGINES", "5"))):
    e = m.Engine(dtype, n, payload=payload)
    e.set_stream(stream.cuda_stream)
    if bits != 4:
        e.set_option(m.OPT_RADIX_BITS, bits)
    engines.append(e)
    if i % 2 == 1:
        pad = torch.empty((37 << 20) + i * 4096, dtype=torch.uint8, device="cuda")   # shift the next allocation
for rep in range(1):
    for i, e in enumerate(engines):
        for _ in range(max(2, sorts // 4)):
            e.sort_from(keys.data_ptr(), n, perm.data_ptr() if payload else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(sorts):
            e.sort_from(keys.data_ptr(), n, perm.data_ptr() if payload else None)
        torch.cuda.synchronize()
        print(f"rep {rep} engine {i}: {(time.perf_counter() - t0) / sorts * 1e3:.3f} ms per sort  result at {e.result_device()[0]:#x} payload at {e.result_device()[1]:#x}", flush=True)
