"""ms per sort at small sizes under a few engine settings (graph on/off, XCD phase, self-scan), device-resident input."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

m = entry.load_package()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
for lg in (int(a) for a in (sys.argv[1:] or ["16", "20", "22"])):
    n = 1 << lg
    keys = torch.from_numpy(np.random.default_rng(1).integers(0, 2**32, size=n, dtype=np.uint32).view(np.int32)).cuda()
    for label, opts in [("default", {}), ("graph off", {m.OPT_GRAPH: 0}), ("phase 0", {m.OPT_XCD_PHASE: 0}), ("self-scan off", {m.OPT_SELF_SCAN: 0}),
                        ("self-scan off, graph off", {m.OPT_SELF_SCAN: 0, m.OPT_GRAPH: 0})]:
        e = m.Engine("uint32", n)
        e.set_stream(stream.cuda_stream)
        for k, v in opts.items():
            e.set_option(k, v)
        for _ in range(20):
            e.sort_from(keys.data_ptr(), n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            e.sort_from(keys.data_ptr(), n)
        torch.cuda.synchronize()
        print(f"2^{lg} {label:<28} {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per sort", flush=True)
        e.close()
