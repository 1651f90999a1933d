"""Why is the scatter slower in the reference's call order (upload -> sort -> download per iteration)?
Run under `rocprofv3 --kernel-trace --memory-copy-trace`; tools/upload_probe_report.py then lists every
kernel of every sort in dispatch order with its duration and the idle gap in front of the sort."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

m = entry.load_package()
n = 1 << 28
keys = np.random.default_rng(1).integers(0, 2**32, size=n, dtype=np.uint32)
with m.Engine("uint32", n) as e:
    e.upload(keys, None); e.sort(); e.sync()
    print("A: upload before every sort", flush=True)
    for _ in range(4):
        e.upload(keys, None); e.sort(); e.sync()
    print("B: host sleeps 30 ms (GPU idle, no copy) before every sort", flush=True)
    for _ in range(4):
        time.sleep(0.03); e.sort(); e.sync()
    print("C: back to back", flush=True)
    for _ in range(4):
        e.sort()
    e.sync()
    print("D: upload, then two sorts back to back", flush=True)
    for _ in range(3):
        e.upload(keys, None); e.sort(); e.sort(); e.sync()
