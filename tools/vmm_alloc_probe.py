"""How long does an engine take to come up with VMM-backed buffers (RSX_ALLOC_MODE=1|2), and does it sort?  Small sizes first, timed, one line each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
m = entry.load_package()
for log2 in (24, 26, 28):
    n = 1 << log2
    t0 = time.perf_counter()
    e = m.Engine("uint32", n, payload=True)
    t1 = time.perf_counter()
    keys = np.random.default_rng(log2).integers(0, 2**32, size=n, dtype=np.uint32)
    e.upload(keys, np.arange(n, dtype=np.uint32))
    e.sort()
    ok = bool(np.array_equal(e.download(), np.sort(keys)))
    t2 = time.perf_counter()
    e.close()
    print(f"mode {os.environ.get('RSX_ALLOC_MODE')} chunk {os.environ.get('RSX_ALLOC_CHUNK_MB')} MB: 2^{log2} keys: create {t1 - t0:.2f} s, upload+sort+download+check {t2 - t1:.2f} s, close {time.perf_counter() - t2:.2f} s, sorted {ok}", flush=True)
