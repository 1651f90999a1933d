"""Does the power-of-two geometry of the headline size cost bandwidth?  ns per key per pass for sizes around 2^28 whose digit regions
(n/16 apart) and XCD ranges (n/8 apart) are not powers of two, next to what plain device copies / reads / fills reach on the same box."""
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
full = 1 << 28
keys = torch.from_numpy(np.random.default_rng(1).integers(0, 2**32, size=full + (1 << 24), dtype=np.uint32).view(np.int32)).cuda()


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


a = torch.empty(full, dtype=torch.int32, device="cuda")
for _ in range(2):
    t = timed(lambda: a.copy_(keys[:full]), 40)
    print(f"copy 1 GiB -> 1 GiB: {t * 1e3:.3f} ms = {2 * full * 4 / t / 1e12:.2f} TB/s", flush=True)
    t = timed(lambda: a.fill_(7), 40)
    print(f"fill 1 GiB: {t * 1e3:.3f} ms = {full * 4 / t / 1e12:.2f} TB/s", flush=True)
    t = timed(lambda: torch.sum(keys[:full]), 40)
    print(f"sum 1 GiB: {t * 1e3:.3f} ms = {full * 4 / t / 1e12:.2f} TB/s", flush=True)
del a
sizes = [full, full - (1 << 20) * 13, full * 15 // 16, full * 17 // 16 , 250_000_000, 268_435_456 - 4096 * 8 * 37, full]
for n in sizes:
    e = m.Engine("uint32", n)
    e.set_stream(stream.cuda_stream)
    t = timed(lambda: e.sort_from(keys.data_ptr(), n), 30)
    print(f"n = {n:>10} ({n / full:.4f} x 2^28): {t * 1e3:.3f} ms per sort, {t * 1e9 / n / 8 * 1e3:.3f} ps per key per pass, {n / t / 1e9:.2f} Gkeys/s", flush=True)
    e.close()
    del e
