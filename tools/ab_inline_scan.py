#!/usr/bin/env python3
"""ms per sort of mid-size inputs: product library vs the experiments build with the table scan inside the reorder launch
(RSX_XOPT_INLINE_SCAN) off and on — interleaved, same box, same process.  Round 4 re-measurement: the INLINE_SCAN kernels are now built for
three waves per SIMD and no longer spill (profiles/r04_kernel_resources_experiments.txt); round 3's numbers were taken with 52-180 bytes of
scratch per lane in the uint32 variants.    python tools/ab_inline_scan.py [log2 sizes ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def ms_per_sort(eng, ptr, n, iters):
    for _ in range(5):
        eng.sort_from(ptr, n)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.sort_from(ptr, n)
    eng.sync()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    import torch
    rsx = entry.load_package()
    xp = rsx.experiments()
    sizes = [int(a) for a in sys.argv[1:]] or [22, 23, 24, 25, 26, 27]
    for dt in ("uint32", "uint64"):
        for lg in sizes:
            n = 1 << lg
            rng = np.random.default_rng(lg)
            keys = torch.from_numpy(rng.integers(0, 2**63 - 1, size=n, dtype=np.int64).astype(np.int64 if dt == "uint64" else np.int32)).cuda()
            iters = 200 if lg <= 24 else 60
            row = []
            with rsx.Engine(dt, n) as a, xp.Engine(dt, n) as b, xp.Engine(dt, n) as c:
                c.set_option(xp.XOPT_INLINE_SCAN, 1)
                c.set_option(xp.XOPT_INLINE_SCAN_MAX_GROUPS, 512)
                for _ in range(2):
                    row += [ms_per_sort(a, keys.data_ptr(), n, iters), ms_per_sort(b, keys.data_ptr(), n, iters), ms_per_sort(c, keys.data_ptr(), n, iters)]
            print(f"{dt} 2^{lg}:  product {row[0]:.4f} {row[3]:.4f}   experiments, scan launches {row[1]:.4f} {row[4]:.4f}   experiments, inline scan {row[2]:.4f} {row[5]:.4f}", flush=True)


if __name__ == "__main__":
    main()
