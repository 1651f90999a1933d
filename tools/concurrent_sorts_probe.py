#!/usr/bin/env python3
"""How fast do MANY small local sorts run when several engines (streams) work side by side?  The sharded sort's waves at a deep partition
(B = 8 at 8 ranks: 32 waves of 2^22 keys, 6 pass units each) are launch-chain-bound one at a time; this probe sorts the 2^27 keys of a rank as
2^27 / 2^lg pieces over `units` pass units with 1, 2, 3, 4 and 6 engines round-robin, and compares with the doubling groups' sizes sorted one
after the other.      python tools/concurrent_sorts_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    rsx = entry.load_package()
    total_lg = 27
    n_total = 1 << total_lg
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=n_total, dtype=np.int32)).cuda()
    dst = torch.empty_like(src)
    for radix_bits, units_list in ((8, (6, 7)), (4, (6, 7))):
        for lg in (22, 23, 24):
            pieces = n_total >> lg
            n = 1 << lg
            for units in units_list:
                row = []
                for nengines in (1, 2, 3, 4, 6):
                    engines, streams = [], []
                    for _ in range(nengines):
                        e = rsx.Engine("uint32", n + 4096)
                        s = torch.cuda.Stream()
                        e.set_stream(s.cuda_stream)
                        if radix_bits != 4:
                            e.set_option(rsx.OPT_RADIX_BITS, radix_bits)
                        engines.append(e)
                        streams.append(s)

                    def step():
                        for p in range(pieces):
                            e = engines[p % nengines]
                            e.sort_from_to(src.data_ptr() + p * n * 4, n, 0, units, dst.data_ptr() + p * n * 4)
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        step()
                    torch.cuda.synchronize()
                    row.append((time.perf_counter() - t0) / 10 * 1e3)
                    for e in engines:
                        e.close()
                print(f"{radix_bits}-bit passes, {pieces:3d} pieces of 2^{lg}, {units} units:  " + "  ".join(f"{k} eng {v:.3f}" for k, v in zip((1, 2, 3, 4, 6), row)) + " ms", flush=True)
        # the doubling groups of B = 6: 2^24, 2^24, 2^25, 2^26 keys, 7 units, one engine
        e = rsx.Engine("uint32", (1 << 26) + 4096)
        if radix_bits != 4:
            e.set_option(rsx.OPT_RADIX_BITS, radix_bits)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        sizes, at = [1 << 24, 1 << 24, 1 << 25, 1 << 26], [0, 1 << 24, 1 << 25, 1 << 26]

        def groups():
            for a, sz in zip(at, sizes):
                e.sort_from_to(src.data_ptr() + a * 4, sz, 0, 7, dst.data_ptr() + a * 4)
        for _ in range(3):
            groups()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            groups()
        torch.cuda.synchronize()
        print(f"{radix_bits}-bit passes, doubling groups 2^24 2^24 2^25 2^26, 7 units, one engine: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
        e.close()


if __name__ == "__main__":
    main()
