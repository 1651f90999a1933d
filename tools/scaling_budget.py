#!/usr/bin/env python3
"""The sharded sort's step time on N GPUs, ON PAPER, from pieces measured on ONE MI355X — reproducible arithmetic, not a measurement.

    python tools/scaling_budget.py [--link-gbs 50] [--total-log2 30]

Inputs (committed): profiles/r04_size_sweep_{4,8}bit.jsonl (ms per 8- / 4-pass sort by size, interpolated in log-log between sizes) and the
count / scatter / contention figures of the forced-exchange runs (profiles/r04_forced_v4, r04_kernel_stats_forced_peer_stores_2p27.csv).
Model of one step on a rank holding n = total / N keys (uint32):
  count      one read of the shard by the 8-bit histogram + scans:        n * 4 B / 3.1 TB/s            (0.17 ms measured at 2^27)
  scatter    read + write of the shard by the 8-bit scatter:             n * 8 B / 4.3 TB/s            (0.25 ms measured at 2^27)
  exchange   (N - 1) / N of the shard leaves, one link per peer, `link_gbs` GB/s per direction; wave 0 (1 / k of it) is exposed, + one fence (0.03 ms);
             the rest hides behind the sorts unless the links are the bottleneck (then the step is exchange-bound)
  sorts      doubling groups {0} {1} {2,3} {4..7} of the k = 8 waves: each a sort of n * g / k keys over `units` of the 8 (4-bit) pass units —
             time = (ms of a full sort of that size from the sweep) * units / 8, * 1.08 for the pushes running beside them (measured 8-11 %)
Linear scaling = N x the single-GPU rate at 2^28 keys from the same sweep.
"""
import argparse
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sweep(path):
    out = {}
    for line in open(path):
        d = json.loads(line)
        out[int(math.log2(d["config"]["keys_per_gpu"]))] = d["ms_per_step"]
    return out


def sort_ms(table, keys):
    """ms of a full sort of `keys` keys: log-log interpolation between the measured sizes."""
    lg = math.log2(keys)
    sizes = sorted(table)
    lo = max([s for s in sizes if s <= lg], default=sizes[0])
    hi = min([s for s in sizes if s >= lg], default=sizes[-1])
    if lo == hi:
        return table[lo] * (keys / 2 ** lo)
    f = (lg - lo) / (hi - lo)
    return math.exp(math.log(table[lo]) * (1 - f) + math.log(table[hi]) * f)


def budget(n_gpus, total_log2, table, radix_bits, link_gbs, bits=None, contention=1.08, fence_ms=0.03):
    n = (1 << total_log2) // n_gpus
    lg_world = (n_gpus - 1).bit_length()
    bits = bits or max(4, min(8, lg_world + 3))
    k = (1 << bits) // n_gpus
    count = n * 4 / 3.1e12 * 1e3
    scatter = n * 8 / 4.3e12 * 1e3
    per_link = n * 4 / n_gpus                                   # bytes to each peer
    exchange = per_link / (link_gbs * 1e9) * 1e3 if n_gpus > 1 else 0.0      # all links at once
    exposed = exchange / k + (fence_ms if n_gpus > 1 else 0.0)
    groups, w = [], 0
    while w < k:
        g = 1 if w < 2 else w
        groups.append(min(g, k - w))
        w += g
    sorts = 0.0
    for g in groups:
        units = -(-(32 - bits + (g - 1).bit_length()) // 4)
        if radix_bits == 8:
            frac = (units // 2 + 0.5 * (units % 2)) / 4            # whole bytes + half a byte pass for the nibble
        else:
            frac = units / 8
        sorts += sort_ms(table, n * g / k) * frac
    sorts *= contention
    step = count + scatter + max(exposed + sorts, exchange)        # exchange-bound when the links take longer than everything they hide behind
    return {"bits": bits, "waves": k, "count": count, "scatter": scatter, "exchange": exchange, "exposed": exposed, "sorts": sorts, "step": step}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbs", type=float, default=50.0, help="achieved GB/s per xGMI link and direction (peak 76.8)")
    ap.add_argument("--total-log2", type=int, default=30)
    args = ap.parse_args()
    t4 = sweep(os.path.join(ROOT, "profiles", "r04_size_sweep_4bit.jsonl"))
    t8 = sweep(os.path.join(ROOT, "profiles", "r04_size_sweep_8bit.jsonl"))
    single = (1 << 28) / t4[28] * 1e-6          # Gkeys/s of the single-GPU 4-bit headline in the same sweep
    print(f"single GPU, 2^28 uint32, 4-bit digits: {t4[28]:.3f} ms = {single:.1f} Gkeys/s (the sweep's box);  links: {args.link_gbs:.0f} GB/s per direction;  total 2^{args.total_log2} keys")
    print(f"{'N':>2} {'local passes':>12} {'B':>2} {'waves':>5} {'count':>6} {'scatter':>7} {'exchange':>8} {'exposed':>7} {'sorts':>6} {'step ms':>8} {'Gkeys/s':>8} {'of N x single':>13}")
    for n_gpus in (1, 2, 4, 8):
        for radix_bits, table in ((4, t4), (8, t8)):
            if n_gpus == 1:
                ms = sort_ms(table, 1 << args.total_log2)
                print(f"{n_gpus:>2} {str(radix_bits) + '-bit':>12} {'-':>2} {'-':>5} {'-':>6} {'-':>7} {'-':>8} {'-':>7} {ms:6.2f} {ms:8.2f} {(1 << args.total_log2) / ms * 1e-6:8.1f} {(1 << args.total_log2) / ms * 1e-6 / single:13.2f}")
                continue
            b = budget(n_gpus, args.total_log2, table, radix_bits, args.link_gbs)
            gk = (1 << args.total_log2) / b["step"] * 1e-6
            print(f"{n_gpus:>2} {str(radix_bits) + '-bit':>12} {b['bits']:>2} {b['waves']:>5} {b['count']:6.2f} {b['scatter']:7.2f} {b['exchange']:8.2f} {b['exposed']:7.2f} {b['sorts']:6.2f} {b['step']:8.2f} {gk:8.1f} {gk / (n_gpus * single):13.2f}")


if __name__ == "__main__":
    main()
