"""CPU stand-in for the device side of bench.py (TEST INFRASTRUCTURE, never a measurement).

bench.py imports this only under RSX_BENCH_REHEARSAL=1: the launcher (`python bench.py --gpus N`
spawning its own ranks) and the whole N>1 rank logic — default workload, collectives, capacity
checks, verification gather, the JSON line — then run under gloo on a machine without GPUs.  The
device work is done by the numpy test double of tests/test_distributed_gloo.py; timings are zero and
the printed line carries `"rehearsal": true` and no value."""
from __future__ import annotations

from types import SimpleNamespace

from test_distributed_gloo import _CpuEngineDouble


class RehearsalEngine(_CpuEngineDouble):
    def set_option(self, option, value):
        pass

    def timings(self, reset=False):
        zero = SimpleNamespace(min_ms=0.0, max_ms=0.0, avg_ms=0.0, sum_ms=0.0, n=0)
        return SimpleNamespace(histogram=zero, scan=zero, paste=zero, reorder=zero, total=zero)

    def download(self):
        return self.result

    def close(self):
        pass
