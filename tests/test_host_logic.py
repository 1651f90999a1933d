"""CPU tests of the host mirror of the reference interface (radix-sort_amd/host):
the C++ self-test binary, and the product's Dataset generators against the golden
vectors produced by the reference's own Dataset.h."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "radix-sort_amd", "host")
KINDS = {"Zeros": 0, "Range": 1, "InvertedRange": 2, "Random": 3, "RandomDistributed": 4}
DTYPES = {"uint32": 0, "int32": 1, "uint64": 2, "int64": 3}


@pytest.fixture(scope="module")
def hostlib():
    lib = C.CDLL(os.path.join(HOST, "libradixsort_host.so"))
    lib.rsxh_dataset_fill.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
    lib.rsxh_dataset_fill.restype = C.c_int
    lib.rsxh_resize.argtypes = [C.c_uint32]
    lib.rsxh_resize.restype = C.c_uint32
    lib.rsxh_default_uniform_seed.restype = C.c_uint64
    return lib


def _fill(lib, kind, dt, n, seed=0):
    out = np.empty(n, dtype=dt)
    assert lib.rsxh_dataset_fill(KINDS[kind], DTYPES[dt], out.ctypes.data, n, seed) == 0
    return out


def test_host_selftest_binary():
    exe = os.path.join(HOST, "bin", "host_selftest")
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "checks passed" in proc.stdout


def test_product_datasets_match_reference_goldens(hostlib, oracle, golden):
    # same inputs as the reference's generators (golden digests come from its Dataset.h)
    for row in golden["datasets"]:
        got = _fill(hostlib, row["kind"], row["dtype"], row["n"])
        assert oracle.digest(got) == row["input_digest"], row
    for row in golden["small_vectors"]:
        assert [int(v) for v in _fill(hostlib, row["kind"], row["dtype"], row["n"])] == row["input"]


def test_product_uniform_dataset_matches_oracle_standin(hostlib, oracle):
    seed = int(hostlib.rsxh_default_uniform_seed())
    for dt in DTYPES:
        a = _fill(hostlib, "RandomDistributed", dt, 4099, seed)
        assert np.array_equal(a, oracle.dataset("SeededUniform", dt, 4099, seed))


def test_contiguous_shards_concatenate_to_the_one_dataset(hostlib):
    """rsxh_dataset_fill_shard: the ranks of a sharded sort each produce their contiguous shard of ONE dataset (BASELINE config 4:
    `Random` sharded contiguously — std::mt19937::discard(rank * n), Dataset.h:110-120); the shards in rank order are the dataset."""
    hostlib.rsxh_dataset_fill_shard.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    hostlib.rsxh_dataset_fill_shard.restype = C.c_int
    total, seed = 10007, 77
    for kind in KINDS:
        for dt in DTYPES:
            whole = _fill(hostlib, kind, dt, total, seed)
            for world in (1, 3, 8):
                cuts = [total * r // world for r in range(world + 1)]
                parts = []
                for a, b in zip(cuts, cuts[1:]):
                    out = np.empty(b - a, dtype=dt)
                    assert hostlib.rsxh_dataset_fill_shard(KINDS[kind], DTYPES[dt], out.ctypes.data, a, b - a, total, seed) == 0
                    parts.append(out)
                assert np.array_equal(np.concatenate(parts), whole), (kind, dt, world)
    out = np.empty(4, dtype=np.uint32)
    assert hostlib.rsxh_dataset_fill_shard(3, 0, out.ctypes.data, total - 2, 4, total, 0) != 0      # past the end


def test_resize_rounds_to_1024(hostlib):
    # RadixSortGPU::Resize (reference src/RadixSortGPU.cpp:288-297)
    for nn, want in [(0, 0), (1, 1024), (1000, 1024), (1024, 1024), (1025, 2048), (1 << 25, 1 << 25), ((1 << 28) + 1, (1 << 28) + 1024)]:
        assert hostlib.rsxh_resize(nn) == want


def test_harness_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(HOST, "bin", "rsx_tests")
    proc = subprocess.run([exe, "--num-elements", "2048"], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 2 and "no CPU fallback" in proc.stderr


def _run_bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    return proc, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks_rehearsal():
    """`python bench.py --gpus 2` from a plain shell: the parent spawns two ranks as CHILD processes
    (no GPU call of its own), they run the complete N>1 rank logic on the CPU test double under gloo,
    and rank 0's line comes back through the parent with n_gpus = 2, cpu_baseline and roofline present."""
    proc, line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--total-log2-keys", "15", "--cpu-sample-log2", "12"],
                            {"RSX_BENCH_REHEARSAL": "1"})
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert line["rehearsal"] is True and line["value"] is None and line["metric"].startswith("REHEARSAL")
    assert line["n_gpus"] == 2 and line["config"]["total_keys"] == 1 << 15 and line["config"]["keys_per_gpu"] == 1 << 14
    assert line["scaling"] == "strong" and "sharded 2x" in line["config"]["workload"]
    assert line["config"]["verified"].startswith("bit-exact vs a host sort of all 32768 keys")
    assert "waves" in line["config"]["parallelism"]
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 0
    assert line["roofline"]["bound"] == "hbm" and "sharded_phases_ms" in line


def test_bench_rehearsal_weak_scaling_and_payload():
    proc, line = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-keys", "13", "--dtype", "int64", "--payload", "--no-cpu-baseline"],
                            {"RSX_BENCH_REHEARSAL": "1"})
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert line["scaling"] == "weak" and line["config"]["keys_per_gpu"] == 1 << 13 and line["n_gpus"] == 2
    assert "cpu_baseline" not in line


def test_bench_launcher_propagates_a_failing_rank():
    """A rank that dies must make `python bench.py --gpus N` exit non-zero and print no result line."""
    proc, line = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--total-log2-keys", "0"], {"RSX_BENCH_REHEARSAL": "1"}, timeout=300)
    assert proc.returncode != 0 and line is None            # 2^0 keys do not divide over two ranks


def test_bench_rehearsal_eight_ranks_is_config_4_shaped():
    """The driver's scaling run ends at `--gpus 8`: eight ranks, two pipelined waves of one top-nibble bucket per rank, the
    capacity verdict from gathered data, the verification gather — rehearsed on the CPU test double (2^17 keys in total)."""
    proc, line = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "1", "--total-log2-keys", "17", "--cpu-sample-log2", "10"],
                            {"RSX_BENCH_REHEARSAL": "1"}, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert line["n_gpus"] == 8 and line["config"]["keys_per_gpu"] == 1 << 14 and line["scaling"] == "strong"
    assert "waves" in line["config"]["parallelism"] and "x8" in line["config"]["parallelism"]
    assert line["config"]["verified"].startswith("bit-exact vs a host sort of all 131072 keys")


def test_xcd_tile_map_visits_every_tile_exactly_once(rsx):
    """rsx_tile_map = the launch geometry of every tile kernel as host arithmetic (rsx::tile_of_block, grid_for): whatever
    the key count, tile size and XCD phase, each tile has exactly one workgroup, surplus workgroups name tiles past the
    end, and with the stagger XCD x (workgroup index mod 8) still walks one contiguous range, entered x * phase tiles in."""
    import numpy as np
    for n, tile_keys in ((1, 4096), (4096, 4096), (4097, 4096), (100003, 1024), (1 << 20, 1024), ((1 << 22) + 77, 4096),
                         (2130003 * 4, 4096), (1 << 26, 4096), (1 << 28, 4096), (250_000_000, 4096)):
        want_tiles = (n + tile_keys - 1) // tile_keys
        for remap in (True, False):
            for phase in (-1, 0, 1, 5, 1024, 1 << 20):
                tiles, ntiles = rsx.tile_map(n, tile_keys, remap, phase)
                assert ntiles == want_tiles
                live = tiles[tiles < ntiles]
                assert live.size == ntiles and np.array_equal(np.sort(live), np.arange(ntiles, dtype=np.uint32)), (n, tile_keys, remap, phase)
                if not remap:
                    assert np.array_equal(tiles, np.arange(ntiles, dtype=np.uint32))
                    continue
                per_xcd = (ntiles + 7) // 8
                assert tiles.size == 8 * per_xcd
                for x in range(8):
                    mine = tiles[x::8].astype(np.int64)
                    assert mine.min() >= x * per_xcd and mine.max() < (x + 1) * per_xcd       # its own contiguous range
                    steps = np.diff(mine)
                    assert np.count_nonzero(steps != 1) <= 1                                   # one wrap at most
    # the default phase at the headline size: an eighth of a range
    tiles, _ = rsx.tile_map(1 << 28)
    assert [int(t) for t in tiles[:8]] == [x * 8192 + x * 1024 for x in range(8)]
    with pytest.raises(rsx.RadixSortError):
        rsx.tile_map(1 << 28, 4096, True, -2)
