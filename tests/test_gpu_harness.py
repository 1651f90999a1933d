"""GPU runs of the C++ host mirror: the reference's integration test shape
(tests/tests.cpp: 4 key types x 5 datasets through CRadixSortTask) and the basic_sort
example, as compiled binaries over the C ABI."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "radix-sort_amd", "host", "bin")


def _run(args, timeout=900):
    return subprocess.run(args, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [1000, 1 << 16, (1 << 20) + 5])
def test_main_test_matrix(n):
    """All 20 (type, dataset) tasks validate: CPU radix == std::sort, GPU == std::sort,
    GPU == CPU radix (src/CRadixSortTask.cpp:225-252 plus the direct comparison)."""
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", str(n)])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert proc.stdout.count("GOLD TEST PASSED!") == 20
    assert "FAILED" not in proc.stdout and "INVALID RESULTS" not in proc.stdout
    assert "20/20 task runs validated" in proc.stdout


def test_main_test_with_permutation_and_stepwise():
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", "50000", "--with-permutation", "--stepwise"])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert proc.stdout.count("Validation of GPU permutation (stable argsort) has passed") == 20


def test_perf_csv_schema_on_stdout():
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", "65536", "--perf-csv-to-stdout", "--perf-to-stdout"])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    header = "NumElements,Datatype,Dataset,avgHistogram,avgScan,avgPaste,avgReorder,avgTotalGPU,avgTotalSTLCPU,avgTotalRDXCPU"
    assert proc.stdout.count(header) == 20
    rows = re.findall(r"^65536,(u?int(?:32|64)_t),([A-Za-z ]+),", proc.stdout, flags=re.M)
    assert len(rows) == 20 and {r[0] for r in rows} == {"uint32_t", "int32_t", "uint64_t", "int64_t"}
    assert "reorder pass:" in proc.stdout and "% of the 8000 GB/s HBM3E peak" in proc.stdout


def test_basic_sort_example():
    proc = _run([os.path.join(BIN, "basic_sort")])
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert "Result: PASSED" in proc.stdout
    proc = _run([os.path.join(BIN, "basic_sort"), "1000003"])
    assert proc.returncode == 0 and "Result: PASSED" in proc.stdout
    proc = _run([os.path.join(BIN, "basic_sort"), "77777", "--int64", "--argsort", "--pinned"])
    assert proc.returncode == 0 and "Result: PASSED" in proc.stdout, proc.stdout[-1000:] + proc.stderr[-1000:]
    # the same five calls on the sharded engine: 8 rank threads on the one GPU, both exchanges
    proc = _run([os.path.join(BIN, "basic_sort"), "1000003", "--ranks", "8", "--argsort"])
    assert proc.returncode == 0 and "Result: PASSED" in proc.stdout and "8 ranks, path waves," in proc.stdout, proc.stdout[-1000:] + proc.stderr[-1000:]
    proc = _run([os.path.join(BIN, "basic_sort"), "500000", "--int64", "--ranks", "4", "--peer-stores"])
    assert proc.returncode == 0 and "Result: PASSED" in proc.stdout and "path waves-p2p" in proc.stdout, proc.stdout[-1000:] + proc.stderr[-1000:]


def test_pinned_transfers_and_sweep_csv(tmp_path):
    """--pinned page-locks the host spans (rsx_pin_host); the sweep script folds per-run CSVs
    into one file with the reference's column order (Performance/performance.csv:1)."""
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", "300000", "--pinned", "--perf-to-stdout"])
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert "20/20 task runs validated" in proc.stdout
    out = tmp_path / "perf.csv"
    proc = _run(["bash", os.path.join(ROOT, "tools", "performance_sweep.sh"), "12", "10", str(out)])
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = out.read_text().strip().splitlines()
    assert lines[0].startswith("NumElements,Datatype,Dataset,avgHistogram,avgScan,avgPaste,avgReorder,avgTotalGPU,avgTotalSTLCPU,avgTotalRDXCPU")
    assert len(lines) == 1 + 3 * 20 and lines[1].startswith("4096,uint32_t,Zeros,")


@pytest.mark.parametrize("mode", ["--overlap", "--zero-copy"])
@pytest.mark.parametrize("n", [3000, (1 << 20) + 5])
def test_overlapped_and_zero_copy_timed_loops(mode, n):
    """End to end beyond upload -> sort -> download in sequence (reference: src/CRadixSortTask.cpp:357-378 times exactly
    that sequence; examples/visualize/visualize.cpp:801-854 sorts out of mapped memory): --overlap keeps two sorts in
    flight on three streams, --zero-copy sorts straight out of / into pinned host memory.  Every result — both
    alternating download buffers, and the array the last pass wrote into host memory — is validated."""
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", str(n), mode, "--with-permutation", "--perf-to-stdout"])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "20/20 task runs validated" in proc.stdout and "FAILED" not in proc.stdout
    assert proc.stdout.count("Validation of GPU permutation (stable argsort) has passed") == 20


def test_pipeline_submit_through_the_binding():
    """rsx_pipeline_submit / rsx_pipeline_wait from Python: six jobs of different inputs through two slots."""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    m = entry.load_package()
    n = 500003
    rng = np.random.default_rng(9)
    with m.Engine("int32", n, payload=True) as e:
        ins = [rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int32) for _ in range(6)]
        outs = [np.empty(n, dtype=np.int32) for _ in range(6)]
        perm = np.arange(n, dtype=np.uint32)
        pouts = [np.empty(n, dtype=np.uint32) for _ in range(6)]
        for a in ins + outs + pouts + [perm]:
            e.pin_host(a)
        for k, o, po in zip(ins, outs, pouts):
            e.pipeline_submit(k, o, perm, po)
        e.pipeline_wait()
        for k, o, po in zip(ins, outs, pouts):
            assert np.array_equal(o, np.sort(k)) and np.array_equal(po, np.argsort(k, kind="stable").astype(np.uint32))
        for a in ins + outs + pouts + [perm]:
            e.unpin_host(a)


def test_harness_with_8bit_digits():
    """--radix-bits 8 through the host mirror: the 4x5 matrix validates against std::sort and RadixSortCPU as with 4-bit digits."""
    proc = _run([os.path.join(BIN, "rsx_tests"), "--num-elements", str((1 << 18) + 3), "--radix-bits", "8", "--with-permutation", "--perf-csv-to-stdout"])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "20/20 task runs validated" in proc.stdout and "FAILED" not in proc.stdout


# --------------------------------------------------------------------------- the sharded engine behind the same harness
@pytest.mark.parametrize("extra", [
    ["--ranks", "8", "--num-elements", str((1 << 20) + 5)],                                              # 8 rank threads, loopback, all-to-all per wave
    ["--ranks", "8", "--num-elements", "300000", "--exchange", "peer-stores", "--with-permutation"],     # peer stores: device-side plan, push + fence per wave
    ["--ranks", "4", "--num-elements", "200000", "--partition-bits", "8", "--radix-bits", "8", "--with-permutation"],
    ["--ranks", "2", "--num-elements", "5000", "--partition-bits", "1", "--exchange", "peer-stores"],
    ["--ranks", "3", "--num-elements", "100000", "--with-permutation"],                                  # not a power of two: splitter path for every dataset
    ["--sharded", "--comm", "rccl", "--num-elements", "100000", "--with-permutation"],                   # ONE rank through real RCCL (ncclCommInitAll, grouped send/recv to self)
    ["--sharded", "--comm", "rccl", "--num-elements", "100000", "--exchange", "peer-stores"],            # ... and the ncclAllGather + ncclAllReduce fence of the peer-store path
])
def test_sharded_harness_matrix(extra):
    """`rsx_tests --gpus N` (here: rank THREADS on the box's one GPU, `--ranks R`): RadixSortMultiGPU<T> behind CRadixSortTask's five
    calls (the reference's seam: src/CRadixSortTask.cpp:289-314, tests/CTestBase.cpp:20-67) — contiguous shards of every dataset,
    one engine + one communication stream per rank, count -> scatter beside the row exchange -> pipelined exchange -> wave sorts, the
    splitter path for inputs that do not balance on their top bits (Zeros, Range, InvertedRange), results concatenated in rank order and
    validated against std::sort AND RadixSortCPU for all 20 (type, dataset) tasks; a failed validation fails the run."""
    proc = _run([os.path.join(BIN, "rsx_tests"), "-v"] + extra)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "20/20 task runs validated" in proc.stdout and "FAILED" not in proc.stdout and "INVALID RESULTS" not in proc.stdout
    if "--with-permutation" in extra:
        assert proc.stdout.count("Validation of GPU permutation (stable argsort) has passed") == 20
    paths = set(re.findall(r"^path: ([a-z0-9-]+)$", proc.stdout, flags=re.M))
    if "--ranks" in extra and extra[extra.index("--ranks") + 1] == "3":
        assert paths == {"split"}
    else:
        assert ("waves-p2p" if "peer-stores" in extra else "waves") in paths and paths <= {"waves", "waves-p2p", "split", "equal"}
    if "rccl" in extra:
        assert "communicator: RCCL" in proc.stdout
    else:
        assert "communicator: loopback" in proc.stdout


def test_sharded_harness_perf_csv_names_the_gpu_count():
    proc = _run([os.path.join(BIN, "rsx_tests"), "--ranks", "4", "--num-elements", "65536", "--perf-csv-to-stdout"])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    rows = [l for l in proc.stdout.splitlines() if l.startswith("65536,")]
    assert len(rows) == 20 and all(r.split(",")[-1] == "1" for r in rows)          # four rank threads, ONE GPU


def test_sharded_harness_refuses_more_gpus_than_the_box_has():
    proc = _run([os.path.join(BIN, "rsx_tests"), "--gpus", "16", "--num-elements", "4096"])
    assert proc.returncode != 0 and "HIP device(s) visible" in proc.stderr
