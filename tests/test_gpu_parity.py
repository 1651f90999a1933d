"""GPU parity tests: the HIP path (through the C ABI) against the oracle.

Bit-exact everywhere — integer work, no tolerance.  Inputs are the reference's dataset
families (Dataset.h:84-137) at sizes the oracle finishes in seconds, the committed
golden digests, edge sizes (empty, 1, non-multiples of the tile and of 1024), and the
full BASELINE sizes through size-independent properties (sortedness + multiset
checksums).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = ["uint32", "int32", "uint64", "int64"]
KINDS = ["Zeros", "Range", "InvertedRange", "SeededUniform", "Random"]   # tests.cpp:20-26


@pytest.fixture(scope="module")
def mod(rsx):
    assert rsx.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return rsx


def _sort(mod, keys, payload=None):
    return mod.sort_host(keys, payload)


# --------------------------------------------------------------------------- whole sort
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kind", KINDS)
def test_sort_matches_oracle_4x5_matrix(mod, oracle, dt, kind):
    """The reference's own test matrix (tests/tests.cpp:18-27,83-87) at 2^16 keys."""
    keys = oracle.dataset(kind, dt, 1 << 16)
    got = _sort(mod, keys)
    assert np.array_equal(got, oracle.radix_sort(keys))       # vs CRadixSortCPU restatement
    assert np.array_equal(got, oracle.std_sort(keys))         # vs std::sort, the reference's ground truth


def test_sort_matches_golden_digests(mod, oracle, golden):
    for row in golden["datasets"]:
        keys = oracle.dataset(row["kind"], row["dtype"], row["n"])
        got = _sort(mod, keys)
        assert oracle.digest(got) == row["sorted_digest"], row


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 12288, 100003])
def test_sort_ragged_sizes(mod, oracle, dt, n):
    rng = np.random.default_rng(n)
    info = np.iinfo(dt)
    keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    assert np.array_equal(_sort(mod, keys), np.sort(keys))


@pytest.mark.parametrize("dt", DT)
def test_sort_few_distinct_and_extremes(mod, oracle, dt):
    rng = np.random.default_rng(5)
    info = np.iinfo(dt)
    pool = np.array([info.min, info.min + 1, -1 if info.min < 0 else 1, 0, 1, 15, 16, 255, 256, info.max - 1, info.max], dtype=dt)
    keys = pool[rng.integers(0, pool.size, size=50000)]
    got = _sort(mod, keys)
    assert np.array_equal(got, np.sort(keys))
    if keys.max() == info.max:      # inside the oracle's correct domain (full round count)
        assert np.array_equal(got, oracle.radix_sort(keys))


def test_sort_empty(mod):
    with mod.Engine("uint32", 16) as e:
        e.upload(np.empty(0, dtype=np.uint32))
        e.sort()
        assert e.download().size == 0


def test_sort_2pow24_random(mod, oracle):
    keys = oracle.dataset("Random", "uint32", 1 << 24)
    got = _sort(mod, keys)
    assert np.array_equal(got, np.sort(keys))


# --------------------------------------------------------------------------- payload
@pytest.mark.parametrize("dt", DT)
def test_payload_is_stable_argsort(mod, oracle, dt):
    """Payload semantics (unpinned in the reference, defined as stable argsort)."""
    rng = np.random.default_rng(11)
    info = np.iinfo(dt)
    n = 70001
    keys = rng.integers(0, 40, size=n, dtype=dt)      # heavy ties exercise stability
    keys[::97] = info.max
    if info.min < 0:
        keys[1::97] = info.min
    perm = np.arange(n, dtype=np.uint32)
    ks, ps = _sort(mod, keys, perm)
    want_k, want_p = oracle.radix_sort(keys, perm)
    assert np.array_equal(ks, want_k)
    assert np.array_equal(ps, want_p)
    assert np.array_equal(ps, np.argsort(keys, kind="stable").astype(np.uint32))


def test_payload_uint64_random_full_entropy(mod, oracle):
    n = 1 << 18
    keys = oracle.dataset("SeededUniform", "uint64", n)
    perm = np.arange(n, dtype=np.uint32)
    ks, ps = _sort(mod, keys, perm)
    assert np.array_equal(ks, np.sort(keys))
    assert np.array_equal(keys[ps], ks)
    assert np.array_equal(ps, oracle.stable_argsort(keys, perm))


# --------------------------------------------------------------------------- single steps
def _digits(keys, pass_, signed_bits):
    u = keys.view(np.uint32 if keys.dtype.itemsize == 4 else np.uint64).copy()
    if signed_bits:
        u ^= (np.uint64(1) << np.uint64(signed_bits - 1)).astype(u.dtype)
    return ((u >> np.array(pass_ * 4, dtype=u.dtype)) & np.array(15, dtype=u.dtype)).astype(np.int64)


@pytest.mark.parametrize("dt,n", [("uint32", 40000), ("int32", 4096 * 3), ("uint64", 10007), ("int64", 65536)])
def test_histogram_scan_paste_reorder_steps(mod, oracle, dt, n):
    """Each kernel against its host definition: table == per-tile digit counts
    (RadixSort.cl:48-70 semantics at tile granularity), scan+paste == exclusive
    prefix of the flattened [digit][tile] table (RadixSort.cl:125-197), reorder == one
    stable counting pass (RadixSort.cl:96-118)."""
    keys = oracle.dataset("SeededUniform", dt, n)
    signed_bits = keys.dtype.itemsize * 8 if keys.dtype.kind == "i" else 0
    with mod.Engine(dt, n) as e:
        e.upload(keys)
        g = e.geometry()
        tile, ntiles = g.tile_keys, g.num_tiles
        for pass_ in (0, 3, g.num_passes - 1):
            d = _digits(keys, pass_, signed_bits)
            want = np.zeros((16, ntiles), dtype=np.uint32)
            for t in range(ntiles):
                want[:, t] = np.bincount(d[t * tile:(t + 1) * tile], minlength=16)
            e.histogram(pass_)
            _, table = e.download(hist_cap=16 * ntiles)
            assert np.array_equal(table.reshape(16, ntiles), want), pass_
            e.scan()
            e.paste()
            _, table = e.download(hist_cap=16 * ntiles)
            flat = want.reshape(-1).astype(np.uint64)
            excl = np.concatenate([[0], np.cumsum(flat)[:-1]]).astype(np.uint32)
            assert np.array_equal(table, excl), pass_
            e.reorder(pass_)
            got = e.download()
            assert np.array_equal(got, keys[np.argsort(d, kind="stable")]), pass_
            e.upload(keys)      # back to the original order for the next pass under test


def test_globsum_is_exclusive_scan_of_block_sums(mod, oracle):
    """Level 2 of the table scan: globsum[digit][group] (a group = scan_block consecutive tiles of
    one digit) holds, after rsx_scan, the exclusive prefix of the group sums in digit-major order
    (the reference's globsum after its second scanhistograms launch, RadixSortGPU.cpp:115-152)."""
    n = (1 << 21) + 12345
    keys = oracle.dataset("Random", "uint32", n)
    with mod.Engine("uint32", n) as e:
        e.upload(keys)
        e.histogram(0)
        g = e.geometry()
        _, counts = e.download(hist_cap=16 * g.num_tiles)
        e.scan()
        _, gs = e.download(globsum_cap=int(g.num_scan_blocks))
        ngroups = g.num_scan_blocks // 16
        rows = counts.reshape(16, g.num_tiles).astype(np.uint64)
        pad = ngroups * g.scan_block - g.num_tiles
        rows = np.concatenate([rows, np.zeros((16, pad), dtype=np.uint64)], axis=1)
        sums = rows.reshape(16, ngroups, g.scan_block).sum(axis=2).reshape(-1)
        assert np.array_equal(gs, np.concatenate([[0], np.cumsum(sums)[:-1]]).astype(np.uint32))


def test_xcd_remap_does_not_change_results(mod, oracle):
    keys = oracle.dataset("Random", "uint32", 300001)
    outs = []
    for remap in (0, 1):
        with mod.Engine("uint32", keys.size) as e:
            e.set_option(mod.OPT_XCD_REMAP, remap)
            e.upload(keys)
            e.sort()
            outs.append(e.download())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], np.sort(keys))


@pytest.mark.parametrize("dt,n,payload", [("uint32", 1 << 20, False), ("int32", 99991, True), ("uint64", 262147, False),
                                          ("int64", 4097, True), ("uint32", 4096, False), ("uint32", 5, True)])
@pytest.mark.parametrize("kind", ["SeededUniform", "Zeros", "InvertedRange"])
def test_lookahead_histogram_equals_separate_histogram_passes(mod, oracle, dt, n, payload, kind):
    """rsx_sort's fused path (reorder of pass p counts pass p+1's digits per output tile)
    against the plain histogram -> scan -> paste -> reorder loop: identical bytes, and the
    last pass's scanned table is identical too."""
    keys = oracle.dataset(kind, dt, n)
    perm = np.arange(n, dtype=np.uint32) if payload else None
    outs = []
    for la, small in ((0, 0), (1, 1), (1, 0), (0, 1)):
        with mod.Engine(dt, n, payload=payload) as e:
            e.set_option(mod.OPT_LOOKAHEAD, la)
            e.set_option(mod.OPT_SMALL_SCAN, small)
            e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 0)      # the table read-back compared below exists in the 4096-key geometry only
            e.upload(keys, perm)
            e.sort()
            ntab = 16 * e.geometry().num_tiles
            outs.append(e.download(want_perm=payload, hist_cap=ntab))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    assert np.array_equal(outs[1][0], np.sort(keys))
    if payload:
        assert np.array_equal(outs[1][1], np.argsort(keys, kind="stable").astype(np.uint32))


@pytest.mark.parametrize("dt,n", [("uint32", 65536), ("int32", 1024), ("uint64", 1 << 20), ("int64", 10240)])
def test_reference_geometry_diagnostics(mod, oracle, dt, n):
    """m_hHistograms / m_hGlobsum as the reference downloads them (RadixSortGPU.cpp:412-428):
    16384-word pasted [digit][group][item] table and 512 scanned block sums of the LAST pass,
    against the host emulation of the reference's pass structure; for Random<uint32> 2^16 also
    the sample values the survey recorded from that emulation."""
    keys = oracle.dataset("Random", dt, n)
    want_sorted, want_table, want_gs = oracle.emulate_reference_gpu(keys)
    for la in (1, 0):
        with mod.Engine(dt, n) as e:
            e.set_option(mod.OPT_LOOKAHEAD, la)
            e.set_option(mod.OPT_REF_DIAGNOSTICS, 1)
            e.upload(keys)
            e.sort()
            got, table, gs = e.download(hist_cap=16384, globsum_cap=512)
        assert np.array_equal(got, want_sorted)
        assert np.array_equal(table, want_table)
        assert np.array_equal(gs, want_gs)
    if dt == "uint32" and n == 65536:
        assert [int(v) for v in table[:4]] == [0, 5, 11, 17] and int(table[16383]) == 65534
        assert int(gs[1]) == 121 and int(gs[511]) == 65400


# --------------------------------------------------------------------------- reference semantics
def test_fill_pad_value_and_rounded_length(mod, oracle):
    """padGPUData writes max()-1 from a byte offset (RadixSortGPU.cpp:270-285); the sort
    covers exactly the uploaded (rounded) length."""
    for dt in DT:
        info = np.iinfo(dt)
        n, rounded = 1000, 1024
        host = np.zeros(rounded, dtype=dt)
        host[:n] = oracle.dataset("Random", dt, n)
        with mod.Engine(dt, rounded) as e:
            e.upload(host)
            e.fill_pad(n * host.dtype.itemsize)
            padded = e.download()
            assert np.array_equal(padded[:n], host[:n])
            assert np.all(padded[n:] == info.max - 1)
            e.sort()
            assert np.array_equal(e.download(), np.sort(padded))
            # the reference's real call order: pad first, upload afterwards overwrites the pad
            e.fill_pad(n * host.dtype.itemsize)
            e.upload(host)
            e.sort()
            assert np.array_equal(e.download(), np.sort(host))


def test_result_lands_in_input_buffer_after_even_passes(mod, oracle):
    """8 / 16 passes end in the buffer named inputKeys (RadixSortGPU.cpp:263-266,394-400)."""
    keys = oracle.dataset("Random", "uint64", 5000)
    with mod.Engine("uint64", 5000) as e:
        e.upload(keys)
        before, _ = e.result_device()
        e.sort()
        after, _ = e.result_device()
        assert before == after
        assert np.array_equal(e.download(), np.sort(keys))


def test_timings_profile_mode(mod, oracle):
    keys = oracle.dataset("Random", "uint32", 1 << 20)
    with mod.Engine("uint32", keys.size) as e:
        e.set_option(mod.OPT_PROFILE, 1)
        e.upload(keys)
        e.sort()
        t = e.timings(reset=True)
        # look-ahead: one histogram launch; 256 tiles: no scan launch at all (every reorder workgroup scans for itself)
        assert t.histogram.n == 1 and t.reorder.n == 8 and t.paste.n == 0 and t.scan.n == 0
        assert np.array_equal(e.download(), np.sort(keys))
        e.set_option(mod.OPT_SELF_SCAN, 0)
        e.upload(keys)
        e.sort()
        t = e.timings(reset=True)
        # ... or the one-workgroup scan: scan+paste in one launch per pass
        assert t.histogram.n == 1 and t.reorder.n == 8 and t.paste.n == 0 and t.scan.n == 8
        assert np.array_equal(e.download(), np.sort(keys))
        e.set_option(mod.OPT_SMALL_SCAN, 0)
        e.upload(keys)
        e.sort()
        t = e.timings(reset=True)
        assert t.histogram.n == 1 and t.reorder.n == 8 and t.paste.n == 0 and t.scan.n == 8      # the fused scan: one launch per pass
        assert np.array_equal(e.download(), np.sort(keys))
        e.set_option(mod.OPT_FUSED_SCAN, 0)
        e.upload(keys)
        e.sort()
        t = e.timings(reset=True)
        assert t.histogram.n == 1 and t.reorder.n == 8 and t.paste.n == 8      # scan #1, then scan #2 + paste in one launch
        assert t.scan.n == 8
        assert np.array_equal(e.download(), np.sort(keys))
        e.set_option(mod.OPT_LOOKAHEAD, 0)
        e.upload(keys)
        e.sort()
        t = e.timings()
        assert t.histogram.n == 8 and t.reorder.n == 8 and t.paste.n == 8
        assert t.scan.n == 16                      # two scan launches per pass (RadixSortGPU.cpp:108,147)
        assert t.total.n == 1 and t.total.sum_ms >= t.reorder.sum_ms > 0
        assert t.reorder.min_ms <= t.reorder.avg_ms <= t.reorder.max_ms
        assert np.array_equal(e.download(), np.sort(keys))


def test_errors_map_to_operation_status(mod):
    with pytest.raises(mod.RadixSortError) as ei:
        mod.Engine("uint32", 0)
    assert ei.value.status == 7                     # RESIZE_FAILED
    with mod.Engine("uint32", 100) as e:
        with pytest.raises(mod.RadixSortError) as ei:
            e.upload(np.zeros(101, dtype=np.uint32))
        assert ei.value.status == 3                 # DATA_UPLOAD_FAILED
        with pytest.raises(mod.RadixSortError) as ei:
            e.histogram(8)
        assert ei.value.status == 4                 # CALCULATION_FAILED
    with mod.Engine("uint32", 100, payload=True) as e:
        with pytest.raises(mod.RadixSortError) as ei:
            e.upload(np.zeros(10, dtype=np.uint32), None)
        assert ei.value.status == 1                 # HOST_BUFFERS_FAILED


# --------------------------------------------------------------------------- device-resident API
def test_sort_from_torch_tensor_leaves_input_untouched(mod, oracle):
    import torch
    keys = oracle.dataset("Random", "int32", 123457)
    t = torch.from_numpy(keys).cuda()
    with mod.Engine("int32", keys.size) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.sort_from(t.data_ptr(), keys.size)
        out = torch.empty_like(t)
        e.copy_result(out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), keys)
        assert np.array_equal(out.cpu().numpy(), np.sort(keys))
        assert np.array_equal(e.download(), np.sort(keys))


@pytest.mark.parametrize("bits", [1, 2, 3, 4])
def test_partition_by_top_bits(mod, oracle, bits):
    import torch
    n = 200003
    keys = oracle.dataset("SeededUniform", "uint32", n)
    perm = np.arange(n, dtype=np.uint32)
    tk, tp = torch.from_numpy(keys).cuda(), torch.from_numpy(perm.view(np.int32)).cuda()
    ok, op = torch.empty_like(tk), torch.empty_like(tp)
    with mod.Engine("uint32", n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        offs = e.partition(tk.data_ptr(), n, 32 - bits, bits, ok.data_ptr(), tp.data_ptr(), op.data_ptr())
        torch.cuda.synchronize()
    d = (keys >> np.uint32(32 - bits)).astype(np.int64)
    order = np.argsort(d, kind="stable")
    assert np.array_equal(ok.cpu().numpy(), keys[order])
    assert np.array_equal(op.cpu().numpy().view(np.uint32), perm[order])
    counts = np.bincount(d, minlength=1 << bits)
    assert offs == [0] + [int(v) for v in np.cumsum(counts)]


# --------------------------------------------------------------------------- BASELINE sizes
def _checksums(a):
    u = a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64).astype(np.uint64, copy=False)
    return int(np.bitwise_xor.reduce(u)), int(u.sum(dtype=np.uint64)), int((u * u).sum(dtype=np.uint64))


@pytest.mark.parametrize("kind", ["Random", "Zeros", "InvertedRange", "SeededUniform"])
def test_full_size_2pow28_uint32_properties(mod, oracle, kind):
    """BASELINE configs 2 and 5 at full size: output is ascending and is a permutation
    of the input (xor / sum / sum-of-squares checksums mod 2^64)."""
    n = 1 << 28
    keys = oracle.dataset(kind, "uint32", n)
    got = _sort(mod, keys)
    assert got.size == n
    assert bool(np.all(got[:-1] <= got[1:]))
    assert _checksums(got) == _checksums(keys)
    if kind == "InvertedRange":
        assert np.array_equal(got, keys[::-1])
    if kind == "Zeros":
        assert not got.any()


def test_full_size_2pow28_uint64_with_payload(mod, oracle):
    """BASELINE config 3: 2^28 uint64 keys + uint32 payload (h_Permut = iota)."""
    n = 1 << 28
    keys = oracle.dataset("SeededUniform", "uint64", n)
    perm = np.arange(n, dtype=np.uint32)
    ks, ps = _sort(mod, keys, perm)
    assert bool(np.all(ks[:-1] <= ks[1:]))
    assert _checksums(ks) == _checksums(keys)
    assert np.array_equal(keys[ps], ks)               # payload followed its key
    ties = np.flatnonzero(ks[:-1] == ks[1:])
    assert bool(np.all(ps[ties] < ps[ties + 1]))      # stability on equal keys


def test_full_size_2pow28_uint32_random_bit_exact_vs_oracle(mod, oracle):
    """BASELINE config 2 at full size against the checker itself (reference: the GPU result is
    memcmp'd with the CPU referees, src/CRadixSortTask.cpp:225-252): every one of the 2^28 output
    keys equals RadixSortCPU's (restatement; and the reference's own headers where oracle/_ref
    was built)."""
    n = 1 << 28
    keys = oracle.dataset("Random", "uint32", n)
    got = _sort(mod, keys)
    want = oracle.radix_sort(keys)
    assert np.array_equal(got, want)
    from _oracle import RefOracle
    if RefOracle.available():
        assert np.array_equal(got, RefOracle().radix_sort(keys))


def test_2pow26_uint64_with_payload_bit_exact_vs_oracle(mod, oracle):
    """BASELINE config 3's shape (uint64 keys with 64-bit entropy + uint32 payload = iota) at 2^26
    against the oracle extended with a payload array: keys AND the permutation, bit for bit."""
    n = 1 << 26
    keys = oracle.dataset("SeededUniform", "uint64", n)
    keys[1::2] = keys[0:-1:2]                       # every key twice: the payload order of ties is checked too
    perm = np.arange(n, dtype=np.uint32)
    ks, ps = _sort(mod, keys, perm)
    wk, wp = oracle.radix_sort(keys, perm)
    assert np.array_equal(ks, wk)
    assert np.array_equal(ps, wp)


@pytest.mark.parametrize("kind", ["Zeros", "InvertedRange", "SeededUniform", "Range"])
def test_config5_inputs_2pow26_bit_exact_vs_oracle(mod, oracle, kind):
    """BASELINE config 5's adversarial inputs (Dataset.h:84-137) at 2^26 against the oracle."""
    keys = oracle.dataset(kind, "uint32", 1 << 26)
    if kind == "Range":
        keys[-1] = np.uint32(0xFFFFFFFF)           # keeps the oracle's round count at 8 (its domain)
    assert np.array_equal(_sort(mod, keys), oracle.radix_sort(keys))


# --------------------------------------------------------------------------- digit width
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kind", KINDS)
def test_8bit_digits_match_oracle_4x5_matrix(mod, oracle, dt, kind):
    """The digit width is a parameter of the reference (_NUM_BITS_PER_RADIX, src/Parameters.h:25); with 8-bit digits the
    reference's own test matrix (tests/tests.cpp:18-27,83-87) must give the same bytes as with 4-bit digits."""
    keys = oracle.dataset(kind, dt, (1 << 16) + 13)
    with mod.Engine(dt, keys.size) as e:
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(keys)
        e.sort()
        got = e.download()
    assert np.array_equal(got, oracle.std_sort(keys))
    if keys.max() == np.iinfo(dt).max:      # inside the oracle's correct domain (full round count)
        assert np.array_equal(got, oracle.radix_sort(keys))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n", [4097, 5000, 8191, 12288, 100003, (1 << 20) + 77])
def test_8bit_digits_ragged_sizes_and_payload(mod, oracle, dt, n):
    rng = np.random.default_rng(n)
    info = np.iinfo(dt)
    keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    keys[::5] = keys[3]                                # ties: the payload order is checked too
    perm = np.arange(n, dtype=np.uint32)
    with mod.Engine(dt, n, payload=True) as e:
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(keys, perm)
        e.sort()
        k, p = e.download(want_perm=True)
        assert np.array_equal(k, np.sort(keys)) and np.array_equal(p, np.argsort(keys, kind="stable").astype(np.uint32))
        e.sort()                                       # again, from the other ping-pong buffer: identity
        k2, p2 = e.download(want_perm=True)
        assert np.array_equal(k2, k) and np.array_equal(p2, p)


def test_8bit_digits_pass_ranges_external_buffers_and_golden(mod, oracle, golden):
    import torch
    for row in golden["datasets"]:
        if row["n"] <= 4096:
            continue
        keys = oracle.dataset(row["kind"], row["dtype"], row["n"])
        with mod.Engine(row["dtype"], keys.size) as e:
            e.set_option(mod.OPT_RADIX_BITS, 8)
            e.upload(keys)
            e.sort()
            assert oracle.digest(e.download()) == row["sorted_digest"], row
    n = 70001
    keys = oracle.dataset("SeededUniform", "int64", n, seed=5)
    t = torch.from_numpy(keys).cuda()
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    u = keys.view(np.uint64) ^ np.uint64(1 << 63)
    with mod.Engine("int64", n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.set_option(mod.OPT_RADIX_BITS, 8)
        # whole bytes -> 8-bit passes; an odd range from a byte boundary -> 8-bit passes + one 4-bit pass for the last nibble
        # (the sharded sort's local passes 0 .. P-2); anything else -> the 4-bit chain
        for first, last in ((0, 16), (0, 14), (2, 8), (1, 16), (0, 5), (0, 15), (2, 7), (0, 3), (4, 15), (0, 1)):
            out = torch.zeros(n + 5, dtype=t.dtype, device="cuda")
            pout = torch.zeros(n + 5, dtype=torch.int32, device="cuda")
            e.sort_from_to(t.data_ptr(), n, first, last, out[2:].data_ptr(), pay.data_ptr(), pout[2:].data_ptr())
            torch.cuda.synchronize()
            field = (u >> np.uint64(4 * first)) & np.uint64((1 << (4 * (last - first))) - 1) if last - first < 16 else u
            order = np.argsort(field, kind="stable")
            got = out.cpu().numpy()
            assert np.array_equal(got[2:2 + n], keys[order]) and not got[:2].any() and not got[2 + n:].any(), (first, last)
            assert np.array_equal(pout.cpu().numpy().view(np.uint32)[2:2 + n], order.astype(np.uint32))
        assert np.array_equal(t.cpu().numpy(), keys)


def test_8bit_digits_full_size_2pow28_bit_exact_vs_oracle(mod, oracle):
    n = 1 << 28
    keys = oracle.dataset("Random", "uint32", n)
    with mod.Engine("uint32", n) as e:
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(keys)
        e.sort()
        got = e.download()
    assert np.array_equal(got, oracle.radix_sort(keys))


@pytest.mark.parametrize("dt,kind", [("uint64", "SeededUniform"), ("uint32", "Random"), ("int64", "Zeros")])
def test_8bit_digits_full_size_2pow28_with_payload(mod, oracle, dt, kind):
    """The payload rows of the 8-bit workload matrix at their real size (BASELINE config 3's shape and the packed uint32 + payload
    scatter): ascending, a permutation of the input, the payload followed its key, equal keys keep their input order."""
    n = 1 << 28
    keys = oracle.dataset(kind, dt, n)
    with mod.Engine(dt, n, payload=True) as e:
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(keys, np.arange(n, dtype=np.uint32))
        e.sort()
        ks, ps = e.download(want_perm=True)
    assert bool(np.all(ks[:-1] <= ks[1:]))
    assert _checksums(ks) == _checksums(keys)
    assert np.array_equal(keys[ps], ks)
    ties = np.flatnonzero(ks[:-1] == ks[1:])
    assert bool(np.all(ps[ties] < ps[ties + 1]))
    if kind == "Zeros":
        assert np.array_equal(ps, np.arange(n, dtype=np.uint32))


# --------------------------------------------------------------------------- table scan variants
@pytest.mark.parametrize("dt,payload", [("uint32", False), ("int64", True)])
@pytest.mark.parametrize("n", [5000, 300001, (1 << 22) + 17, (1 << 24) + 4097])
def test_fused_scan_equals_the_separate_launches(mod, oracle, dt, payload, n):
    """The one-launch table scan (group sums handed over as tagged granules inside the launch) against scan #1 +
    scan #2/paste as separate launches: same keys, same payload, same final table and group sums."""
    keys = oracle.dataset("SeededUniform", dt, n, seed=n)
    perm = np.arange(n, dtype=np.uint32) if payload else None
    seen = []
    for fused, small in ((1, 0), (0, 0), (1, 1)):
        with mod.Engine(dt, n, payload=payload) as e:
            e.set_option(mod.OPT_SELF_SCAN, 0)
            e.set_option(mod.OPT_FUSED_SCAN, fused)
            e.set_option(mod.OPT_SMALL_SCAN, small)
            e.upload(keys, perm)
            for _ in range(3):                    # epochs advance; granules of earlier launches must never be taken for current ones
                e.sort()
            g = e.geometry()
            # (the one-workgroup scan of small tables leaves no group sums: asking for them is an error, checked below)
            has_sums = not (small and g.num_tiles <= 1024)
            out = e.download(want_perm=payload, hist_cap=int(g.table_len), globsum_cap=int(g.num_scan_blocks) if has_sums else 0)
            if not has_sums:
                with pytest.raises(mod.RadixSortError):
                    e.download(globsum_cap=int(g.num_scan_blocks))
                out = out + (None,)
            seen.append(out if isinstance(out, tuple) else (out,))
    want = np.sort(keys)
    for got in seen:
        assert np.array_equal(got[0], want)
    for a, b in zip(seen[0], seen[1]):            # fused vs separate launches: everything, incl. table and group sums
        assert np.array_equal(a, b)
    for a, b in zip(seen[0][:-1], seen[2][:-1]):  # the one-workgroup scan of small tables leaves no group sums
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dt,payload", [("uint32", False), ("int32", True), ("uint64", True), ("int64", False)])
@pytest.mark.parametrize("n", [4097, 8192, 70001, (1 << 20) + 5, 1 << 22])
def test_self_scan_equals_the_scan_launches(mod, oracle, dt, payload, n):
    """Tables of up to 1024 tiles take no scan launch: every reorder workgroup derives its own bases from the raw counts.
    Keys, payload and the last pass's table must equal what the chain with scan launches leaves; repeated sorts rotate
    the three count buffers through every phase."""
    keys = oracle.dataset("SeededUniform", dt, n, seed=n + 1)
    keys[::7] = keys[2]
    perm = np.arange(n, dtype=np.uint32) if payload else None
    seen = []
    for self_scan in (1, 0):
        with mod.Engine(dt, n, payload=payload) as e:
            e.set_option(mod.OPT_SELF_SCAN, self_scan)
            e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 0)      # (tiles of 1024 keys: test_small_tiles_sort_the_same)
            e.upload(keys, perm)
            for _ in range(4):
                e.sort()
            g = e.geometry()
            out = e.download(want_perm=payload, hist_cap=int(g.table_len))
            seen.append(out)
    assert np.array_equal(seen[0][0], np.sort(keys))
    if payload:
        e2 = np.argsort(keys, kind="stable").astype(np.uint32)
        with mod.Engine(dt, n, payload=True) as e:
            e.upload(keys, perm)
            e.sort()
            assert np.array_equal(e.download(want_perm=True)[1], e2)
    for a, b in zip(seen[0], seen[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dt,payload", [("uint32", False), ("int32", True), ("uint64", True), ("int64", False)])
@pytest.mark.parametrize("n", [4097, 5000, 65536, 100003, 1 << 20])
def test_small_tiles_sort_the_same(mod, oracle, dt, payload, n):
    """Self-scan sorts on tiles of 1024 keys (4 per thread): same keys, same stable payload order; also through
    rsx_sort_from_to with a partial pass range, and with the reference-geometry diagnostics of the host mirror."""
    keys = oracle.dataset("SeededUniform", dt, n, seed=n + 3)
    keys[::3] = keys[1]
    perm = np.arange(n, dtype=np.uint32) if payload else None
    with mod.Engine(dt, n, payload=payload) as e:
        e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 1 << 20)
        e.upload(keys, perm)
        for _ in range(4):
            e.sort()
        out = e.download(want_perm=payload)
    k = out[0] if payload else out
    assert np.array_equal(k, np.sort(keys))
    if payload:
        assert np.array_equal(out[1], np.argsort(keys, kind="stable").astype(np.uint32))
    if n % 1024 == 0 and not payload:
        want_sorted, want_table, want_gs = oracle.emulate_reference_gpu(keys)
        with mod.Engine(dt, n) as e:
            e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 1 << 20)
            e.set_option(mod.OPT_REF_DIAGNOSTICS, 1)
            e.upload(keys)
            e.sort()
            got, table, gs = e.download(hist_cap=16384, globsum_cap=512)
        assert np.array_equal(got, want_sorted) and np.array_equal(table, want_table) and np.array_equal(gs, want_gs)


def test_self_scan_tile_limit_option(mod, oracle):
    """RSX_OPT_SELF_SCAN_MAX_TILES moves the border between the self-scan chain and the chain with scan launches
    (clamped to 1024 tiles): same keys and table on either side of it."""
    n = (1 << 20) + 77
    keys = oracle.dataset("SeededUniform", "uint32", n, seed=5)
    seen = []
    for limit in (64, 1 << 20):
        with mod.Engine("uint32", n) as e:
            e.set_option(mod.OPT_SELF_SCAN_MAX_TILES, limit)
            e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 0)
            e.upload(keys)
            e.sort()
            seen.append(e.download(hist_cap=int(e.geometry().table_len)))
    assert np.array_equal(seen[0][0], np.sort(keys))
    for a, b in zip(seen[0], seen[1]):
        assert np.array_equal(a, b)
    with mod.Engine("uint32", 4096) as e:
        with pytest.raises(mod.RadixSortError):
            e.set_option(mod.OPT_SELF_SCAN_MAX_TILES, -1)


# --------------------------------------------------------------------------- XCD phase stagger (placement only)
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dt,payload", [("uint32", False), ("int64", True)])
@pytest.mark.parametrize("n", [(1 << 22) + 4096 * 8 + 77, 2130003, 1 << 23])
def test_xcd_phase_is_placement_only(mod, oracle, dt, payload, n, bits):
    """RSX_OPT_XCD_PHASE moves where each XCD enters its tile range (rsx::tile_of_block); every tile must still be
    sorted exactly once whatever the phase: lockstep (0), the default (-1), one tile, the largest phase the
    range admits, and one past it (falls back to lockstep).  Sizes: a ragged last range, an odd tile count, 2^23."""
    keys = oracle.dataset("SeededUniform", dt, n, seed=n + bits)
    keys[::5] = keys[2]
    perm = np.arange(n, dtype=np.uint32) if payload else None
    want = np.sort(keys)
    want_perm = np.argsort(keys, kind="stable").astype(np.uint32) if payload else None
    tiles_per_xcd = ((n + 4095) // 4096 + 7) // 8
    limit = (tiles_per_xcd - 1) // 7
    for phase in (0, -1, 1, limit, limit + 1):
        with mod.Engine(dt, n, payload=payload) as e:
            e.set_option(mod.OPT_RADIX_BITS, bits)
            e.set_option(mod.OPT_XCD_PHASE, phase)
            e.upload(keys, perm)
            e.sort()
            out = e.download(want_perm=payload)
        k = out[0] if payload else out
        assert np.array_equal(k, want), phase
        if payload:
            assert np.array_equal(out[1], want_perm), phase
    with mod.Engine(dt, 4096) as e:
        with pytest.raises(mod.RadixSortError):
            e.set_option(mod.OPT_XCD_PHASE, -2)


# --------------------------------------------------------------------------- one-workgroup sort of small inputs
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n", [1, 2, 17, 1000, 1024, 4095, 4096])
def test_tile_sort_equals_the_pass_chain(mod, oracle, dt, n):
    """Inputs of at most one tile take ONE launch (every pass inside LDS).  Everything observable must equal
    what the multi-launch chain leaves behind: keys, payload (stable argsort), the last pass's table, and —
    for the reference-compatible host mirror — the reference-geometry diagnostics."""
    rng = np.random.default_rng(n)
    info = np.iinfo(dt)
    keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    keys[: n // 3] = keys[0]                                    # ties: the payload order matters
    perm = np.arange(n, dtype=np.uint32)
    seen = {}
    for tile_sort in (1, 0):
        with mod.Engine(dt, 4096, payload=True) as e:
            e.set_option(mod.OPT_TILE_SORT, tile_sort)
            e.upload(keys, perm)
            e.sort()
            k, p, table = e.download(want_perm=True, hist_cap=16)
            seen[tile_sort] = (k, p, table)
    assert np.array_equal(seen[1][0], np.sort(keys)) and np.array_equal(seen[1][1], np.argsort(keys, kind="stable").astype(np.uint32))
    for a, b in zip(seen[1], seen[0]):
        assert np.array_equal(a, b)
    if n % 1024 == 0:
        want_sorted, want_table, want_gs = oracle.emulate_reference_gpu(keys)
        with mod.Engine(dt, n) as e:
            e.set_option(mod.OPT_REF_DIAGNOSTICS, 1)
            e.upload(keys)
            e.sort()
            got, table, gs = e.download(hist_cap=16384, globsum_cap=512)
        assert np.array_equal(got, want_sorted) and np.array_equal(table, want_table) and np.array_equal(gs, want_gs)


@pytest.mark.parametrize("dt", ["uint32", "int64"])
def test_tile_sort_pass_ranges_and_external_buffers(mod, oracle, dt):
    """The one-launch path behind rsx_sort_from / rsx_sort_from_to: external input left untouched, partial and
    odd pass ranges, output into a caller buffer, and repeated internal sorts (ping-pong parity)."""
    import torch
    n = 3001
    keys = oracle.dataset("SeededUniform", dt, n, seed=11)
    signed = {"uint32": np.int32, "uint64": np.int64}.get(np.dtype(dt).name)
    t = torch.from_numpy(keys.view(signed) if signed else keys).cuda()
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    bits = keys.dtype.itemsize * 8
    u = keys.view(np.uint32 if bits == 32 else np.uint64)
    if keys.dtype.kind == "i":
        u = u ^ u.dtype.type(1 << (bits - 1))
    for tile_sort in (1, 0):
        with mod.Engine(dt, 4096, payload=True) as e:
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.set_option(mod.OPT_TILE_SORT, tile_sort)
            e.sort_from(t.data_ptr(), n, pay.data_ptr())
            k, p = e.download(want_perm=True)
            assert np.array_equal(k, np.sort(keys)) and np.array_equal(p, np.argsort(keys, kind="stable").astype(np.uint32))
            assert np.array_equal(t.cpu().numpy().view(keys.dtype), keys)
            for first, last in ((0, 3), (2, 3), (1, bits // 4), (0, bits // 4 - 1)):
                out = torch.zeros(n + 9, dtype=t.dtype, device="cuda")
                pout = torch.zeros(n + 9, dtype=torch.int32, device="cuda")
                e.set_option(mod.OPT_FIRST_PASS, 0)
                e.sort_from_to(t.data_ptr(), n, first, last, out[3:].data_ptr(), pay.data_ptr(), pout[3:].data_ptr())
                torch.cuda.synchronize()
                field = (u >> u.dtype.type(4 * first)) & u.dtype.type((1 << (4 * (last - first))) - 1)
                order = np.argsort(field, kind="stable")
                got = out.cpu().numpy().view(keys.dtype)
                assert np.array_equal(got[3:3 + n], keys[order]) and not got[:3].any() and not got[3 + n:].any(), (tile_sort, first, last)
                assert np.array_equal(pout.cpu().numpy().view(np.uint32)[3:3 + n], order.astype(np.uint32))
            # internal sorts: the result buffer alternates as in the chain, and sorting the result again is the identity
            e.sort_from(t.data_ptr(), n, pay.data_ptr())
            for _ in range(3):
                dk, dp = e.result_device()
                e.sort_from(dk, n, dp)
            k2, p2 = e.download(want_perm=True)
            assert np.array_equal(k2, k) and np.array_equal(p2, p)


# --------------------------------------------------------------------------- aliasing and stale state
def test_sort_from_accepts_the_engines_own_result_buffer(mod, oracle):
    """rsx_sort_from on the pointer rsx_result_device just returned (one of the engine's ping-pong
    buffers) must not reorder in place: it runs through the internal ping-pong.  Interior pointers
    and outputs overlapping the engine's buffers are refused."""
    import torch
    n = 70001
    keys = oracle.dataset("SeededUniform", "uint32", n)
    t = torch.from_numpy(keys.view(np.int32)).cuda()
    with mod.Engine("uint32", 2 * n, payload=True) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        pay = torch.arange(n, dtype=torch.int32, device="cuda")
        e.sort_from(t.data_ptr(), n, pay.data_ptr())
        first_k, first_p = e.download(want_perm=True)
        assert np.array_equal(first_k, np.sort(keys))
        # scramble the result in place (low 12 bits descending), then sort the engine's own buffer again
        dk, dp = e.result_device()
        assert dk and dp
        e.sort_from(dk, n, dp)
        k2, p2 = e.download(want_perm=True)
        assert np.array_equal(k2, first_k) and np.array_equal(p2, first_p)      # sorting sorted data: identical, stable
        for _ in range(3):                                                       # and again from whichever buffer holds it now
            dk, dp = e.result_device()
            e.sort_from(dk, n, dp)
        k3, p3 = e.download(want_perm=True)
        assert np.array_equal(k3, first_k) and np.array_equal(p3, first_p)
        dk, dp = e.result_device()
        with pytest.raises(mod.RadixSortError) as ei:
            e.sort_from(dk + 64, n, dp + 64)                                    # interior pointer
        assert ei.value.status == 1
        with pytest.raises(mod.RadixSortError):
            e.sort_from(dk, n, pay.data_ptr())                                   # keys inside, payload outside
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        with pytest.raises(mod.RadixSortError):
            e.sort_from_to(t.data_ptr(), n, 0, 7, dk, pay.data_ptr(), dp)        # output = engine buffer
        with pytest.raises(mod.RadixSortError):
            e.sort_from_to(t.data_ptr(), n, 0, 7, t.data_ptr(), pay.data_ptr(), out.data_ptr())   # output = input
        # after rsx_sort_from_to the engine holds no result
        pout = torch.empty_like(out)
        e.sort_from_to(t.data_ptr(), n, 0, 8, out.data_ptr(), pay.data_ptr(), pout.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), first_k)
        assert e.result_device() == (0, 0)
        with pytest.raises(mod.RadixSortError) as ei:
            e.download()
        assert ei.value.status == 5                                              # DATA_DOWNLOAD_FAILED
        with pytest.raises(mod.RadixSortError):
            e.copy_result(out.data_ptr(), pout.data_ptr())
        e.sort_from(t.data_ptr(), n, pay.data_ptr())                             # the next sort restores it
        assert np.array_equal(e.download(), first_k)


def test_count_then_foreign_sort_then_scatter_is_refused(mod, oracle):
    """rsx_partition_count leaves a raw table for exactly one input; any call that overwrites the table
    in between (a sort, a histogram, another partition) must void it, or the scatter would scan a
    foreign table and write out of bounds."""
    import torch
    n = 60000
    a = torch.from_numpy(oracle.dataset("SeededUniform", "int32", n)).cuda()
    b = torch.from_numpy(oracle.dataset("Random", "int32", n // 2)).cuda()
    out = torch.empty_like(a)
    with mod.Engine("int32", n) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        for clobber in ("sort_from", "histogram", "partition", "count_other"):
            e.partition_count(a.data_ptr(), n, 28, 4)
            if clobber == "sort_from":
                e.sort_from(b.data_ptr(), b.numel())
            elif clobber == "histogram":
                e.upload(b.cpu().numpy())
                e.histogram(0)
            elif clobber == "partition":
                e.partition(b.data_ptr(), b.numel(), 28, 4, out.data_ptr())
            else:
                e.partition_count(b.data_ptr(), b.numel(), 28, 4)
            with pytest.raises(mod.RadixSortError) as ei:
                e.partition_scatter(a.data_ptr(), n, 28, 4, out.data_ptr())
            assert ei.value.status == 4                                          # CALCULATION_FAILED
        splitters = [1 << 30, 3 << 30]
        e.partition_count_split(a.data_ptr(), n, splitters)
        e.sort_from(b.data_ptr(), b.numel())
        with pytest.raises(mod.RadixSortError):
            e.partition_scatter_split(a.data_ptr(), n, out.data_ptr())
        # rsx_msd_count borrows the 8-BIT tables: a 4-bit sort in between leaves them alone, an 8-bit one voids the count
        row = torch.zeros(259, dtype=torch.int64, device="cuda")
        e.msd_count(a.data_ptr(), n, 6, 4, row.data_ptr())
        e.sort_from(b.data_ptr(), b.numel())
        e.msd_scatter(a.data_ptr(), n, out.data_ptr())
        e.msd_count(a.data_ptr(), n, 6, 4, row.data_ptr())
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.sort_from(b.data_ptr(), b.numel())
        e.set_option(mod.OPT_RADIX_BITS, 4)
        with pytest.raises(mod.RadixSortError):
            e.msd_scatter(a.data_ptr(), n, out.data_ptr())
        # and the guarded sequence itself still works
        counts = e.partition_count(a.data_ptr(), n, 28, 4)
        e.partition_scatter(a.data_ptr(), n, 28, 4, out.data_ptr())
        torch.cuda.synchronize()
        assert sum(counts) == n


# --------------------------------------------------------------------------- extremes
def test_maximum_length_2pow32_minus_1024(mod):
    """Largest length the uint32_t API admits (Resize keeps n a multiple of 1024 below 2^32): every
    32-bit slot computation runs at its edge.  Keys are generated and checked on the device
    (torch is the checker's plumbing here); sortedness + sum/xor checksums, chunked."""
    import torch
    n = (1 << 32) - 1024
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs ~50 GiB of HBM")
    gen = torch.Generator(device="cuda").manual_seed(1234)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    chunk = 1 << 28
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        keys[lo:hi] = torch.randint(-(1 << 31), (1 << 31) - 1, (hi - lo,), generator=gen, dtype=torch.int64, device="cuda").to(torch.int32)
    out = torch.empty_like(keys)
    with mod.Engine("uint32", n) as e:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.sort_from(keys.data_ptr(), n)
        e.copy_result(out.data_ptr())
        torch.cuda.synchronize()
    sign = torch.tensor(-(1 << 31), dtype=torch.int32, device="cuda")
    s_in = s_out = 0
    prev_last = None
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        a, b = keys[lo:hi], out[lo:hi]
        s_in += int(a.to(torch.int64).sum()); s_out += int(b.to(torch.int64).sum())
        u = torch.bitwise_xor(b, sign)                      # unsigned order as signed order
        assert bool((u[:-1] <= u[1:]).all()), lo
        if prev_last is not None:
            assert prev_last <= int(u[0])
        prev_last = int(u[-1])
    assert s_in == s_out                                     # same multiset sum
    # exact multiset check on a sample of values: counts of a few probe keys agree
    for probe in (0, 1, -1, 123456789, -(1 << 31), (1 << 31) - 1):
        assert int((keys == probe).sum()) == int((out == probe).sum())


def test_engine_reuse_and_concurrent_engines(mod, oracle):
    """One engine sorts inputs of different lengths back to back; two engines on two streams run
    at the same time without sharing state (SURVEY §8b threading: handles share nothing)."""
    import torch
    with mod.Engine("uint32", 1 << 20) as e:
        for n in (1 << 20, 1000, 4097, 1 << 18, 1):
            keys = oracle.dataset("SeededUniform", "uint32", n, seed=n)
            e.upload(keys)
            e.sort()
            assert np.array_equal(e.download(), np.sort(keys)), n
    a = oracle.dataset("Random", "int64", 300000)
    b = oracle.dataset("SeededUniform", "uint32", 500001)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ta = torch.from_numpy(a).cuda()
    tb = torch.from_numpy(b.view(np.int32)).cuda()
    torch.cuda.synchronize()
    with mod.Engine("int64", a.size) as e1, mod.Engine("uint32", b.size) as e2:
        e1.set_stream(s1.cuda_stream)
        e2.set_stream(s2.cuda_stream)
        for _ in range(3):
            e1.sort_from(ta.data_ptr(), a.size)
            e2.sort_from(tb.data_ptr(), b.size)
        torch.cuda.synchronize()
        assert np.array_equal(e1.download(), np.sort(a))
        assert np.array_equal(e2.download(), np.sort(b))


def test_graph_replay_small_sorts(mod, oracle):
    """Sorts of <= 2^22 keys replay a captured hipGraph (RSX_OPT_GRAPH): same bytes as the eager
    path, across repeated calls, changing inputs, both entry points and a payload."""
    import torch
    n = 200003
    a = oracle.dataset("SeededUniform", "uint32", n, seed=1)
    b = oracle.dataset("SeededUniform", "uint32", n, seed=2)
    for graph in (1, 0):
        with mod.Engine("uint32", n, payload=True) as e:
            e.set_option(mod.OPT_GRAPH, graph)
            for keys in (a, b, a):
                e.upload(keys, np.arange(n, dtype=np.uint32))
                e.sort()
                ks, ps = e.download(want_perm=True)
                assert np.array_equal(ks, np.sort(keys)) and np.array_equal(ps, np.argsort(keys, kind="stable").astype(np.uint32))
            ta = torch.from_numpy(a.view(np.int32)).cuda()
            tp = torch.arange(n, dtype=torch.int32, device="cuda")
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(4):                      # alternates between the two internal buffers
                e.sort_from(ta.data_ptr(), n, tp.data_ptr())
            torch.cuda.synchronize()
            ks, ps = e.download(want_perm=True)
            assert np.array_equal(ks, np.sort(a)) and np.array_equal(ps, np.argsort(a, kind="stable").astype(np.uint32))
            ta.copy_(torch.from_numpy(b.view(np.int32)).cuda())     # same pointer, new contents: the graph must not cache data
            e.sort_from(ta.data_ptr(), n, tp.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(e.download(), np.sort(b))


def test_fused_scan_group_limit_comes_from_the_occupancy_query(mod, oracle):
    """The fused table scan's workgroups wait for each other inside one launch, so only tables the device holds at once may
    take it: the limit is derived from hipOccupancyMaxActiveBlocksPerMultiprocessor x CU count at rsx_create (half of it),
    RSX_OPT_FUSED_SCAN_MAX_GROUPS moves it (clamped to what is resident), and a table beyond it takes the two-launch scan
    — same keys, same table, same group sums; the time-out word stays clear."""
    n = (1 << 24) + 4097                    # 4098 tiles = 17 scan groups: beyond the self-scan, inside the default limit
    keys = oracle.dataset("SeededUniform", "uint32", n, seed=3)
    want = np.sort(keys)
    seen = []
    for limit in (-1, 4, 0, 1 << 20):
        with mod.Engine("uint32", n) as e:
            g = e.geometry()
            assert g.fused_scan_resident >= 256 and 0 < g.fused_scan_max_groups <= min(512, g.fused_scan_resident // 2)
            e.set_option(mod.OPT_FUSED_SCAN_MAX_GROUPS, limit)
            now = e.geometry().fused_scan_max_groups
            assert now == {-1: g.fused_scan_max_groups, 4: 4, 0: 0, 1 << 20: min(512, g.fused_scan_resident)}[limit]
            e.set_option(mod.OPT_PROFILE, 1)
            e.upload(keys)
            for _ in range(2):
                e.sort()
            e.sync()                                                       # reports a timed-out scan: must not
            rt = e.timings()
            # fused: one scan launch per pass; beyond the limit: scan #1 (timeScan) + scan #2/paste (timePaste)
            fused = 17 <= now
            assert (rt.paste.n == 0) == fused, (limit, now, rt.scan.n, rt.paste.n)
            g = e.geometry()
            seen.append(e.download(hist_cap=int(g.table_len), globsum_cap=int(g.num_scan_blocks)))
    for got in seen:
        assert np.array_equal(got[0], want)
        for a, b in zip(seen[0], got):
            assert np.array_equal(a, b)
    with mod.Engine("uint32", 4096) as e:
        with pytest.raises(mod.RadixSortError):
            e.set_option(mod.OPT_FUSED_SCAN_MAX_GROUPS, -2)


def test_two_engines_run_fused_scans_at_the_same_time(mod, oracle):
    """Two engines of >= 2^24 keys on two streams, both through the fused scan (each may use half of what the device holds
    at once): results exact, the time-out word clear on both."""
    import torch
    na, nb = (1 << 24) + 12345, (1 << 25) + 1
    a = oracle.dataset("SeededUniform", "uint32", na, seed=21)
    b = oracle.dataset("SeededUniform", "uint64", nb, seed=22)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ta = torch.from_numpy(a.view(np.int32)).cuda()
    tb = torch.from_numpy(b.view(np.int64)).cuda()
    torch.cuda.synchronize()
    with mod.Engine("uint32", na) as e1, mod.Engine("uint64", nb) as e2:
        assert e1.geometry().fused_scan_max_groups >= 17 and e2.geometry().fused_scan_max_groups >= 33
        e1.set_stream(s1.cuda_stream)
        e2.set_stream(s2.cuda_stream)
        for _ in range(4):
            e1.sort_from(ta.data_ptr(), na)
            e2.sort_from(tb.data_ptr(), nb)
        e1.sync()
        e2.sync()
        e1.check_status()
        e2.check_status()
        assert np.array_equal(e1.download(), np.sort(a))
        assert np.array_equal(e2.download(), np.sort(b))


def test_download_refuses_tables_the_last_sort_did_not_produce(mod, oracle):
    """rsx_download hands out the engine's own [digit][tile] table / group sums only when the last sort produced them: sorts on
    1024-key tiles and 8-bit passes leave none (an earlier sort's would be silently stale); the reference-geometry
    diagnostics work on every path."""
    n = 1 << 16
    keys = oracle.dataset("Random", "uint32", n)
    with mod.Engine("uint32", 1 << 21) as e:
        e.upload(keys)
        e.sort()                                                           # default: self-scan on tiles of 1024 keys
        assert np.array_equal(e.download(), np.sort(keys))
        with pytest.raises(mod.RadixSortError):
            e.download(hist_cap=16 * 16)
        e.set_option(mod.OPT_SMALL_TILE_MAX_KEYS, 0)                       # self-scan on 4096-key tiles: a table, no group sums
        e.upload(keys)
        e.sort()
        _, table = e.download(hist_cap=16 * 16)
        assert int(table[0]) == 0
        with pytest.raises(mod.RadixSortError):
            e.download(globsum_cap=16)
        big = oracle.dataset("SeededUniform", "uint32", 1 << 21, seed=8)
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(big)
        e.sort()
        assert np.array_equal(e.download(), np.sort(big))
        with pytest.raises(mod.RadixSortError):
            e.download(hist_cap=16)
        e.set_option(mod.OPT_RADIX_BITS, 4)
        e.set_option(mod.OPT_SELF_SCAN, 0)
        e.set_option(mod.OPT_SMALL_SCAN, 0)                                # (the one-workgroup scan of small tables leaves no group sums either)
        e.upload(big)
        e.sort()
        _, table, gs = e.download(hist_cap=16 * 512, globsum_cap=32)
        assert int(table[0]) == 0 and int(gs[0]) == 0
    want_sorted, want_table, want_gs = oracle.emulate_reference_gpu(keys)
    with mod.Engine("uint32", n) as e:
        e.set_option(mod.OPT_REF_DIAGNOSTICS, 1)
        e.upload(keys)
        e.sort()
        got, table, gs = e.download(hist_cap=16384, globsum_cap=512)
        assert np.array_equal(got, want_sorted) and np.array_equal(table, want_table) and np.array_equal(gs, want_gs)
        # the reference-geometry diagnostics describe a 4-BIT last pass: after an 8-bit sort they are refused, not recomputed from stale state
    with mod.Engine("uint32", 1 << 20) as e:                              # (8-bit passes run from 2^19 keys; below, the 4-bit self-scan chain serves either setting)
        wide = oracle.dataset("SeededUniform", "uint32", 1 << 20, seed=12)
        e.set_option(mod.OPT_REF_DIAGNOSTICS, 1)
        e.set_option(mod.OPT_RADIX_BITS, 8)
        e.upload(wide)
        e.sort()
        with pytest.raises(mod.RadixSortError) as err:
            e.download(hist_cap=16384, globsum_cap=512)
        assert "4-BIT pass" in str(err.value)
        assert np.array_equal(e.download(), np.sort(wide))                 # the keys themselves are fine
        e.set_option(mod.OPT_RADIX_BITS, 4)
        e.upload(wide)
        e.sort()
        _, table, gs = e.download(hist_cap=16384, globsum_cap=512)
        _, want_table2, want_gs2 = oracle.emulate_reference_gpu(wide)
        assert np.array_equal(table, want_table2) and np.array_equal(gs, want_gs2)


def test_graph_capture_with_8bit_digits_allocates_outside_the_capture(mod, oracle):
    """RSX_OPT_GRAPH with 8-bit digits: the 8-bit tables are allocated before the capture begins (an allocation inside
    hipStreamBeginCapture invalidates it), so the first call already captures and every call sorts."""
    n = 1 << 21
    keys = oracle.dataset("SeededUniform", "uint32", n, seed=6)
    want = np.sort(keys)
    for order in ("graph_first", "bits_first"):
        with mod.Engine("uint32", n) as e:
            if order == "graph_first":
                e.set_option(mod.OPT_GRAPH, 1)
                e.set_option(mod.OPT_RADIX_BITS, 8)
            else:
                e.set_option(mod.OPT_RADIX_BITS, 8)
                e.set_option(mod.OPT_GRAPH, 1)
            for _ in range(3):
                e.upload(keys)
                e.sort()
                assert np.array_equal(e.download(), want)


@pytest.mark.parametrize("dt,n", [("uint32", (1 << 21) + 4099), ("int32", 70001), ("uint64", (1 << 21) - 4097), ("int64", 3 * (1 << 19) + 77)])
def test_8bit_scatter_on_512_thread_workgroups_is_exact(mod, oracle, dt, n):
    """RSX_R8_WIDE (read at rsx_create; -1 = the policy: 512 threads for 64-bit keys without payload): kernel 1 of the 8-bit passes on workgroups of
    512 threads x 8 keys — the same 4096-key tiles and tables, rows of 8 slots in the LDS image — and on 256 x 16 give the same keys and the same
    stable payload order; keys only, payload (packed for 32-bit keys), payload kept apart, ragged last tile, ties, constant data."""
    import os
    keys = oracle.dataset("SeededUniform", dt, n, seed=n % 19)
    keys[::4] = keys[5]
    for data in (keys, np.full(n, keys[9], dtype=dt)):
        want_k = np.sort(data)
        want_p = np.argsort(data, kind="stable").astype(np.uint32)
        for payload, packed, wide in [(False, "1", "1"), (False, "1", "0"), (True, "1", "1")] + ([(True, "0", "1")] if np.dtype(dt).itemsize == 4 else []):
            os.environ["RSX_R8_PACKED"] = packed
            os.environ["RSX_R8_WIDE"] = wide
            try:
                e = mod.Engine(dt, n, payload=payload)
            finally:
                del os.environ["RSX_R8_PACKED"], os.environ["RSX_R8_WIDE"]
            with e:
                e.set_option(mod.OPT_RADIX_BITS, 8)
                if payload:
                    e.upload(data, np.arange(n, dtype=np.uint32))
                    e.sort()
                    ks, ps = e.download(want_perm=True)
                    assert np.array_equal(ps, want_p)
                else:
                    e.upload(data)
                    e.sort()
                    ks = e.download()
                assert np.array_equal(ks, want_k)


@pytest.mark.parametrize("dt", ["uint64", "int64"])
@pytest.mark.parametrize("n", [(1 << 23) + 4099, 5 * 4096 * 300 + 1])
def test_4bit_reorder_of_64bit_keys_with_payload_on_either_workgroup_shape(mod, oracle, dt, n):
    """RSX_REORDER_WIDE (read at rsx_create; -1 = policy: on): the 4-bit reorder of 64-bit keys with payload on 512 threads x 8 keys and on
    256 x 16 — same tiles and tables — gives the same keys, the same stable payload order and the same final table."""
    import os
    keys = oracle.dataset("SeededUniform", dt, n, seed=n % 23)
    keys[::6] = keys[1]
    want_p = np.argsort(keys, kind="stable").astype(np.uint32)
    tables = []
    for wide in ("1", "0"):
        os.environ["RSX_REORDER_WIDE"] = wide
        try:
            e = mod.Engine(dt, n, payload=True)
        finally:
            del os.environ["RSX_REORDER_WIDE"]
        with e:
            e.upload(keys, np.arange(n, dtype=np.uint32))
            e.sort()
            g = e.geometry()
            ks, ps, hist = e.download(want_perm=True, hist_cap=int(g.table_len))
        assert np.array_equal(ks, keys[want_p]) and np.array_equal(ps, want_p)
        tables.append(hist)
    assert np.array_equal(tables[0], tables[1])
