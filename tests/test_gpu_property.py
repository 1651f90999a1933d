"""Property-based parity (hypothesis): random lengths, key types, value distributions with heavy
ties / narrow ranges / extremes, optional payload, both sort modes — every case bit-exact against
numpy's stable sort (= std::sort order for keys, stable argsort for payloads)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

DT = ["uint32", "int32", "uint64", "int64"]


@st.composite
def sort_case(draw):
    dt = draw(st.sampled_from(DT))
    n = draw(st.one_of(st.integers(0, 70), st.integers(71, 5000), st.integers(4000, 20000), st.sampled_from([4095, 4096, 4097, 8192, 12289])))
    info = np.iinfo(dt)
    shape = draw(st.sampled_from(["full", "narrow", "ties", "sorted", "reverse", "constant", "extremes", "low_bits", "high_bits"]))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    if shape == "full":
        keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    elif shape == "narrow":
        span = int(info.max) - int(info.min) - 2000
        lo = int(info.min) + (int(rng.integers(0, 1 << 20)) * span >> 20)          # python ints: no overflow
        keys = np.array([lo + int(v) for v in rng.integers(0, 1000, size=n)], dtype=dt)
    elif shape == "ties":
        keys = rng.integers(0, 7, size=n).astype(dt)
    elif shape == "sorted":
        keys = np.sort(rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True))
    elif shape == "reverse":
        keys = np.sort(rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True))[::-1].copy()
    elif shape == "constant":
        keys = np.full(n, int(rng.integers(info.min, info.max, dtype=dt, endpoint=True)), dtype=dt)
    elif shape == "extremes":
        keys = rng.choice(np.array([info.min, info.min + 1, 0, 1, info.max - 1, info.max], dtype=dt), size=n)
    elif shape == "low_bits":
        keys = rng.integers(0, 16, size=n).astype(dt)                       # only the first pass does anything
    else:                                                                    # only the last pass does anything
        if dt.startswith("u"):
            keys = (rng.integers(0, 16, size=n).astype(np.uint64) << np.uint64(info.bits - 4)).astype(dt)
        else:
            keys = (rng.integers(-8, 8, size=n).astype(np.int64) << (info.bits - 4)).astype(dt)
    return dt, keys, draw(st.booleans()), draw(st.booleans())


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case=sort_case())
def test_sort_property(rsx, case):
    dt, keys, payload, lookahead = case
    n = keys.size
    perm = np.arange(n, dtype=np.uint32) if payload else None
    with rsx.Engine(dt, max(n, 1), payload=payload) as e:
        e.set_option(rsx.OPT_LOOKAHEAD, int(lookahead))
        e.upload(keys, perm)
        e.sort()
        out = e.download(want_perm=True) if payload else (e.download(), None)
    assert np.array_equal(out[0], np.sort(keys, kind="stable"))
    if payload:
        assert np.array_equal(out[1], np.argsort(keys, kind="stable").astype(np.uint32))
