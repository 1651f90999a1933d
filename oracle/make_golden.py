#!/usr/bin/env python3
"""Generate tests/golden/oracle_golden.json from the REFERENCE's own headers.

TEST INFRASTRUCTURE.  Run in the build container (needs /root/reference):

    make -C oracle ref && python oracle/make_golden.py

Every expected value below is produced by oracle/_ref/libref_oracle.so, i.e. by
/root/reference/src/CRadixSortCPU.h and Dataset.h compiled as they lie — never by
this repo's restatement.  The file holds data only (inputs, outputs, digests).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import Oracle, RefOracle, build_oracle  # noqa: E402

DT = ["uint32", "int32", "uint64", "int64"]
KINDS = ["Zeros", "Range", "InvertedRange", "Random"]
SIZES = [1000, 1024, 65536]

# Inputs on which the reference's round count is short (SURVEY §8c "oracle's correct
# domain"): exact powers of the base, max == 1, signed data with a small raw maximum.
KNOWN_ANSWER_INPUTS = [
    ("uint32", [1, 0]),
    ("uint32", [8, 1, 0, 7]),
    ("uint32", [64, 9, 63, 1]),
    ("uint32", [512, 9, 511, 65, 1]),
    ("uint64", [16, 1, 0, 15]),
    ("uint64", [256, 17, 1, 0, 255]),
    ("int32", [5, -3, 100, -100, 0]),
    ("int64", [5, -3, 100, -100, 0]),
    ("uint32", [3, 2, 1, 0, 7, 6, 5, 4]),
    ("int32", [-1, -2, -3, 2147483647, -2147483648, 0]),
    ("uint64", [18446744073709551615, 0, 9223372036854775808, 4294967296, 4294967295]),
    ("int64", [9223372036854775807, -9223372036854775808, -1, 0, 1]),
]


def main() -> None:
    build_oracle(ref=True)
    if not RefOracle.available():
        raise SystemExit("oracle/_ref/libref_oracle.so missing: needs /root/reference")
    ref = RefOracle()
    dig = Oracle().digest   # FNV-1a-64 is plain arithmetic; shared helper

    out = {
        "_generator": "oracle/make_golden.py via oracle/_ref (reference headers CRadixSortCPU.h, Dataset.h)",
        "digest": "FNV-1a-64 over little-endian bytes",
        "datasets": [],
        "random_first8": {},
        "known_answers": [],
        "small_vectors": [],
    }
    for dt in DT:
        out["random_first8"][dt] = [int(v) for v in ref.dataset("Random", dt, 8)]
        for kind in KINDS:
            for n in SIZES:
                src = ref.dataset(kind, dt, n)
                srt = ref.radix_sort(src)
                assert np.array_equal(srt, np.sort(src)), (dt, kind, n)
                out["datasets"].append({
                    "dtype": dt, "kind": kind, "n": n,
                    "input_digest": dig(src), "sorted_digest": dig(srt),
                    "first": int(srt[0]), "mid": int(srt[n // 2]), "last": int(srt[-1]),
                })
        src = ref.dataset("Random", dt, 64)
        out["small_vectors"].append({
            "dtype": dt, "kind": "Random", "n": 64,
            "input": [int(v) for v in src], "sorted": [int(v) for v in ref.radix_sort(src)],
        })
    for dt, vals in KNOWN_ANSWER_INPUTS:
        src = np.array(vals, dtype=dt)
        out["known_answers"].append({"dtype": dt, "input": vals, "output": [int(v) for v in ref.radix_sort(src)]})

    path = os.path.join(ROOT, "tests", "golden", "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
