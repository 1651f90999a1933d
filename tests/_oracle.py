"""ctypes access to the CPU checker in oracle/ (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
`Oracle` wraps this repo's restatement (oracle/liboracle.so); `RefOracle` wraps the
reference's own headers compiled into oracle/_ref/libref_oracle.so (present only
when built in the container that holds /root/reference; it travels to the GPU box
as a prebuilt .so).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

DTYPES = {"uint32": 0, "int32": 1, "uint64": 2, "int64": 3}
KINDS = {"Zeros": 0, "Range": 1, "InvertedRange": 2, "Random": 3, "SeededUniform": 4}
DEFAULT_SEED = 0x5EEDCAFEF00D


def _code(dtype) -> int:
    return DTYPES[np.dtype(dtype).name]


def build_oracle(ref: bool = True) -> None:
    """Compile oracle/liboracle.so (and oracle/_ref when the reference tree exists)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)


class Oracle:
    def __init__(self) -> None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle(ref=False)
        self.lib = C.CDLL(path)
        L = self.lib
        L.oracle_radix_sort.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_radix_sort.restype = C.c_int
        L.oracle_round_count.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
        L.oracle_round_count.restype = C.c_uint64
        L.oracle_std_sort.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
        L.oracle_std_sort.restype = C.c_int
        L.oracle_stable_argsort.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_stable_argsort.restype = C.c_int
        L.oracle_dataset.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
        L.oracle_dataset.restype = C.c_int
        L.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_fnv1a64.restype = C.c_uint64
        L.oracle_emulate_reference_gpu.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_emulate_reference_gpu.restype = C.c_int
        for name in ("oracle_time_radix_sort", "oracle_time_std_sort"):
            f = getattr(L, name)
            f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
            f.restype = C.c_int

    # -- sorting -----------------------------------------------------------
    def radix_sort(self, keys: np.ndarray, payload: np.ndarray | None = None):
        """RadixSortCPU restated; returns sorted copies (keys[, payload])."""
        k = np.ascontiguousarray(keys).copy()
        p = None
        if payload is not None:
            p = np.ascontiguousarray(payload, dtype=np.uint32).copy()
        rc = self.lib.oracle_radix_sort(_code(k.dtype), k.ctypes.data, p.ctypes.data if p is not None else None, k.size)
        assert rc == 0
        return k if p is None else (k, p)

    def round_count(self, keys: np.ndarray) -> int:
        k = np.ascontiguousarray(keys)
        return int(self.lib.oracle_round_count(_code(k.dtype), k.ctypes.data, k.size))

    def std_sort(self, keys: np.ndarray) -> np.ndarray:
        k = np.ascontiguousarray(keys).copy()
        assert self.lib.oracle_std_sort(_code(k.dtype), k.ctypes.data, k.size) == 0
        return k

    def stable_argsort(self, keys: np.ndarray, payload: np.ndarray) -> np.ndarray:
        k = np.ascontiguousarray(keys)
        p = np.ascontiguousarray(payload, dtype=np.uint32).copy()
        assert self.lib.oracle_stable_argsort(_code(k.dtype), k.ctypes.data, p.ctypes.data, k.size) == 0
        return p

    # -- inputs ------------------------------------------------------------
    def dataset(self, kind: str, dtype, n: int, seed: int = DEFAULT_SEED) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        assert self.lib.oracle_dataset(KINDS[kind], _code(dtype), out.ctypes.data, n, seed) == 0
        return out

    def digest(self, arr: np.ndarray) -> str:
        a = np.ascontiguousarray(arr)
        return f"{self.lib.oracle_fnv1a64(a.ctypes.data, a.nbytes):016x}"

    def emulate_reference_gpu(self, keys: np.ndarray):
        k = np.ascontiguousarray(keys).copy()
        table = np.zeros(16 * 1024, dtype=np.uint32)
        globsum = np.zeros(512, dtype=np.uint32)
        rc = self.lib.oracle_emulate_reference_gpu(_code(k.dtype), k.ctypes.data, k.size, table.ctypes.data, globsum.ctypes.data)
        assert rc == 0, rc
        return k, table, globsum

    # -- baseline timing -----------------------------------------------------
    def time_radix_sort(self, keys: np.ndarray, iters: int = 1, return_sorted: bool = False):
        """Milliseconds per (copy-in + sort); with return_sorted also the array it sorted."""
        k = np.ascontiguousarray(keys)
        scratch = np.empty_like(k)
        ms = C.c_double(0.0)
        assert self.lib.oracle_time_radix_sort(_code(k.dtype), k.ctypes.data, scratch.ctypes.data, k.size, iters, C.byref(ms)) == 0
        return (ms.value, scratch) if return_sorted else ms.value

    def time_std_sort(self, keys: np.ndarray, iters: int = 1) -> float:
        k = np.ascontiguousarray(keys)
        scratch = np.empty_like(k)
        ms = C.c_double(0.0)
        assert self.lib.oracle_time_std_sort(_code(k.dtype), k.ctypes.data, scratch.ctypes.data, k.size, iters, C.byref(ms)) == 0
        return ms.value


class RefOracle:
    """The reference's own RadixSortCPU / Dataset headers (oracle/_ref)."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libref_oracle.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self) -> None:
        self.lib = C.CDLL(self.PATH)
        L = self.lib
        L.ref_radix_sort.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
        L.ref_radix_sort.restype = C.c_int
        L.ref_dataset.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        L.ref_dataset.restype = C.c_int
        L.ref_time_radix_sort.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        L.ref_time_radix_sort.restype = C.c_int

    def radix_sort(self, keys: np.ndarray) -> np.ndarray:
        k = np.ascontiguousarray(keys).copy()
        assert self.lib.ref_radix_sort(_code(k.dtype), k.ctypes.data, k.size) == 0
        return k

    def dataset(self, kind: str, dtype, n: int) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        assert self.lib.ref_dataset(KINDS[kind], _code(dtype), out.ctypes.data, n) == 0
        return out

    def time_radix_sort(self, keys: np.ndarray, iters: int = 1, return_sorted: bool = False):
        k = np.ascontiguousarray(keys)
        scratch = np.empty_like(k)
        ms = C.c_double(0.0)
        assert self.lib.ref_time_radix_sort(_code(k.dtype), k.ctypes.data, scratch.ctypes.data, k.size, iters, C.byref(ms)) == 0
        return (ms.value, scratch) if return_sorted else ms.value
