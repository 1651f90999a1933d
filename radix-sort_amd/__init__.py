"""radix-sort_amd — MI355X-native LSD radix sort behind the reference's RadixSortGPU API.

This module is the Python-side binding of the C ABI in include/radixsort_hip.h
(ctypes over radix-sort_amd/libradixsort_hip.so).  It carries no sort logic: every
method is one C-ABI call.  The C++20 host mirror of the reference interface
(RadixSortGPU<T>, CRadixSortTask<T>, Dataset<T>, ...) lives in radix-sort_amd/host/.

There is no CPU fallback: if the HIP library is missing, or no GPU is present when an
engine is created, this raises.

The directory name contains a hyphen, so load it with `importlib` (see
__graft_entry__.load_package()) under the module name `radix_sort_amd`.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSX_LIB") or os.path.join(_HERE, "libradixsort_hip.so")   # RSX_LIB: A/B builds while tuning

# OperationStatus (reference src/OperationStatus.h:4-17)
STATUS_NAMES = [
    "OK", "HOST_BUFFERS_FAILED", "INITIALIZATION_FAILED", "DATA_UPLOAD_FAILED", "CALCULATION_FAILED",
    "DATA_DOWNLOAD_FAILED", "CLEANUP_FAILED", "RESIZE_FAILED", "KERNEL_CREATION_FAILED",
    "PROGRAM_CREATION_FAILED", "NO_SOURCE_FOUND", "LOADING_SOURCE_FAILED",
]

# rsx_option (include/radixsort_hip.h)
OPT_PROFILE, OPT_XCD_REMAP, OPT_FIRST_PASS, OPT_LAST_PASS, OPT_LOOKAHEAD, OPT_REF_DIAGNOSTICS, OPT_GRAPH, OPT_SMALL_SCAN, OPT_TILE_SORT, OPT_FUSED_SCAN = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_RADIX_BITS, OPT_SELF_SCAN, OPT_SMALL_TILE_MAX_KEYS, OPT_XCD_PHASE, OPT_SELF_SCAN_MAX_TILES, OPT_FUSED_SCAN_MAX_GROUPS = 10, 11, 12, 13, 14, 15
# rsx_experimental_option (include/radixsort_hip_experiments.h): known to the EXPERIMENTS build only (experiments()); the product library refuses them
XOPT_DEBUG_RAISE_SCAN_TIMEOUT, XOPT_INLINE_SCAN, XOPT_INLINE_SCAN_MAX_GROUPS, XOPT_REORDER8_KERNEL, XOPT_REORDER8_STAY = 16, 17, 18, 19, 20
EXPERIMENTS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_variants", "libradixsort_hip_experiments.so")

# every symbol include/radixsort_hip.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "rsx_device_count", "rsx_device_name", "rsx_last_error", "rsx_version",
    "rsx_create", "rsx_destroy", "rsx_set_stream", "rsx_get_stream", "rsx_set_option", "rsx_get_geometry", "rsx_resize",
    "rsx_upload", "rsx_fill_pad", "rsx_download", "rsx_pin_host", "rsx_unpin_host", "rsx_pipeline_submit", "rsx_pipeline_wait", "rsx_host_device_pointer",
    "rsx_histogram", "rsx_scan", "rsx_paste", "rsx_reorder", "rsx_sort", "rsx_sync", "rsx_check_status",
    "rsx_sort_from", "rsx_partition", "rsx_partition_count", "rsx_partition_scatter", "rsx_sample_keys", "rsx_partition_count_split", "rsx_partition_scatter_split", "rsx_peer_alloc", "rsx_peer_free", "rsx_peer_open", "rsx_peer_close", "rsx_peer_enable", "rsx_sort_from_to", "rsx_msd_count", "rsx_msd_scatter", "rsx_msd_plan", "rsx_msd_plan_wait", "rsx_msd_push", "rsx_copy_to_device", "rsx_copy_from_device", "rsx_copy_on_device", "rsx_wait_for", "rsx_record_mark", "rsx_wait_mark", "rsx_key_range", "rsx_partition_range", "rsx_result_device", "rsx_copy_result", "rsx_tile_map", "rsx_timings",
]


class PhaseStat(C.Structure):
    _fields_ = [("min_ms", C.c_double), ("max_ms", C.c_double), ("avg_ms", C.c_double), ("sum_ms", C.c_double), ("n", C.c_uint64)]


class Runtimes(C.Structure):   # RuntimesGPU (reference src/RadixSortGPU.h:18-24)
    _fields_ = [("histogram", PhaseStat), ("scan", PhaseStat), ("paste", PhaseStat), ("reorder", PhaseStat), ("total", PhaseStat)]


class Geometry(C.Structure):
    _fields_ = [
        ("tile_threads", C.c_uint32), ("keys_per_thread", C.c_uint32), ("tile_keys", C.c_uint32), ("scan_block", C.c_uint32),
        ("num_keys", C.c_uint64), ("capacity", C.c_uint64), ("num_tiles", C.c_uint64), ("table_len", C.c_uint64),
        ("num_scan_blocks", C.c_uint64), ("num_passes", C.c_uint32), ("key_bytes", C.c_uint32),
        ("fused_scan_resident", C.c_uint32), ("fused_scan_max_groups", C.c_uint32),
    ]


class RadixSortError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        name = STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else str(status)
        super().__init__(f"{where}: OperationStatus::{name}" + (f" ({detail})" if detail else ""))


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so
    (soname libamdhip64.so.7, but its users ask for it as `libamdhip64.so`); if this
    library pulled in /opt/rocm's copy first, a later `import torch` would map a second
    runtime and find no GPU.  When PyTorch is installed, map its copy first so both sides
    bind the same one.  RSX_NO_TORCH_RUNTIME=1 opts out (pure C/C++ hosts never get here)."""
    if os.environ.get("RSX_NO_TORCH_RUNTIME") or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> C.CDLL:
    """dlopen the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    P, U64, I = C.c_void_p, C.c_uint64, C.c_int
    sig = {
        "rsx_device_count": ([C.POINTER(I)], I),
        "rsx_device_name": ([I, C.c_char_p, C.c_size_t], I),
        "rsx_last_error": ([], C.c_char_p),
        "rsx_version": ([], C.c_char_p),
        "rsx_create": ([C.POINTER(P), I, I, I, I, U64], I),
        "rsx_destroy": ([P], I),
        "rsx_set_stream": ([P, P], I),
        "rsx_get_stream": ([P, C.POINTER(P)], I),
        "rsx_set_option": ([P, I, C.c_int64], I),
        "rsx_get_geometry": ([P, C.POINTER(Geometry)], I),
        "rsx_resize": ([P, U64], I),
        "rsx_upload": ([P, P, P, U64], I),
        "rsx_fill_pad": ([P, U64], I),
        "rsx_download": ([P, P, P, P, U64, P, U64], I),
        "rsx_pin_host": ([P, P, U64], I),
        "rsx_unpin_host": ([P, P], I),
        "rsx_pipeline_submit": ([P, P, P, U64, P, P], I),
        "rsx_pipeline_wait": ([P], I),
        "rsx_host_device_pointer": ([P, P, C.POINTER(P)], I),
        "rsx_histogram": ([P, I], I),
        "rsx_scan": ([P], I),
        "rsx_paste": ([P], I),
        "rsx_reorder": ([P, I], I),
        "rsx_sort": ([P], I),
        "rsx_sync": ([P], I),
        "rsx_check_status": ([P], I),
        "rsx_sort_from": ([P, P, P, U64], I),
        "rsx_partition": ([P, P, P, U64, I, I, P, P, C.POINTER(U64)], I),
        "rsx_partition_count": ([P, P, U64, I, I, C.POINTER(U64)], I),
        "rsx_partition_scatter": ([P, P, P, U64, I, I, P, P], I),
        "rsx_sample_keys": ([P, P, U64, C.c_uint32, C.POINTER(U64)], I),
        "rsx_partition_count_split": ([P, P, U64, C.POINTER(U64), I, C.POINTER(U64)], I),
        "rsx_partition_scatter_split": ([P, P, P, U64, P, P], I),
        "rsx_peer_alloc": ([P, U64, C.POINTER(P), P], I),
        "rsx_peer_free": ([P, P], I),
        "rsx_peer_open": ([P, P, C.POINTER(P)], I),
        "rsx_peer_close": ([P, P], I),
        "rsx_peer_enable": ([P, I], I),
        "rsx_sort_from_to": ([P, P, P, U64, I, I, P, P], I),
        "rsx_msd_count": ([P, P, U64, I, I, P], I),
        "rsx_msd_scatter": ([P, P, P, U64, P, P], I),
        "rsx_msd_plan": ([P, P, C.c_uint32, C.c_uint32, I, I, P], I),
        "rsx_msd_plan_wait": ([P, C.POINTER(U64), C.POINTER(U64), C.POINTER(U64), C.POINTER(U64)], I),
        "rsx_msd_push": ([P, I, P, P, P, P, I, P], I),
        "rsx_copy_to_device": ([P, P, P, U64], I),
        "rsx_copy_from_device": ([P, P, P, U64], I),
        "rsx_copy_on_device": ([P, P, P, U64], I),
        "rsx_wait_for": ([P, P], I),
        "rsx_record_mark": ([P, I], I),
        "rsx_wait_mark": ([P, P, I], I),
        "rsx_key_range": ([P, P, U64, C.POINTER(U64), C.POINTER(U64)], I),
        "rsx_partition_range": ([P, P, P, U64, U64, I, U64, P, P, C.POINTER(U64)], I),
        "rsx_result_device": ([P, C.POINTER(P), C.POINTER(P)], I),
        "rsx_copy_result": ([P, P, P], I),
        "rsx_timings": ([P, C.POINTER(Runtimes), I], I),
        "rsx_tile_map": ([C.c_uint64, C.c_uint32, I, C.c_int64, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)], I),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def variant(lib_path: str):
    """A second, independent instance of this binding over ANOTHER build of the library (A/B builds, the experiments build):
    its own module object, its own dlopen; engines of the two never mix."""
    if not os.path.exists(lib_path):
        raise FileNotFoundError(f"{lib_path} is missing (tools/build_variant.sh, or __graft_entry__.build())")
    name = __name__ + "_variant_" + os.path.splitext(os.path.basename(lib_path))[0]
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, "__init__.py"), submodule_search_locations=[_HERE])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.LIB_PATH = lib_path
    return mod


def experiments():
    """The binding over the experiments build (-DRSX_EXPERIMENTS: rejected kernel variants, inline table scan, test hooks)."""
    return variant(EXPERIMENTS_LIB_PATH)


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.rsx_device_count(C.byref(n))
    if rc != 0:
        raise RadixSortError(rc, "rsx_device_count", lib.rsx_last_error().decode())
    return n.value


def device_name(device: int = 0) -> str:
    lib = load_library()
    buf = C.create_string_buffer(256)
    rc = lib.rsx_device_name(device, buf, 256)
    if rc != 0:
        raise RadixSortError(rc, "rsx_device_name", lib.rsx_last_error().decode())
    return buf.value.decode()


_KEY_DTYPES = {"uint32": (4, 0), "int32": (4, 1), "uint64": (8, 0), "int64": (8, 1)}


class Engine:
    """One device + one stream + one buffer set: the C-ABI `rsx_engine`."""

    def __init__(self, dtype, capacity: int, payload: bool = False, device: int = 0):
        self.lib = load_library()
        self.dtype = np.dtype(dtype)
        if self.dtype.name not in _KEY_DTYPES:
            raise TypeError(f"unsupported key type {self.dtype}")
        kb, sg = _KEY_DTYPES[self.dtype.name]
        self.payload = bool(payload)
        self.capacity = int(capacity)
        self.device = int(device)
        self._h = C.c_void_p()
        rc = self.lib.rsx_create(C.byref(self._h), device, kb, sg, int(self.payload), self.capacity)
        if rc != 0:
            raise RadixSortError(rc, "rsx_create", self.lib.rsx_last_error().decode())

    # -- plumbing ----------------------------------------------------------
    def _check(self, rc: int, where: str) -> None:
        if rc != 0:
            raise RadixSortError(rc, where, self.lib.rsx_last_error().decode())

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rsx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: int) -> None:
        self._check(self.lib.rsx_set_stream(self._h, C.c_void_p(hip_stream)), "rsx_set_stream")

    def get_stream(self) -> int:
        s = C.c_void_p()
        self._check(self.lib.rsx_get_stream(self._h, C.byref(s)), "rsx_get_stream")
        return int(s.value or 0)

    def set_option(self, option: int, value: int) -> None:
        self._check(self.lib.rsx_set_option(self._h, option, value), "rsx_set_option")

    def geometry(self) -> Geometry:
        g = Geometry()
        self._check(self.lib.rsx_get_geometry(self._h, C.byref(g)), "rsx_get_geometry")
        return g

    # -- reference-shaped steps ---------------------------------------------
    def resize(self, n: int) -> None:
        self._check(self.lib.rsx_resize(self._h, n), "rsx_resize")

    def upload(self, keys: np.ndarray, perm: np.ndarray | None = None) -> None:
        k = np.ascontiguousarray(keys, dtype=self.dtype)
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint32)
        self._check(self.lib.rsx_upload(self._h, k.ctypes.data, p.ctypes.data if p is not None else None, k.size), "rsx_upload")

    def pin_host(self, arr: np.ndarray) -> None:
        self._check(self.lib.rsx_pin_host(self._h, arr.ctypes.data, arr.nbytes), "rsx_pin_host")

    def unpin_host(self, arr: np.ndarray) -> None:
        self._check(self.lib.rsx_unpin_host(self._h, arr.ctypes.data), "rsx_unpin_host")

    def pipeline_submit(self, keys: np.ndarray, out: np.ndarray, perm: np.ndarray | None = None, perm_out: np.ndarray | None = None) -> None:
        """Asynchronous upload -> sort -> download of one job; `keys`/`out` (pinned) must stay alive and untouched until pipeline_wait()."""
        assert keys.dtype == self.dtype and out.dtype == self.dtype and out.size >= keys.size and keys.flags.c_contiguous and out.flags.c_contiguous
        self._check(self.lib.rsx_pipeline_submit(self._h, keys.ctypes.data, perm.ctypes.data if perm is not None else None, keys.size,
                                                 out.ctypes.data, perm_out.ctypes.data if perm_out is not None else None), "rsx_pipeline_submit")

    def pipeline_wait(self) -> None:
        self._check(self.lib.rsx_pipeline_wait(self._h), "rsx_pipeline_wait")

    def host_device_pointer(self, arr: np.ndarray) -> int:
        p = C.c_void_p()
        self._check(self.lib.rsx_host_device_pointer(self._h, arr.ctypes.data, C.byref(p)), "rsx_host_device_pointer")
        return int(p.value or 0)

    def fill_pad(self, byte_offset: int) -> None:
        self._check(self.lib.rsx_fill_pad(self._h, byte_offset), "rsx_fill_pad")

    def histogram(self, pass_: int) -> None:
        self._check(self.lib.rsx_histogram(self._h, pass_), "rsx_histogram")

    def scan(self) -> None:
        self._check(self.lib.rsx_scan(self._h), "rsx_scan")

    def paste(self) -> None:
        self._check(self.lib.rsx_paste(self._h), "rsx_paste")

    def reorder(self, pass_: int) -> None:
        self._check(self.lib.rsx_reorder(self._h, pass_), "rsx_reorder")

    def sort(self) -> None:
        self._check(self.lib.rsx_sort(self._h), "rsx_sort")

    def sync(self) -> None:
        self._check(self.lib.rsx_sync(self._h), "rsx_sync")

    def check_status(self) -> None:
        """Raises if a fused table scan of a sort that has already finished timed out (no synchronisation; reported once)."""
        self._check(self.lib.rsx_check_status(self._h), "rsx_check_status")

    def download(self, want_perm: bool = False, hist_cap: int = 0, globsum_cap: int = 0):
        n = self.geometry().num_keys
        keys = np.empty(n, dtype=self.dtype)
        perm = np.empty(n, dtype=np.uint32) if (want_perm and self.payload) else None
        hist = np.zeros(hist_cap, dtype=np.uint32) if hist_cap else None
        gs = np.zeros(globsum_cap, dtype=np.uint32) if globsum_cap else None
        self._check(self.lib.rsx_download(
            self._h, keys.ctypes.data, perm.ctypes.data if perm is not None else None,
            hist.ctypes.data if hist is not None else None, hist_cap,
            gs.ctypes.data if gs is not None else None, globsum_cap), "rsx_download")
        out = [keys]
        if want_perm:
            out.append(perm)
        if hist_cap:
            out.append(hist)
        if globsum_cap:
            out.append(gs)
        return out[0] if len(out) == 1 else tuple(out)

    # -- device-resident callers ----------------------------------------------
    def sort_from(self, d_keys: int, n: int, d_payload: int | None = None) -> None:
        self._check(self.lib.rsx_sort_from(self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n), "rsx_sort_from")

    def partition(self, d_keys: int, n: int, shift: int, bits: int, d_keys_out: int,
                  d_payload: int | None = None, d_payload_out: int | None = None) -> list[int]:
        offs = (C.c_uint64 * ((1 << bits) + 1))()
        self._check(self.lib.rsx_partition(
            self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n, shift, bits,
            C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None, offs), "rsx_partition")
        return [int(v) for v in offs]

    def partition_count(self, d_keys: int, n: int, shift: int, bits: int) -> list[int]:
        counts = (C.c_uint64 * (1 << bits))()
        self._check(self.lib.rsx_partition_count(self._h, C.c_void_p(d_keys), n, shift, bits, counts), "rsx_partition_count")
        return [int(v) for v in counts]

    def partition_scatter(self, d_keys: int, n: int, shift: int, bits: int, d_keys_out: int,
                          d_payload: int | None = None, d_payload_out: int | None = None) -> None:
        self._check(self.lib.rsx_partition_scatter(
            self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n, shift, bits,
            C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None), "rsx_partition_scatter")

    def sample_keys(self, d_keys: int, n: int, count: int) -> list[int]:
        out = (C.c_uint64 * count)()
        self._check(self.lib.rsx_sample_keys(self._h, C.c_void_p(d_keys), n, count, out), "rsx_sample_keys")
        return [int(v) for v in out]

    def partition_count_split(self, d_keys: int, n: int, splitters: list[int]) -> list[int]:
        m = len(splitters)
        arr = (C.c_uint64 * max(m, 1))(*splitters)
        counts = (C.c_uint64 * (2 * m + 1))()
        self._check(self.lib.rsx_partition_count_split(self._h, C.c_void_p(d_keys), n, arr, m, counts), "rsx_partition_count_split")
        return [int(v) for v in counts]

    def partition_scatter_split(self, d_keys: int, n: int, d_keys_out: int, d_payload: int | None = None, d_payload_out: int | None = None) -> None:
        self._check(self.lib.rsx_partition_scatter_split(
            self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n,
            C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None), "rsx_partition_scatter_split")

    def peer_alloc(self, nbytes: int) -> tuple[int, bytes]:
        """A device buffer other ranks may write to: (address, IPC handle for other processes)."""
        p, h = C.c_void_p(), C.create_string_buffer(64)
        self._check(self.lib.rsx_peer_alloc(self._h, nbytes, C.byref(p), h), "rsx_peer_alloc")
        return int(p.value), h.raw

    def peer_free(self, d_ptr: int) -> None:
        self._check(self.lib.rsx_peer_free(self._h, C.c_void_p(d_ptr)), "rsx_peer_free")

    def peer_open(self, handle: bytes) -> int:
        p = C.c_void_p()
        self._check(self.lib.rsx_peer_open(self._h, C.create_string_buffer(handle, 64), C.byref(p)), "rsx_peer_open")
        return int(p.value)

    def peer_close(self, d_ptr: int) -> None:
        self._check(self.lib.rsx_peer_close(self._h, C.c_void_p(d_ptr)), "rsx_peer_close")

    def peer_enable(self, peer_device: int) -> None:
        self._check(self.lib.rsx_peer_enable(self._h, peer_device), "rsx_peer_enable")

    def sort_from_to(self, d_keys: int, n: int, first_pass: int, last_pass: int, d_keys_out: int,
                     d_payload: int | None = None, d_payload_out: int | None = None) -> None:
        """Device-to-device sort over passes [first_pass, last_pass); the last pass writes to d_keys_out."""
        self._check(self.lib.rsx_sort_from_to(
            self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n, first_pass, last_pass,
            C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None), "rsx_sort_from_to")

    # -- exchange step of the sharded sort on the top B <= 8 bits ----------------
    def msd_count(self, d_keys: int, n: int, bits: int, world: int, d_counts: int) -> None:
        """Keys per bucket of the top `bits` bits into device memory (256 x uint64 at d_counts, natural order), asynchronously."""
        self._check(self.lib.rsx_msd_count(self._h, C.c_void_p(d_keys), n, bits, world, C.c_void_p(d_counts)), "rsx_msd_count")

    def msd_scatter(self, d_keys: int, n: int, d_staging: int, d_payload: int | None = None, d_staging_payload: int | None = None) -> None:
        self._check(self.lib.rsx_msd_scatter(self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n, C.c_void_p(d_staging),
                                             C.c_void_p(d_staging_payload) if d_staging_payload else None), "rsx_msd_scatter")

    def msd_plan(self, d_table: int, stride: int, cap_at: int, rank: int, hip_stream: int = 0, grouping: int = 0) -> None:
        self._check(self.lib.rsx_msd_plan(self._h, C.c_void_p(d_table), stride, cap_at, rank, grouping, C.c_void_p(hip_stream) if hip_stream else None), "rsx_msd_plan")

    def msd_plan_wait(self, waves: int, world: int) -> tuple[list[int], list[int], list[int], int]:
        """(first slot of every wave in this rank's receive buffer, keys of every wave, keys every rank ends up with, verdict bits)"""
        ws, wc, ld, v = (C.c_uint64 * waves)(), (C.c_uint64 * waves)(), (C.c_uint64 * world)(), C.c_uint64()
        self._check(self.lib.rsx_msd_plan_wait(self._h, ws, wc, ld, C.byref(v)), "rsx_msd_plan_wait")
        return [int(x) for x in ws], [int(x) for x in wc], [int(x) for x in ld], int(v.value)

    def msd_push(self, wave: int, d_staging: int, d_peer_keys: int, d_staging_payload: int | None = None, d_peer_payload: int | None = None, parts: int = 0,
                 hip_stream: int = 0) -> None:
        """Wave `wave` of the staging buffer into the owners' receive buffers, on hip_stream (0: the engine's stream)."""
        self._check(self.lib.rsx_msd_push(self._h, wave, C.c_void_p(d_staging), C.c_void_p(d_staging_payload) if d_staging_payload else None,
                                          C.c_void_p(d_peer_keys), C.c_void_p(d_peer_payload) if d_peer_payload else None, parts,
                                          C.c_void_p(hip_stream) if hip_stream else None), "rsx_msd_push")

    def wait_for(self, other: "Engine") -> None:
        """This engine's stream waits for everything enqueued on `other`'s stream so far."""
        self._check(self.lib.rsx_wait_for(self._h, other._h), "rsx_wait_for")

    def record_mark(self, slot: int) -> None:
        """Marks "everything enqueued on this engine's stream so far" under `slot` (0..255)."""
        self._check(self.lib.rsx_record_mark(self._h, slot), "rsx_record_mark")

    def wait_mark(self, other: "Engine", slot: int) -> None:
        """This engine's stream waits for `other`'s mark `slot`."""
        self._check(self.lib.rsx_wait_mark(self._h, other._h, slot), "rsx_wait_mark")

    def key_range(self, d_keys: int, n: int) -> tuple[int, int]:
        lo, hi = C.c_uint64(), C.c_uint64()
        self._check(self.lib.rsx_key_range(self._h, C.c_void_p(d_keys), n, C.byref(lo), C.byref(hi)), "rsx_key_range")
        return int(lo.value), int(hi.value)

    def partition_range(self, d_keys: int, n: int, lo: int, shift: int, mul: int, d_keys_out: int,
                        d_payload: int | None = None, d_payload_out: int | None = None) -> list[int]:
        offs = (C.c_uint64 * 17)()
        self._check(self.lib.rsx_partition_range(
            self._h, C.c_void_p(d_keys), C.c_void_p(d_payload) if d_payload else None, n, lo, shift, mul,
            C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None, offs), "rsx_partition_range")
        return [int(v) for v in offs]

    def result_device(self) -> tuple[int, int]:
        k, p = C.c_void_p(), C.c_void_p()
        self._check(self.lib.rsx_result_device(self._h, C.byref(k), C.byref(p)), "rsx_result_device")
        return int(k.value or 0), int(p.value or 0)

    def copy_result(self, d_keys_out: int, d_payload_out: int | None = None) -> None:
        self._check(self.lib.rsx_copy_result(self._h, C.c_void_p(d_keys_out), C.c_void_p(d_payload_out) if d_payload_out else None), "rsx_copy_result")

    def timings(self, reset: bool = False) -> Runtimes:
        r = Runtimes()
        self._check(self.lib.rsx_timings(self._h, C.byref(r), int(reset)), "rsx_timings")
        return r


def tile_map(num_keys: int, tile_keys: int = 4096, xcd_remap: bool = True, xcd_phase: int = -1) -> tuple[np.ndarray, int]:
    """Tile of every workgroup of a launch over num_keys keys (rsx_tile_map: host arithmetic, no GPU) and the tile count."""
    lib = load_library()
    blocks, ntiles = C.c_uint32(0), C.c_uint32(0)
    rc = lib.rsx_tile_map(num_keys, tile_keys, int(xcd_remap), xcd_phase, None, 0, C.byref(blocks), C.byref(ntiles))
    if rc != 0:
        raise RadixSortError(rc, "rsx_tile_map", lib.rsx_last_error().decode())
    out = np.empty(blocks.value, dtype=np.uint32)
    rc = lib.rsx_tile_map(num_keys, tile_keys, int(xcd_remap), xcd_phase, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(blocks), C.byref(ntiles))
    if rc != 0:
        raise RadixSortError(rc, "rsx_tile_map", lib.rsx_last_error().decode())
    return out, ntiles.value


def sort_host(keys: np.ndarray, payload: np.ndarray | None = None, device: int = 0):
    """upload -> sort -> download of a host array (the shape of ExecuteTask,
    reference src/CRadixSortTask.cpp:289-314).  Returns sorted keys (and payload)."""
    k = np.ascontiguousarray(keys)
    with Engine(k.dtype, max(k.size, 1), payload=payload is not None, device=device) as e:
        e.upload(k, payload)
        e.sort()
        if payload is None:
            return e.download()
        return e.download(want_perm=True)
