"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/radixsort_hip.h declares; the status enum carries OperationStatus' values; and
— no GPU here — the product path fails loudly instead of falling back to a CPU sort."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "radixsort_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(rsx):
    lib = rsx.load_library()
    declared = _declared_functions()
    assert len(declared) >= 20
    assert sorted(rsx.SYMBOLS) == declared          # the binding covers the whole header
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", rsx.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (rsx_[a-z_0-9]+)", out))
    assert set(declared) <= exported


def test_status_enum_matches_operation_status(rsx):
    # reference src/OperationStatus.h:4-17, same order
    text = open(HEADER).read()
    names = re.findall(r"RSX_([A-Z_]+)\s*=\s*(\d+)", text)
    got = [n for n, _ in sorted(((n, int(v)) for n, v in names if n not in ("RADIX_BITS", "RADIX")), key=lambda t: t[1])
           if not n.startswith("OPT_")]
    assert got == rsx.STATUS_NAMES
    assert rsx.STATUS_NAMES == [
        "OK", "HOST_BUFFERS_FAILED", "INITIALIZATION_FAILED", "DATA_UPLOAD_FAILED", "CALCULATION_FAILED",
        "DATA_DOWNLOAD_FAILED", "CLEANUP_FAILED", "RESIZE_FAILED", "KERNEL_CREATION_FAILED",
        "PROGRAM_CREATION_FAILED", "NO_SOURCE_FOUND", "LOADING_SOURCE_FAILED"]


def test_option_constants_match_the_header(rsx):
    """every RSX_OPT_* of include/radixsort_hip.h has its OPT_* twin in the binding with the same value, and no value is used twice"""
    text = open(HEADER).read()
    opts = {n: int(v) for n, v in re.findall(r"RSX_OPT_([A-Z0-9_]+)\s*=\s*(\d+)", text)}
    assert len(opts) >= 16 and len(set(opts.values())) == len(opts)
    for name, value in opts.items():
        assert getattr(rsx, "OPT_" + name) == value, name
    assert {n[4:] for n in dir(rsx) if n.startswith("OPT_")} == set(opts)
    # the experiments build's options live in a header of their own, beside values the product does not use
    xtext = open(os.path.join(ROOT, "include", "radixsort_hip_experiments.h")).read()
    xopts = {n: int(v) for n, v in re.findall(r"RSX_XOPT_([A-Z0-9_]+)\s*=\s*(\d+)", xtext)}
    assert len(xopts) == 5 and not set(xopts.values()) & set(opts.values())
    for name, value in xopts.items():
        assert getattr(rsx, "XOPT_" + name) == value, name
    assert "RSX_XOPT" not in text and "DEBUG_RAISE" not in text          # no test hook in the public enum


def test_library_is_gfx950_code_object(rsx):
    out = subprocess.run(["strings", "-n", "6", rsx.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "gfx950" in out
    assert rsx.load_library().rsx_version().decode().startswith("radixsort_hip")


def test_no_cpu_fallback_without_gpu(rsx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be observed")
    import numpy as np
    with pytest.raises(rsx.RadixSortError) as ei:
        rsx.sort_host(np.arange(10, dtype=np.uint32))
    assert ei.value.status == 2     # INITIALIZATION_FAILED: no device, no silent CPU path


def test_product_does_not_reference_oracle():
    # the oracle is test infrastructure: nothing under radix-sort_amd/ or include/ may name it
    bad = []
    for base in ("radix-sort_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                p = os.path.join(dirpath, f)
                try:
                    txt = open(p, errors="ignore").read()
                except OSError:
                    continue
                if re.search(r"liboracle|oracle/|_oracle|ref_shim|libref_oracle", txt):
                    bad.append(p)
    assert not bad, bad


def test_rccl_side_library_exports_what_shardcomm_declares():
    """libradixsort_rccl.so (RcclComm.hip, hipcc): the five collectives + the error text ShardComm.h declares, linked against librccl.
    Looked at with nm — it is loaded on demand by RadixSortMultiGPU only, never by a single-GPU run or by the Python binding."""
    lib = os.path.join(ROOT, "radix-sort_amd", "host", "libradixsort_rccl.so")
    assert os.path.exists(lib), "run __graft_entry__.build()"
    header = open(os.path.join(ROOT, "radix-sort_amd", "host", "ShardComm.h")).read()
    declared = set(re.findall(r"\b(rsxc_rccl_[a-z_]+)\(", header))
    assert declared == {"rsxc_rccl_create", "rsxc_rccl_destroy", "rsxc_rccl_all_gather", "rsxc_rccl_all_to_all_v", "rsxc_rccl_fence", "rsxc_rccl_last_error"}
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if " T " in line}
    assert declared <= exported
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    for sym in ("ncclCommInitAll", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclAllReduce"):
        assert sym in undefined
    host = os.path.join(ROOT, "radix-sort_amd", "host", "libradixsort_host.so")
    needed = subprocess.run(["readelf", "-d", host], capture_output=True, text=True, check=True).stdout
    assert "librccl" not in needed and "libradixsort_rccl" not in needed        # on demand only
