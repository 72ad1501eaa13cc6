"""The C-ABI shared library: loads, exports every declared symbol, fails loudly
without a GPU.  No compute calls (CPU-only suite)."""

import ctypes
import os
import re

import pytest

from fdtd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fdtd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdtdhip_\w+)\s*\(", text)))


def test_library_loads_and_exports_header_symbols():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/fdtd_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.fdtdhip_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.load()
    handle = ctypes.c_void_p()
    rc = lib.fdtdhip_plan_create(ctypes.byref(handle), 0, 4, 4, _lib.F64, 0.5, 0)
    assert rc < 0 and b"extents" in lib.fdtdhip_last_error()
    rc = lib.fdtdhip_plan_create(ctypes.byref(handle), 4, 4, 4, 7, 0.5, 0)
    assert rc < 0 and b"dtype" in lib.fdtdhip_last_error()
    assert lib.fdtdhip_update_E(None, 0) < 0
    with pytest.raises(_lib.NativeError):
        _lib.check(lib.fdtdhip_run(None, 0, 1))


def test_no_device_is_a_loud_error():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    assert lib.fdtdhip_device_count() == 0
    handle = ctypes.c_void_p()
    rc = lib.fdtdhip_plan_create(ctypes.byref(handle), 4, 4, 4, _lib.F64, 0.5, 0)
    assert rc < 0 and b"no HIP device" in lib.fdtdhip_last_error()
    import fdtd_amd

    with pytest.raises(RuntimeError):
        fdtd_amd.set_backend("hip")
    with pytest.raises(ValueError):
        fdtd_amd.set_backend("numpy")


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fdtd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"
