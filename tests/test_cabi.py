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


@pytest.mark.gpu
def test_bad_arguments_on_a_live_plan_return_errors():
    """On a real device: everything a caller can get wrong about shapes, indices and order of
    calls comes back as a negative status with a message -- before any kernel could read or
    write outside an array."""
    import torch

    lib = _lib.load()
    h = ctypes.c_void_p()
    accepted = []

    def refused(what, rc):
        if rc >= 0:
            accepted.append(what)

    refused("device ordinal", lib.fdtdhip_plan_create(ctypes.byref(h), 6, 5, 4, _lib.F64, 0.5, 99))
    assert lib.fdtdhip_plan_create(ctypes.byref(h), 6, 5, 4, _lib.F64, 0.5, 0) == 0
    try:
        refused("update_E without fields", lib.fdtdhip_update_E(h, 0))
        refused("run without fields", lib.fdtdhip_run(h, 0, 3))
        refused("bulk without fields", lib.fdtdhip_update_E_bulk(h, 0, 6))
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")            # noqa: E731
        E, H, ie, im = (zeros(6, 5, 4, 3) for _ in range(4))
        p = lambda t: ctypes.c_void_p(t.data_ptr())                                       # noqa: E731
        refused("set_fields with a null array", lib.fdtdhip_plan_set_fields(h, p(E), p(H), p(ie), None))
        assert lib.fdtdhip_plan_set_fields(h, p(E), p(H), p(ie), p(im)) == 0
        refused("bulk x range beyond nx", lib.fdtdhip_update_E_bulk(h, 2, 9))
        refused("bulk negative x", lib.fdtdhip_update_E_bulk(h, -1, 3))
        refused("bulk reversed x", lib.fdtdhip_update_H_bulk(h, 4, 2))
        b = _lib.doubles([0.5] * 8)
        psi = zeros(2, 8, 5, 4)
        refused("pml thicker than the grid", lib.fdtdhip_plan_add_pml(h, 0, 0, 7, b, b, b, b, p(psi), p(psi)))
        refused("pml axis", lib.fdtdhip_plan_add_pml(h, 3, 0, 2, b, b, b, b, p(psi), p(psi)))
        refused("pml side", lib.fdtdhip_plan_add_pml(h, 0, 2, 2, b, b, b, b, p(psi), p(psi)))
        refused("pml null array", lib.fdtdhip_plan_add_pml(h, 0, 0, 2, b, b, b, b, None, p(psi)))
        refused("pml thickness 0", lib.fdtdhip_plan_add_pml(h, 0, 0, 0, b, b, b, b, p(psi), p(psi)))
        refused("periodic axis", lib.fdtdhip_plan_add_periodic(h, 5))
        inv = zeros(2, 2, 2, 3)
        box = _lib.int32s
        refused("object beyond nx", lib.fdtdhip_plan_add_object(h, 0, box([0, 7, 0, 2, 0, 2]), p(inv), None))
        refused("object reversed", lib.fdtdhip_plan_add_object(h, 0, box([3, 2, 0, 2, 0, 2]), p(inv), None))
        refused("object negative", lib.fdtdhip_plan_add_object(h, 0, box([0, 2, -1, 2, 0, 2]), p(inv), None))
        refused("object kind", lib.fdtdhip_plan_add_object(h, 9, box([0, 2, 0, 2, 0, 2]), p(inv), None))
        refused("object null tensor", lib.fdtdhip_plan_add_object(h, 0, box([0, 2, 0, 2, 0, 2]), None, None))
        refused("absorbing without factor",
                lib.fdtdhip_plan_add_object(h, _lib.OBJ_ABSORBING, box([0, 2, 0, 2, 0, 2]), p(inv), None))
        amp = _lib.doubles([1.0])
        wave = (9.0, 0.0, 0, 1.0 / 9.0, 4.5, 5.0)
        add_src = lib.fdtdhip_plan_add_source
        refused("source x beyond", add_src(h, _lib.SRC_ADD_POINTS, 2, 2, 1, box([6, 0, 0]), amp, *wave))
        refused("source y negative", add_src(h, _lib.SRC_ADD_POINTS, 2, 2, 1, box([0, -1, 0]), amp, *wave))
        refused("source component", add_src(h, _lib.SRC_ADD_POINTS, 3, 2, 1, box([0, 0, 0]), amp, *wave))
        refused("source kind", add_src(h, 17, 2, 2, 1, box([0, 0, 0]), amp, *wave))
        refused("source null cells", add_src(h, _lib.SRC_ADD_POINTS, 2, 2, 1, None, amp, *wave))
        refused("plane source box beyond", add_src(h, _lib.SRC_SET_BOX, 2, 1, 1, box([0, 6, 0, 6, 0, 1]), amp, *wave))
        add_det = lib.fdtdhip_plan_add_detector
        refused("detector y beyond", add_det(h, _lib.DET_CELLS, 1, box([0, 5, 0]), 1.0))
        refused("detector z beyond", add_det(h, _lib.DET_CELLS, 1, box([0, 0, 4]), 1.0))
        refused("detector kind", add_det(h, 5, 1, box([0, 0, 0]), 1.0))
        det = add_det(h, _lib.DET_CELLS, 1, box([1, 1, 1]), 1.0)
        assert det >= 0
        refused("step without detector output", lib.fdtdhip_update_E(h, 0))
        refused("output of an unknown detector", lib.fdtdhip_plan_set_detector_output(h, det + 3, None, None, 4))
        refused("adjoint with a null array", lib.fdtdhip_adjoint_run(h, 2, None, p(H), p(E)))
        assert not accepted, f"accepted instead of refused: {accepted}"
        assert lib.fdtdhip_last_error()                                                    # a message is kept
        # and the plan still works after all of that
        outE, outH = zeros(2, 1, 3), zeros(2, 1, 3)
        assert lib.fdtdhip_plan_set_detector_output(h, det, p(outE), p(outH), 2) == 0
        assert lib.fdtdhip_run(h, 0, 2) == 0, lib.fdtdhip_last_error()
        torch.cuda.synchronize()
        assert float(E.abs().max()) == 0.0
    finally:
        assert lib.fdtdhip_plan_destroy(h) == 0
