"""The C-ABI shared library: loads, exports every declared symbol, fails loudly
without a GPU.  No compute calls (CPU-only suite)."""

import ctypes
import os
import re
import sys

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
    assert lib.fdtdhip_step_stage(None, 0, 0) < 0
    assert lib.fdtdhip_plan_whole_step_kind(None, 1, None, None) < 0
    assert lib.fdtdhip_plan_get_timing_parts(None, None, None) < 0
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
        refused("stage out of range", lib.fdtdhip_step_stage(h, 0, 4))
        refused("stage out of order", lib.fdtdhip_step_stage(h, 0, 2))
        refused("staged step without second buffers", lib.fdtdhip_step_stage(h, 0, 0))
        assert lib.fdtdhip_plan_whole_step_kind(h, 1, None, None) == 0    # no whole-step kernel for this plan
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


_STAGED = r"""
import ctypes, sys
import torch
sys.path.insert(0, %r)
from fdtd_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
gen = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda", generator=gen) - 0.5
shape = (12, 9, 70)
E0, H0 = rnd(*shape, 3), rnd(*shape, 3)
below, above = rnd(shape[1], shape[2], 2), rnd(shape[1], shape[2], 2)
ie, im = torch.full((*shape, 3), 0.8, dtype=torch.float64, device="cuda"), torch.ones(*shape, 3, dtype=torch.float64, device="cuda")
results = []
for staged in (False, True):
    h = ctypes.c_void_p()
    assert lib.fdtdhip_plan_create(ctypes.byref(h), *shape, _lib.F64, 0.5, 0) == 0
    E, H = E0.clone(), H0.clone()
    E1, H1 = torch.empty_like(E), torch.empty_like(H)
    assert lib.fdtdhip_plan_set_fields(h, p(E), p(H), p(ie), p(im)) == 0
    assert lib.fdtdhip_plan_set_halo(h, p(below), p(above)) == 0
    assert lib.fdtdhip_plan_scan_materials(h, 1) == 3
    if staged:
        assert lib.fdtdhip_plan_set_step_buffers(h, p(E1), p(H1)) == 0
        core = (ctypes.c_int32 * 6)()
        assert lib.fdtdhip_plan_whole_step_kind(h, 1, core, None) == 2, lib.fdtdhip_last_error()
        assert list(core) == [1, 11, 0, 9, 0, 70], list(core)
        assert lib.fdtdhip_plan_whole_step_kind(h, 0, None, None) == 0      # open faces: not inside fdtdhip_run
        assert lib.fdtdhip_step_stage(h, 0, 1) < 0                           # stage 0 is due
        assert lib.fdtdhip_step_stage(h, 0, 0) == 0, lib.fdtdhip_last_error()
        assert lib.fdtdhip_run(h, 0, 2) < 0 and b"in progress" in lib.fdtdhip_last_error()
        assert lib.fdtdhip_update_E(h, 0) < 0
        assert lib.fdtdhip_step_stage(h, 0, 2) < 0 and b"stage 1 is due" in lib.fdtdhip_last_error()
        for stage in (1, 2, 3):
            assert lib.fdtdhip_step_stage(h, 0, stage) == 0, lib.fdtdhip_last_error()
        for stage in range(4):                                               # back into the first buffers
            assert lib.fdtdhip_step_stage(h, 1, stage) == 0, lib.fdtdhip_last_error()
    else:
        for q in range(2):
            assert lib.fdtdhip_update_E(h, q) == 0 and lib.fdtdhip_update_H(h, q) == 0
    torch.cuda.synchronize()
    results.append((E.clone(), H.clone()))
    assert lib.fdtdhip_plan_destroy(h) == 0
assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
assert not torch.equal(results[0][0], E0)
print("staged ok")
"""


@pytest.mark.gpu
def test_whole_steps_in_stages_through_the_c_abi():
    """fdtdhip_step_stage on a slab with both x faces open, driven through ctypes alone: the
    order of the stages is enforced, other stepping calls are refused while a step is in
    progress, and two staged steps equal two pairs of half steps bit for bit (the ghost planes
    held fixed).  FDTDHIP_FUSED=2 lets a grid this small take the marching kernel."""
    import subprocess

    done = subprocess.run([sys.executable, "-c", _STAGED % ROOT], env=dict(os.environ, FDTDHIP_FUSED="2"),
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "staged ok" in done.stdout, done.stdout[-2000:] + done.stderr[-3000:]
