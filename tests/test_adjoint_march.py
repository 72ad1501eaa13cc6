"""Whole reverse steps (csrc/yee_adjoint_march.hpp) against the two-pass reverse sweep.

With second adjoint arrays the reverse sweep undoes a step in one marching pass over the cells
away from PML slabs and objects and one pass per half step over the others (slabs no object
comes near: a kernel that takes the faces as arguments; otherwise, and around the objects,
one that walks the plan's tables).  Same sums on the same operands: loss gradients with respect to the start fields and the learnable
tensor must equal those of the two-pass sweep bit for bit, and both must agree with finite
differences.  (Reference: torch autograd over fdtd/grid.py:267-273, examples/learnable_focus.py:117-128.)
"""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd

pytestmark = pytest.mark.gpu

_CASES = [("float32", (100, 170, 250)), ("float64", (61, 90, 140)), ("float32", (48, 64, 1024))]
_FACES = {
    "no-pml": [],
    "six-faces": [(2, "low", 7), (2, "high", 7), (0, "low", 7), (0, "high", 7), (1, "low", 7), (1, "high", 7)],
    "four-faces": [(1, "high", 5), (0, "low", 9), (2, "low", 1), (2, "high", 11)],
}


def _gradients(dtype, shape, faces, objects, steps, second_buffers):
    fdtd.set_backend(f"hip.{dtype}")
    bd = fdtd.backend
    nx, ny, nz = shape
    rs = np.random.RandomState(5)
    g = fdtd.Grid(shape)
    E0 = bd.array(1e-2 * rs.randn(*shape, 3)).requires_grad_(True)
    H0 = bd.array(1e-2 * rs.randn(*shape, 3)).requires_grad_(True)
    g.E, g.H = E0, H0
    tensor = None
    if objects:
        learn = fdtd.LearnableAnisotropicObject(permittivity=1.8)
        g[12:30, 20:44, 30:80] = learn
        with torch.no_grad():
            learn.inverse_permittivity += bd.array(0.03 * rs.randn(18, 24, 50, 3, 3))
        tensor = learn.inverse_permittivity
        g[25:36, 30:60, 60:120] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=3e-3)
        if objects != "inside":
            g[2:12, ny - 30:ny - 2, 20:60] = fdtd.Object(permittivity=2.5)   # reaches into faces
    for axis, side, t in _FACES[faces]:
        loc = [slice(None)] * 3
        loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
        g[tuple(loc)] = fdtd.PML()
    g[nx // 2, ny // 2, nz // 2] = fdtd.PointSource(period=11)
    g[0, 0, 0] = fdtd.PointSource(period=7)
    fdtd.Grid._adjoint_second_buffers = second_buffers
    try:
        g.run(steps, progress_bar=False)
        wE, wH = bd.array(rs.randn(*shape, 3)), bd.array(rs.randn(*shape, 3))
        loss = (wE * g.E * g.E).sum() + (wH * g.H).sum()
        loss.backward()
    finally:
        fdtd.Grid._adjoint_second_buffers = True
    torch.cuda.synchronize()
    out = {"dE0": E0.grad, "dH0": H0.grad}
    if tensor is not None:
        out["dtensor"] = tensor.grad
    return out


def _march_case(dtype, shape, faces, objects, steps):
    want = _gradients(dtype, shape, faces, objects, steps, second_buffers=False)
    got = _gradients(dtype, shape, faces, objects, steps, second_buffers=True)
    for name in want:
        assert float(want[name].abs().max()) > 0, name
        assert torch.equal(got[name], want[name]), f"{name}: {dtype} {shape} {faces} objects={objects}"


def _run_child(code):
    here = os.path.dirname(os.path.abspath(__file__))
    code = f"import sys\nsys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n" + code
    # FDTDHIP_ADJ_MARCH=2: whole reverse steps wherever the plan allows, at any size (read once
    # per process, hence the child)
    done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDTDHIP_ADJ_MARCH="2"),
                          capture_output=True, text=True, timeout=280, cwd=here)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    return done.stdout


def test_whole_reverse_steps_equal_the_two_pass_sweep_bitwise():
    out = _run_child(
        "import test_adjoint_march as t\n"
        "for dtype, shape in t._CASES:\n"
        "    for faces, objects, steps in (('no-pml', False, 4), ('six-faces', False, 5), ('four-faces', True, 7),\n"
        "                                  ('no-pml', True, 3), ('six-faces', True, 6), ('six-faces', 'inside', 4),\n"
        "                                  ('four-faces', 'inside', 5)):\n"
        "        t._march_case(dtype, shape, faces, objects, steps)\n"
        "print('marched backwards')\n")
    assert "marched backwards" in out


def test_whole_reverse_steps_against_finite_differences():
    out = _run_child(
        "import test_adjoint_march as t\n"
        "t._finite_difference_case()\n"
        "print('agrees with finite differences')\n")
    assert "agrees with finite differences" in out


def _finite_difference_case():
    """Directional derivative of a quadratic loss along random directions of the start fields and
    the learnable tensor, float64, six faces and objects: adjoint against central differences."""
    fdtd.set_backend("hip.float64")
    bd = fdtd.backend
    shape, steps = (40, 56, 130), 6
    rs = np.random.RandomState(9)
    E0n, H0n = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    dE, dH, dM = rs.randn(*shape, 3), rs.randn(*shape, 3), rs.randn(10, 12, 40, 3, 3)
    wE, wH = rs.randn(*shape, 3), rs.randn(*shape, 3)
    off = 0.03 * rs.randn(10, 12, 40, 3, 3)

    def run(eps, grad):
        g = fdtd.Grid(shape)
        E0 = bd.array(E0n + eps * 1e-2 * dE).requires_grad_(grad)
        H0 = bd.array(H0n + eps * 1e-2 * dH).requires_grad_(grad)
        g.E, g.H = E0, H0
        learn = fdtd.LearnableAnisotropicObject(permittivity=1.8)
        g[14:24, 20:32, 30:70] = learn
        with torch.no_grad():
            learn.inverse_permittivity += bd.array(off + eps * 0.1 * dM)
        g[20:30, 28:40, 60:100] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=3e-3)
        for axis, side, t in _FACES["six-faces"]:
            loc = [slice(None)] * 3
            loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
            g[tuple(loc)] = fdtd.PML()
        g[20, 28, 65] = fdtd.PointSource(period=11)
        if not grad:
            with torch.no_grad():
                g.run(steps, progress_bar=False)
        else:
            g.run(steps, progress_bar=False)
        loss = (bd.array(wE) * g.E * g.E).sum() + (bd.array(wH) * g.H * g.H).sum()
        return loss, E0, H0, learn.inverse_permittivity

    loss, E0, H0, M = run(0.0, True)
    loss.backward()
    adjoint = float((E0.grad * bd.array(1e-2 * dE)).sum() + (H0.grad * bd.array(1e-2 * dH)).sum()
                    + (M.grad * bd.array(0.1 * dM)).sum())
    h = 1e-4
    numeric = float(run(h, False)[0] - run(-h, False)[0]) / (2 * h)
    assert abs(adjoint - numeric) <= 1e-6 * abs(numeric), (adjoint, numeric)


def _recorded_pairs_case():
    from fdtd_amd import _lib

    lib = _lib.load()
    for dtype, shape in _CASES:
        for faces, objects, steps in (("six-faces", True, 10), ("four-faces", "inside", 6), ("no-pml", True, 9)):
            before = lib.fdtdhip_two_step_launches()
            got = _gradients(dtype, shape, faces, objects, steps, second_buffers=True)
            assert lib.fdtdhip_two_step_launches() - before == 2 * ((steps - 1) // 4), (dtype, faces, steps)
            os.environ["FDTDHIP_MARCH2"] = "0"      # (read at every call)
            try:
                want = _gradients(dtype, shape, faces, objects, steps, second_buffers=True)
            finally:
                del os.environ["FDTDHIP_MARCH2"]
            for name in want:
                assert float(want[name].abs().max()) > 0, name
                assert torch.equal(got[name], want[name]), f"{name}: {dtype} {shape} {faces} objects={objects}"


def test_recorded_runs_take_two_steps_per_pass_and_give_the_same_gradients():
    """A recorded run advances two steps per pass like any other where the plan allows it (the
    E half steps of the objects' box, which the shell launches take, write the rows of the curl
    traces one after the other).  Gradients with respect to the initial fields and to the
    learnable tensor against the same run with FDTDHIP_MARCH2=0, bit for bit; FDTDHIP_FUSED=2
    lets grids of these sizes march whatever their share of PML cells (read once per process,
    hence the child)."""
    here = os.path.dirname(os.path.abspath(__file__))
    code = (f"import sys\nsys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
            "import test_adjoint_march as t\nt._recorded_pairs_case()\nprint('same gradients')\n")
    done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDTDHIP_FUSED="2"),
                          capture_output=True, text=True, timeout=280, cwd=here)
    assert done.returncode == 0 and "same gradients" in done.stdout, done.stdout[-2000:] + done.stderr[-2000:]
