"""Backward-through-time adjoint kernels against the reference's autograd gradients.

Golden values (tests/golden/adjoint_*.npz) come from the reference's torch backend
back-propagating the same loss (tests/golden/make_golden_adjoint.py).  Tolerance:
rel-L2 <= 1e-10 in float64 (reverse-time summation order differs from autograd's),
<= 2e-4 in float32 against the float64 reference.
"""

import os

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd
import scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel())


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 2e-4)])
@pytest.mark.parametrize("name", sorted(scenes.ADJOINT_SCENES))
def test_gradients_match_reference_autograd(name, dtype, tol):
    fdtd.set_backend(f"hip.{dtype}")
    scene = scenes.ADJOINT_SCENES[name]
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    grid, made = scenes.build_api(scene, fdtd)
    for n in scene["chunks"]:
        grid.run(n, progress_bar=False)
    assert grid.time_steps_passed == scene["steps"]
    loss = scenes.adjoint_loss(scene, grid, torch)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_l2(grid.E.detach().cpu().numpy(), z["E"]) <= (1e-12 if dtype == "float64" else 1e-5)
    assert abs(loss.item() - float(z["loss"])) <= (1e-11 if dtype == "float64" else 1e-4) * abs(float(z["loss"]))
    seen = 0
    for n, (kind, t, _, _) in enumerate(made):
        if kind in ("learnable", "substrate"):
            g = t.inverse_permittivity.grad
            assert g is not None and tuple(g.shape) == z[f"grad{n}"].shape
            err = rel_l2(g.cpu().numpy(), z[f"grad{n}"])
            assert err <= tol, f"{name}/{dtype}: object {n} gradient rel-L2 {err:.3e}"
            seen += 1
        if kind == "substrate":   # the modifier's convolution weights learn too
            for got, key in ((t.nonlin_conv.weight.grad, f"gradw{n}"), (t.nonlin_conv.bias.grad, f"gradb{n}")):
                assert got is not None and tuple(got.shape) == z[key].shape
                err = rel_l2(got.cpu().numpy(), z[key])
                assert err <= tol, f"{name}/{dtype}: {key} rel-L2 {err:.3e}"
        if kind == "cortical":    # every seed tensor of the source and its gate's weights
            for key, leaf in t.leaves.items():
                want = z[f"src{n}_{key}"]
                assert leaf.grad is not None and tuple(leaf.grad.shape) == want.shape, key
                err = rel_l2(leaf.grad.cpu().numpy(), want)
                assert err <= tol, f"{name}/{dtype}: source {key} gradient rel-L2 {err:.3e}"
    assert seen >= 1


def test_gradient_against_finite_differences():
    """Directional derivative of the loss w.r.t. the learnable tensor and the initial
    fields, central differences in float64."""
    fdtd.set_backend("hip.float64")
    scene = scenes.ADJOINT_SCENES["adjoint_pml3d"]
    rs = np.random.RandomState(3)

    def run(delta_M, delta_E, eps):
        grid, made = scenes.build_api(scene, fdtd)
        obj = [t for k, t, _, _ in made if k == "learnable"][0]
        with torch.no_grad():
            obj.inverse_permittivity += eps * delta_M
        grid.E = (eps * delta_E).clone().requires_grad_(True)
        E0 = grid.E
        grid.run(12, progress_bar=False)
        return scenes.adjoint_loss(scene, grid, torch), obj, E0

    grid0, made0 = scenes.build_api(scene, fdtd)
    obj0 = [t for k, t, _, _ in made0 if k == "learnable"][0]
    dM = fdtd.backend.array(rs.randn(*obj0.inverse_permittivity.shape))
    dE = fdtd.backend.array(1e-2 * rs.randn(*grid0.E.shape))
    loss, obj, E0 = run(dM, dE, 0.0)
    loss.backward()
    analytic = (obj.inverse_permittivity.grad * dM).sum().item() + (E0.grad * dE).sum().item()
    h = 1e-6
    with torch.no_grad():
        lp, _, _ = run(dM, dE, h)
        lm, _, _ = run(dM, dE, -h)
    numeric = (lp.item() - lm.item()) / (2 * h)
    assert abs(analytic - numeric) <= 1e-6 * max(abs(numeric), 1e-30), (analytic, numeric)


def test_no_grad_run_leaves_no_graph():
    fdtd.set_backend("hip.float64")
    scene = scenes.ADJOINT_SCENES["adjoint_focus2d"]
    grid, made = scenes.build_api(scene, fdtd)
    with torch.no_grad():
        grid.run(5, progress_bar=False)
    assert grid.E.grad_fn is None and not grid.E.requires_grad
    grid.run(5, progress_bar=False)
    assert grid.E.grad_fn is not None
    grid.update_E()                      # half steps are recorded too
    assert grid.E.grad_fn is not None


@pytest.mark.parametrize("name", ["adjoint_focus2d", "adjoint_pml3d", "adjoint_substrate2d", "adjoint_cortical2d_b",
                                  "adjoint_softfeed"])
def test_half_steps_driven_by_hand_give_the_reference_gradients(name):
    """grid.update_E() / grid.update_H() under autograd (reference tests/test_detectors.py:62-70
    drives the grid that way; the reference records every operation, fdtd/backend.py:57-59):
    the first chunk of the scene advanced by explicit half steps, the rest by run(), must
    give the gradients of the reference's autograd for the whole run."""
    fdtd.set_backend("hip.float64")
    scene = scenes.ADJOINT_SCENES[name]
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    grid, made = scenes.build_api(scene, fdtd)
    by_hand = scene["chunks"][0] if len(scene["chunks"]) > 1 else scene["steps"] // 2
    for _ in range(by_hand):
        grid.update_E()
        grid.update_H()
        grid.time_steps_passed += 1
    grid.run(scene["steps"] - by_hand, progress_bar=False)
    loss = scenes.adjoint_loss(scene, grid, torch)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_l2(grid.E.detach().cpu().numpy(), z["E"]) <= 1e-12
    assert abs(loss.item() - float(z["loss"])) <= 1e-11 * abs(float(z["loss"]))
    seen = 0
    for n, (kind, t, _, _) in enumerate(made):
        if kind in ("learnable", "substrate"):
            err = rel_l2(t.inverse_permittivity.grad.cpu().numpy(), z[f"grad{n}"])
            assert err <= 1e-10, f"{name}: object {n} gradient rel-L2 {err:.3e}"
            seen += 1
        if kind == "substrate":
            assert rel_l2(t.nonlin_conv.weight.grad.cpu().numpy(), z[f"gradw{n}"]) <= 1e-10
        if kind == "cortical":
            for key, leaf in t.leaves.items():
                err = rel_l2(leaf.grad.cpu().numpy(), z[f"src{n}_{key}"])
                assert err <= 1e-10, f"{name}: source {key} gradient rel-L2 {err:.3e}"
    assert seen


def test_field_write_between_recorded_half_steps():
    """A value written into E between the half steps (reference tests/test_detectors.py:64-68)
    is part of the recorded computation: the loss's derivative with respect to that value,
    from autograd, equals the finite difference."""
    fdtd.set_backend("hip.float64")

    def run(v):
        g = fdtd.Grid((14, 13, 12))
        g[0:3, :, :] = fdtd.PML()
        g[4:9, 3:8, 2:9] = fdtd.LearnableAnisotropicObject(permittivity=2.0)
        g[7, 6, 6] = fdtd.PointSource(period=8)
        poke = torch.tensor(v, dtype=torch.float64, device=g.E.device, requires_grad=True)
        for s in range(6):
            g.update_E()
            if s == 2:
                g.E[5, 6, 7, 1] = poke * 1.0
            g.update_H()
            g.time_steps_passed += 1
        return (g.E ** 2).sum() + (g.H ** 2).sum(), poke

    loss, poke = run(0.3)
    loss.backward()
    h = 1e-6
    with torch.no_grad():
        lp, lm = run(0.3 + h)[0].item(), run(0.3 - h)[0].item()
    numeric = (lp - lm) / (2 * h)
    assert abs(poke.grad.item() - numeric) <= 1e-6 * abs(numeric), (poke.grad.item(), numeric)


def test_substrate_forward_without_grad():
    """The energy-modulated object outside autograd: same fields as the reference; and the
    fused whole-step kernel (taken by ``run`` when no periodic boundary is present) agrees
    bit for bit with separate half steps."""
    fdtd.set_backend("hip.float64")
    scene = scenes.ADJOINT_SCENES["adjoint_substrate2d"]
    z = np.load(os.path.join(GOLDEN, "adjoint_substrate2d.npz"))
    with torch.no_grad():
        grid, _ = scenes.build_api(scene, fdtd)
        for n in scene["chunks"]:
            grid.run(n, progress_bar=False)
        assert grid.E.grad_fn is None
        assert rel_l2(grid.E.cpu().numpy(), z["E"]) <= 1e-12
        assert rel_l2(grid.H.cpu().numpy(), z["H"]) <= 1e-12
        open_scene = dict(scene, items=[it for it in scene["items"] if it[0] != "periodic"])
        a, _ = scenes.build_api(open_scene, fdtd)
        b, _ = scenes.build_api(open_scene, fdtd)
        a.run(scene["steps"], progress_bar=False)
        for _ in range(scene["steps"]):
            b.update_E()
            b.update_H()
            b.time_steps_passed += 1
        torch.cuda.synchronize()
        assert float(a.E.abs().max()) > 0
        assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)


def test_cortical_source_forward_without_grad():
    """The energy-gated column source outside autograd (inputs prepared per run under
    no_grad), stepped as the scene prescribes and by half steps: the reference's fields."""
    fdtd.set_backend("hip.float64")
    scene = scenes.ADJOINT_SCENES["adjoint_cortical2d"]
    z = np.load(os.path.join(GOLDEN, "adjoint_cortical2d.npz"))
    with torch.no_grad():
        grid, _ = scenes.build_api(scene, fdtd)
        for n in scene["chunks"]:
            grid.run(n, progress_bar=False)
        assert grid.E.grad_fn is None
        assert rel_l2(grid.E.cpu().numpy(), z["E"]) <= 1e-12
        assert rel_l2(grid.H.cpu().numpy(), z["H"]) <= 1e-12
        half, _ = scenes.build_api(scene, fdtd)
        for _ in range(scene["steps"]):
            half.update_E()
            half.update_H()
            half.time_steps_passed += 1
        assert rel_l2(half.E.cpu().numpy(), z["E"]) <= 1e-12
        # an unseeded source does nothing (reference fdtd/sources.py:583-585)
        idle, made = scenes.build_api(scene, fdtd)
        for kind, t, _, _ in made:
            if kind == "cortical":
                t.cc_dirs = None
        idle.run(3, progress_bar=False)
        assert torch.isfinite(idle.E).all()


@pytest.mark.parametrize("absorbing", [True, False], ids=["absorbing", "dielectric"])
def test_large_plane_adjoint_against_finite_differences(absorbing):
    """Planes of 64 x 64 cells engage the row-mapped adjoint kernels (and, forward, the
    large-grid half-step kernels with curl traces): directional derivative against central
    differences in float64, PMLs on four faces, a learnable object overlapped by an
    absorbing or a plain dielectric one."""
    fdtd.set_backend("hip.float64")
    rs = np.random.RandomState(9)
    shape = (14, 64, 64)

    def build(eps, dM):
        g = fdtd.Grid(shape)
        g[0:3, :, :] = fdtd.PML()
        g[-3:, :, :] = fdtd.PML()
        g[:, 0:4, :] = fdtd.PML()
        g[:, :, -4:] = fdtd.PML()
        obj = fdtd.LearnableAnisotropicObject(permittivity=2.5)
        g[4:10, 20:40, 10:50] = obj
        if absorbing:
            g[5:9, 30:50, 40:60] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=3e4)
        else:
            g[5:9, 30:50, 40:60] = fdtd.Object(permittivity=1.5)
        g[7, 32, 30] = fdtd.PointSource(period=9)
        with torch.no_grad():
            obj.inverse_permittivity += eps * dM
        return g, obj

    def loss_of(g):
        w = torch.as_tensor(np.random.RandomState(4).rand(*shape, 3), dtype=g.E.dtype, device=g.E.device)
        return (w * g.E ** 2).sum() + (w * g.H ** 2).sum()

    g0, obj0 = build(0.0, 0.0)
    dM = fdtd.backend.array(rs.randn(*obj0.inverse_permittivity.shape))
    g0.run(13, progress_bar=False)
    loss_of(g0).backward()
    analytic = (obj0.inverse_permittivity.grad * dM).sum().item()
    h = 1e-6
    with torch.no_grad():
        gp, _ = build(h, dM)
        gp.run(13, progress_bar=False)
        gm, _ = build(-h, dM)
        gm.run(13, progress_bar=False)
        numeric = (loss_of(gp).item() - loss_of(gm).item()) / (2 * h)
    assert abs(analytic) > 0
    assert abs(analytic - numeric) <= 1e-6 * abs(numeric), (analytic, numeric)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_single_recorded_steps_are_taken_out_of_place(dtype):
    """A recorded run advanced one step per autograd node (reference cortical/autoencoder.py:108-112)
    takes each step in one launch from the node's input tensors into new ones
    (fdtdhip_step_into) instead of copying the fields first: the inputs stay what they were,
    fields and memory variables equal those of one node over all steps bit for bit, and so do
    the gradients.  2-D grid with four PMLs (the one-lane-per-cell kernel: psi_E goes into new
    arrays too), a learnable object, a source and a detector."""
    fdtd.set_backend(f"hip.{dtype}")
    rs = np.random.RandomState(12)
    E0n, off = 1e-2 * rs.randn(60, 50, 1, 3), 0.05 * rs.randn(20, 14, 1, 3, 3)

    def build():
        g = fdtd.Grid((60, 50, 1))
        g[0:6, :, :] = fdtd.PML()
        g[-6:, :, :] = fdtd.PML()
        g[:, 0:6, :] = fdtd.PML()
        g[:, -6:, :] = fdtd.PML()
        obj = fdtd.LearnableAnisotropicObject(permittivity=2.0)
        g[20:40, 18:32, 0] = obj
        with torch.no_grad():
            obj.inverse_permittivity += fdtd.backend.array(off)
        g[12, 25, 0] = fdtd.PointSource(period=9)
        det = fdtd.LineDetector()
        g[45, 10:40, 0] = det
        E0 = fdtd.backend.array(E0n).requires_grad_(True)
        g.E = E0
        return g, obj, det, E0

    steps = 9
    a, obj_a, det_a, E0a = build()
    a.run(steps, progress_bar=False)
    b, obj_b, det_b, E0b = build()
    held = []
    for _ in range(steps):
        before = b.E
        snapshot = before.detach().clone()
        b.run(1, progress_bar=False)
        assert b.E is not before and torch.equal(before.detach(), snapshot)   # the input is left alone
        held.append(b.E)
    torch.cuda.synchronize()
    assert b.time_steps_passed == steps
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)
    for pa, pb in zip(a.boundaries, b.boundaries):
        assert torch.equal(pa._psiE, pb._psiE) and torch.equal(pa._psiH, pb._psiH)
    assert torch.equal(torch.stack(list(det_a.E)), torch.stack(list(det_b.E)))
    w = fdtd.backend.array(rs.randn(60, 50, 1, 3))
    (w * a.E * a.E).sum().backward()
    # (the loss of the stepwise run also reads the E of an earlier node: still there)
    ((w * b.E * b.E).sum() + 0.0 * held[3].sum()).backward()
    assert float(obj_a.inverse_permittivity.grad.abs().max()) > 0
    tol = 1e-12 if dtype == "float64" else 1e-5
    assert rel_l2(obj_b.inverse_permittivity.grad.cpu(), obj_a.inverse_permittivity.grad.cpu()) <= tol
    assert rel_l2(E0b.grad.cpu(), E0a.grad.cpu()) <= tol
