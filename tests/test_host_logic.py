"""Host-side logic of the engine (registration, index arithmetic, PML profiles,
error conventions) checked without a GPU, with arrays in host staging memory.

Expected values come from the golden fixtures (what the reference resolved for
the same user-level keys) and from the reference's own API tests.
"""

import os

import numpy as np
import pytest

import fdtd_amd as fdtd
import scenes
from oracle import yee_numpy as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def staging():
    fdtd.set_backend("hip.float64", device="host-staging")
    yield


@pytest.fixture
def grid():  # reference tests/fixtures.py:15-24
    return fdtd.Grid(shape=(10, 10, 10), grid_spacing=100e-9)


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_registration_resolves_like_the_reference(name):
    scene = scenes.SCENES[name]
    g, made = scenes.build_api(scene, fdtd)
    res, _ = scenes.resolve(scene, g, made)
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ref = scenes.loads(z["resolved"])
    assert res["grid"] == ref["grid"]
    assert res["items"] == ref["items"]
    assert np.array_equal(g.inverse_permittivity.numpy(), z["inv_eps"])
    for n, (kind, t, _, _) in enumerate(made):
        if kind in ("object", "absorbing", "anisotropic"):
            assert np.array_equal(t.inverse_permittivity.detach().numpy(), z[f"obj{n}_inv"])
        elif kind == "line":
            assert np.array_equal(t.profile, z[f"src{n}_profile"])


def test_pml_profiles_equal_the_oracle():
    g = fdtd.Grid((12, 11, 13))
    og = orc.OracleGrid((12, 11, 13))
    for axis, side, t, key in [(0, "low", 5, (slice(0, 5),)), (1, "high", 4, (slice(None), slice(-4, None))),
                               (2, "low", 6, (slice(None), slice(None), slice(0, 6)))]:
        pml = fdtd.PML(a=1e-7)
        g[key] = pml
        o = og.add_pml(axis, side, t, a=1e-7)
        assert type(pml).__name__ == f"_PML{'XYZ'[axis]}{side}"
        for mine, theirs in ((pml.bE, o.bE), (pml.cE, o.cE), (pml.bH, o.bH), (pml.cH, o.cH),
                             (pml.sigmaE, o.sigmaE), (pml.sigmaH, o.sigmaH)):
            assert np.array_equal(mine.numpy(), theirs)
        assert pml.phi_E.shape == o.phi_E.shape and pml.psi_Ex.shape == o.psi_E[0].shape


# --- error conventions of the reference's API tests -------------------------
def test_grid_shapes_and_courant():  # reference tests/test_grid.py:17-44
    assert fdtd.Grid((3, 3, 3)).shape == (3, 3, 3)
    assert fdtd.Grid((10.0e-9, 10.0e-9, 10.0e-9), grid_spacing=5.0e-9).shape == (2, 2, 2)
    assert fdtd.Grid((10.0e-9, 10.0e-9, 3), grid_spacing=5.0e-9).shape == (2, 2, 3)
    assert fdtd.Grid((3, 1, 1)).courant_number == pytest.approx(1.0, rel=0.02)
    assert fdtd.Grid((3, 3, 1)).courant_number == pytest.approx(0.5 ** 0.5, rel=0.02)
    assert fdtd.Grid((3, 3, 3)).courant_number == pytest.approx((1 / 3) ** 0.5, rel=0.02)
    with pytest.raises(ValueError):
        fdtd.Grid((3, 3, 3), courant_number=0.9)
    with pytest.raises(ValueError):
        fdtd.Grid((3, 3))
    with pytest.raises(KeyError):
        fdtd.Grid((3, 3, 3))[0, 0, 0, 0] = fdtd.PointSource()


def test_boundary_placement_errors(grid):  # reference tests/test_boundaries.py:19-54
    pb = fdtd.PeriodicBoundary(name="periodic_boundary")
    grid[0, :, :] = pb
    assert pb in grid.boundaries and grid.periodic_boundary is pb
    with pytest.raises(IndexError):
        grid[0:2, :, :] = fdtd.PeriodicBoundary()
    with pytest.raises(IndexError):
        grid[2, :, :] = fdtd.PeriodicBoundary()
    with pytest.raises(AttributeError):  # second boundary on the same faces
        grid[-1, :, :] = fdtd.PeriodicBoundary()
    with pytest.raises(AttributeError):
        grid[0:3, :, :] = fdtd.PML()
    pml = fdtd.PML(name="PML")
    grid[:, 0:3, :] = pml
    assert pml in grid.boundaries
    with pytest.raises(IndexError):
        grid[:, :, 2:4] = fdtd.PML()
    with pytest.raises(ValueError):
        grid[:, :, 0:3] = fdtd.PML(name="PML")  # duplicate name


def test_domain_border_pml(grid):  # reference tests/test_boundaries.py:49-54
    with pytest.raises(IndexError):
        fdtd.DomainBorderPML(grid, grid.Nx // 2 + 1)
    g2 = fdtd.Grid((10, 10, 10))
    fdtd.DomainBorderPML(g2, 3)
    assert sorted(type(b).__name__ for b in g2.boundaries) == sorted(
        f"_PML{a}{s}" for a in "XYZ" for s in ("low", "high"))


def test_plane_source_polarisation(grid):  # reference tests/test_sources.py:17-43
    for key, pol in [((0, slice(None), slice(None)), "x"), ((slice(None), 0, slice(None)), "y"),
                     ((slice(None), slice(None), 0), "z")]:
        with pytest.raises(ValueError):
            grid[key] = fdtd.PlaneSource(polarization=pol)
    table = [((0, slice(None), slice(None)), "y", 1, 2), ((0, slice(None), slice(None)), "z", 2, 1),
             ((slice(None), 0, slice(None)), "x", 0, 2), ((slice(None), 0, slice(None)), "z", 2, 0),
             ((slice(None), slice(None), 0), "x", 0, 1), ((slice(None), slice(None), 0), "y", 1, 0)]
    for key, pol, e, h in table:
        ps = fdtd.PlaneSource(polarization=pol)
        grid[key] = ps
        assert (ps._Epol, ps._Hpol) == (e, h)


def test_source_and_object_errors(grid):
    with pytest.raises(ValueError):
        grid[0:2, 0, 0] = fdtd.PointSource()
    with pytest.raises(ValueError):
        grid[3:4, 3:4, 3:4] = fdtd.LineSource()  # fewer than two points
    with pytest.raises(IndexError):
        grid[[1, 2], [1], [1, 2]] = fdtd.LineSource()
    with pytest.raises(IndexError):
        grid[[1, 2], :, :] = fdtd.Object(permittivity=2.0)
    with pytest.raises(IndexError):
        grid[0:4:2, :, :] = fdtd.Object(permittivity=2.0)
    with pytest.raises(ValueError):
        grid[0:2, :, :] = fdtd.Object(permittivity=2.0, name="shape")  # name taken by the grid


def test_stepping_without_a_device_raises(grid):
    with pytest.raises(RuntimeError):
        grid.update_E()
    with pytest.raises(RuntimeError):
        grid.run(3, progress_bar=False)
    with pytest.raises(RuntimeError):
        fdtd.curl_E(grid.E)


def test_grid_text(grid):
    grid[0:3, :, :] = fdtd.PML(name="pml_xlow")
    grid[5, 5, 5] = fdtd.PointSource(period=12, name="src")
    text = str(grid)
    assert text.startswith("Grid(shape=(10,10,10), grid_spacing=1.00e-07, courant_number=0.57)")
    assert "PML(name='pml_xlow')\n        @ x=0:3, y=:, z=:" in text
    assert "PointSource(period=12, amplitude=1.0, phase_shift=0.0, name='src')" in text
