"""The numpy oracle replayed against the golden vectors captured from the reference.

Equality is exact (bitwise): the oracle performs the reference's floating-point
operations in the reference's order.  No GPU and no reference checkout needed.
"""

import glob
import os

import numpy as np
import pytest

import scenes
from oracle import yee_numpy as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENE_FILES = sorted(
    os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
    if os.path.basename(p) not in ("curls.npz",) and not os.path.basename(p).startswith("adjoint")
)


def test_curl_known_answers():
    z = np.load(os.path.join(GOLDEN, "curls.npz"))
    for name in ("kat0", "kat1", "big", "flat", "line"):
        F = z[f"{name}_in"]
        assert np.array_equal(orc.curl_E(F), z[f"{name}_curl_E"])
        assert np.array_equal(orc.curl_H(F), z[f"{name}_curl_H"])


def test_curl_reference_test_vectors():
    """Spot values and the zero pattern of the reference's own curl tests
    (reference tests/test_grid.py:47-104 for curl_E, 107-164 for curl_H)."""
    cE = orc.curl_E(np.random.RandomState(0).randn(3, 3, 3, 3))
    assert cE[0, 0, 0] == pytest.approx([-0.99186526, -0.01377993, 2.48607585])
    assert cE[1, 1, 1] == pytest.approx([-2.47477698, -1.91668701, 2.82293032])
    assert cE[0, 2, 2] == pytest.approx([0.0, 1.2263907, 0.94356054])
    assert cE[2, 0, 2] == pytest.approx([0.94212919, 0.0, -0.72440584])
    assert cE[2, 2, 0] == pytest.approx([-0.36402867, -1.82421078, 0.0])
    assert not cE[2, 2, 2].any()
    cH = orc.curl_H(np.random.RandomState(1).randn(3, 3, 3, 3))
    assert not cH[0, 0, 0].any()
    assert cH[0, 0, 1] == pytest.approx([-1.47716404, -2.69731399, 0.0])
    assert cH[0, 1, 0] == pytest.approx([-1.53196896, 0.0, 1.87371574])
    assert cH[1, 0, 0] == pytest.approx([0.0, -1.05852722, 0.34386833])
    assert cH[1, 1, 1] == pytest.approx([0.22579499, 3.88075292, -1.24148611])
    assert cH[2, 2, 2] == pytest.approx([-0.78607747, 2.12528137, 0.27793831])


def test_readme_example_checksums():
    """Scalars recorded from the reference run of its README example (200 steps)."""
    z = np.load(os.path.join(GOLDEN, "c1_readme.npz"))
    assert float((z["E"] ** 2).sum()) == pytest.approx(5.668938171999647, rel=1e-13)
    assert float((z["H"] ** 2).sum()) == pytest.approx(5.805371108944498, rel=1e-13)
    assert float(np.abs(z["E"]).max()) == pytest.approx(0.16008116798751112, rel=1e-13)


@pytest.mark.parametrize("name", SCENE_FILES)
def test_scene_bitwise(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    resolved = scenes.loads(z["resolved"])
    grid, made = scenes.build_oracle(resolved, z)
    assert np.array_equal(grid.inverse_permittivity, z["inv_eps"])
    scenes.run_oracle(resolved, grid)
    assert np.array_equal(grid.E, z["E"])
    assert np.array_equal(grid.H, z["H"])
    for n, (kind, t) in enumerate(made):
        if kind == "pml":
            u, v, a = (t.axis + 1) % 3, (t.axis + 2) % 3, t.axis
            live = np.stack([t.psi_E[u][..., a], t.psi_E[v][..., a], t.psi_H[u][..., a], t.psi_H[v][..., a]])
            assert np.array_equal(live, z[f"psi{n}"])
        elif kind in ("linedet", "blockdet"):
            assert np.array_equal(np.array(t.E), z[f"det{n}_E"])
            assert np.array_equal(np.array(t.H), z[f"det{n}_H"])
        elif kind == "currentdet":
            assert np.array_equal(np.array(t.I), z[f"det{n}_I"])


def test_slab_extension_matches_single_grid():
    """Ghost-plane extension of the oracle: two x-slabs stepped side by side with
    one-plane exchanges reproduce the single-grid run (no reference counterpart)."""
    full = orc.OracleGrid((16, 9, 7))
    full.add_point_source((7, 4, 3), period=9)
    full.add_point_source((8, 4, 3), period=5)
    lo, hi = orc.OracleGrid((8, 9, 7)), orc.OracleGrid((8, 9, 7))
    for g in (lo, hi):  # the Courant number of a slab is the full grid's
        g.courant_number = full.courant_number
    lo.add_point_source((7, 4, 3), period=9)
    hi.add_point_source((0, 4, 3), period=5)
    lo.halo_hi_E = np.zeros((9, 7, 3))
    hi.halo_lo_H = np.zeros((9, 7, 3))
    for _ in range(30):
        full.step()
        lo.update_E(); hi.update_E()
        lo.halo_hi_E = hi.E[0].copy()
        lo.update_H(); hi.update_H()
        hi.halo_lo_H = lo.H[-1].copy()
        lo.time_steps_passed += 1; hi.time_steps_passed += 1
    got = np.concatenate([lo.E, hi.E]), np.concatenate([lo.H, hi.H])
    assert np.allclose(got[0], full.E, rtol=0, atol=1e-15)
    assert np.allclose(got[1], full.H, rtol=0, atol=1e-15)
    assert np.abs(full.E).max() > 1e-3
