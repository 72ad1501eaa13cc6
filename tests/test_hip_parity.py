"""Parity of the HIP engine with the reference, on a real MI355X.

Every scene of tests/scenes.py is built through the fdtd_amd API (which calls
libfdtd_hip.so through its C ABI), run, and compared with

  * the golden vectors captured from the reference (tests/golden/*.npz), and
  * the numpy oracle run in the same process on the same inputs.

Tolerances (relative L2 over each array): float64 <= 1e-12, float32 <= 1e-5
against the float64 reference.
"""

import os

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd
import scenes
from oracle import yee_numpy as orc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"float64": 1e-12, "float32": 1e-5}


def rel_l2(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    num = np.linalg.norm((got - want).ravel())
    return num / den if den > 0 else num


def host(t):
    if isinstance(t, (list, tuple)):
        return torch.stack(list(t)).cpu().numpy() if len(t) else np.zeros((0,))
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_curl_known_answers(dtype):
    fdtd.set_backend(f"hip.{dtype}")
    z = np.load(os.path.join(GOLDEN, "curls.npz"))
    for name in ("kat0", "kat1", "big", "flat", "line"):
        F = fdtd.backend.array(z[f"{name}_in"])
        for fn, key in ((fdtd.curl_E, "curl_E"), (fdtd.curl_H, "curl_H")):
            got = host(fn(F))
            if dtype == "float64":
                assert np.array_equal(got, z[f"{name}_{key}"]), (name, key)
            else:
                assert rel_l2(got, z[f"{name}_{key}"]) < 1e-6


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_scene_matches_reference(name, dtype):
    fdtd.set_backend(f"hip.{dtype}")
    tol = TOL[dtype]
    scene = scenes.SCENES[name]
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    grid, made = scenes.build_api(scene, fdtd)
    scenes.run_scene(scene, grid)
    torch.cuda.synchronize()
    assert grid.time_steps_passed == scene["steps"]
    errs = {"E": rel_l2(host(grid.E), z["E"]), "H": rel_l2(host(grid.H), z["H"])}
    for n, (kind, t, _, _) in enumerate(made):
        if kind == "pml":
            live = np.concatenate([host(t._psiE), host(t._psiH)])
            errs[f"psi{n}"] = rel_l2(live, z[f"psi{n}"])
        elif kind in ("linedet", "blockdet"):
            assert len(t.E) == len(t.H) == scene["steps"]
            errs[f"det{n}_E"] = rel_l2(host(t.E), z[f"det{n}_E"])
            errs[f"det{n}_H"] = rel_l2(host(t.H), z[f"det{n}_H"])
        elif kind == "currentdet":
            assert len(t.I) == scene["steps"]
            errs[f"det{n}_I"] = rel_l2(host(t.I), z[f"det{n}_I"])
        elif kind == "soft":   # both voltage records and the loop current that feeds back
            assert len(t.input_voltage) == len(t.source_voltage) == scene["steps"]
            errs[f"src{n}_vin"] = rel_l2(host(t.input_voltage), z[f"src{n}_vin"])
            errs[f"src{n}_vout"] = rel_l2(host(t.source_voltage), z[f"src{n}_vout"])
            errs[f"src{n}_I"] = rel_l2(host(t.current_detector.I), z[f"src{n}_I"])
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, f"{name}/{dtype}: rel-L2 above {tol}: {bad}"
    if dtype == "float64":
        print(f"{name}: max rel-L2 {max(errs.values()):.2e}, "
              f"E bitwise equal: {np.array_equal(host(grid.E), z['E'])}")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_random_state_against_oracle(dtype):
    """Seeded random fields and materials, every half-step feature on, oracle run here."""
    fdtd.set_backend(f"hip.{dtype}")
    rs = np.random.RandomState(11)
    shape = (19, 23, 21)
    g = fdtd.Grid(shape)
    o = orc.OracleGrid(shape)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    ie, im = 1.0 / (1 + rs.rand(*shape, 3)), 1.0 / (1 + 0.2 * rs.rand(*shape, 3))
    o.E[...], o.H[...], o.inverse_permittivity[...], o.inverse_permeability[...] = E0, H0, ie, im
    g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
    g.inverse_permittivity, g.inverse_permeability = fdtd.backend.array(ie), fdtd.backend.array(im)
    for axis, side, t in [(0, "low", 4), (0, "high", 3), (2, "low", 4), (2, "high", 4)]:
        key = [slice(None)] * 3
        key[axis] = slice(0, t) if side == "low" else slice(-t, None)
        g[tuple(key)] = fdtd.PML()
        o.add_pml(axis, side, t)
    g[:, 0, :] = fdtd.PeriodicBoundary()
    o.add_periodic(1)
    g[5:11, 6:15, 7:12] = fdtd.AnisotropicObject(permittivity=2.0)
    oa = o.add_object((5, 11, 6, 15, 7, 12), 2.0, kind="anisotropic")
    off = 0.03 * rs.randn(6, 9, 5, 3, 3)
    oa.inverse_permittivity += off
    with torch.no_grad():
        g.objects[0].inverse_permittivity += fdtd.backend.array(off)
    g[9, 11, 10] = fdtd.PointSource(period=9)
    o.add_point_source((9, 11, 10), period=9)
    g.run(25, progress_bar=False)
    o.run(25)
    torch.cuda.synchronize()
    assert rel_l2(host(g.E), o.E) <= TOL[dtype]
    assert rel_l2(host(g.H), o.H) <= TOL[dtype]


def test_user_mutation_between_half_steps_and_rebinding():
    """Tensors stay live: pokes between half steps and re-bound fields are honoured
    (reference tests/test_detectors.py:62-70, fdtd/grid.py:327-333)."""
    fdtd.set_backend("hip.float64")
    g = fdtd.Grid((12, 12, 12))
    o = orc.OracleGrid((12, 12, 12))
    g[6, 6, 6] = fdtd.PointSource(period=8)
    o.add_point_source((6, 6, 6), period=8)
    for s in range(6):
        g.update_E(); o.update_E()
        g.E[3, 4, 5, 1] += 0.25; o.E[3, 4, 5, 1] += 0.25
        g.update_H(); o.update_H()
        g.time_steps_passed += 1; o.time_steps_passed += 1
    assert rel_l2(host(g.E), o.E) <= 1e-12
    g.reset()
    assert g.time_steps_passed == 0 and not host(g.E).any() and not host(g.H).any()
    o.E[...] = 0; o.H[...] = 0; o.time_steps_passed = 0
    g.inverse_permittivity = g.inverse_permittivity * 0.5      # new tensor
    o.inverse_permittivity *= 0.5
    g.run(7, progress_bar=False); o.run(7)
    assert rel_l2(host(g.E), o.E) <= 1e-12 and rel_l2(host(g.H), o.H) <= 1e-12


@pytest.mark.parametrize("name", ["pml3d", "objects3d", "c1_readme", "detectors"])
def test_run_and_half_steps_agree_bitwise(name):
    """``run(n)`` may take whole steps in one fused kernel where the scene allows it;
    ``update_E``/``update_H`` never do.  Both orders of evaluation round identically."""
    fdtd.set_backend("hip.float64")
    scene = scenes.SCENES[name]
    a, _ = scenes.build_api(scene, fdtd)
    b, _ = scenes.build_api(scene, fdtd)
    a.run(scene["steps"], progress_bar=False)
    for _ in range(scene["steps"]):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    torch.cuda.synchronize()
    assert np.array_equal(host(a.E), host(b.E)) and np.array_equal(host(a.H), host(b.H))


def test_impedance_spectrum_matches_reference():
    """FrequencyRoutines.impedance of the voltage source with series resistance (reference
    fdtd/fourier.py:216-260, numbers from the reference itself in soft_feed.npz); the
    transforms run on the device.  Relative L2 <= 1e-9 (FFT summation order differs)."""
    from fdtd_amd.fourier import FrequencyRoutines

    fdtd.set_backend("hip.float64")
    scene = scenes.SCENES["soft_feed"]
    z = np.load(os.path.join(GOLDEN, "soft_feed.npz"))
    grid, made = scenes.build_api(scene, fdtd)
    fr0 = FrequencyRoutines(grid, made[8][1])
    assert fr0.impedance() == ([], [])           # nothing recorded yet
    scenes.run_scene(scene, grid)
    for n, (kind, t, _, _) in enumerate(made):
        if kind == "soft":
            freqs, spectrum = FrequencyRoutines(grid, t).impedance()
            assert rel_l2(np.abs(host(freqs)), np.abs(z[f"src{n}_imp_freqs"])) <= 1e-12
            got, want = host(spectrum), z[f"src{n}_imp"]
            assert got.shape == want.shape
            assert np.linalg.norm(got - want) <= 1e-9 * np.linalg.norm(want)
    det = [t for k, t, _, _ in made if k == "linedet"][0]
    f, s = FrequencyRoutines(grid, det).FFT()
    assert f.shape == s.shape and s.shape[0] == 2 * scene["steps"]
    with pytest.raises(ValueError):
        FrequencyRoutines(grid, "nonsense").FFT()


def test_save_data_and_energy_plane(tmp_path, monkeypatch):
    """Detector records on disk in the reference's file layout (reference
    fdtd/grid.py:487-502) and the plane energies of its plots (visualization.py:119-146),
    against the golden README scene."""
    fdtd.set_backend("hip.float64")
    scene = scenes.SCENES["c1_readme"]
    z = np.load(os.path.join(GOLDEN, "c1_readme.npz"))
    grid, made = scenes.build_api(scene, fdtd)
    with pytest.raises(Exception, match="Save location not initialized"):
        grid.save_data()
    scenes.run_scene(scene, grid)
    monkeypatch.chdir(tmp_path)
    folder = grid.save_simulation("readme")
    assert os.path.isdir(folder) and folder.endswith("(readme)")
    grid.save_data()
    saved = np.load(os.path.join(folder, "detector_readings.npz"))
    n = [i for i, m in enumerate(made) if m[0] == "linedet"][0]
    assert sorted(saved.files) == ["detector (E)", "detector (H)"]
    assert rel_l2(saved["detector (E)"], z[f"det{n}_E"]) <= 1e-12
    assert rel_l2(saved["detector (H)"], z[f"det{n}_H"]) <= 1e-12
    eE, eH = grid.energy_plane(z=0)
    assert rel_l2(host(eE), (z["E"] ** 2).sum(-1)[:, :, 0]) <= 1e-12
    assert rel_l2(host(eH), (z["H"] ** 2).sum(-1)[:, :, 0]) <= 1e-12
    with pytest.raises(ValueError):
        grid.energy_plane()
    with pytest.raises(ValueError):
        grid.energy_plane(x=1, y=2)
    # the other two orientations on a 3-D state
    g3 = fdtd.Grid((7, 9, 11))
    rs = np.random.RandomState(2)
    E, H = rs.randn(7, 9, 11, 3), rs.randn(7, 9, 11, 3)
    g3.E, g3.H = fdtd.backend.array(E), fdtd.backend.array(H)
    a, b = g3.energy_plane(x=3)
    assert rel_l2(host(a), (E ** 2).sum(-1)[3]) <= 1e-14 and rel_l2(host(b), (H ** 2).sum(-1)[3]) <= 1e-14
    a, b = g3.energy_plane(y=-2)
    assert rel_l2(host(a), (E ** 2).sum(-1)[:, -2, :].T) <= 1e-14
    assert rel_l2(host(b), (H ** 2).sum(-1)[:, -2, :].T) <= 1e-14


def test_backend_precision_switch_between_runs():
    """set_backend after a grid has stepped: the grid follows the new precision (arrays are
    converted, the native plan is rebuilt for it) and keeps matching the oracle."""
    fdtd.set_backend("hip.float64")
    shape = (14, 12, 10)
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    g[0:3, :, :] = fdtd.PML()
    o.add_pml(0, "low", 3)
    g[4:9, 3:8, 2:7] = fdtd.AbsorbingObject(permittivity=2.0, conductivity=1e3)
    o.add_object((4, 9, 3, 8, 2, 7), 2.0, kind="absorbing", conductivity=1e3)
    g[7, 6, 5] = fdtd.PointSource(period=8)
    o.add_point_source((7, 6, 5), period=8)
    g.run(10, progress_bar=False); o.run(10)
    assert rel_l2(host(g.E), o.E) <= 1e-12
    fdtd.set_backend("hip.float32")
    g.run(10, progress_bar=False); o.run(10)
    assert g.E.dtype == torch.float32 and rel_l2(host(g.E), o.E) <= 1e-5
    fdtd.set_backend("hip.float64")
    g.run(5, progress_bar=False); o.run(5)
    assert g.E.dtype == torch.float64 and rel_l2(host(g.H), o.H) <= 1e-5


def test_cells_outside_the_grid_raise_at_the_first_step():
    """A source or detector cell beyond the grid is accepted at registration and raises
    IndexError in the first step, as numpy does for the reference (fdtd/sources.py:111,
    fdtd/detectors.py:117); negative indices count from the end."""
    fdtd.set_backend("hip.float64")
    for bad in ("source", "detector"):
        g = fdtd.Grid((8, 7, 6))
        if bad == "source":
            g[10, 0, 0] = fdtd.PointSource(period=9)
        else:
            g[8, 3, 3] = fdtd.BlockDetector()
        with pytest.raises(IndexError):
            g.run(1, progress_bar=False)
    g, o = fdtd.Grid((8, 7, 6)), orc.OracleGrid((8, 7, 6))
    g[-1, -2, -3] = fdtd.PointSource(period=9)
    o.add_point_source((7, 5, 3), period=9)
    g.run(6, progress_bar=False)
    o.run(6)
    torch.cuda.synchronize()
    assert np.array_equal(host(g.E), o.E) and np.array_equal(host(g.H), o.H)
