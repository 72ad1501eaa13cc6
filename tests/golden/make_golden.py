"""Generate the golden fixtures in this directory from the real reference.

Run only where the reference checkout is importable (the development
container):

    PYTHONPATH=/root/repo python tests/golden/make_golden.py [/root/reference]

For every scene of ``tests/scenes.py`` the script
  1. builds and runs the scene with the reference's numpy backend,
  2. reads back the integer coordinates the reference resolved,
  3. builds and runs the numpy oracle from those integers,
  4. REQUIRES bitwise equality of every output (fields, material arrays,
     live PML memory variables, detector traces), and
  5. writes inputs + reference outputs to ``<scene>.npz``.

Nothing of the reference is stored: fixtures hold numbers only.
"""

import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")

import fdtd as ref  # noqa: E402  (the real reference)
from fdtd.grid import curl_E as ref_curl_E, curl_H as ref_curl_H  # noqa: E402

import scenes  # noqa: E402
from oracle import yee_numpy as orc  # noqa: E402


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        raise SystemExit(f"oracle differs from reference: {what} "
                         f"(max abs diff {np.abs(a - b).max() if a.shape == b.shape else 'shape'})")


def live_psi(pml, axis, psiE, psiH):
    """(4, slab) array: the only entries of the reference's 18 psi components that
    can be non-zero for a PML whose normal is ``axis``; checks the rest are 0."""
    u, v = (axis + 1) % 3, (axis + 2) % 3
    for fam in (psiE, psiH):
        for comp in range(3):
            for d in range(3):
                if not (d == axis and comp in (u, v)):
                    assert not fam[comp][..., d].any(), "unexpected live psi entry"
    return np.stack([psiE[u][..., axis], psiE[v][..., axis], psiH[u][..., axis], psiH[v][..., axis]])


def curl_kats():
    out = {}
    cases = {"kat0": (0, (3, 3, 3)), "kat1": (1, (3, 3, 3)), "big": (2, (7, 6, 5)),
             "flat": (3, (9, 8, 1)), "line": (4, (11, 1, 1))}
    for name, (seed, shp) in cases.items():
        F = np.random.RandomState(seed).randn(*shp, 3)
        cE, cH = ref_curl_E(F), ref_curl_H(F)
        same(orc.curl_E(F), cE, f"curl_E {name}")
        same(orc.curl_H(F), cH, f"curl_H {name}")
        out[f"{name}_in"], out[f"{name}_curl_E"], out[f"{name}_curl_H"] = F, cE, cH
    np.savez_compressed(os.path.join(HERE, "curls.npz"), **out)
    print("curls.npz ok")


def one_scene(name, scene):
    ref.set_backend("numpy")
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints in pulse mode
        rgrid, rmade = scenes.build_api(scene, ref)
        resolved, arrays = scenes.resolve(scene, rgrid, rmade)
        inv_eps0 = rgrid.inverse_permittivity.copy()
        scenes.run_scene(scene, rgrid)
    ogrid, omade = scenes.build_oracle(resolved, arrays)
    same(ogrid.inverse_permittivity, inv_eps0, f"{name}: inverse_permittivity after registration")
    scenes.run_oracle(resolved, ogrid)

    out = dict(arrays)
    out["resolved"] = np.array(scenes.dumps(resolved))
    out["E"], out["H"], out["inv_eps"] = rgrid.E, rgrid.H, inv_eps0
    same(ogrid.E, rgrid.E, f"{name}: E")
    same(ogrid.H, rgrid.H, f"{name}: H")
    assert ogrid.time_steps_passed == rgrid.time_steps_passed
    for n, ((kind, rt, _, _), (_, ot)) in enumerate(zip(rmade, omade)):
        if kind == "pml":
            r = live_psi(rt, ot.axis, [rt.psi_Ex, rt.psi_Ey, rt.psi_Ez], [rt.psi_Hx, rt.psi_Hy, rt.psi_Hz])
            o = live_psi(ot, ot.axis, ot.psi_E, ot.psi_H)
            same(o, r, f"{name}: psi of boundary {n}")
            out[f"psi{n}"] = r
        elif kind in ("object", "absorbing", "anisotropic"):
            same(ot.inverse_permittivity, rt.inverse_permittivity, f"{name}: object {n} tensor")
            out[f"obj{n}_inv"] = np.asarray(rt.inverse_permittivity)
            if kind == "absorbing":
                same(ot.absorption_factor, rt.absorption_factor, f"{name}: object {n} absorption")
        elif kind == "line":
            same(ot.profile, rt.profile, f"{name}: line source profile {n}")
            out[f"src{n}_profile"] = np.asarray(rt.profile)
        elif kind in ("linedet", "blockdet"):
            for f in "EH":
                r = np.array(getattr(rt, f))
                same(np.array(getattr(ot, f)), r, f"{name}: detector {n} {f}")
                out[f"det{n}_{f}"] = r
        elif kind == "currentdet":
            r = np.array(rt.I)
            same(np.array(ot.I), r, f"{name}: detector {n} I")
            out[f"det{n}_I"] = r
        elif kind == "soft":
            for key, attr in (("vin", "input_voltage"), ("vout", "source_voltage")):
                r = np.array(getattr(rt, attr), dtype=np.float64)
                same(np.array(getattr(ot, attr), dtype=np.float64), r, f"{name}: source {n} {attr}")
                out[f"src{n}_{key}"] = r
            r = np.array(rt.current_detector.I)
            same(np.array(ot.current_detector.I), r, f"{name}: source {n} loop current")
            out[f"src{n}_I"] = r
            # the reference's own impedance spectrum of this node (fdtd/fourier.py:216-260)
            from fdtd.fourier import FrequencyRoutines
            with contextlib.redirect_stdout(io.StringIO()):
                freqs, spectrum = FrequencyRoutines(rgrid, rt).impedance()
            out[f"src{n}_imp_freqs"], out[f"src{n}_imp"] = np.asarray(freqs), np.asarray(spectrum)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz ok  ({os.path.getsize(path) / 1024:.0f} KiB)  sumE2={float((rgrid.E ** 2).sum()):.6e}")


if __name__ == "__main__":
    curl_kats()
    only = sys.argv[2:]  # scene names; all scenes when none is given
    for name, scene in scenes.SCENES.items():
        if not only or name in only:
            one_scene(name, scene)
