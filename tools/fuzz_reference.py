"""Random scenes (tests/fuzz_scenes.py) through the REAL reference and through the numpy
oracle: every output must be bitwise equal.  Runs only where the reference checkout is
importable (the development container), on the CPU:

    python tools/fuzz_reference.py [first_seed] [count] [/root/reference]

Pins the oracle beyond the hand-written golden scenes; nothing of the reference is stored.
"""
import contextlib
import io
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

EXTRAS = os.environ.get("FUZZ_EXTRAS") == "1"   # soft and direction-vector point sources too
CROWDED = os.environ.get("FUZZ_CROWDED") == "1"  # many objects, sources and detectors per scene

import fdtd as ref  # noqa: E402  (the real reference)

import fuzz_scenes  # noqa: E402
import scenes  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import live_psi  # noqa: E402


class Differs(Exception):
    pass


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
        raise Differs(f"{what}: max abs diff "
                      f"{np.nanmax(np.abs(a - b)) if a.shape == b.shape else 'shape'}")


def one(seed):
    scene = fuzz_scenes.random_scene(seed, extras=EXTRAS, crowded=CROWDED)
    ref.set_backend("numpy")
    with contextlib.redirect_stdout(io.StringIO()):
        rgrid, rmade = scenes.build_api(scene, ref)
        resolved, arrays = scenes.resolve(scene, rgrid, rmade)
        inv_eps0 = rgrid.inverse_permittivity.copy()
        scenes.run_scene(scene, rgrid)
    ogrid, omade = scenes.build_oracle(resolved, arrays)
    same(ogrid.inverse_permittivity, inv_eps0, "inverse_permittivity after registration")
    scenes.run_oracle(resolved, ogrid)
    same(ogrid.E, rgrid.E, "E")
    same(ogrid.H, rgrid.H, "H")
    for n, ((kind, rt, _, _), (_, ot)) in enumerate(zip(rmade, omade)):
        if kind == "pml":
            r = live_psi(rt, ot.axis, [rt.psi_Ex, rt.psi_Ey, rt.psi_Ez], [rt.psi_Hx, rt.psi_Hy, rt.psi_Hz])
            same(live_psi(ot, ot.axis, ot.psi_E, ot.psi_H), r, f"psi of boundary {n}")
        elif kind in ("object", "absorbing", "anisotropic"):
            same(ot.inverse_permittivity, rt.inverse_permittivity, f"object {n} tensor")
        elif kind in ("linedet", "blockdet"):
            for f in "EH":
                same(np.array(getattr(ot, f)), np.array(getattr(rt, f)), f"detector {n} {f}")
        elif kind == "currentdet":
            same(np.array(ot.I), np.array(rt.I), f"detector {n} I")
        elif kind == "soft":
            for attr in ("input_voltage", "source_voltage"):
                same(np.array(getattr(ot, attr), dtype=np.float64),
                     np.array(getattr(rt, attr), dtype=np.float64), f"source {n} {attr}")
            same(np.array(ot.current_detector.I), np.array(rt.current_detector.I), f"source {n} loop current")
    return scene, bool(np.isfinite(rgrid.E).all())


bad = skipped = 0
for seed in range(first, first + count):
    try:
        scene, finite = one(seed)
        if not finite:
            print(f"seed {seed}: non-finite fields (still equal)")
    except Differs as e:
        bad += 1
        print(f"seed {seed}: ORACLE DIFFERS FROM REFERENCE: {e}\n   {fuzz_scenes.random_scene(seed, extras=EXTRAS, crowded=CROWDED)}")
    except Exception:  # noqa: BLE001 -- a scene the reference itself refuses is a generator bug
        skipped += 1
        print(f"seed {seed}: construction/run error\n{traceback.format_exc(limit=3)}")
print(f"{count} scenes: {bad} differ, {skipped} errored")
sys.exit(1 if bad else 0)
