"""Energy-modulated substrate objects: the REAL reference (torch backend, CPU, float64) against
the numpy oracle on random 2-D scenes (tests/fuzz_scenes.py::random_substrate_scene).
Development container only.  The modifier goes through torch's convolution and sigmoid in the
reference and through numpy expressions in the oracle, so equality is to rounding (relative
L2 <= 1e-13), not bitwise.

    python tools/fuzz_substrate_reference.py [first_seed] [count] [/root/reference]
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import torch  # noqa: E402
import fdtd as ref  # noqa: E402  (the real reference)

import fuzz_scenes  # noqa: E402
import scenes  # noqa: E402

ref.set_backend("torch")


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    n = np.linalg.norm(np.ravel(b))
    return float(np.linalg.norm(np.ravel(a - b)) / (n if n > 0 else 1.0))


bad = refused = 0
worst = 0.0
for seed in range(first, first + count):
    scene = fuzz_scenes.random_substrate_scene(seed)
    try:
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            rgrid, rmade = scenes.build_api(scene, ref)
            resolved, arrays = scenes.resolve(scene, rgrid, rmade)
            scenes.run_scene(scene, rgrid)
    except Exception as e:  # noqa: BLE001 -- the reference's torch backend does not run every scene
        refused += 1
        print(f"seed {seed}: the reference's torch backend fails: {type(e).__name__}: {str(e)[:120]}", flush=True)
        continue
    ogrid, omade = scenes.build_oracle(resolved, arrays)
    scenes.run_oracle(resolved, ogrid)
    if not np.isfinite(ogrid.E).all():
        continue
    errs = {"E": rel(rgrid.E, ogrid.E), "H": rel(rgrid.H, ogrid.H)}
    for n, ((kind, rt, _, _), (_, ot)) in enumerate(zip(rmade, omade)):
        if kind in ("linedet", "blockdet"):
            errs[f"det{n}"] = rel(torch.stack([torch.as_tensor(np.array(v)) if not torch.is_tensor(v) else v
                                               for v in rt.E]) if len(rt.E) else np.zeros(0), np.array(ot.E))
    worst = max(worst, max(errs.values()))
    if max(errs.values()) > 1e-13:
        bad += 1
        print(f"seed {seed}: {errs}\n   {scene}", flush=True)
print(f"{count} substrate scenes: {bad} above 1e-13, worst {worst:.2e}; {refused} not run by the reference")
sys.exit(1 if bad else 0)
