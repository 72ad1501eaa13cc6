"""Where does the float32 engine's distance from the float64 oracle come from?

    python tools/fuzz_fp32_twin.py [first_seed] [count] [big] [extras] [crowded] [out.json]

For every seeded random scene (tests/fuzz_scenes.py) three trajectories are compared: the
HIP engine in float32, the oracle's float32 twin (same slice algebra on float32 arrays,
oracle/yee_numpy.py) and the float64 oracle.  Per quantity (E, H, detector records, loop
currents, live PML memory variables) the relative L2 distances
    engine32 - oracle64,   engine32 - twin32,   twin32 - oracle64
are printed for the scenes whose engine32 - oracle64 distance exceeds 1e-5 on a field
quantity, and summarised over all scenes.  If the engine's arithmetic is that of the
reference carried out in float32, engine32 - twin32 is far below the other two.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fdtd_amd as fdtd  # noqa: E402
import fuzz_scenes  # noqa: E402
import scenes  # noqa: E402
from test_fuzz import host, live_psi, rel_l2  # noqa: E402


def records(made, grid):
    """name -> array for everything a scene produces (engine or oracle side)."""
    out = {"E": host(grid.E) if torch.is_tensor(grid.E) else grid.E,
           "H": host(grid.H) if torch.is_tensor(grid.H) else grid.H}
    for n, entry in enumerate(made):
        kind, t = entry[0], entry[1]
        if kind == "pml":
            if hasattr(t, "_psiE"):
                out[f"psi{n}"] = np.concatenate([host(t._psiE), host(t._psiH)])
            else:
                out[f"psi{n}"] = live_psi(t)
        elif kind in ("linedet", "blockdet"):
            out[f"det{n}_E"], out[f"det{n}_H"] = host(t.E), host(t.H)
        elif kind == "currentdet":
            out[f"det{n}_I"] = host(t.I)
        elif kind == "soft":
            out[f"src{n}_I"] = host(t.current_detector.I)
    return {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}


def three_way(seed, big=False, extras=False, crowded=False):
    fdtd.set_backend("hip.float32")
    scene = fuzz_scenes.random_scene(seed, big=big, extras=extras, crowded=crowded)
    grid, made = scenes.build_api(scene, fdtd)
    resolved, arrays = scenes.resolve(scene, grid, made)
    o64, m64 = scenes.build_oracle(resolved, arrays)
    o32, m32 = scenes.build_oracle(resolved, arrays, dtype=np.float32)
    with torch.no_grad():
        scenes.run_scene(scene, grid)
    scenes.run_oracle(resolved, o64)
    scenes.run_oracle(resolved, o32)
    torch.cuda.synchronize()
    if not np.isfinite(o64.E).all():
        return None
    assert o32.E.dtype == np.float32 and o32.H.dtype == np.float32
    e, t, r = records(made, grid), records(m32, o32), records(m64, o64)
    return {k: (rel_l2(e[k], r[k]), rel_l2(e[k], t[k]), rel_l2(t[k], r[k])) for k in r}, scene


if __name__ == "__main__":
    args = sys.argv[1:]
    first = int(args[0]) if args else 0
    count = int(args[1]) if len(args) > 1 else 100
    flags = args[2:]
    big, extras, crowded = "big" in flags, "extras" in flags, "crowded" in flags
    path = next((f for f in flags if f.endswith(".json")), None)
    worst = {"field": [0, 0, 0], "diff": [0, 0, 0]}   # max over scenes of the three distances
    above, rows = [], {}
    for seed in range(first, first + count):
        out = three_way(seed, big, extras, crowded)
        if out is None:
            continue
        dist, scene = out
        rows[seed] = {k: [float(x) for x in v] for k, v in dist.items()}
        for k, v in dist.items():
            cls = "diff" if k.endswith("_I") or k.startswith("psi") else "field"
            for n in range(3):
                worst[cls][n] = max(worst[cls][n], v[n])
        over = {k: v for k, v in dist.items()
                if not (k.endswith("_I") or k.startswith("psi")) and v[0] > 1e-5}
        if over:
            above.append(seed)
            print(f"seed {seed} shape {scene['grid']['shape']} steps {scene['steps']}: "
                  + ", ".join(f"{k}: e-o64 {v[0]:.2e} e-t32 {v[1]:.2e} t32-o64 {v[2]:.2e}"
                              for k, v in over.items()), flush=True)
    print(f"seeds {first}..{first + count - 1}{' ' + ' '.join(f for f in flags if not f.endswith('.json')) if flags else ''}: "
          f"{len(rows)} scenes, {len(above)} above 1e-5 on a field quantity {above}")
    for cls, what in (("field", "E, H, field detectors"), ("diff", "loop currents, PML memory variables")):
        w = worst[cls]
        print(f"  worst {what}: engine32-oracle64 {w[0]:.2e}, engine32-twin32 {w[1]:.2e}, twin32-oracle64 {w[2]:.2e}")
    if path:
        json.dump({"first": first, "count": count, "flags": flags, "above": above, "worst": worst,
                   "rows": rows}, open(path, "w"))
