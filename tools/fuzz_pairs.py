"""Random plans that may take two steps per pass (constant materials, PML faces of any
thickness on any subset of the faces, objects of the three plain kinds anywhere, up to eight
source points, cell detectors) run three ways on the HIP engine and compared bit for bit --
fields, PML memory variables, detector readings:

    a  grid.run(...) in two calls, two steps per pass wherever the plan allows;
    b  the same with FDTDHIP_MARCH2=0 (one step per pass);
    c  explicit half steps (update_E / update_H).

    FDTDHIP_FUSED=2 python tools/fuzz_pairs.py [first_seed] [count] [float32|float64]

(FDTDHIP_FUSED=2 lets grids of this size march at all.)  Prints how many scenes took pairs.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd import _lib  # noqa: E402


def scene(seed):
    rs = np.random.RandomState(seed)
    # (every fifth scene: few planes or rows, down to where no room is left for the two-step kernel)
    lo = 4 if seed % 5 == 2 else 20
    shape = (int(rs.randint(lo, 61)), int(rs.randint(lo, 61)), int(rs.randint(64, 161)))
    faces = [(axis, side, int(rs.randint(1, min(8, shape[axis] // 2 + 1)))) for axis in range(3)
             for side in ("low", "high") if rs.rand() < 0.6]
    # every fourth scene: a periodic boundary instead of the PMLs of one axis, registered after
    # the PMLs of the other axes (a PML registered after it keeps the plan on half steps)
    periodic = None
    if seed % 4 == 1:
        periodic = int(rs.randint(3))
        faces = [f for f in faces if f[0] != periodic]
    objects = []
    for _ in range(int(rs.randint(0, 4))):
        box = []
        for n in shape:
            a = int(rs.randint(0, max(n - 2, 1)))
            box.append((a, int(rs.randint(a + 1, min(n, a + 24) + 1))))
        objects.append((("plain", "anisotropic", "absorbing")[int(rs.randint(3))], box,
                        1.2 + 2 * rs.rand(), int(rs.randint(1 << 30))))
    sources = [(tuple(int(rs.randint(0, n)) for n in shape), int(rs.randint(4, 20)))
               for _ in range(int(rs.randint(1, 9)))]
    sources = list({c: (c, p) for c, p in sources}.values())
    # every third scene: a line source of ten points or more and / or a plane source -- sources
    # that are applied by launches of their own between the half steps, their box stepped by
    # the shell launches
    big = []
    if seed % 3 == 0:
        if rs.rand() < 0.7:
            axis = int(rs.randint(3)) if min(shape) >= 20 else 2      # (an axis long enough for ten points)
            cell = [int(rs.randint(0, n)) for n in shape]
            a = int(rs.randint(0, shape[axis] - 12))
            key = list(cell)
            key[axis] = slice(a, int(rs.randint(a + 10, min(shape[axis], a + 40))))
            big.append(("line", tuple(key), int(rs.randint(5, 20))))
        if rs.rand() < 0.5:
            axis = int(rs.randint(3))
            key = [slice(None)] * 3
            key[axis] = int(rs.randint(0, shape[axis]))
            pol = "xyz"[(axis + 1 + int(rs.randint(2))) % 3]
            big.append(("plane", tuple(key), int(rs.randint(5, 20)), pol))
        if not big:
            big.append(("line", (shape[0] // 2, shape[1] // 2, slice(3, 15)), 9))
    detectors = []
    for _ in range(int(rs.randint(0, 5))):
        axis = int(rs.randint(3))
        cell = [int(rs.randint(0, n)) for n in shape]
        if rs.rand() < 0.7:
            key = list(cell)
            a = int(rs.randint(0, shape[axis] - 1))
            key[axis] = slice(a, int(rs.randint(a + 1, shape[axis])))
            detectors.append(("line", tuple(key)))
        else:
            # (a block detector's slices include their stop, reference fdtd/detectors.py)
            key = tuple(slice(min(c, n - 2), min(n - 1, c + int(rs.randint(1, 6)))) for c, n in zip(cell, shape))
            detectors.append(("block", key))
    steps = (int(rs.randint(4, 11)), int(rs.randint(1, 10)))
    return dict(shape=shape, faces=faces, objects=objects, sources=sources, big=big, detectors=detectors, periodic=periodic,
                steps=steps, permittivity=1.0 + rs.rand(), rows=bool(rs.rand() < 0.7), seed=seed)


def build(sc, dtype):
    fdtd.set_backend(f"hip.{dtype}")
    rs = np.random.RandomState(sc["seed"] + 1)
    g = fdtd.Grid(sc["shape"], permittivity=sc["permittivity"])
    g.E = fdtd.backend.array(1e-2 * rs.randn(*sc["shape"], 3))
    g.H = fdtd.backend.array(1e-2 * rs.randn(*sc["shape"], 3))
    for kind, box, eps, oseed in sc["objects"]:
        key = tuple(slice(a, b) for a, b in box)
        if kind == "plain":
            g[key] = fdtd.Object(permittivity=eps)
        elif kind == "absorbing":
            g[key] = fdtd.AbsorbingObject(permittivity=eps, conductivity=3e-3 * eps)
        else:
            ani = fdtd.AnisotropicObject(permittivity=eps)
            g[key] = ani
            off = 0.03 * np.random.RandomState(oseed).randn(*[b - a for a, b in box], 3, 3)
            with torch.no_grad():
                ani.inverse_permittivity += fdtd.backend.array(off)
    for axis, side, t in sc["faces"]:
        loc = [slice(None)] * 3
        loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
        g[tuple(loc)] = fdtd.PML()
    if sc["periodic"] is not None:
        loc = [slice(None)] * 3
        loc[sc["periodic"]] = 0
        g[tuple(loc)] = fdtd.PeriodicBoundary()
    for cell, period in sc["sources"]:
        g[cell] = fdtd.PointSource(period=period)
    for item in sc["big"]:
        if item[0] == "line":
            g[item[1]] = fdtd.LineSource(period=item[2])
        else:
            g[item[1]] = fdtd.PlaneSource(period=item[2], polarization=item[3])
    g.dets = []
    for kind, key in sc["detectors"]:
        det = fdtd.LineDetector() if kind == "line" else fdtd.BlockDetector()
        g[key] = det
        g.dets.append(det)
    return g


def state(g):
    out = [g.E, g.H]
    for b in g.boundaries:
        if hasattr(b, "thickness"):
            out += [b._psiE, b._psiH]
    for d in g.dets:
        out += [torch.stack(list(d.E)), torch.stack(list(d.H))]
    return out


def check(seed, dtype):
    sc = scene(seed)
    lib = _lib.load()
    runs = {}
    before = lib.fdtdhip_two_step_launches()
    for way in "abc":
        env = {"b": {"FDTDHIP_MARCH2": "0"}, "a": {} if sc["rows"] else {"FDTDHIP_MARCH2_ROWS": "0"}}.get(way, {})
        os.environ.update(env)
        try:
            g = build(sc, dtype)
            if way == "c":
                for _ in range(sum(sc["steps"])):
                    g.update_E()
                    g.update_H()
                    g.time_steps_passed += 1
            else:
                for n in sc["steps"]:
                    g.run(n, progress_bar=False)
        finally:
            for k in env:
                del os.environ[k]
        runs[way] = state(g)
    torch.cuda.synchronize()
    paired = lib.fdtdhip_two_step_launches() > before
    for way in "ab":
        for n, (x, y) in enumerate(zip(runs[way], runs["c"])):
            if not torch.equal(x, y):
                return f"{way} differs from half steps in tensor {n}: max |diff| {float((x - y).abs().max()):.3e}", sc, paired
    if not float(runs["c"][0].abs().max()) > 0:
        return "empty field", sc, paired
    return None, sc, paired


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
    bad = pairs = 0
    for seed in range(first, first + count):
        err, sc, paired = check(seed, dtype)
        pairs += paired
        if err:
            bad += 1
            print(f"seed {seed}: {err}\n   {sc}", flush=True)
    print(f"{dtype} seeds {first}..{first + count - 1}: {bad} differ, {pairs} scenes took two steps per pass", flush=True)
    sys.exit(1 if bad else 0)
