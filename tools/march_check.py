"""Whole-step marching kernel against explicit half steps on small scenes, one PML face set at a
time (debugging aid).  Run with FDTDHIP_FUSED=2 so that every eligible plan takes the kernel."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402

CASES = {
    "none": [],
    "xlow": [(0, "low", 5)],
    "xhigh": [(0, "high", 5)],
    "ylow": [(1, "low", 5)],
    "yhigh": [(1, "high", 5)],
    "zlow": [(2, "low", 5)],
    "zhigh": [(2, "high", 5)],
    "all": [(0, "low", 5), (0, "high", 5), (1, "low", 5), (1, "high", 5), (2, "low", 5), (2, "high", 5)],
}


def run(shape, dtype, faces, steps=4):
    fdtd.set_backend(f"hip.{dtype}")
    rs = np.random.RandomState(3)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)

    def build():
        g = fdtd.Grid(shape)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        for axis, side, t in faces:
            if shape[axis] < 3 * t:
                continue
            loc = [slice(None)] * 3
            loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
            g[tuple(loc)] = fdtd.PML()
        return g

    a, b = build(), build()
    a.run(steps, progress_bar=False)
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    torch.cuda.synchronize()
    out = [float((a.E - b.E).abs().max()), float((a.H - b.H).abs().max())]
    for pa, pb in zip(a.boundaries, b.boundaries):
        out.append(float((pa._psiE - pb._psiE).abs().max()))
        out.append(float((pa._psiH - pb._psiH).abs().max()))
    return out


if __name__ == "__main__":
    shapes = [(40, 50, 1), (40, 50, 70), (40, 50, 150)]
    for shape, dtype, name in itertools.product(shapes, ("float64", "float32"), CASES):
        print(shape, dtype, name, ["%.2e" % v for v in run(shape, dtype, CASES[name])], flush=True)
