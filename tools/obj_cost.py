"""Step time with and without objects: 6 PML(10) plus an anisotropic and an absorbing 64^3
object, float32 at 512^3 and float64 at 256^3 (milliseconds per step over 100 steps)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402

S = slice


def build(n, objects):
    g = fdtd.Grid((n, n, n))
    for axis in range(3):
        lo, hi = [S(None)] * 3, [S(None)] * 3
        lo[axis], hi[axis] = S(0, 10), S(-10, None)
        g[tuple(lo)] = fdtd.PML()
        g[tuple(hi)] = fdtd.PML()
    g[n // 2, n // 2, n // 2] = fdtd.PointSource(period=20)
    if objects:
        a = n // 4
        ani = fdtd.AnisotropicObject(permittivity=2.5)
        g[a:a + 64, a:a + 64, a:a + 64] = ani
        off = 0.05 * np.random.RandomState(0).randn(64, 64, 64, 3, 3)
        for c in range(3):
            off[..., c, c] = 0
        with torch.no_grad():
            ani.inverse_permittivity += fdtd.backend.array(off)
        g[n // 2 + 20:n // 2 + 84, a:a + 64, a:a + 64] = fdtd.AbsorbingObject(permittivity=2.0, conductivity=1e-3)
    return g


for dtype, n in (("float32", 512), ("float64", 256)):
    fdtd.set_backend("hip." + dtype, device=0)
    for objects in (False, True):
        g = build(n, objects)
        g.run(20, progress_bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(100, progress_bar=False)
        torch.cuda.synchronize()
        print(f"{dtype} {n}^3 {'with objects' if objects else 'PML only   '}: "
              f"{(time.perf_counter() - t0) * 10:.4f} ms/step", flush=True)
        del g
        torch.cuda.empty_cache()
