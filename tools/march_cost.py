"""Step time of a 3-D grid with chosen PML faces and objects (which kernel runs is the library's
choice; FDTDHIP_FUSED=0/2 forces two half-step launches / the marching step).

    python tools/march_cost.py [n] [dtype] case...      cases: vac x y z xyz obj xyz+obj
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402


def build(n, case, t=10):
    g = fdtd.Grid((n, n, n))
    axes = [a for a, c in enumerate("xyz") if c in case.split("+")[0]] if case.split("+")[0] not in ("vac", "obj") else []
    for a in axes:
        for key in (slice(0, t), slice(-t, None)):
            loc = [slice(None)] * 3
            loc[a] = key
            g[tuple(loc)] = fdtd.PML()
    if "obj" in case:
        g[n // 4:n // 4 + 64, n // 4:n // 4 + 64, n // 4:n // 4 + 64] = fdtd.AnisotropicObject(permittivity=2.0)
        g[n // 2:n // 2 + 64, n // 2:n // 2 + 64, n // 2:n // 2 + 64] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=1e-3)
    g[n // 2, n // 2, n // 2] = fdtd.PointSource(period=20)
    return g


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
    fdtd.set_backend(f"hip.{dtype}")
    for case in sys.argv[3:] or ["vac", "x", "y", "z", "xyz", "obj", "xyz+obj"]:
        g = build(n, case)
        g.run(6, progress_bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(20, progress_bar=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{n}^3 {dtype} {case:8s} {ms:7.3f} ms/step  {n ** 3 / ms / 1e6:6.1f} Gcells/s", flush=True)
        del g
        torch.cuda.empty_cache()
