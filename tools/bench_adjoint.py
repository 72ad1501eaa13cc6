"""Timing of differentiable runs (forward with traces + reverse-time adjoint).

    python tools/bench_adjoint.py [small|large|stepwise] [--repeat R]

small    97x97x1 float64, periodic z, point source, learnable 40x20 box, 200 steps forward
         and backward in one autograd node (the learnable_focus set-up)
stepwise the same, advanced one step per autograd node (how reference
         cortical/autoencoder.py:108-109 drives the grid)
large    256^3 float32, 6 PML(10), learnable 64^3 box, 100 steps forward and backward
xlarge   the same at 512^3, 40 steps
plane2d  3000x3000x1 float32, 4 PML(10), learnable 64x64 box, 40 steps
Prints one JSON line per workload: milliseconds per step for forward and for backward.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402


def build(kind):
    if kind in ("small", "stepwise"):
        fdtd.set_backend("hip.float64")
        g = fdtd.Grid((97, 97, 1))
        g[:, :, 0] = fdtd.PeriodicBoundary()
        g[48, 20, 0] = fdtd.PointSource(period=14)
        g[28:68, 40:60, 0:1] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
        return g, 200, (slice(40, 56), 80, slice(0, 1))
    fdtd.set_backend("hip.float32")
    if kind == "plane2d":
        g = fdtd.Grid((3000, 3000, 1))
        g[0:10, :, :] = fdtd.PML()
        g[-10:, :, :] = fdtd.PML()
        g[:, 0:10, :] = fdtd.PML()
        g[:, -10:, :] = fdtd.PML()
        g[1500, 1400, 0] = fdtd.PointSource(period=20)
        g[1468:1532, 1468:1532, 0:1] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
        return g, 40, (slice(1490, 1510), 1540, slice(0, 1))
    if kind == "xlarge":
        g = fdtd.Grid((512, 512, 512))
        for axis in range(3):
            lo, hi = [slice(None)] * 3, [slice(None)] * 3
            lo[axis], hi[axis] = slice(0, 10), slice(-10, None)
            g[tuple(lo)] = fdtd.PML()
            g[tuple(hi)] = fdtd.PML()
        g[256, 180, 256] = fdtd.PointSource(period=20)
        g[224:288, 224:288, 224:288] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
        return g, 40, (slice(240, 272), 330, slice(240, 272))
    g = fdtd.Grid((256, 256, 256))
    for axis in range(3):
        lo, hi = [slice(None)] * 3, [slice(None)] * 3
        lo[axis], hi[axis] = slice(0, 10), slice(-10, None)
        g[tuple(lo)] = fdtd.PML()
        g[tuple(hi)] = fdtd.PML()
    g[128, 90, 128] = fdtd.PointSource(period=20)
    g[96:160, 96:160, 96:160] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
    return g, 100, (slice(120, 136), 110, slice(120, 136))


def once(kind):
    g, steps, window = build(kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "stepwise":
        for _ in range(steps):
            g.run(1, progress_bar=False)
    else:
        g.run(steps, progress_bar=False)
    loss = -(g.E[window] ** 2 + g.H[window] ** 2).sum()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    grad = g.objects[0].inverse_permittivity.grad
    assert grad is not None and bool(torch.isfinite(grad).all())
    cells = g.Nx * g.Ny * g.Nz
    return dict(workload=kind, cells=cells, steps=steps,
                forward_ms_per_step=(t1 - t0) / steps * 1e3, backward_ms_per_step=(t2 - t1) / steps * 1e3,
                forward_Gcells_s=cells * steps / (t1 - t0) / 1e9, backward_Gcells_s=cells * steps / (t2 - t1) / 1e9,
                grad_norm=float(grad.norm()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kinds", nargs="*", default=["small", "stepwise", "large"])
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    for kind in a.kinds:
        best = None
        for _ in range(a.repeat):   # first pass warms up allocator and code objects
            r = once(kind)
            total = r["forward_ms_per_step"] + r["backward_ms_per_step"]
            if best is None or total < best["forward_ms_per_step"] + best["backward_ms_per_step"]:
                best = r
        print(json.dumps({k: (float(f"{v:.5g}") if isinstance(v, float) else v) for k, v in best.items()}), flush=True)
