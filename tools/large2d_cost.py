"""Step time of large 2-D grids (the README example's set-up scaled up): milliseconds per
step and Gcells/s, float64 and float32, with the share of the bulk kernels' ideal traffic time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402


def build(scale):
    nx, ny = 161 * scale, 97 * scale
    g = fdtd.Grid((nx, ny, 1))
    s = scale
    g[11 * s:32 * s, 30 * s:84 * s, 0] = fdtd.Object(permittivity=1.7 ** 2)
    g[84 * s:116 * s, 32 * s:52 * s, 0] = fdtd.Object(permittivity=1.5 ** 2)
    g[48 * s:52 * s, 76 * s:84 * s, 0] = fdtd.LineSource(period=20)
    g[77 * s, :, 0] = fdtd.LineDetector()
    g[0:10, :, :] = fdtd.PML()
    g[-10:, :, :] = fdtd.PML()
    g[:, 0:10, :] = fdtd.PML()
    g[:, -10:, :] = fdtd.PML()
    g[:, :, 0] = fdtd.PeriodicBoundary()
    return g


for dtype, es in (("float64", 8), ("float32", 4)):
    fdtd.set_backend("hip." + dtype, device=0)
    build(4).run(300, progress_bar=False)   # first steps of a process: code objects, clocks
    for scale in (4, 10, 20, 40):
        g = build(scale)
        g.run(50, progress_bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(200, progress_bar=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        cells = g.Nx * g.Ny
        print(f"{dtype} {g.Nx}x{g.Ny}x1: {dt * 1e3:.4f} ms/step, {cells / dt / 1e9:.1f} Gcells/s, "
              f"{cells * 18 * es / dt / 1e9:.0f} GB/s of field traffic", flush=True)
        del g
        torch.cuda.empty_cache()
