"""Step latency of the small-grid (one launch per step) path by ingredient: the README
example's grid with one thing added at a time (microseconds per step, 4000 steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402

fdtd.set_backend("hip." + os.environ.get("PREC", "float64"), device=0)
S = slice


def build(what):
    g = fdtd.Grid((161, 97, 1))
    if "o" in what:
        g[11:32, 30:84, 0] = fdtd.Object(permittivity=1.7 ** 2)
        g[S(13e-6, 18e-6), S(5e-6, 8e-6), 0] = fdtd.Object(permittivity=1.5 ** 2)
    if "s" in what:
        g[S(7.5e-6, 8.0e-6), S(11.8e-6, 13.0e-6), 0] = fdtd.LineSource(period=1550e-9 / 3e8)
    else:
        g[80, 48, 0] = fdtd.PointSource(period=20)
    if "d" in what:
        g[12e-6, :, 0] = fdtd.LineDetector()
    if "x" in what:
        g[0:10, :, :] = fdtd.PML()
        g[-10:, :, :] = fdtd.PML()
    if "y" in what:
        g[:, 0:10, :] = fdtd.PML()
        g[:, -10:, :] = fdtd.PML()
    if "z" in what:
        g[:, :, 0] = fdtd.PeriodicBoundary()
    return g


CASES = [("point source only", ""), ("line source", "s"), ("+ detector", "sd"), ("+ objects", "sdo"),
         ("pml x", "x"), ("pml xy", "xy"), ("pml xy + periodic z", "xyz"), ("README example", "sdoxy"), ("README example + periodic z", "sdoxyz")]
only = sys.argv[1:]
for name, what in CASES:
    if only and name not in only:
        continue
    g = build(what)
    g.run(3000, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(4000, progress_bar=False)
    torch.cuda.synchronize()
    print(f"{name:24s} {(time.perf_counter() - t0) / 4000 * 1e6:.2f} us/step", flush=True)
