"""Step time of mid-size grids on the one-launch path (below 2^19 cells): microseconds per
step for a few shapes with four or six PML faces, a point source and an object."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd import _lib  # noqa: E402

if os.environ.get("EXPLIB"):          # an experimental build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["EXPLIB"])

fdtd.set_backend("hip." + os.environ.get("PREC", "float64"), device=0)
SHAPES = os.environ.get("SHAPES")
shapes = ([tuple(int(v) for v in t.split("x")) for t in SHAPES.split(",")] if SHAPES else
          [(161, 97, 1), (300, 300, 1), (512, 512, 1), (720, 720, 1), (48, 48, 48), (80, 80, 80)])
for shape in shapes:
    g = fdtd.Grid(shape)
    g[0:10, :, :] = fdtd.PML()
    g[-10:, :, :] = fdtd.PML()
    g[:, 0:10, :] = fdtd.PML()
    g[:, -10:, :] = fdtd.PML()
    if shape[2] > 1:
        g[:, :, 0:10] = fdtd.PML()
        g[:, :, -10:] = fdtd.PML()
    g[shape[0] // 2, shape[1] // 2, shape[2] // 2] = fdtd.PointSource(period=20)
    g[15:25, 15:30, 0:1] = fdtd.Object(permittivity=2.0)
    g.run(500, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(2000, progress_bar=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    cells = shape[0] * shape[1] * shape[2]
    print(f"{shape}: {dt * 1e6:.2f} us/step, {cells / dt / 1e9:.2f} Gcells/s", flush=True)
