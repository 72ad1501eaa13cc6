"""Where the step time of the 512^3 float32 workload with PMLs and objects goes:
times the same grid with one ingredient at a time (ms/step, 60 timed steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd import _lib  # noqa: E402

if os.environ.get("EXPLIB"):          # an experimental build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["EXPLIB"])

PREC = os.environ.get("PREC", "float32")
fdtd.set_backend("hip." + PREC, device=0)
N, T = int(os.environ.get("N", "512")), 10
STEPS = int(os.environ.get("STEPS", "60"))


def faces(g, which):
    if "x" in which:
        g[0:T, :, :] = fdtd.PML()
        g[-T:, :, :] = fdtd.PML()
    if "y" in which:
        g[:, 0:T, :] = fdtd.PML()
        g[:, -T:, :] = fdtd.PML()
    if "z" in which:
        g[:, :, 0:T] = fdtd.PML()
        g[:, :, -T:] = fdtd.PML()


def objects(g, which):
    if "a" in which:
        g[128:192, 128:192, 128:192] = fdtd.AnisotropicObject(permittivity=2.5)
    if "p" in which:
        g[128:192, 128:192, 128:192] = fdtd.Object(permittivity=2.5)
    if "s" in which:   # same cell count as the cube, spread over all z segments
        g[128:192, 128:136, 0:512] = fdtd.AnisotropicObject(permittivity=2.5)
    if "t" in which:   # one cell: the cost of carrying the object code at all
        g[128:129, 128:129, 128:129] = fdtd.Object(permittivity=2.5)
    if "q" in which:   # a quarter of the cube's planes
        g[128:144, 128:192, 128:192] = fdtd.Object(permittivity=2.5)
    if "b" in which:
        g[320:384, 128:192, 128:192] = fdtd.AbsorbingObject(permittivity=2.0, conductivity=1e-3)


def detectors(g):
    g[12:500, 256, 256] = fdtd.LineDetector()
    g[256, 12:500, 256] = fdtd.LineDetector()
    g[256, 256, 12:500] = fdtd.LineDetector()


CASES = [("vacuum", "", "", False), ("pml x", "x", "", False), ("pml y", "y", "", False),
         ("pml z", "z", "", False), ("pml xyz", "xyz", "", False), ("aniso", "", "a", False),
         ("absorbing", "", "b", False), ("both objects", "", "ab", False),
         ("plain object", "", "p", False), ("one-cell object", "", "t", False),
         ("16-plane object", "", "q", False), ("aniso z-spread", "", "s", False),
         ("detectors", "", "", True), ("pml xyz + objects", "xyz", "ab", False),
         ("all", "xyz", "ab", True)]
only = sys.argv[1:]
for name, f, o, d in CASES:
    if only and name not in only:
        continue
    g = fdtd.Grid((N, N, N))
    faces(g, f)
    objects(g, o)
    if d:
        detectors(g)
    g[N // 2, N // 2, N // 2] = fdtd.PointSource(period=20)
    g.run(10, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(STEPS, progress_bar=False)
    torch.cuda.synchronize()
    print(f"{name:20s} {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms/step", flush=True)
    del g
    torch.cuda.empty_cache()
