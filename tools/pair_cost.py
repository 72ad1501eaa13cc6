"""Time of the two-step kernel and of the launches beside it on 512^3 float32 scenes of growing
content: vacuum; six PML faces; + two 64^3 objects; + three line detectors (the benchmark's m3);
a line source of 200 points and a plane source over a whole plane instead of the point source
(applied by launches of their own, their boxes stepped by the shell launches); wall time per
step beside the kernels' own.

    python tools/pair_cost.py [steps]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fdtd.set_backend("hip.float32")
import time  # noqa: E402

for scene in ("vacuum", "pml", "pml+objects", "pml+objects+detectors", "pml(x only)", "pml(y only)", "pml(z only)",
              "vacuum+line source", "pml+line source", "pml+plane source", "pml(x, y) + periodic z"):
    g = fdtd.Grid((512, 512, 512))
    if scene.startswith("pml"):
        axes = {"pml(x only)": [0], "pml(y only)": [1], "pml(z only)": [2],
                "pml(x, y) + periodic z": [0, 1]}.get(scene, [0, 1, 2])
        for ax in axes:
            lo, hi = [slice(None)] * 3, [slice(None)] * 3
            lo[ax], hi[ax] = slice(0, 10), slice(-10, None)
            g[tuple(lo)] = fdtd.PML()
            g[tuple(hi)] = fdtd.PML()
    if "periodic z" in scene:
        g[:, :, 0] = fdtd.PeriodicBoundary()
    if "line source" in scene:
        g[156:356, 256, 256] = fdtd.LineSource(period=20)
    elif "plane source" in scene:
        g[100, :, :] = fdtd.PlaneSource(period=20, polarization="y")
    else:
        g[256, 256, 256] = fdtd.PointSource(period=20)
    if "objects" in scene:
        g[128:192, 128:192, 128:192] = fdtd.AnisotropicObject(permittivity=2.5)
        g[320:384, 128:192, 128:192] = fdtd.AbsorbingObject(permittivity=2.0, conductivity=1e-3)
    if "detectors" in scene:
        g[12:500, 256, 256] = fdtd.LineDetector()
        g[256, 12:500, 256] = fdtd.LineDetector()
        g[256, 256, 12:500] = fdtd.LineDetector()
    gen = torch.Generator(device=g.E.device).manual_seed(1)
    with torch.no_grad():
        for f in (g.E, g.H):
            f.uniform_(-1e-3, 1e-3, generator=gen)
        g.run(8, progress_bar=False)
        lib = g._ensure_plan()
        _lib.check(lib.fdtdhip_plan_enable_timing(g._plan, 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(steps, progress_bar=False)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
    ms, n = (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    _lib.check(lib.fdtdhip_plan_get_timing_parts(g._plan, ms, n))
    inner = (ctypes.c_int32 * 6)()
    two = lib.fdtdhip_plan_two_step_box(g._plan, 0, inner)
    cells = (inner[1] - inner[0]) * (inner[3] - inner[2]) * (inner[5] - inner[4]) if two else 0
    print(f"{scene:24s} two-step launches {n[4]:3d} x {ms[4] / max(n[4], 1):.3f} ms over {cells / 1e6:6.1f} M cells "
          f"({ms[4] / max(n[4], 1) / max(cells, 1) * 1e9:.2f} ps per cell and launch); shell {ms[3] / steps:.3f} ms per step; "
          f"one-step {n[2]} x {ms[2] / max(n[2], 1):.3f}; wall {wall:.3f} ms per step = {512 ** 3 / wall / 1e6:.1f} Gcells/s",
          flush=True)
    del g
    torch.cuda.empty_cache()
