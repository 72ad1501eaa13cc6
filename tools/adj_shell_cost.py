"""What the cells next to PML faces cost the reverse sweep, by orientation of the faces.

    rocprofv3 --kernel-trace --stats ... -- python3 tools/adj_shell_cost.py <axes> [n] [obj]

512^3 float32 (or n^3), ten-cell PML faces normal to the axes named in <axes> ("x", "yz",
"xyz", "-" for none), `obj`: a 64^3 learnable box in the middle; 6 recorded steps and their
reverse sweep.  Read the kernel times with tools/kstats.py.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402

axes = sys.argv[1] if len(sys.argv) > 1 else "xyz"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fdtd.set_backend("hip.float32")
g = fdtd.Grid((n, n, n))
for axis, name in enumerate("xyz"):
    if name in axes:
        lo, hi = [slice(None)] * 3, [slice(None)] * 3
        lo[axis], hi[axis] = slice(0, 10), slice(-10, None)
        g[tuple(lo)] = fdtd.PML()
        g[tuple(hi)] = fdtd.PML()
g[n // 2, n // 3, n // 2] = fdtd.PointSource(period=20)
if "obj" in sys.argv:
    m = n // 2
    g[m - 32:m + 32, m - 32:m + 32, m - 32:m + 32] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
g.E = torch.randn_like(g.E).mul_(1e-2).requires_grad_(True)
g.run(6, progress_bar=False)
loss = (g.E ** 2).sum() + (g.H ** 2).sum()
loss.backward()
torch.cuda.synchronize()
print("done", axes, n)
