"""Does the marching kernel's rate depend on where its four arrays lie relative to one
another?  E, H and the two second buffers are cut out of one allocation at chosen byte
offsets from a 2 MiB-aligned spacing.

    python tools/march_align.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd as fdtd  # noqa: E402

fdtd.set_backend("hip.float32")
n = 512
shape = (n, n, n, 3)
count = n * n * n * 3
span = ((count * 4 + (2 << 20) - 1) // (2 << 20) + 1) * (2 << 20)      # bytes between array starts, 2 MiB multiple
pool = torch.empty(4 * span // 4 + (64 << 20) // 4, dtype=torch.float32, device="cuda")
base = (-pool.data_ptr()) % (2 << 20) // 4


def view(slot, off_bytes):
    start = base + slot * span // 4 + off_bytes // 4
    return pool[start:start + count].view(shape)


for offs in ((0, 0, 0, 0), (0, 4096, 8192, 12288), (0, 1 << 20, 0, 1 << 20), (0, 6144, 12288, 18432),
             (0, 256, 512, 768), (0, 65536, 131072, 196608), (0, 1536, 3072, 4608), (0, 3 << 20, 6 << 20, 9 << 20),
             (0, 768 * 1024, 1536 * 1024, 2304 * 1024)):
    g = fdtd.Grid((n, n, n))
    g[n // 2, n // 2, n // 2] = fdtd.PointSource(period=20)
    E, H, E1, H1 = (view(s, o) for s, o in enumerate(offs))
    gen = torch.Generator(device="cuda").manual_seed(1)
    E.uniform_(-1e-3, 1e-3, generator=gen)
    H.uniform_(-1e-3, 1e-3, generator=gen)
    g.E, g.H = E, H
    g._step_buffers = (E1, H1)
    g.run(10, progress_bar=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(100, progress_bar=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"offsets {offs}: {ms:.4f} ms/step  (E at {E.data_ptr() % (1 << 30):#x}, H {H.data_ptr() % (1 << 30):#x}, "
          f"E1 {E1.data_ptr() % (1 << 30):#x}, H1 {H1.data_ptr() % (1 << 30):#x})", flush=True)
    del g
