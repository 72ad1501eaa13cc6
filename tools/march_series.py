"""Step time of the headline grid over consecutive runs of the same grid (does the rate hold?).

    python tools/march_series.py [runs] [steps per run] [workload]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd import _lib  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
workload = sys.argv[3] if len(sys.argv) > 3 else "m3v"
lib = _lib.load()
grid, cells, es, dt, desc = bench.build_workload(fdtd, workload)
with torch.no_grad():
    grid.run(10, progress_bar=False)
torch.cuda.synchronize()
for n in range(runs):
    elapsed, timing = bench.timed_run(grid, lib, _lib, steps, torch.cuda.synchronize)
    emax = float(grid.E.abs().max())
    print(f"run {n}: {elapsed / steps * 1e3:.4f} ms/step, kernel {timing['ms'][2] / max(timing['whole_steps'], 1):.4f} ms, "
          f"max|E| {emax:.3e}", flush=True)
