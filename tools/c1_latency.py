"""Step latency of the README example (launch-bound): microseconds per step over repeated runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import fdtd_amd as fdtd  # noqa: E402

grid, cells, es, dt, desc = bench.build_workload(fdtd, "c1")
grid.run(3000, progress_bar=False)
torch.cuda.synchronize()
out = []
for _ in range(4):
    t0 = time.perf_counter()
    grid.run(3000, progress_bar=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append(((t2 - t0) / 3000 * 1e6, (t1 - t0) / 3000 * 1e6))
print(desc, "FDTDHIP_FUSED=%s" % os.environ.get("FDTDHIP_FUSED", "auto"),
      "us/step (total, host enqueue):", ["%.1f/%.1f" % o for o in out])
