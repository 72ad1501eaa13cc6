"""Cost of driving the steps from Python (SlabGrid) against the library's own loop (Grid),
one rank, 512^3 float32: what a multi-GPU rank pays before any exchange."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd.parallel import SlabGrid  # noqa: E402

fdtd.set_backend("hip.float32", device=0)
for cls in (fdtd.Grid, SlabGrid):
    g = cls((512, 512, 512))
    g[256, 256, 256] = fdtd.PointSource(period=20)
    kw = {} if cls is SlabGrid else {"progress_bar": False}
    g.run(10, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(100, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cls.__name__}: {(t2 - t0) * 10:.3f} ms/step, host enqueue {(t1 - t0) * 10:.3f} ms/step", flush=True)
    del g
    torch.cuda.empty_cache()
dist.destroy_process_group()
