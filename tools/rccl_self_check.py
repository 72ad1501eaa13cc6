"""One-GPU check of the RCCL point-to-point path SlabGrid uses: a one-rank "nccl"
group, a plane sent to the same rank with batch_isend_irecv while a bulk kernel runs
on the compute stream, then the stream-side wait.  Run as a plain script."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options()
opts.is_high_priority_stream = True     # as bench.py initialises RCCL
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                        pg_options=opts)
import fdtd_amd as fdtd
from fdtd_amd.parallel import SlabGrid

fdtd.set_backend("hip.float32", device=0)
shape = (16, 70, 64)
g, ref = SlabGrid(shape), fdtd.Grid(shape)
for grid in (g, ref):
    grid[8, 35, 32] = fdtd.PointSource(period=9)
g.run(6)
ref.run(6, progress_bar=False)
torch.cuda.synchronize()
assert float(g.H.abs().max()) > 0 and torch.equal(g.E, ref.E) and torch.equal(g.H, ref.H)
print("one-rank SlabGrid equals Grid", flush=True)
recv = torch.zeros_like(g.H[-1])
ex = g._start_exchange(g.H[-1], 0, recv, 0)
g._engine.prepare()
g._engine.bulk_E(1, g.Nx)
ex.wait()
g._engine.bulk_E(0, 1)
torch.cuda.synchronize()
assert torch.equal(recv, g.H[-1])
print("RCCL self send/recv of a plane: ok", flush=True)
dist.destroy_process_group()
