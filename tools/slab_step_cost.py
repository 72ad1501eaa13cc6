"""Step time of ONE interior slab of the 8-GPU configuration on one GPU, with its plane
exchanges: 256 x 1024 x 1024 float32, PMLs on the y and z faces, both x faces open.  The one
rank is its own neighbour on both sides (a one-rank RCCL group sending the planes to
itself), so the exchanges cost what RCCL's send/recv kernels and their stream
synchronisation cost on one device -- no xGMI, but every launch, wait and copy of the real
schedule.  FDTDHIP_FUSED=0 in the environment: two half-step launches per step (with the
interior / boundary-plane split); FDTDHIP_MARCH2=0: whole steps in stages; default: two steps per
pass in stages where the plan allows.

    python tools/slab_step_cost.py [steps] [workload: m4 | m4end (the first slab: a PML on its
                                                     low x face, one neighbour) | m3v | small]
                                   [adjoint [recorded steps]]

`adjoint`: then a recorded run (a learnable 32^3 box in the middle of the slab) and its reverse
sweep, with the plane exchanges of the real schedule: milliseconds per reverse step and a
checksum of the gradients' bit patterns.  FDTDHIP_ADJ_MARCH=0: the two-pass sweep, one plane
exchanged per half step between its passes; default: whole reverse steps in stages.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29549")
torch.cuda.set_device(0)
# (RCCL's stream with high priority, as bench.py asks for it: a plane's send/recv kernel should not
# queue behind the kernel it is meant to overlap)
_opts = dist.ProcessGroupNCCL.Options()
_opts.is_high_priority_stream = os.environ.get("SLAB_PLAIN_PRIORITY") is None
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=_opts)
import fdtd_amd as fdtd  # noqa: E402
from fdtd_amd.backend import backend as bd  # noqa: E402
from fdtd_amd.parallel import SlabGrid  # noqa: E402


class InteriorSlab(SlabGrid):
    """A slab that believes it is rank 1 of 3 and exchanges planes with itself."""

    def __init__(self, shape, first=False):
        super().__init__(shape)
        self._rank, self._world = (0 if first else 1), 3
        plane = (self.Ny, self.Nz, 2)
        self._H_below, self._E_above = (None if first else bd.zeros(plane)), bd.zeros(plane)
        self._halo = (self._H_below, self._E_above)
        self._send_H, self._send_E = bd.zeros(plane), (None if first else bd.zeros(plane))

    def _start_exchange(self, send, dst, recv, src):
        # (the first slab of a chain sends H up and receives E from above: with itself as the
        # only peer both directions have to exist, so it sends into a spare plane)
        if self._rank == 0 and (send is None) != (recv is None):
            if recv is None:
                recv = self._spare = getattr(self, "_spare", None) if getattr(self, "_spare", None) is not None else bd.zeros(send.shape)
            else:
                send = self._spare = getattr(self, "_spare", None) if getattr(self, "_spare", None) is not None else bd.zeros(recv.shape)
        return super()._start_exchange(send, dst, recv, src)

    def _global_rank(self, r):
        return 0

    def _whole_steps_agreed(self, eng, n):
        if not (n >= 2 and getattr(self, "_step_buffers", None) is not None and eng.whole_steps_possible()):
            return 0
        return 2 if n >= 4 and eng.pairs_possible() else 1


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
workload = sys.argv[2] if len(sys.argv) > 2 else "m4"
fdtd.set_backend("hip.float32", device=0)
shape = {"m4": (256, 1024, 1024), "m4end": (256, 1024, 1024), "small": (64, 128, 512)}.get(workload, (512, 512, 512))
g = InteriorSlab(shape, first=workload == "m4end")
if workload == "small":      # the smallest slab the default policy marches, thin PMLs (tests)
    for key in ((slice(None), slice(0, 4)), (slice(None), slice(-4, None)),
                (slice(None), slice(None), slice(0, 4)), (slice(None), slice(None), slice(-4, None))):
        g[key] = fdtd.PML()
if workload == "m4end":      # the first slab of the chain: the low x face of the simulation
    g[0:10, :, :] = fdtd.PML()
if workload in ("m4", "m4end"):
    for key in ((slice(None), slice(0, 10)), (slice(None), slice(-10, None)),
                (slice(None), slice(None), slice(0, 10)), (slice(None), slice(None), slice(-10, None))):
        g[key] = fdtd.PML()
g[shape[0] // 2, shape[1] // 2, shape[2] // 2] = fdtd.PointSource(period=20)
adjoint = len(sys.argv) > 3 and sys.argv[3] == "adjoint"
if adjoint:
    m = [n // 2 for n in shape]
    g[m[0] - 16:m[0] + 16, m[1] + 20:m[1] + 52, m[2] - 16:m[2] + 16] = fdtd.LearnableAnisotropicObject(permittivity=2.5)
gen = torch.Generator(device=g.E.device).manual_seed(1)
with torch.no_grad():
    for f in (g.E, g.H):
        f.uniform_(-1e-3, 1e-3, generator=gen)
    g.run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(steps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
cells = shape[0] * shape[1] * shape[2]
# a checksum of the state's bit patterns: the same for both ways of stepping, or the schedule of
# one of them reads a ghost plane too early / sends a plane too late
bits = int(g.E.view(torch.int32).to(torch.int64).sum()) ^ (int(g.H.view(torch.int32).to(torch.int64).sum()) << 1)
print(f"state checksum {bits:#x}, max|E| {float(g.E.abs().max()):.6e}")
print(f"{workload} interior slab, FDTDHIP_FUSED={os.environ.get('FDTDHIP_FUSED', 'default')}: "
      f"{(t2 - t0) / steps * 1e3:.3f} ms/step = {cells * steps / (t2 - t0) / 1e9:.1f} Gcells/s, "
      f"host enqueue {(t1 - t0) / steps * 1e3:.3f} ms/step, whole steps taken {g.whole_steps_taken}, "
      f"of them two per pass {g.pair_steps_taken}",
      flush=True)
if adjoint:
    nb = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    for _ in range(2):      # the first pass warms up allocator and code objects
        g.objects[0].inverse_permittivity.grad = None
        E0 = g.E.detach().clone().requires_grad_(True)
        g.E = E0
        g.H = g.H.detach()
        for b in g.boundaries:      # the memory variables of the pass before belong to its graph
            if getattr(b, "_psiE", None) is not None:
                b._psiE, b._psiH = b._psiE.detach(), b._psiH.detach()
        taken0 = getattr(g, "whole_reverse_steps_taken", 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(nb)
        loss = (g.E ** 2).sum() + (g.H ** 2).sum()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g.backward(loss)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    grad = g.objects[0].inverse_permittivity.grad
    gbits = int(E0.grad.view(torch.int32).to(torch.int64).sum()) ^ (int(grad.view(torch.int32).to(torch.int64).sum()) << 1)
    print(f"gradient checksum {gbits:#x}, |dL/dE0| {float(E0.grad.norm()):.6e}, |dL/dtensor| {float(grad.norm()):.6e}")
    print(f"{workload} interior slab, FDTDHIP_ADJ_MARCH={os.environ.get('FDTDHIP_ADJ_MARCH', 'default')}: recorded forward "
          f"{(t1 - t0) / nb * 1e3:.3f} ms/step, backward {(t2 - t1) / nb * 1e3:.3f} ms/step = "
          f"{cells * nb / (t2 - t1) / 1e9:.1f} Gcells/s, whole reverse steps taken "
          f"{getattr(g, 'whole_reverse_steps_taken', 0) - taken0}", flush=True)
dist.destroy_process_group()
