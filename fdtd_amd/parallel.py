"""Multi-GPU runs: x-slab domain decomposition, one process per GPU.

The reference is single device (no counterpart).  What decomposes: along x,
``update_E`` at plane i needs Hy, Hz of plane i-1 (reference fdtd/grid.py:71,73) and
``update_H`` at plane i needs Ey, Ez of plane i+1 (fdtd/grid.py:46,48); nothing else
crosses an x face.  So every rank owns a contiguous range of x planes of all four
field arrays and, per half step, receives the two tangential components of ONE plane
from one neighbour (``Ny * Nz * 2`` values):

    before update_E : (Hy, Hz) of the last plane of rank r-1  ->  ``H_below`` of rank r
    before update_H : (Ey, Ez) of the first plane of rank r+1 ->  ``E_above`` of rank r

The sender packs the two components into a contiguous ``(Ny, Nz, 2)`` buffer (one small
copy), the receiver's ghost buffer has the same shape and the kernels read it in place of
the domain-edge mask (``fdtdhip_plan_set_halo``).

Overlap: the exchange is started first, the interior planes (which do not need the
ghost) are enqueued on the compute stream, then the stream waits for the exchange
and the one boundary plane is computed.  With the RCCL backend the transfer runs on
RCCL's own stream over xGMI while the interior kernel runs.

``SlabGrid`` takes keys in the coordinates of the WHOLE simulation
(``grid[x, y, z] = thing`` as for a plain ``Grid``, executed on every rank); each
component keeps only the part that lies in the rank's planes.
"""

import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .backend import backend as bd
from .grid import Grid


class _NativeEngine:
    """Half-step pieces of one slab executed by libfdtd_hip."""

    def __init__(self, grid):
        self.grid = grid

    def prepare(self, n=1):
        self.lib = self.grid._ensure_plan(steps=n)
        self.plan = self.grid._plan
        self.grid._prepare_sources(self.lib, n)

    def reserve_detectors(self, nE, nH):
        for det in self.grid.detectors:
            det._reserve(self.lib, self.plan, nE, nH)

    def commit_detectors(self, nE, nH):
        for det in self.grid.detectors:
            det._commit(nE, nH)

    def bulk_E(self, x0, x1):
        _lib.check(self.lib.fdtdhip_update_E_bulk(self.plan, x0, x1))

    def post_E(self, q):
        _lib.check(self.lib.fdtdhip_update_E_post(self.plan, q))

    def bulk_H(self, x0, x1):
        _lib.check(self.lib.fdtdhip_update_H_bulk(self.plan, x0, x1))

    def post_H(self, q):
        _lib.check(self.lib.fdtdhip_update_H_post(self.plan, q))

    def post_E_stage(self, q, stage):
        _lib.check(self.lib.fdtdhip_update_E_post_stage(self.plan, q, stage))

    def post_H_stage(self, q, stage):
        _lib.check(self.lib.fdtdhip_update_H_post_stage(self.plan, q, stage))

    def whole_steps_possible(self):
        """Whether this slab's plan can advance whole steps in stages (``fdtdhip_step_stage``):
        second buffers provided, nothing between the half steps that the whole-step kernel
        cannot do, and a grid large enough for it to pay."""
        return self.lib.fdtdhip_plan_whole_step_kind(self.plan, 1, None, None) == 2

    def step_stage(self, q, stage):
        _lib.check(self.lib.fdtdhip_step_stage(self.plan, q, stage))

    def pairs_possible(self):
        """Whether this slab's plan can advance two steps per pass in stages
        (``fdtdhip_pair_stage``): whole steps possible, constant materials, detectors of the cell
        kind only, objects away from the open faces (recorded runs included)."""
        return self.lib.fdtdhip_plan_two_step_box(self.plan, 1, None) == 1

    def pair_stage(self, q, stage):
        _lib.check(self.lib.fdtdhip_pair_stage(self.plan, q, stage))

    def defer_currents(self, plane):
        """Current detectors wait for ``plane`` (the H plane under this slab) instead of
        being read with the other detectors; None: read them with the others."""
        ptr = ctypes.c_void_p(plane.data_ptr()) if plane is not None else None
        for det in self.grid.detectors:
            if det._kind == _lib.DET_CURRENT:
                _lib.check(self.lib.fdtdhip_plan_set_detector_below(self.plan, det._id, ptr))

    def read_deferred_currents(self):
        _lib.check(self.lib.fdtdhip_read_deferred_currents(self.plan))


class _Exchange:
    """One in-flight neighbour exchange (send one plane, receive one plane)."""

    def __init__(self, works, staged=None):
        self.works, self.staged = works, staged

    def wait(self):
        for w in self.works:
            w.wait()  # RCCL: makes the current stream wait; gloo: blocks the host
        if self.staged is not None:
            host, device = self.staged
            device.copy_(host)


class SlabGrid(Grid):
    """One x-slab of a decomposed simulation.  ``shape`` is the shape of the whole
    simulation; planes are dealt to the ranks of ``group`` in rank order, as evenly
    as possible.  Requires an initialised ``torch.distributed`` process group
    ("nccl", i.e. RCCL, for GPUs)."""

    def __init__(self, shape, grid_spacing=155e-9, permittivity=1.0, permeability=1.0,
                 courant_number=None, group=None, engine_factory=None, planes=None):
        if not dist.is_initialized():
            raise RuntimeError("SlabGrid needs torch.distributed.init_process_group first")
        # planes: x planes of every rank, in rank order (the same list on every rank), for
        # slabs of unequal cost -- e.g. fewer planes for the two end slabs, which carry the
        # PMLs on the x faces of the simulation; None: as even as possible
        self._planes = None if planes is None else [int(n) for n in planes]
        self._group = group
        self._rank = dist.get_rank(group)
        self._world = dist.get_world_size(group)
        self._backend_name = dist.get_backend(group)
        super().__init__(shape, grid_spacing, permittivity, permeability, courant_number)
        plane = (self.Ny, self.Nz, 2)     # the two components tangential to an x face
        self._H_below = bd.zeros(plane) if self._rank > 0 else None
        self._E_above = bd.zeros(plane) if self._rank < self._world - 1 else None
        self._halo = (self._H_below, self._E_above)
        self._send_H = bd.zeros(plane) if self._rank < self._world - 1 else None
        self._send_E = bd.zeros(plane) if self._rank > 0 else None
        self._engine = (engine_factory or _NativeEngine)(self)
        self.whole_steps_taken = 0    # steps advanced by the whole-step kernel (diagnostics)
        self.pair_steps_taken = 0     # of those: advanced two per pass

    # ------------------------------------------------------------ decomposition
    def _plane_counts(self):
        """x planes of every rank, in rank order."""
        n, p = self.Nx_global, self._world
        if self._planes is not None:
            if len(self._planes) != p or sum(self._planes) != n or min(self._planes) < 1:
                raise ValueError(f"planes={self._planes} does not deal {n} x planes to {p} ranks "
                                 "(one entry per rank, each at least 1)")
            return list(self._planes)
        if n < p:
            raise ValueError(f"cannot split {n} x planes over {p} ranks")
        base, extra = divmod(n, p)
        return [base + (1 if r < extra else 0) for r in range(p)]

    def _partition(self):
        counts = self._plane_counts()
        self.Nx = counts[self._rank]
        self.x_offset = sum(counts[:self._rank])

    def _planes_of_end_slab(self, side):
        counts = self._plane_counts()
        return counts[0] if side == "low" else counts[-1]

    def _global_rank(self, r):
        return r if self._group is None else dist.get_global_rank(self._group, r)

    def _border_inv_eps_x(self, y, z):
        """Collective: every rank gets the last rank's ``inverse_permittivity[-1, y, z, 0]``."""
        shape = self.inverse_permittivity[-1, y, z, 0].shape
        last = self._world - 1
        if self._rank == last:
            buf = self.inverse_permittivity[-1, y, z, 0].clone().contiguous()
        else:
            buf = bd.zeros(tuple(shape))
        if self._backend_name != "nccl" and buf.is_cuda:
            host = buf.cpu()
            dist.broadcast(host, src=self._global_rank(last), group=self._group)
            buf.copy_(host)
        else:
            dist.broadcast(buf, src=self._global_rank(last), group=self._group)
        return buf

    @property
    def _x_wrap(self):
        """A periodic boundary along x on a grid of several slabs."""
        return self._world > 1 and any(
            getattr(b, "axis", None) == 0 and not hasattr(b, "thickness") for b in self.boundaries)

    @property
    def _currents_cross_slabs(self):
        """Whether a CurrentDetector holds a cell on the first plane of some rank's slab.
        Its loop reads H of plane x-1 (reference fdtd/detectors.py:452-459; index -1 wraps
        to the last plane of the simulation), which is the last plane of the rank below,
        as that rank leaves it at the end of the H half step.  The same answer on every rank."""
        if self._world == 1:
            return False
        counts = self._plane_counts()
        firsts = {sum(counts[:r]) for r in range(self._world)}
        return any(getattr(d, "_kind", None) == _lib.DET_CURRENT
                   and any(self._cell_index(x, self.Nx_global) in firsts for x in d.x)
                   for d in self.detectors)

    # ------------------------------------------------------------ halo exchange
    def _start_exchange(self, send, dst, recv, src):
        """Send plane ``send`` to rank ``dst`` and receive into ``recv`` from rank ``src``
        (either may be None at the ends of the chain)."""
        if send is None and recv is None:
            return None
        staged = None
        if self._backend_name != "nccl" and (send if send is not None else recv).is_cuda:
            # transports without device-memory support (gloo): stage through the host
            if send is not None:
                send = send.cpu()
            if recv is not None:
                staged = (torch.empty(recv.shape, dtype=recv.dtype), recv)
                recv = staged[0]
        ops = []
        if send is not None:
            ops.append(dist.P2POp(dist.isend, send, self._global_rank(dst), self._group))
        if recv is not None:
            ops.append(dist.P2POp(dist.irecv, recv, self._global_rank(src), self._group))
        return _Exchange(dist.batch_isend_irecv(ops), staged)

    @staticmethod
    def _tangential(plane, buf):
        """Components y and z of an x plane, packed into the send buffer."""
        if buf.dtype != plane.dtype or buf.device != plane.device:
            buf = torch.empty(plane.shape[:-1] + (2,), dtype=plane.dtype, device=plane.device)
        buf.copy_(plane.detach()[..., 1:])
        return buf

    # ----------------------------------------------------------------- stepping
    def _half_step_E(self, eng, q):
        r, p = self._rank, self._world
        ex = self._start_exchange(self._tangential(self.H[-1], self._send_H) if r < p - 1 else None,
                                  r + 1, self._H_below, r - 1)
        first = 1 if r > 0 else 0
        eng.bulk_E(first, self.Nx)       # planes that do not need the ghost plane
        if ex is not None:
            ex.wait()
        if first:
            eng.bulk_E(0, 1)             # the plane that reads H_below
        if self._x_wrap:
            # E[0] = E[-1] of the whole simulation (reference fdtd/boundaries.py:177),
            # at its place among the boundaries
            eng.post_E_stage(q, 0)
            ex = self._start_exchange(self.E[-1] if r == p - 1 else None, 0,
                                      self.E[0] if r == 0 else None, p - 1)
            if ex is not None:
                ex.wait()
            eng.post_E_stage(q, 1)
        else:
            eng.post_E(q)

    def _half_step_H(self, eng, q):
        r, p = self._rank, self._world
        cross = self._currents_cross_slabs
        if cross:
            if getattr(self, "_H_ring", None) is None:
                self._H_ring = bd.zeros((self.Ny, self.Nz, 3))
            eng.defer_currents(self._H_ring)   # before the post pieces: they leave these out
        ex = self._start_exchange(self._tangential(self.E[0], self._send_E) if r > 0 else None,
                                  r - 1, self._E_above, r + 1)
        last = self.Nx - 1 if r < p - 1 else self.Nx
        eng.bulk_H(0, last)
        if ex is not None:
            ex.wait()
        if last < self.Nx:
            eng.bulk_H(last, self.Nx)    # the plane that reads E_above
        if self._x_wrap:
            # H[-1] = H[0] of the whole simulation (reference fdtd/boundaries.py:182)
            eng.post_H_stage(q, 0)
            ex = self._start_exchange(self.H[0] if r == 0 else None, p - 1,
                                      self.H[-1] if r == p - 1 else None, 0)
            if ex is not None:
                ex.wait()
            eng.post_H_stage(q, 1)
        else:
            eng.post_H(q)
        if cross:
            ex = self._start_exchange(self.H[-1], (r + 1) % p, self._H_ring, (r - 1) % p)
            ex.wait()
            eng.read_deferred_currents()

    def update_E(self):
        if not self._half_step_recorded(True):
            self._native_half_step(None, True)

    def update_H(self):
        if not self._half_step_recorded(False):
            self._native_half_step(None, False)

    def _native_half_step(self, lib, phase_E):
        eng = self._engine
        eng.prepare()
        if phase_E:
            eng.reserve_detectors(1, 0)
            self._half_step_E(eng, self.time_steps_passed)
            eng.commit_detectors(1, 0)
            self._finish_sources(1)
        else:
            eng.reserve_detectors(0, 1)
            self._half_step_H(eng, self.time_steps_passed)
            eng.commit_detectors(0, 1)

    def _half_step_recorded(self, phase_E):
        """As on an undivided grid, and collective like ``_advance``: either every rank records
        the half step or none does."""
        from . import autograd

        if bd.staging or not self._any_rank_needs_grad():
            return False
        autograd.differentiable_half_step(self, phase_E)
        return True

    def _advance(self, n):
        if n <= 0:
            return
        from . import autograd

        if not bd.staging and self._any_rank_needs_grad():
            autograd.differentiable_advance(self, n)
            return
        if n >= 2 and not bd.staging:
            self._provide_step_buffers(slab=True)
        self._native_run(None, n)
        self.time_steps_passed += n

    def _provide_step_buffers(self, slab=True):
        super()._provide_step_buffers(slab=True)

    def _any_rank_needs_grad(self):
        """Whether this run has to be recorded for a backward pass.  Collective when gradients
        are enabled: the reverse sweep exchanges planes between neighbours, so either every
        rank records the run or none does -- also a rank that holds no cell of any learnable
        object.  Such a rank ties the run to its E field, whose adjoint the sweep needs anyway."""
        from . import autograd

        if not torch.is_grad_enabled():
            return False
        mine = autograd.needs_grad(self)
        if self._world == 1:
            return mine
        on_device = self._backend_name == "nccl"
        flag = torch.tensor([1 if mine else 0], device=bd.device if on_device else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._group)
        wanted = bool(flag.item())
        if wanted and not mine:
            self.E = self.E.detach().requires_grad_(True)
        return wanted

    def _whole_step(self, eng, q, cur, alt):
        """One step from the buffers ``cur`` = (E, H) into ``alt`` in four pieces around the two
        plane exchanges (``fdtdhip_step_stage``): while (Hy, Hz) of the rank below travel, the
        cells that do not read them are advanced -- the E half step next to PMLs and objects
        and half of the whole-step kernel; then plane 0 takes its E half step and its new
        (Ey, Ez) travel to the rank below behind the other half of the whole-step kernel; the
        last plane takes its H half step when (Ey, Ez) of the rank above have arrived."""
        r, p = self._rank, self._world
        ex = self._start_exchange(self._tangential(cur[1][-1], self._send_H) if r < p - 1 else None,
                                  r + 1, self._H_below, r - 1)
        eng.step_stage(q, 0)
        if ex is not None:
            ex.wait()
        eng.step_stage(q, 1)
        ex = self._start_exchange(self._tangential(alt[0][0], self._send_E) if r > 0 else None,
                                  r - 1, self._E_above, r + 1)
        eng.step_stage(q, 2)
        if ex is not None:
            ex.wait()
        eng.step_stage(q, 3)

    def _whole_steps_agreed(self, eng, n):
        """Collective: how the coming ``n`` steps are taken -- 0: half steps, 1: whole steps in
        stages, 2: two steps per pass in stages where four or more remain, whole steps
        otherwise.  The ways of stepping exchange planes at different moments, so every rank has
        to take the same one: the fastest that every rank's plan can."""
        if n < 2 or not hasattr(eng, "whole_steps_possible"):
            return 0
        mine = 0
        if (getattr(self, "_step_buffers", None) is not None and not self._x_wrap
                and not self._currents_cross_slabs and eng.whole_steps_possible()):
            mine = 2 if n >= 4 and hasattr(eng, "pairs_possible") and eng.pairs_possible() else 1
        if self._world == 1:
            return mine
        on_device = self._backend_name == "nccl"
        flag = torch.tensor([mine], device=bd.device if on_device else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
        return int(flag.item())

    def _pair_of_steps(self, eng, q, cur, alt):
        """Steps ``q`` and ``q + 1`` from the buffers ``cur`` = (E, H) into ``alt`` in eight pieces
        around the four plane exchanges (``fdtdhip_pair_stage``).  The two-step kernel takes the
        cells a few planes away from the slab's open faces (and from PMLs) in one pass
        and needs no neighbour's plane; the others take the two steps one after the other, the
        first in place in ``cur``, each plane exchange hidden behind the launches that do not
        wait for it."""
        r, p = self._rank, self._world
        up = lambda plane: self._tangential(plane, self._send_H) if r < p - 1 else None      # noqa: E731
        down = lambda plane: self._tangential(plane, self._send_E) if r > 0 else None        # noqa: E731
        ex = self._start_exchange(up(cur[1][-1]), r + 1, self._H_below, r - 1)
        eng.pair_stage(q, 0)
        if ex is not None:
            ex.wait()
        eng.pair_stage(q, 1)
        ex = self._start_exchange(down(cur[0][0]), r - 1, self._E_above, r + 1)
        eng.pair_stage(q, 2)
        if ex is not None:
            ex.wait()
        eng.pair_stage(q, 3)
        ex = self._start_exchange(up(cur[1][-1]), r + 1, self._H_below, r - 1)
        eng.pair_stage(q, 4)
        if ex is not None:
            ex.wait()
        eng.pair_stage(q, 5)
        ex = self._start_exchange(down(alt[0][0]), r - 1, self._E_above, r + 1)
        eng.pair_stage(q, 6)
        if ex is not None:
            ex.wait()
        eng.pair_stage(q, 7)

    def _native_run(self, lib, n):
        eng = self._engine
        eng.prepare(n)
        eng.reserve_detectors(n, n)
        s = 0
        # (recorded runs too: the E half step of the box around the objects, which the shell
        # launches take, writes the curl traces the reverse sweep reads)
        how = self._whole_steps_agreed(eng, n)
        if how:
            # pairs of whole steps, first -> second buffers and back: the state ends in the
            # tensors the user holds; an odd last step runs as two half steps
            cur = (self.E.detach(), self.H.detach())
            alt = self._step_buffers
            while how == 2 and s + 3 < n:      # two steps per pass, in twos
                self._pair_of_steps(eng, self.time_steps_passed + s, cur, alt)
                self._pair_of_steps(eng, self.time_steps_passed + s + 2, alt, cur)
                s += 4
                self.pair_steps_taken += 4
            while s + 1 < n:
                self._whole_step(eng, self.time_steps_passed + s, cur, alt)
                self._whole_step(eng, self.time_steps_passed + s + 1, alt, cur)
                s += 2
            self.whole_steps_taken += s
        for s in range(s, n):
            q = self.time_steps_passed + s
            self._half_step_E(eng, q)
            self._half_step_H(eng, q)
        eng.commit_detectors(n, n)
        self._finish_sources(n)

    # ----------------------------------------------------------------- adjoint
    def _native_adjoint(self, lib, n, lamE, lamH, work, half=None):
        """Reverse sweep on a slab.  The forward x differences across a slab face read the
        neighbour's plane, so their adjoints flow back across it: undoing the H half step,
        plane 0 gathers from the LAST plane of difference adjoints of the rank below;
        undoing the E half step, the last plane gathers from the FIRST plane of the rank
        above.  One packed ``(Ny, Nz, 2)`` plane per half step and neighbour, as in the forward
        pass."""
        r, p, plan = self._rank, self._world, self._plan
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        planes = work.view(self.Nx, self.Ny, self.Nz, 6)
        # what crosses a face: the adjoints of the two differences along x, entries 0 and 1 of a
        # cell's six -- a packed (Ny, Nz, 2) plane, the size of the forward halo
        from_below = bd.zeros((self.Ny, self.Nz, 2)) if r > 0 else None
        from_above = bd.zeros((self.Ny, self.Nz, 2)) if r < p - 1 else None
        send = bd.zeros((self.Ny, self.Nz, 2)) if p > 1 else None

        def across(plane):
            send.copy_(plane[..., :2])
            return send

        wrap = self._x_wrap
        wrapped = bd.zeros((self.Ny, self.Nz, 3)) if wrap and r in (0, p - 1) else None

        def undo_wrap(phase, s, lam):
            """The post pieces of a half step around the transpose of the periodic copy along x
            (reference fdtd/boundaries.py:173-184: E[0] = E[-1] after the E half step, H[-1] =
            H[0] after the H half step, first and last plane of the whole simulation): the
            adjoint of the copied plane is added to the adjoint of the plane it was copied
            from, on the other end rank, and cleared."""
            _lib.check(lib.fdtdhip_adjoint_post_stage(plan, phase, s, 1, ptr(lamE), ptr(lamH)))
            dst_rank, src_rank = (0, p - 1) if phase else (p - 1, 0)      # of the forward copy
            dst_plane = lam[0] if phase else lam[-1]
            src_plane = lam[-1] if phase else lam[0]
            ex = self._start_exchange(dst_plane if r == dst_rank else None, src_rank,
                                      wrapped if r == src_rank else None, dst_rank)
            if ex is not None:
                ex.wait()
            if r == src_rank:
                src_plane += wrapped
            if r == dst_rank:
                dst_plane.zero_()
            _lib.check(lib.fdtdhip_adjoint_post_stage(plan, phase, s, 0, ptr(lamE), ptr(lamH)))

        def undo_H(s):
            if wrap:
                undo_wrap(0, s, lamH)
            _lib.check(lib.fdtdhip_adjoint_cells(plan, 0, s, ptr(lamE), ptr(lamH), ptr(work)))
            ex = self._start_exchange(across(planes[-1]) if r < p - 1 else None, r + 1, from_below, r - 1)
            if ex is not None:
                ex.wait()
            _lib.check(lib.fdtdhip_adjoint_gather(plan, 0, ptr(lamE), ptr(lamH), ptr(work),
                                                  ptr(from_below)))

        # a source fed back from its loop current on the first plane of a slab: the loop reads H
        # of plane x - 1 (reference fdtd/detectors.py:452-459), another rank's last plane, and the
        # adjoint of Ez at the cell flows back there -- collected in a plane of its own, passed
        # down the ring (as the forward pass passes the H plane up) and added on arrival
        soft = [src for src in self.sources if hasattr(src, "current_detector") and src.impedance > 0]
        counts = self._plane_counts()
        firsts = {sum(counts[:q]) for q in range(p)}
        soft_cross = p > 1 and any(self._cell_index(src.x, self.Nx_global) in firsts for src in soft)
        below = ring_in = None
        if soft_cross:
            below, ring_in = bd.zeros((self.Ny, self.Nz, 3)), bd.zeros((self.Ny, self.Nz, 3))
            for src in soft:
                if src._native_id is not None and self._local_x(self._cell_index(src.x, self.Nx_global)) == 0:
                    _lib.check(lib.fdtdhip_plan_set_soft_source_adjoint_below(plan, src._native_id, ptr(below)))

        def undo_E(s):
            if wrap:
                undo_wrap(1, s, lamE)
            _lib.check(lib.fdtdhip_adjoint_cells(plan, 1, s, ptr(lamE), ptr(lamH), ptr(work)))
            if soft_cross:
                ex = self._start_exchange(below, (r - 1) % p, ring_in, (r + 1) % p)
                ex.wait()
                lamH[-1] += ring_in
                below.zero_()
            ex = self._start_exchange(across(planes[0]) if r > 0 else None, r - 1, from_above, r + 1)
            if ex is not None:
                ex.wait()
            _lib.check(lib.fdtdhip_adjoint_gather(plan, 1, ptr(lamE), ptr(lamH), ptr(work),
                                                  ptr(from_above)))

        try:
            self._reverse_sweep(lib, n, lamE, lamH, work, half, undo_E, undo_H, send, from_below, from_above)
        finally:
            if soft_cross:
                for src in soft:
                    if src._native_id is not None:
                        _lib.check(lib.fdtdhip_plan_set_soft_source_adjoint_below(plan, src._native_id, None))

    def _reverse_sweep(self, lib, n, lamE, lamH, work, half, undo_E, undo_H, send, from_below, from_above):
        if half is not None:      # the adjoint of one hand-driven half step
            (undo_E if half == "E" else undo_H)(0)
            return
        s = n - 1
        alt = self._whole_reverse_steps_agreed(lib, n, lamE, lamH)
        if alt is not None:
            # pairs of whole reverse steps in stages, first -> second adjoint arrays and back (an
            # odd first one in the two-pass form): the mirror image of _whole_step
            try:
                if n % 2:
                    undo_H(s)
                    undo_E(s)
                    s -= 1
                first, second = (lamE, lamH), alt[:2]
                while s >= 1:
                    self._whole_reverse_step(lib, s, 1, first, second, work, send, from_below, from_above)
                    self._whole_reverse_step(lib, s - 1, 2, second, first, work, send, from_below, from_above)
                    s -= 2
                self.whole_reverse_steps_taken = getattr(self, "whole_reverse_steps_taken", 0) + n - n % 2
            finally:
                self._withdraw_adjoint_buffers(lib)
            return
        for s in range(n - 1, -1, -1):
            undo_H(s)
            undo_E(s)

    def _whole_reverse_steps_agreed(self, lib, n, lamE, lamH):
        """Collective: second adjoint arrays if every rank's plan can undo whole steps in stages
        (``fdtdhip_adjoint_step_stage``), else None.  The two ways of sweeping exchange planes
        at different moments, so all ranks take the same one."""
        mine, alt = 0, None
        big = self._adjoint_march_forced() or self.Nx * self.Ny * self.Nz >= 1 << 22
        if n >= 2 and self._adjoint_second_buffers and big and not self._x_wrap:
            ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            pmls = [b for b in self.boundaries if hasattr(b, "thickness") and b._active]
            alt = (torch.empty_like(lamE), torch.empty_like(lamH),
                   [(torch.empty_like(b._psiE.detach()), torch.empty_like(b._psiH.detach())) for b in pmls])
            _lib.check(lib.fdtdhip_plan_set_adjoint_buffers(self._plan, ptr(alt[0]), ptr(alt[1])))
            for b, (ae, ah) in zip(pmls, alt[2]):
                _lib.check(lib.fdtdhip_plan_set_pml_adjoint_buffers(self._plan, b._native_id, ptr(ae), ptr(ah)))
            mine = _lib.check(lib.fdtdhip_plan_adjoint_whole_steps(self._plan))
        if self._world > 1:
            on_device = self._backend_name == "nccl"
            flag = torch.tensor([mine], device=bd.device if on_device else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
            mine = int(flag.item())
        if not mine:
            if alt is not None:
                self._withdraw_adjoint_buffers(lib)
            return None
        return alt

    def _withdraw_adjoint_buffers(self, lib):
        _lib.check(lib.fdtdhip_plan_set_adjoint_buffers(self._plan, None, None))
        for b in self.boundaries:
            if hasattr(b, "thickness") and b._active:
                _lib.check(lib.fdtdhip_plan_set_pml_adjoint_buffers(self._plan, b._native_id, None, None))

    def _whole_reverse_step(self, lib, s, flip, cur, new, work, send, from_below, from_above):
        """One reverse step from the adjoint arrays ``cur`` = (lamE, lamH) into ``new`` in four
        pieces around its two plane exchanges.  Undoing the H half step, plane 0 gathers from
        the last plane of the rank below: every rank packs the two difference adjoints along x
        of its last plane (they depend on lamH as it stands) and sends them upwards first; the
        cells that do not wait for them are undone meanwhile, with half of the marching kernel.
        Undoing the E half step, the last plane gathers from plane 0 of the rank above, whose
        new lamE exists once its stage 1 has run: packed, sent downwards behind the other half
        of the marching kernel."""
        r, p, plan = self._rank, self._world, self._plan
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        args = (flip, ptr(cur[0]), ptr(cur[1]), ptr(new[0]), ptr(new[1]), ptr(work))
        if r < p - 1:
            _lib.check(lib.fdtdhip_adjoint_pack_plane(plan, 0, s, flip, ptr(cur[0]), ptr(cur[1]), ptr(send)))
        ex = self._start_exchange(send if r < p - 1 else None, r + 1, from_below, r - 1)
        _lib.check(lib.fdtdhip_adjoint_step_stage(plan, s, 0, *args, None))
        if ex is not None:
            ex.wait()
        _lib.check(lib.fdtdhip_adjoint_step_stage(plan, s, 1, *args, ptr(from_below)))
        if r > 0:
            _lib.check(lib.fdtdhip_adjoint_pack_plane(plan, 1, s, flip, ptr(new[0]), ptr(cur[1]), ptr(send)))
        ex = self._start_exchange(send if r > 0 else None, r - 1, from_above, r + 1)
        _lib.check(lib.fdtdhip_adjoint_step_stage(plan, s, 2, *args, None))
        if ex is not None:
            ex.wait()
        _lib.check(lib.fdtdhip_adjoint_step_stage(plan, s, 3, *args, ptr(from_above)))

    def backward(self, loss):
        """Back-propagate this rank's part of the loss (the total loss is the sum over
        ranks).  Collective: the reverse sweep exchanges planes with the neighbours, so
        every rank calls it, also one whose part of the loss does not depend on the
        fields -- the zero-weight term below ties every rank's loss to its slab."""
        anchor = 0.0 * (self.E.sum() + self.H.sum())
        (loss + anchor).backward()

    def run(self, total_time, progress_bar: bool = False):
        if isinstance(total_time, float):
            total_time /= self.time_step
        self._advance(int(total_time))

    # ------------------------------------------------------------------ output
    def gather(self, name="E"):
        """The whole-simulation array ``name`` ("E", "H", ...) as a numpy array on every
        rank (host copies; meant for checks and small outputs, not for the hot path)."""
        mine = bd.numpy(getattr(self, name))
        parts = [None] * self._world
        dist.all_gather_object(parts, mine, group=self._group)
        return np.concatenate(parts, axis=0)

    def __repr__(self):
        return (f"{self.__class__.__name__}(shape=({self.Nx_global},{self.Ny},{self.Nz}), "
                f"planes={self.x_offset}:{self.x_offset + self.Nx} on rank {self._rank}/{self._world}, "
                f"grid_spacing={self.grid_spacing:.2e}, courant_number={self.courant_number:.2f})")
