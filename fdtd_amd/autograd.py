"""Differentiable time stepping: ``grid.run(n)`` as one autograd node.

The reference obtains gradients by letting torch autograd record every slice
operation of every step (reference examples/learnable_focus.py:98-128 through
fdtd/objects.py:315-349).  Here a run of ``n`` steps is a single
``torch.autograd.Function``:

* forward  -- the usual native run on fresh copies of the state (fields and PML
  memory variables), with the E half step recording, per step, the curl of H
  inside every learnable anisotropic object (the only forward quantity the
  gradient of their tensors needs);
* backward -- ``fdtdhip_adjoint_run``: hand-written reverse-time kernels
  (``csrc/yee_adjoint.hpp``) propagate dL/dE, dL/dH and dL/dpsi back through the
  n steps and accumulate dL/d(inverse_permittivity) of the learnable objects.

Differentiable inputs: ``grid.E``, ``grid.H``, the PML memory variables (so that
consecutive runs chain) and ``obj.inverse_permittivity`` of anisotropic objects
that require grad.  Material arrays of the grid and sources are constants; the feedback of
a ``SoftArbitraryPointSource`` (its voltage depends on H through its loop current) is part
of the adjoint.

Half steps driven by hand (``grid.update_E()`` / ``grid.update_H()``, as reference
tests/test_detectors.py:62-70 does, possibly with writes into the fields in between) are
recorded the same way, one node per half step (the reference records them like any other
operation: fdtd/grid.py:275-325 under fdtd/backend.py:57-59); its backward is the adjoint
of that half step alone (``fdtdhip_adjoint_cells`` + ``fdtdhip_adjoint_gather``).

On x-slabs (``fdtd_amd.parallel.SlabGrid``) every rank records the same chain of
nodes for its planes (whole runs or hand-driven half steps); the reverse sweep exchanges one plane of difference adjoints
with a neighbour per half step, so ``backward`` is a collective: every rank must
reach it (``SlabGrid.backward(loss)`` makes sure of that).
"""

import ctypes

import torch

from . import _lib
from .backend import backend as bd


def _learnable(grid):
    return [o for o in grid.objects
            if o._native_kind in (_lib.OBJ_ANISOTROPIC, _lib.OBJ_ANISOTROPIC_MODULATED) and o.Nx > 0
            and torch.is_tensor(o.inverse_permittivity) and o.inverse_permittivity.requires_grad]


def _modulated(objs):
    """Those of ``objs`` whose tensor is scaled by the energy-dependent modifier."""
    return [o for o in objs if o._native_kind == _lib.OBJ_ANISOTROPIC_MODULATED]


def _modulated_sources(grid):
    """Seeded energy-modulated sources with cells on this rank."""
    return [s for s in grid.sources
            if hasattr(s, "_torch_inputs") and s.cc_dirs is not None and s._native_id is not None]


def _active_pmls(grid):
    return [b for b in grid.boundaries if hasattr(b, "thickness") and b._active]


def needs_grad(grid):
    """True when a run must be recorded for a later backward pass."""
    if not torch.is_grad_enabled():
        return False
    if grid.E.requires_grad or grid.H.requires_grad or _learnable(grid):
        return True
    if any(hasattr(s, "_torch_inputs") and s.cc_dirs is not None for s in grid.sources):
        return True
    return any(p._psiE.requires_grad or p._psiH.requires_grad for p in _active_pmls(grid))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _Run(torch.autograd.Function):
    """``n`` full steps (n > 0), or one half step: n = -1 the E half step, n = -2 the H one."""

    @staticmethod
    def forward(ctx, grid, n, E, H, *rest):
        half, n = (None, n) if n > 0 else ("E" if n == -1 else "H", 1)
        ctx.half = half
        objs, pmls, mods, srcs = grid._tape_lists   # (differentiable_advance made them)
        # object tensors, (weight, bias) per modulated object, (P, osc, weight, bias) per source
        k = len(objs) + 2 * len(mods) + 4 * len(srcs)
        src_in = rest[len(objs) + 2 * len(mods):k]
        # the run works on fresh state tensors; the inputs keep their values for autograd.
        # A single step can be taken out of place instead of copying both fields first
        # (fdtdhip_step_into: one launch from the input tensors into new ones) where the plan
        # has a whole-step kernel for it -- the case of a run advanced one step per node
        # (reference cortical/autoencoder.py:108-112).
        psis = [t.detach().clone() for t in rest[k:]]
        for n_p, p in enumerate(pmls):
            p._psiE, p._psiH = psis[2 * n_p], psis[2 * n_p + 1]
        into = None
        if half is None and n == 1 and grid._single_steps_out_of_place():
            # (the caller validated E and H: their detached aliases need no second look)
            grid._bind_state(E.detach(), H.detach())
            into = grid._step_buffers = (torch.empty_like(grid.E), torch.empty_like(grid.H))
            for p in pmls:
                if getattr(p, "_psiE_alt", None) is None:
                    p._psiE_alt = torch.empty_like(p._psiE)
        if into is None:
            grid.E, grid.H = E.detach().clone(), H.detach().clone()
        lib = grid._ensure_plan(steps=n)   # binds the state tensors
        plan = grid._plan
        if into is not None and _lib.check(lib.fdtdhip_plan_whole_step_kind(plan, 0, None, None)) == 0:
            # no whole-step kernel for this plan after all: in place on copies
            into = grid._step_buffers = None
            grid.E, grid.H = grid.E.clone(), grid.H.clone()
            lib = grid._ensure_plan(steps=n)
        traces, etraces = [], []
        for o in objs:
            tr = torch.empty((n, o.Nx * o.Ny * o.Nz, 3), dtype=bd.float, device=bd.device)   # every row is written
            _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, _ptr(tr), n))
            traces.append(tr)
        for o in mods:
            et = bd.zeros((n, o.Nx * o.Ny * o.Nz, 2))
            o._mod_buffers = (et, None)
            o._native_refresh(lib, plan)
            etraces.append(et)
        q0, src_state = grid.time_steps_passed, []
        for n_s, src in enumerate(srcs):
            P, osc = src_in[4 * n_s].detach().contiguous(), src_in[4 * n_s + 1].detach().contiguous()
            box = src._local_box()
            cells = (box[1] - box[0]) * (box[3] - box[2]) * (box[5] - box[4])
            state = [P, osc, src._packed_weights(), bd.zeros((n, cells, 2))]
            src._recording = (*state, None, None, None, q0)
            src_state.append(state)
        try:
            if into is not None:
                kind = grid._native_run(lib, 1, into=True)
                # the roles are exchanged: the new tensors hold the state
                grid._bind_state(*into)
                grid._step_buffers = None      # (the inputs: not ours to write to later)
                if kind == 1:                  # psi_E went into the PMLs' second buffers
                    for n_p, p in enumerate(pmls):
                        p._psiE, p._psiE_alt = p._psiE_alt, None
                        psis[2 * n_p] = p._psiE
            elif half is None:
                grid._native_run(lib, n)
            else:
                grid._native_half_step(lib, half == "E")
        finally:
            for src in srcs:
                src._recording = None
            for o in objs:
                _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, None, 0))
            for o in mods:
                o._mod_buffers = (None, None)
                o._native_refresh(lib, plan)
        ctx.grid, ctx.n, ctx.k, ctx.traces, ctx.etraces = grid, n, k, traces, etraces
        ctx.mods = mods
        ctx.srcs, ctx.src_state, ctx.q0 = srcs, src_state, q0
        # weights as they were during this run (an optimiser may change them before backward)
        ctx.mod_weights = [o._packed_weights().clone() for o in mods]
        ctx.objs, ctx.pmls = objs, pmls
        ctx.save_for_backward(*rest[:len(objs)])
        return (grid.E, grid.H, *psis)

    @staticmethod
    def backward(ctx, gE, gH, *gpsi):
        grid, n, objs, pmls, mods = ctx.grid, ctx.n, ctx.objs, ctx.pmls, ctx.mods
        params = ctx.saved_tensors[:len(objs)]
        full = (grid.Nx, grid.Ny, grid.Nz, 3)

        def start(g, shape):
            if g is None:
                return bd.zeros(shape)
            if g.dtype == bd.float:
                return g.detach().clone(memory_format=torch.contiguous_format)
            return g.detach().to(dtype=bd.float).contiguous().clone()

        lamE, lamH = start(gE, full), start(gH, full)
        lam_psi = [start(g, (2,) + p.shape) for g, p in
                   zip(gpsi, [q for p in pmls for q in (p, p)])]
        grads = [torch.zeros_like(p.detach()) for p in params]
        # (every entry of the work array that a reverse step reads it has written before: the
        # array of the node before this one serves as it is)
        work = getattr(grid, "_adjoint_work", None)
        if work is None or work.dtype != bd.float or work.device != bd.device \
                or work.shape[0] != grid.Nx * grid.Ny * grid.Nz:
            work = grid._adjoint_work = bd.zeros((grid.Nx * grid.Ny * grid.Nz, 6))
        lib = grid._ensure_plan()
        plan = grid._plan
        for o, tr, gr in zip(objs, ctx.traces, grads):
            _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, _ptr(tr), n))
            _lib.check(lib.fdtdhip_plan_set_object_grad(plan, o._native_id, _ptr(gr)))
        for n_p, p in enumerate(pmls):
            _lib.check(lib.fdtdhip_plan_set_pml_adjoint(
                plan, p._native_id, _ptr(lam_psi[2 * n_p]), _ptr(lam_psi[2 * n_p + 1])))
        src_grads = []
        for src, (P, osc, wb, trace) in zip(ctx.srcs, ctx.src_state):
            gP, gosc, gwb = torch.zeros_like(P), torch.zeros_like(osc), torch.zeros_like(wb)
            src._recording = (P, osc, wb, trace, gP, gosc, gwb, ctx.q0)
            src._native_prepare(lib, plan, ctx.q0, n)
            src_grads.append((gP, gosc, gwb))
        for src in grid.sources:   # sources fed back from their loop current: the run's first time step
            if hasattr(src, "current_detector") and getattr(src, "_native_id", None) is not None:
                _lib.check(lib.fdtdhip_plan_set_soft_source_adjoint(plan, src._native_id, ctx.q0))
        gmods = []
        for o, et, wb in zip(mods, ctx.etraces, ctx.mod_weights):
            gm = bd.zeros((o.Nx * o.Ny * o.Nz, 9, 3))
            _lib.check(lib.fdtdhip_plan_set_object_modifier(plan, o._native_id, _ptr(wb), _ptr(et),
                                                            _ptr(gm)))
            gmods.append(gm)
        try:
            grid._native_adjoint(lib, n, lamE, lamH, work, ctx.half)
        finally:
            for o in objs:
                _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, None, 0))
                _lib.check(lib.fdtdhip_plan_set_object_grad(plan, o._native_id, None))
            for o in mods:
                o._weights_key = None
                o._native_refresh(lib, plan)
            for src in ctx.srcs:
                src._recording = None
            for p in pmls:
                _lib.check(lib.fdtdhip_plan_set_pml_adjoint(plan, p._native_id, None, None))
        wgrads = []
        for o, gm in zip(mods, gmods):
            total = gm.sum(0)                       # (9, 3): d/dw_E, d/dw_H, d/dbias per entry
            w, b = o.nonlin_conv.weight, o.nonlin_conv.bias
            wgrads.append(total[:, :2].reshape(w.shape).to(w.dtype))
            wgrads.append(total[:, 2].reshape(b.shape).to(b.dtype))
        sgrads = []
        for src, (gP, gosc, gwb) in zip(ctx.srcs, src_grads):
            w, b = src.nonlin_conv.weight, src.nonlin_conv.bias
            sgrads += [gP, gosc, gwb[:, :2].reshape(w.shape).to(w.dtype),
                       gwb[:, 2].reshape(b.shape).to(b.dtype)]
        return (None, None, lamE, lamH, *grads, *wgrads, *sgrads, *lam_psi)


def differentiable_half_step(grid, phase_E):
    """One half step recorded as an autograd node (the step counter is the caller's)."""
    differentiable_advance(grid, -1 if phase_E else -2)


def differentiable_advance(grid, n):
    """``n`` full steps (or, n < 0, one half step) recorded as one autograd node; re-binds
    the grid's state tensors to the node's outputs."""
    for src in grid.sources:
        if getattr(src, "_no_autograd", None):
            raise NotImplementedError(f"differentiable runs do not cover {src._no_autograd}")
    objs, pmls = _learnable(grid), _active_pmls(grid)
    grid._fields()       # (E and H validated here; the node works on aliases of them)
    if n >= 2:
        grid._provide_step_buffers()   # small grids: one launch per recorded step as well
    inputs = [o.inverse_permittivity for o in objs]
    for o in _modulated(objs):
        inputs += [o.nonlin_conv.weight, o.nonlin_conv.bias]
    if any(hasattr(s, "_torch_inputs") for s in grid.sources):
        grid._ensure_plan()  # decides which sources have cells on this rank
    srcs = _modulated_sources(grid)
    for src in srcs:
        P, osc = src._torch_inputs(grid.time_steps_passed, max(n, 1))
        inputs += [P, osc, src.nonlin_conv.weight, src.nonlin_conv.bias]
    for p in pmls:
        inputs += [p._psiE, p._psiH]
    grid._tape_lists = (objs, pmls, _modulated(objs), srcs)
    out = _Run.apply(grid, n, grid.E, grid.H, *inputs)
    if out[0].is_contiguous() and out[1].is_contiguous():
        grid._bind_state(out[0], out[1])
    else:
        grid.E, grid.H = out[0], out[1]
    for n_p, p in enumerate(pmls):
        p._psiE, p._psiH = out[2 + 2 * n_p], out[3 + 2 * n_p]
    if n > 0:
        grid.time_steps_passed += n
