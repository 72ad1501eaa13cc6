"""fdtdhip_adjoint_run_noalloc against fdtdhip_adjoint_run, through the C ABI.

The no-allocation entry point keeps the work arrays and the descriptor tables of the
reverse sweep in a static device array instead of allocated memory; its results must be
those of the existing entry point bit for bit.  Two calls in a row, the second on a smaller
grid (the array then still holds the first call's contents), in both precisions; sizes
beyond the array are refused with a message.
"""

import ctypes

import numpy as np
import pytest

from fdtd_amd import _lib

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _Scene:
    """A small differentiable scene built with plain C-ABI calls: two PML faces, an
    anisotropic object whose tensor is learnable, an isotropic object next to it, a point
    source; `steps` recorded forward steps."""

    def __init__(self, lib, shape, dtype, steps, seed, scan, periodic_z=False):
        import torch

        self.lib, self.steps = lib, steps
        nx, ny, nz = shape
        tdt = torch.float32 if dtype == _lib.F32 else torch.float64
        gen = torch.Generator(device="cuda").manual_seed(seed)
        rnd = lambda *s: torch.rand(*s, dtype=tdt, device="cuda", generator=gen) - 0.5      # noqa: E731
        zeros = lambda *s: torch.zeros(*s, dtype=tdt, device="cuda")                        # noqa: E731
        self.rnd, self.zeros = rnd, zeros
        sc = 0.99 / np.sqrt(sum(n > 1 for n in shape))
        self.h = ctypes.c_void_p()
        _lib.check(lib.fdtdhip_plan_create(ctypes.byref(self.h), nx, ny, nz, dtype, sc, 0))
        self.E, self.H = rnd(nx, ny, nz, 3) * 1e-2, rnd(nx, ny, nz, 3) * 1e-2
        self.ie, self.im = torch.full((nx, ny, nz, 3), 0.8, dtype=tdt, device="cuda"), zeros(nx, ny, nz, 3) + 1
        # objects: x0,x1,y0,y1,z0,z1
        self.abox = (nx // 3, nx // 3 + 5, ny // 3, ny // 3 + 4, 0, min(nz, 3))
        self.ibox = (nx // 3 + 3, nx // 3 + 7, ny // 2, ny // 2 + 3, 0, nz)
        for b in (self.abox, self.ibox):
            self.ie[b[0]:b[1], b[2]:b[3], b[4]:b[5]] = 0          # the caller's duty (objects.py:90)
        _lib.check(lib.fdtdhip_plan_set_fields(self.h, _p(self.E), _p(self.H), _p(self.ie), _p(self.im)))
        t = 4
        prof = [_lib.doubles(list(v)) for v in (np.linspace(0.7, 0.95, t), np.linspace(-0.2, -0.02, t),
                                                 np.linspace(0.75, 0.97, t), np.linspace(-0.15, -0.01, t))]
        self.psi = []
        for axis, side in ((0, 0), (1, 1)):
            slab = [nx, ny, nz]
            slab[axis] = t
            pe, ph = zeros(2, *slab), zeros(2, *slab)
            pid = _lib.check(lib.fdtdhip_plan_add_pml(self.h, axis, side, t, *prof, _p(pe), _p(ph)))
            self.psi.append((pid, pe, ph))
        if periodic_z:   # after the PMLs, which then keep their correction inside the bulk kernel
            _lib.check(lib.fdtdhip_plan_add_periodic(self.h, 2))
        ab, ib = self.abox, self.ibox
        cells = (ab[1] - ab[0]) * (ab[3] - ab[2]) * (ab[5] - ab[4])
        self.tensor = rnd(cells, 3, 3) * 0.3 + torch.eye(3, dtype=tdt, device="cuda") * 0.5
        self.aid = _lib.check(lib.fdtdhip_plan_add_object(self.h, _lib.OBJ_ANISOTROPIC, _lib.int32s(ab),
                                                           _p(self.tensor), None))
        self.iso = zeros(ib[1] - ib[0], ib[3] - ib[2], ib[5] - ib[4], 3) + 0.4
        _lib.check(lib.fdtdhip_plan_add_object(self.h, _lib.OBJ_ISOTROPIC, _lib.int32s(ib), _p(self.iso), None))
        _lib.check(lib.fdtdhip_plan_add_source(self.h, _lib.SRC_ADD_POINTS, 2, 2, 1,
                                               _lib.int32s([nx // 2, ny // 2, nz // 2]), _lib.doubles([1.0]),
                                               9.0, 0.0, 0, 1.0 / 9.0, 4.5, 5.0))
        if scan:
            assert _lib.check(lib.fdtdhip_plan_scan_materials(self.h, 1)) == 3
        self.trace = zeros(steps, cells, 3)
        _lib.check(lib.fdtdhip_plan_set_object_trace(self.h, self.aid, _p(self.trace), steps))
        _lib.check(lib.fdtdhip_run(self.h, 0, steps))
        _lib.check(lib.fdtdhip_plan_set_object_trace(self.h, self.aid, None, 0))
        # where every reverse sweep starts from
        self.lam0 = [rnd(nx, ny, nz, 3), rnd(nx, ny, nz, 3)]
        self.lampsi0 = [(rnd(*pe.shape), rnd(*ph.shape)) for _, pe, ph in self.psi]

    def sweep(self, noalloc):
        """One reverse sweep from the common starting point; returns every output."""
        import torch

        lib, h = self.lib, self.h
        lamE, lamH = (t.clone() for t in self.lam0)
        lampsi = [(a.clone(), b.clone()) for a, b in self.lampsi0]
        grad = torch.zeros_like(self.tensor)
        _lib.check(lib.fdtdhip_plan_set_object_trace(h, self.aid, _p(self.trace), self.steps))
        _lib.check(lib.fdtdhip_plan_set_object_grad(h, self.aid, _p(grad)))
        for (pid, _, _), (le, lh) in zip(self.psi, lampsi):
            _lib.check(lib.fdtdhip_plan_set_pml_adjoint(h, pid, _p(le), _p(lh)))
        if noalloc:
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            _lib.check(lib.fdtdhip_adjoint_run_noalloc(h, self.steps, _p(lamE), _p(lamH)))
            torch.cuda.synchronize()
            self.device_bytes_taken = free_before - torch.cuda.mem_get_info()[0]
        else:
            work = self.zeros(self.E.numel() // 3, 6)
            _lib.check(lib.fdtdhip_adjoint_run(h, self.steps, _p(lamE), _p(lamH), _p(work)))
        torch.cuda.synchronize()
        _lib.check(lib.fdtdhip_plan_set_object_trace(h, self.aid, None, 0))
        _lib.check(lib.fdtdhip_plan_set_object_grad(h, self.aid, None))
        out = {"lamE": lamE, "lamH": lamH, "grad": grad}
        for n, (le, lh) in enumerate(lampsi):
            out[f"lam_psiE{n}"], out[f"lam_psiH{n}"] = le, lh
        return {k: v.cpu().numpy() for k, v in out.items()}

    def prepare_noalloc(self):
        """Fresh adjoint arrays for a sweep (made on torch's stream: synchronise before
        enqueue_noalloc)."""
        import torch

        lib, h = self.lib, self.h
        lamE, lamH = (t.clone() for t in self.lam0)
        lampsi = [(a.clone(), b.clone()) for a, b in self.lampsi0]
        grad = torch.zeros_like(self.tensor)
        _lib.check(lib.fdtdhip_plan_set_object_trace(h, self.aid, _p(self.trace), self.steps))
        _lib.check(lib.fdtdhip_plan_set_object_grad(h, self.aid, _p(grad)))
        for (pid, _, _), (le, lh) in zip(self.psi, lampsi):
            _lib.check(lib.fdtdhip_plan_set_pml_adjoint(h, pid, _p(le), _p(lh)))
        out = {"lamE": lamE, "lamH": lamH, "grad": grad}
        for n, (le, lh) in enumerate(lampsi):
            out[f"lam_psiE{n}"], out[f"lam_psiH{n}"] = le, lh
        return out

    def enqueue_noalloc(self, stream, arrays):
        """The no-allocation sweep enqueued on `stream`, without waiting for it."""
        lib, h = self.lib, self.h
        _lib.check(lib.fdtdhip_plan_set_stream(h, ctypes.c_void_p(stream.cuda_stream)))
        _lib.check(lib.fdtdhip_adjoint_run_noalloc(h, self.steps, _p(arrays["lamE"]), _p(arrays["lamH"])))

    def close(self):
        assert self.lib.fdtdhip_plan_destroy(self.h) == 0


def _compare(scene, check_memory=True):
    want = scene.sweep(noalloc=False)
    got = scene.sweep(noalloc=True)
    assert np.abs(want["grad"]).max() > 0 and np.abs(want["lamE"]).max() > 0
    for name in want:
        assert np.array_equal(got[name], want[name]), f"{name} differs from fdtdhip_adjoint_run"
    if check_memory:   # (not on the first call of the process, which loads the code object)
        assert scene.device_bytes_taken == 0, "the no-alloc call changed the device's free memory"
    # and the existing entry point is not disturbed by the tables having lived elsewhere
    again = scene.sweep(noalloc=False)
    for name in want:
        assert np.array_equal(again[name], want[name]), f"{name}: fdtdhip_adjoint_run changed after the no-alloc call"


def test_noalloc_reverse_sweep_equals_the_allocating_one_twice_in_a_row():
    lib = _lib.load()
    # first call: 3-D, float64, the plain two-pass sweep (6 912 cells)
    big = _Scene(lib, (24, 18, 16), _lib.F64, steps=7, seed=11, scan=True, periodic_z=True)
    # second call right after it, on a smaller grid: 2-D, float32, paired launches
    # (the scratch array still holds the first call's difference adjoints and tables)
    small = _Scene(lib, (20, 17, 1), _lib.F32, steps=9, seed=12, scan=False)
    try:
        _compare(big, check_memory=False)
        _compare(small)
        _compare(big)        # and back: tables of the small plan are not taken for the big one's
    finally:
        big.close()
        small.close()


def test_noalloc_calls_on_two_plans_and_two_streams_back_to_back():
    """Every call on a GPU shares that GPU's scratch array.  A second call -- another plan,
    another non-blocking stream -- made while the first sweep is still running must wait for
    it instead of clearing the array under it."""
    import torch

    lib = _lib.load()
    long = _Scene(lib, (48, 40, 36), _lib.F64, steps=60, seed=21, scan=True, periodic_z=True)
    short = _Scene(lib, (30, 22, 1), _lib.F32, steps=12, seed=22, scan=False)
    try:
        want_long, want_short = long.sweep(noalloc=False), short.sweep(noalloc=False)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        got_long, got_short = long.prepare_noalloc(), short.prepare_noalloc()
        torch.cuda.synchronize()
        long.enqueue_noalloc(s1, got_long)      # some milliseconds of kernels on s1 ...
        short.enqueue_noalloc(s2, got_short)    # ... and the next user of the array right behind
        torch.cuda.synchronize()
        for scene in (long, short):
            _lib.check(lib.fdtdhip_plan_set_stream(scene.h, None))
        for got, want in ((got_long, want_long), (got_short, want_short)):
            for name in want:
                assert np.array_equal(got[name].cpu().numpy(), want[name]), f"{name} differs"
    finally:
        long.close()
        short.close()


def test_noalloc_refuses_what_does_not_fit():
    import torch

    lib = _lib.load()
    h = ctypes.c_void_p()
    # 349 355 cells in float64: one more than the static array holds
    _lib.check(lib.fdtdhip_plan_create(ctypes.byref(h), 349355, 1, 1, _lib.F64, 0.5, 0))
    try:
        dummy = torch.zeros(8, device="cuda")
        rc = lib.fdtdhip_adjoint_run_noalloc(h, 1, _p(dummy), _p(dummy))
        assert rc < 0
        msg = lib.fdtdhip_last_error().decode()
        assert "scratch" in msg and "fdtdhip_adjoint_run" in msg
        assert lib.fdtdhip_adjoint_run_noalloc(h, 1, None, _p(dummy)) < 0
    finally:
        assert lib.fdtdhip_plan_destroy(h) == 0
    # the largest float64 grid that fits is accepted up to the point where fields are needed
    _lib.check(lib.fdtdhip_plan_create(ctypes.byref(h), 349354, 1, 1, _lib.F64, 0.5, 0))
    try:
        rc = lib.fdtdhip_adjoint_run_noalloc(h, 1, _p(dummy), _p(dummy))
        assert rc < 0 and b"fields not set" in lib.fdtdhip_last_error()
    finally:
        assert lib.fdtdhip_plan_destroy(h) == 0
