"""What one recorded step costs at the least from Python: a bare loop of fdtdhip_plan_bind,
fdtdhip_step_into and two empty_like per step on the stepwise workload of bench_adjoint.py
(97 x 97 x 1, one learnable box), with the row of the curl traces bound once or re-bound for
every step, and with nothing but the launch.

    python tools/stepwise_floor.py
"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_adjoint
from fdtd_amd import _lib
from fdtd_amd.grid import _stream_ptr
from fdtd_amd.backend import backend as bd

for variant in ("bound-once", "set-each-step", "bound-once", "set-each-step", "launch-only"):
    g, steps, window = bench_adjoint.build("stepwise")
    with torch.no_grad():
        g.run(2, progress_bar=False)
    lib = g._ensure_plan()
    plan = g._plan
    o = g.objects[0]
    cells = o.Nx * o.Ny * o.Nz
    tr = torch.empty((steps, cells, 3), dtype=bd.float, device=bd.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, p(tr), steps))
    E, H = g.E, g.H
    ie, im = g.inverse_permittivity, g.inverse_permeability
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        E1, H1 = torch.empty_like(E), torch.empty_like(H)
        if variant != "launch-only" or k == 0:
            _lib.check(lib.fdtdhip_plan_bind(plan, _stream_ptr(bd.device), E.data_ptr(), H.data_ptr(), ie.data_ptr(), im.data_ptr(),
                                             E1.data_ptr(), H1.data_ptr(), None, None))
        if variant == "set-each-step":
            _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, ctypes.c_void_p(tr.data_ptr() + k * cells * 3 * tr.element_size()), 1))
        _lib.check(lib.fdtdhip_step_into(plan, 2 + k))
        if variant == "set-each-step":
            _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, None, 0))
        if variant != "launch-only":
            E, H = E1, H1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{variant}: host {(t1 - t0) / steps * 1e6:.1f} us/step, with drain {(t2 - t0) / steps * 1e6:.1f} us/step", flush=True)
    _lib.check(lib.fdtdhip_plan_set_object_trace(plan, o._native_id, None, 0))
