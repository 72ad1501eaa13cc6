"""Benchmark of the Yee time-stepping hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Prints ONE JSON line (rank 0).  A "step" is one full Yee step (E half step + H
half step) of the whole grid; `value` is cell-updates per second summed over all
GPUs with the fields resident in HBM.

Workloads (all synthetic, fields start at small random values, a point source drives them):
  m3v   512^3 float32 per GPU, vacuum, PointSource    (default: the "512^3 float32 Yee
        update" the headline metric is quoted on; with --gpus N a (512 N) x 512 x 512 grid in
        N x-slabs, the same work per GPU)
  m3    512^3 float32, 6 PML(10), anisotropic + absorbing objects, 3 LineDetectors
  m2    256^3 float64, 6 PML(10), PointSource
  m3v64 512^3 float64, vacuum, PointSource
  c1    161x97x1 float64, the reference's README example
  m4    256x1024x1024 float32 per GPU, 6 PML(10), PointSource (with --gpus 8 the
        2048x1024x1024 x-slab configuration; measured after the default workload of a run on
        several GPUs and reported as also.m4)
With --gpus N > 1 the grid is N slabs of the single-GPU size along x (weak scaling), one
process per GPU, the two tangential components of one halo plane exchanged per half step
over RCCL.  Started as a plain `python bench.py --gpus N` the script launches its N ranks
itself (`python -m torch.distributed.run`, as child processes, before anything touches a
GPU) and relays rank 0's line; started by torch.distributed.run it is one of the ranks.

Materials: the grids are homogeneous outside their objects, as the reference creates
them, and the library then passes the material arrays to its kernels as constants
(it proves this with a scan whenever the arrays change).  `roofline` counts only the
bytes the kernels then have to move.  --materials arrays turns the feature off (every
kernel reads both material arrays).

Extra objects in the JSON line:
  roofline      achieved algorithmic GB/s of the bulk kernels from HIP-event
                timings taken inside the timed region, against the 8 TB/s HBM peak
  cpu_baseline  the numpy oracle (a restatement of the reference's numpy backend)
                timed on this machine's host cores on bounded samples (128^3 and 256^3)
  also          (one GPU, default workload, unless --no-also) the other benchmark
                configurations measured by the same run: the headline grid with both material
                arrays read, the headline grid over 1000 steps ("sustained"), m3, m2, m3v64, m4,
                differentiable runs with their reverse sweeps ("adjoint" at 256^3, "adjoint_512",
                "m5": the reference's learnable_focus set-up in one autograd node and one node
                per step) and the README example ("c1", steps per second)
"""

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (spec)
PMC_NOTE = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of an earlier run of this workload, not this run)"


def build_workload(fdtd, name, n_slabs=1, rank=0, group=None):
    """Returns (grid, cells_per_rank, element size, dtype label, description)."""
    import numpy as np
    import torch

    if name in ("m3v", "m3"):
        fdtd.set_backend("hip.float32")
        shape, es, dt = (512, 512, 512), 4, "f32"
    elif name == "m4":
        fdtd.set_backend("hip.float32")
        shape, es, dt = (256, 1024, 1024), 4, "f32"
    elif name == "m2":
        fdtd.set_backend("hip.float64")
        shape, es, dt = (256, 256, 256), 8, "f64"
    elif name == "m3v64":
        fdtd.set_backend("hip.float64")
        shape, es, dt = (512, 512, 512), 8, "f64"
    elif name == "c1":
        fdtd.set_backend("hip.float64")
        shape, es, dt = (161, 97, 1), 8, "f64"
    else:
        raise SystemExit(f"unknown workload {name}")
    if n_slabs > 1:
        from fdtd_amd.parallel import SlabGrid

        grid = SlabGrid((shape[0] * n_slabs, shape[1], shape[2]), group=group,
                        planes=slab_planes(name, shape[0], n_slabs))
    else:
        grid = fdtd.Grid(shape)
    nx, ny, nz = grid.Nx_global, grid.Ny, grid.Nz  # extents of the whole simulation
    desc = f"{nx}x{ny}x{nz} {dt}"
    if name in ("m3v", "m3v64"):
        grid[nx // 2, ny // 2, nz // 2] = fdtd.PointSource(period=20)
        desc += " vacuum + PointSource"
    elif name in ("m2", "m4"):
        for key in ((slice(0, 10),), (slice(-10, None),), (slice(None), slice(0, 10)),
                    (slice(None), slice(-10, None)), (slice(None), slice(None), slice(0, 10)),
                    (slice(None), slice(None), slice(-10, None))):
            grid[key] = fdtd.PML()
        grid[nx // 2, ny // 2, nz // 2] = fdtd.PointSource(period=20)
        desc += " 6xPML(10) + PointSource"
    elif name == "m3":
        for key in ((slice(0, 10),), (slice(-10, None),), (slice(None), slice(0, 10)),
                    (slice(None), slice(-10, None)), (slice(None), slice(None), slice(0, 10)),
                    (slice(None), slice(None), slice(-10, None))):
            grid[key] = fdtd.PML()
        grid[nx // 2, ny // 2, nz // 2] = fdtd.PointSource(period=20)
        ani = fdtd.AnisotropicObject(permittivity=2.5)
        grid[128:192, 128:192, 128:192] = ani
        off = 0.05 * np.random.RandomState(0).randn(64, 64, 64, 3, 3)
        for c in range(3):
            off[..., c, c] = 0
        with torch.no_grad():
            ani.inverse_permittivity += fdtd.backend.array(off)
        grid[320:384, 128:192, 128:192] = fdtd.AbsorbingObject(permittivity=2.0, conductivity=1e-3)
        grid[12:500, 256, 256] = fdtd.LineDetector()
        grid[256, 12:500, 256] = fdtd.LineDetector()
        grid[256, 256, 12:500] = fdtd.LineDetector()
        desc += " 6xPML(10) + anisotropic 64^3 + absorbing 64^3 + 3 LineDetectors"
    elif name == "c1":
        S = slice
        grid[11:32, 30:84, 0] = fdtd.Object(permittivity=1.7 ** 2)
        grid[S(13e-6, 18e-6), S(5e-6, 8e-6), 0] = fdtd.Object(permittivity=1.5 ** 2)
        grid[S(7.5e-6, 8.0e-6), S(11.8e-6, 13.0e-6), 0] = fdtd.LineSource(period=1550e-9 / 3e8)
        grid[12e-6, :, 0] = fdtd.LineDetector()
        grid[0:10, :, :] = fdtd.PML()
        grid[-10:, :, :] = fdtd.PML()
        grid[:, 0:10, :] = fdtd.PML()
        grid[:, -10:, :] = fdtd.PML()
        desc += " README example"
    if name != "c1":
        # non-trivial data everywhere (all-zero fields would flatter power and clocks)
        gen = torch.Generator(device=grid.E.device).manual_seed(1234 + rank)
        for f in (grid.E, grid.H):
            f.uniform_(-1e-3, 1e-3, generator=gen)
    cells = shape[0] * shape[1] * shape[2]
    return grid, cells, es, dt, desc


def slab_planes(name, per_rank, n_slabs):
    """x planes of every rank.  The two end slabs of a simulation with PMLs on its x faces
    carry ten planes of PML cells each, which cost more than plain ones (measured on one GPU,
    tools/slab_step_cost.py, two steps per pass: the first slab of m4 2.265 ms per step against
    2.127 ms for an interior one, i.e. 16.6 planes' worth): with three or more ranks the end
    slabs hand 16.6 (n - 2) / n planes each to the interior ones (twelve on eight ranks), so
    that every rank's step takes about the same time.  The simulation and its total number of cells stay
    what they are."""
    if name != "m4" or n_slabs < 3:
        return None
    trim = round(16.6 * (n_slabs - 2) / n_slabs)
    planes = [per_rank] * n_slabs
    planes[0] -= trim
    planes[-1] -= trim
    for n in range(2 * trim):
        planes[1 + n % (n_slabs - 2)] += 1
    return planes


def half_step_bytes(grid, cells, es, mask):
    """Bytes one E and one H half step of one GPU have to move: the advanced field read
    and written, the driving field read, the material array read unless the library
    holds it as a constant (mask bit 1: inverse permittivity, 2: inverse permeability),
    and the two live memory variables of every PML cell read and written."""
    psi = sum(int(b._psiE.numel()) for b in grid.boundaries if getattr(b, "_psiE", None) is not None)
    e = cells * (9 if mask & 1 else 12) * es + 2 * psi * es
    h = cells * (9 if mask & 2 else 12) * es + 2 * psi * es
    return e, h


def torch_no_grad():
    import torch

    return torch.no_grad()


def timed_run(grid, lib, _lib, steps, sync, events=True):
    """Time `steps` steps; returns (seconds, timing) with timing = {"ms": [H bulk, E bulk,
    whole-step kernels, shell launches], "n": launches of each, "whole_steps": steps advanced
    by a whole-step kernel, "core" / "hole": the boxes that kernel covers}.
    events=False leaves out the HIP events around every launch (they cost a launch-bound
    step several microseconds); the kernel times are then zero."""
    plan = grid._plan
    _lib.check(lib.fdtdhip_plan_enable_timing(plan, 1 if events else 0))
    taken0 = getattr(grid, "whole_steps_taken", None)
    sync()
    t0 = time.perf_counter()
    with torch_no_grad():     # plain time stepping: nothing to record, no collective about it
        grid.run(steps, progress_bar=False)
    sync()
    elapsed = time.perf_counter() - t0
    ms, n = (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    _lib.check(lib.fdtdhip_plan_get_timing_parts(plan, ms, n))
    _lib.check(lib.fdtdhip_plan_enable_timing(plan, 0))
    core, hole, inner = (ctypes.c_int32 * 6)(), (ctypes.c_int32 * 6)(), (ctypes.c_int32 * 6)()
    staged = 1 if taken0 is not None and grid._world > 1 else 0
    two_step = _lib.check(lib.fdtdhip_plan_two_step_box(plan, staged, inner))
    _lib.check(lib.fdtdhip_plan_whole_step_kind(plan, staged, core, hole))
    whole = n[2] if taken0 is None or grid._world == 1 else grid.whole_steps_taken - taken0
    return elapsed, {"ms": list(ms), "n": list(n), "whole_steps": int(whole),
                     "core": list(core), "hole": list(hole),      # ms[4], n[4]: two-step launches
                     "inner": list(inner) if two_step else None}


def use_material_arrays(grid, lib, _lib):
    """Make the kernels read the material arrays whatever they hold."""
    grid._ensure_plan()
    grid._material_scans = grid._MAX_MATERIAL_SCANS + 1
    _lib.check(lib.fdtdhip_plan_scan_materials(grid._plan, 0))


def roofline(grid, cells, es, timing, mask, steps, workload):
    """The dominant kernel is the bulk half-step kernel, or -- where the library takes whole
    steps in one pass (large 3-D grids; the README example) -- the whole-step kernel.

    Half steps: one half step of one GPU = `cells` cells = one bulk launch on a single GPU;
    on a slab it is two launches (interior planes, then the plane that waits for the halo),
    so durations are summed per half step before dividing.  Algorithmic bytes: the advanced
    field read and written, the driving field read, the material array unless it is a
    constant, the two memory variables of every PML cell read and written.

    Whole steps: the marching kernel advances E and H of its box (`core`: the whole grid, or
    the cells no PML reaches and no open slab face touches) in one launch per step (two on
    an x-slab, around the plane exchanges; summed per step).  Its algorithmic bytes are E
    and H read once and written once, 12 values per cell (9 in the box of the objects, whose
    E half step is the shell launches'), plus the material arrays where they are read.  The
    half-step launches over the shell (PML cells, objects, slab faces) are reported next to
    it as `shell_ms_per_step`; a run of an odd number of steps ends with one step of two
    half-step launches, reported as `half_step_launches`."""
    ms, n, whole = timing["ms"], timing["n"], timing["whole_steps"]
    be, bh = half_step_bytes(grid, cells, es, mask)
    mat = {0: "arrays", 1: "inverse permittivity constant", 2: "inverse permeability constant",
           3: "constants"}[mask]
    roof = {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 0.0,
            "traffic": None, "traffic_source": None}
    if n[4] > 0 and 2 * n[4] >= steps - 3:
        # two steps per launch: E and H read once and written once per TWO steps (12 values per
        # cell and launch, plus the material arrays where they are read)
        b = timing.get("inner") or [0, 0, 0, 0, 0, 0]
        inner = (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]) or cells
        per_launch = 12 * inner * es      # (constant materials: the kernel's condition)
        avg_ms = ms[4] / n[4]
        pass_ms = (ms[4] + ms[3]) / n[4]
        roof.update({
            "kernel": ("two-step marching kernel (E and H advanced twice in one pass)" if inner == cells else
                       "two-step marching kernel over the cells two or three layers away from PML faces, objects and "
                       "open slab faces; two steps of half-step launches over the others"),
            "kernel_cells": inner, "grid_cells": cells,
            "shell_ms_per_step": round(ms[3] / steps, 4),
            "steps_per_launch": 2, "launches_per_step": round(n[4] / steps, 2),
            "other_launches": n[0] + n[1] + n[2],
            # the time of a pass over the WHOLE grid (with a shell: its launches included)
            # priced at the one-step kernels' 12 values per cell and STEP -- what a kernel that
            # passes over memory once per step would have to sustain -- and at the single-sweep
            # figure of 24 values per cell and step (every array read once per half step), the
            # rate that sets the 100 % mark of 8 TB/s / 96 B = 83.3 Gcells/s in float32 (41.7 in
            # float64).  Two steps per pass move a quarter of the latter.
            "one_pass_per_step_equivalent_GBps": round(2 * cells * 12 * es / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else 0.0,
            "all_arrays_equivalent_GBps": round(2 * cells * 24 * es / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else 0.0,
            "single_sweep_roofline_Gcells_s": round(HBM_PEAK_GBPS / (24 * es), 1),
            "vs_single_sweep_roofline": round(2 * cells / (pass_ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS / (24 * es)), 3) if pass_ms > 0 else 0.0,
        })
    elif whole > 0:
        c, h = timing["core"], timing["hole"]
        core = (c[1] - c[0]) * (c[3] - c[2]) * (c[5] - c[4])
        hole = (h[1] - h[0]) * (h[3] - h[2]) * (h[5] - h[4])
        values = 12 * core - 3 * hole
        if not mask & 1:
            values += 3 * (core - hole)
        if not mask & 2:
            values += 3 * core
        per_launch = values * es
        avg_ms = ms[2] / whole
        roof.update({
            "kernel": ("one-lane-per-cell whole-step kernel" if workload == "c1" else
                       "marching whole-step kernel (E and H in one pass)" if core == cells else
                       "marching whole-step kernel over the cells away from PMLs, objects and slab "
                       "faces; half-step launches over the others"),
            "kernel_cells": core, "grid_cells": cells,
            "launches_per_step": round(n[2] / whole, 2),
            "shell_ms_per_step": round(ms[3] / whole, 4),
            "half_step_launches": n[0] + n[1],
            # the same duration priced at 24 values per cell and step whatever the materials
            # are (every array counted as read): an effective rate, not HBM traffic
            "all_arrays_equivalent_GBps": round(core * 24 * es / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0,
        })
    else:
        avg_ms = (ms[0] + ms[1]) / (2 * steps)
        per_launch = (be + bh) / 2
        roof.update({
            "kernel": "bulk half-step kernel (E and H)",
            "all_arrays_equivalent_GBps": round(cells * 12 * es / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0,
            "avg_launch_ms_E": round(ms[1] / steps, 4),
            "avg_launch_ms_H": round(ms[0] / steps, 4),
        })
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roof.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
                 "materials": mat, "bytes_per_launch": per_launch, "avg_launch_ms": round(avg_ms, 4)})
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        key = f"{workload}:{mat}" + (":two-steps" if "steps_per_launch" in roof else ":whole-step" if whole > 0 else "")
        rec = json.load(open(pmc)).get(key)
        if rec:
            roof["traffic"] = rec.get("bytes_per_launch")
            roof["traffic_source"] = PMC_NOTE
            if "shell" in rec:      # the half-step launches beside the marching kernel, both per step
                roof["shell_traffic"] = rec["shell"]["bytes_per_step"]
                roof["shell_cells"] = rec["shell"]["cells"]
    return roof


def measure_single(workload, steps, warmup, materials="auto"):
    """One workload on this process's GPU; returns a compact record and releases the grid."""
    import torch

    import fdtd_amd as fdtd
    from fdtd_amd import _lib

    lib = _lib.load()
    grid, cells, es, dt, desc = build_workload(fdtd, workload)
    if materials == "arrays":
        use_material_arrays(grid, lib, _lib)
    with torch_no_grad():
        grid.run(warmup, progress_bar=False)
    torch.cuda.synchronize()
    mask = 0 if materials == "arrays" else _lib.check(lib.fdtdhip_plan_scan_materials(grid._plan, 1))
    elapsed, timing = timed_run(grid, lib, _lib, steps, torch.cuda.synchronize)
    roof = roofline(grid, cells, es, timing, mask, steps, workload)
    rec = {
        "workload": desc, "name": workload, "dtype": dt, "steps": steps, "warmup": warmup,
        "value": round(cells * steps / elapsed / 1e9, 3), "unit": "Gcells/s",
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "avg_launch_ms": roof["avg_launch_ms"], "achieved": roof["achieved"], "frac": roof["frac"],
        "materials": roof["materials"], "kernel": roof["kernel"],
        "bytes_per_launch": roof["bytes_per_launch"], "traffic": roof["traffic"],
        **{k: roof[k] for k in ("kernel_cells", "shell_ms_per_step", "half_step_launches", "shell_traffic", "steps_per_launch",
                                   "one_pass_per_step_equivalent_GBps") if k in roof},
    }
    del grid
    torch.cuda.empty_cache()
    return rec


def measure_adjoint(kind="large", passes=2):
    """configs[4]: a differentiable run recorded in one autograd node and its reverse sweep;
    milliseconds per step of each (tools/bench_adjoint.py).  `large`: 256^3 float32, six PMLs,
    a learnable 64^3 anisotropic box, in the size of the other legs; `xlarge`: the same at
    512^3; `small`: the 97x97x1 float64 set-up of the reference's examples/learnable_focus.py:98-128,
    200 steps; `stepwise`: that set-up advanced one step per autograd node, as the reference's
    cortical/autoencoder.py:108-112 drives the grid."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_adjoint

    best = None
    for _ in range(passes):      # the first pass warms up allocator and code objects
        r = bench_adjoint.once(kind)
        if best is None or r["forward_ms_per_step"] + r["backward_ms_per_step"] < \
                best["forward_ms_per_step"] + best["backward_ms_per_step"]:
            best = r
    what = {"large": "256x256x256 f32 6xPML(10) + learnable anisotropic 64^3",
            "xlarge": "512x512x512 f32 6xPML(10) + learnable anisotropic 64^3",
            "small": "97x97x1 f64 periodic z + learnable anisotropic 40x20 (learnable_focus set-up), one autograd node",
            "stepwise": "97x97x1 f64 periodic z + learnable anisotropic 40x20 (learnable_focus set-up), "
                        "one autograd node per step"}[kind]
    return {"workload": f"{what}, {best['steps']} steps forward (curl traces recorded) and backward "
                        "(reverse-time adjoint)",
            "forward_ms_per_step": round(best["forward_ms_per_step"], 5),
            "backward_ms_per_step": round(best["backward_ms_per_step"], 5),
            "forward_Gcells_s": round(best["forward_Gcells_s"], 3),
            "backward_Gcells_s": round(best["backward_Gcells_s"], 3), "unit": "Gcells/s"}


def measure_c1(steps=5000, warmup=500):
    """configs[0] on the GPU: the README example (reference README.md:137-296), launch-bound;
    steps per second without per-launch events."""
    import torch

    import fdtd_amd as fdtd

    grid, cells, _, dt, desc = build_workload(fdtd, "c1")
    with torch_no_grad():
        grid.run(warmup, progress_bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grid.run(steps, progress_bar=False)
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    del grid
    return {"workload": desc, "name": "c1", "dtype": dt, "steps": steps, "warmup": warmup,
            "steps_per_s": round(steps / elapsed, 1), "us_per_step": round(elapsed / steps * 1e6, 3),
            "value": round(cells * steps / elapsed / 1e9, 4), "unit": "Gcells/s",
            "kernel": "one-lane-per-cell whole-step kernel, one launch per step"}


def cpu_baseline(name, budget_s=12.0):
    """The numpy oracle on bounded samples of the same scene (rank 0, N=1 only): 128^3 for
    about `budget_s` seconds, then 256^3 for 5 steps."""
    from oracle.yee_numpy import OracleGrid  # CPU baseline leg: allowed use of the oracle

    def sample(shape, budget, min_steps, max_steps):
        g = OracleGrid(shape)
        if name in ("m2", "m3", "m4"):
            for axis in range(3):
                g.add_pml(axis, "low", 10)
                g.add_pml(axis, "high", 10)
        if name == "c1":
            for axis in range(2):
                g.add_pml(axis, "low", 10)
                g.add_pml(axis, "high", 10)
            g.add_object((11, 32, 30, 84, 0, 1), 1.7 ** 2)
        g.add_point_source((shape[0] // 2, shape[1] // 2, shape[2] // 2), period=20)
        g.step()  # warm-up
        t0, steps = time.perf_counter(), 0
        while steps < min_steps or (time.perf_counter() - t0 < budget and steps < max_steps):
            g.step()
            steps += 1
        dt = time.perf_counter() - t0
        cells = shape[0] * shape[1] * shape[2]
        return cells * steps / dt / 1e9, steps, dt

    if name == "c1":
        v, steps, dt = sample((161, 97, 1), budget_s, 1, 2000)
        shape_txt, extra = "161x97x1", {}
    else:
        v, steps, dt = sample((128, 128, 128), budget_s, 1, 2000)
        shape_txt = "128x128x128"
        v2, steps2, dt2 = sample((256, 256, 256), 0.0, 5, 5)
        extra = {"at_256": {"value": v2, "unit": "Gcells/s",
                            "sample": f"256x256x256 cells, {steps2} steps in {dt2:.1f} s"}}
    return dict(
        value=v, unit="Gcells/s", cores=1, kind="port",
        sample=f"numpy float64 oracle, {shape_txt} cells, same boundaries and "
               f"source, {steps} steps in {dt:.1f} s (numpy elementwise ops use one core; "
               f"host has {os.cpu_count()} logical CPUs)", **extra)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes
    through torch.distributed.run and relay rank 0's JSON line.  Nothing in this process
    has touched a GPU (torch is not even imported)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            print(line.rstrip("\n"), flush=True)
        else:
            sys.stderr.write(line)
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (half a second or more of timed kernels by default: a sampling monitor beside the run sees it)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, help="default: m3v (on several GPUs also m4, as also.m4)")
    ap.add_argument("--materials", choices=("auto", "arrays"), default="auto")
    ap.add_argument("--no-also", action="store_true",
                    help="leave out the `also` records (other configurations, sustained run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # under rocprofv3 the profiler's preloaded library has initialised the GPU already, and
        # starting the ranks from here would be a program started by a process that holds it
        preload = os.environ.get("LD_PRELOAD", "")
        if "rocprof" in preload.lower() or any(k.startswith("ROCPROF_") for k in os.environ):
            sys.exit("bench.py --gpus N does not start its own ranks under a profiler: profile one rank "
                     "(--gpus 1) or start the ranks with torch.distributed.run and profile from outside")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    default_run = args.workload is None
    workload = args.workload or "m3v"
    # one rank per GPU; FDTD_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
    backend = os.environ.get("FDTD_DIST_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    ctl = data_group = None      # control plane (barriers, timing) / halo planes
    transport = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            # RCCL's stream gets high priority: a halo plane's send/recv kernel should start as
            # soon as it is enqueued, not queue behind the interior kernel it is meant to overlap
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
            except Exception:  # noqa: BLE001 -- an option of convenience only
                opts = None
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), pg_options=opts)
            ctl = dist.new_group(backend="gloo")   # barriers and the timing reduction on the host
            transport = "rccl"
            # one small neighbour exchange over RCCL before anything is built; if it cannot
            # be done on this machine the planes go through the host instead of the run dying
            ok = 1
            try:
                probe = torch.full((4,), float(rank), device="cuda")
                got = torch.zeros(4, device="cuda")
                ops = [dist.P2POp(dist.isend, probe, (rank + 1) % world),
                       dist.P2POp(dist.irecv, got, (rank - 1) % world)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                torch.cuda.synchronize()
                ok = int(float(got[0].item()) == float((rank - 1) % world))
            except Exception as exc:  # noqa: BLE001 -- any transport failure means "fall back"
                print(f"rank {rank}: RCCL point-to-point probe failed: {exc}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
            if int(flag.item()) == 0:
                data_group, transport = ctl, "gloo (host-staged; RCCL point-to-point unavailable)"
        else:
            dist.init_process_group(backend)
            transport = backend + " (host-staged)"

    import fdtd_amd as fdtd
    from fdtd_amd import _lib

    lib = _lib.load()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        return float(t.item())

    def measure(workload, steps, warmup):
        """One workload on all ranks of this job (a grid on one GPU, x-slabs on several);
        rank 0 returns the record, the others None."""
        # on several GPUs rank 0 first times, on its own, a one-GPU grid of the per-GPU size of the
        # same workload (the other ranks wait).  It is NOT one of the run's slabs: it carries the
        # boundaries of all six faces and exchanges nothing, where a slab has at most one closed x
        # face and one or two neighbours (tools/slab_step_cost.py times real slabs with their
        # exchanges on one GPU)
        one_slab = None
        if world > 1:
            if rank == 0 and workload != "c1":
                one_slab = measure_single(workload, steps, warmup, args.materials)
            sync()
        grid, cells, es, dt, desc = build_workload(fdtd, workload, world, rank, data_group)
        if args.materials == "arrays":
            use_material_arrays(grid, lib, _lib)
        with torch_no_grad():
            grid.run(warmup, progress_bar=False)
        sync()
        mask = 0 if args.materials == "arrays" else _lib.check(lib.fdtdhip_plan_scan_materials(grid._plan, 1))
        if workload == "c1":
            # launch-bound: the step rate is timed without per-launch events, the kernel
            # durations come from a second pass of the same steps with them
            elapsed = timed_run(grid, lib, _lib, steps, sync, events=False)[0]
            _, timing = timed_run(grid, lib, _lib, steps, sync)
        else:
            elapsed, timing = timed_run(grid, lib, _lib, steps, sync)
        elapsed = max_over_ranks(elapsed)
        out = None
        if rank == 0:
            value = cells * world * steps / elapsed / 1e9
            # (this rank's own cells: the slabs of a run may differ in thickness)
            roof = roofline(grid, grid.Nx * grid.Ny * grid.Nz, es, timing, mask, steps, workload)
            roof["launches"] = sum(timing["n"])
            out = {
                "metric": "Yee cell-updates/s", "value": round(value, 3), "unit": "Gcells/s",
                "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
                "config": {"workload": desc, "name": workload,
                           "parallelism": f"x-slab x{world}" if world > 1 else "single GPU",
                           "cells_per_gpu": cells,
                           **({"transport": transport,
                               "halo": f"{grid.Ny}x{grid.Nz}x2 {dt} per half step and neighbour",
                               "planes_per_rank": slab_planes(workload, cells // (grid.Ny * grid.Nz), world)
                                                  or "even"} if transport else {})},
                "roofline": roof,
            }
            if one_slab is not None:
                one_slab["note"] = ("a one-GPU grid of the per-GPU size with the boundaries of all six faces "
                                    "and no exchange; not one of this run's slabs")
                out["one_gpu_grid_of_slab_size"] = one_slab
        del grid
        torch.cuda.empty_cache()
        return out

    # On several GPUs the series stays the one-GPU headline's: every GPU holds a 512^3 float32
    # vacuum slab of a (512 N) x 512 x 512 grid, the same work per GPU as `--gpus 1` times --
    # weak scaling of one workload.  The 8-GPU configuration proper (256 x 1024 x 1024 per GPU
    # with the PMLs of the whole simulation, `m4`) is measured by the same ranks right after it
    # and reported as also.m4 (with its own one-GPU point).
    out = measure(workload, args.steps, args.warmup)
    if world > 1 and default_run and not args.no_also:
        m4 = measure("m4", min(args.steps, 200), min(args.warmup, 10))
        if rank == 0:
            for k in ("metric", "unit", "higher_is_better", "vs_baseline", "data"):
                m4.pop(k, None)
            m4["note"] = ("the 8-GPU configuration's slabs (256 x 1024 x 1024 cells per GPU, PMLs of the whole "
                          "simulation): its weak-scaling series starts at one_gpu_grid_of_slab_size.value in "
                          "this record (= also.m4 of the one-GPU line)")
            out["also"] = {"m4": m4}
    if rank == 0 and world == 1 and default_run and not args.no_also:
        # the other benchmark configurations, measured by this same run
        also = {}
        also["array_materials"] = measure_single("m3v", args.steps, args.warmup, "arrays")
        also["sustained"] = measure_single("m3v", 1000, args.warmup)
        also["m3"] = measure_single("m3", 200, 20)
        also["m3_array_materials"] = measure_single("m3", 100, 10, "arrays")
        also["m2"] = measure_single("m2", 1000, 20)
        also["m3v64"] = measure_single("m3v64", 100, 10)
        also["m4"] = measure_single("m4", 100, 10)     # one GPU's share of the 8-GPU configuration
        also["adjoint"] = measure_adjoint("large")
        also["adjoint_512"] = measure_adjoint("xlarge")
        also["c1"] = measure_c1()
        also["m5"] = {"one_node": measure_adjoint("small", 3), "node_per_step": measure_adjoint("stepwise", 3)}
        out["also"] = also
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
