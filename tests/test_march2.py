"""Two steps per launch (csrc/yee_march2.hpp) against explicit half steps and the oracle.

A large 3-D grid with nothing between its steps but additive point sources -- no PML, object,
detector or recorded curl, materials the scan proves constant -- is advanced two steps per pass over memory: the state after the
first of them exists in registers only.  Same sums on the same operands: the fields must equal
those of explicit half steps (reference fdtd/grid.py:267-295) bit for bit, and the oracle's within
the engine's tolerances.  Extents that are no multiples of the tile heights (6 or 4 rows, float64: 2 or 3),
of the 61 cells of a z row a wave owns or of the chunk length; sources on the corners of the
grid, on both sides of a wave's first and last owned cells and of a chunk's first plane;
numbers of steps that leave 0 to 3 steps to the one-step kernels and half steps; random start
fields; with material arrays the one-step kernels must run instead.
"""

import ctypes
import os

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd
from fdtd_amd import _lib
from oracle import yee_numpy as orc
from test_hip_parity import TOL, host, rel_l2

pytestmark = pytest.mark.gpu

_CASES = [("float32", (70, 66, 1000), 9, False), ("float32", (131, 130, 260), 8, False),
          ("float64", (67, 129, 500), 11, False), ("float32", (64, 128, 512), 6, True),
          ("float64", (64, 70, 940), 4, True), ("float32", (150, 120, 244), 8, False),
          ("float32", (160, 150, 183), 5, False)]


def _pair_launches(grid, lib):
    ms, n = (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    _lib.check(lib.fdtdhip_plan_get_timing_parts(grid._plan, ms, n))
    return int(n[4]), int(n[2]), int(n[0])


@pytest.mark.parametrize("dtype,shape,steps,dense", _CASES,
                         ids=[f"{d}-{'x'.join(map(str, s))}-{n}{'-arrays' if a else ''}" for d, s, n, a in _CASES])
def test_two_steps_per_launch_equal_half_steps_bitwise(dtype, shape, steps, dense):
    fdtd.set_backend(f"hip.{dtype}")
    nx, ny, nz = shape
    rs = np.random.RandomState(31)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    ie = 1.0 / (1 + rs.rand(*shape, 3)) if dense else None
    im = 1.0 / (1 + rs.rand(*shape, 3)) if dense else None
    zs = min(nz - 1, 122)
    sources = (((nx // 2, ny // 2, nz // 2), 11), ((0, 0, 0), 7), ((nx - 1, ny - 1, nz - 1), 9),
               ((33, 4, 60 % nz), 13), ((33, 3, 61 % nz), 5), ((min(nx - 1, 64), ny - 2, zs), 6),
               ((min(nx - 1, 63), 1, zs - 1), 8))

    def build():
        g = fdtd.Grid(shape)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        if dense:
            g.inverse_permittivity = fdtd.backend.array(ie)
            g.inverse_permeability = fdtd.backend.array(im)
        for cell, period in sources:
            g[cell] = fdtd.PointSource(period=period)
        return g

    a = build()
    a._provide_step_buffers()
    lib = a._ensure_plan()
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 1))
    a.run(steps, progress_bar=False)
    pairs, singles, halves = _pair_launches(a, lib)
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 0))
    if dense:   # material arrays would have to be read twice per launch: one step per launch
        assert pairs == 0 and singles == 2 * (steps // 2) and halves == steps % 2
    else:
        assert pairs == 2 * (steps // 4), "the two-step kernel did not run"
        assert singles == 2 * ((steps % 4) // 2) and halves == steps % 2
    b = build()
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    torch.cuda.synchronize()
    assert float(a.E.abs().max()) > 0
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)
    # the one-step kernels with the switch off: the same bits again
    os.environ["FDTDHIP_MARCH2"] = "0"
    try:
        c = build()
        c._provide_step_buffers()
        c._ensure_plan()
        _lib.check(lib.fdtdhip_plan_enable_timing(c._plan, 1))
        c.run(steps, progress_bar=False)
        assert _pair_launches(c, lib)[0] == 0
        _lib.check(lib.fdtdhip_plan_enable_timing(c._plan, 0))
    finally:
        del os.environ["FDTDHIP_MARCH2"]
    assert torch.equal(a.E, c.E) and torch.equal(a.H, c.H)
    if nz <= 512 and not dense:
        # rows of up to 512 cells take the kernel with a workgroup per tile; the kernel of
        # independent waves (longer rows) on the same grid: the same bits once more
        os.environ["FDTDHIP_MARCH2_ROWS"] = "0"
        try:
            d = build()
            d.run(steps, progress_bar=False)
        finally:
            del os.environ["FDTDHIP_MARCH2_ROWS"]
        assert torch.equal(a.E, d.E) and torch.equal(a.H, d.H)
    if nx * ny * nz > 5_000_000:
        return      # (the oracle comparison of the other shapes covers the arithmetic)
    o = orc.OracleGrid(shape)
    o.E[...], o.H[...] = E0, H0
    if dense:
        o.inverse_permittivity[...] = ie
        o.inverse_permeability[...] = im
    for cell, period in sources:
        o.add_point_source(cell, period=period)
    o.run(steps)
    assert rel_l2(host(a.E), o.E) <= TOL[dtype] and rel_l2(host(a.H), o.H) <= TOL[dtype]


def test_detectors_read_every_step_of_a_pair():
    """A detector reads E after the sources and H after the H half step of every step
    (reference fdtd/detectors.py:114-124).  With two steps per pass the state after the first
    step of a pair never exists in memory: a small kernel evaluates it once more for the cells
    the detectors read.  Lines along the three axes, a slanted line, a block, a line through a
    source cell, on a vacuum grid of the benchmark's kind: records equal to those of explicit
    half steps bit for bit."""
    fdtd.set_backend("hip.float32")
    shape, steps = (64, 128, 512), 10
    rs = np.random.RandomState(3)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)

    def build():
        g = fdtd.Grid(shape)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        g[32, 64, 256] = fdtd.PointSource(period=9)
        g[0, 0, 0] = fdtd.PointSource(period=7)
        dets = [fdtd.LineDetector() for _ in range(5)] + [fdtd.BlockDetector()]
        g[0:64, 64, 200] = dets[0]
        g[10, 0:128, 61] = dets[1]
        g[63, 127, 0:512] = dets[2]
        g[5:60, 10:120, 300] = dets[3]          # a slanted line
        g[32, 64, 250:260] = dets[4]            # through the source cell
        g[30:34, 62:66, 254:258] = dets[5]
        return g, dets

    a, da = build()
    a._provide_step_buffers()
    lib = a._ensure_plan()
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 1))
    a.run(steps, progress_bar=False)
    pairs, singles, _ = _pair_launches(a, lib)
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 0))
    assert pairs == 4 and singles == 2
    b, db = build()
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    torch.cuda.synchronize()
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)
    for n, (x, y) in enumerate(zip(da, db)):
        assert len(x.E) == len(x.H) == steps
        assert torch.equal(torch.stack(list(x.E)), torch.stack(list(y.E))), n
        assert torch.equal(torch.stack(list(x.H)), torch.stack(list(y.H))), n
        assert float(torch.stack(list(x.H)).abs().max()) > 0


def _small_case(seed):
    """A random small grid (extents of one cell, fewer rows than a tile, fewer planes than the
    march looks ahead, z rows just long enough, just over 512 cells, ...) advanced two steps per
    launch by both kernels: equal to explicit half steps bit for bit and to the oracle."""
    rs = np.random.RandomState(seed)
    dtype = ("float64", "float32")[seed % 2]
    fdtd.set_backend(f"hip.{dtype}")
    nx = int(rs.choice([1, 2, 3, 4, 5, 7, 20, 33]))
    ny = int(rs.choice([1, 2, 3, 5, 6, 7, 13, 25]))
    nz = int(rs.choice([61, 62, 64, 65, 122, 123, 130, 200, 512, 513, 600]))
    shape, steps = (nx, ny, nz), int(rs.choice([4, 5, 8, 11]))
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    sources = [((int(rs.randint(nx)), int(rs.randint(ny)), int(rs.randint(nz))), int(rs.randint(5, 15)))
               for _ in range(int(rs.randint(0, 4)))]
    sources = list({c: p for c, p in sources}.items())      # (a repeated cell would be added serially)
    eps = float(rs.choice([1.0, 1.7]))

    def build():
        g = fdtd.Grid(shape, permittivity=eps)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        for cell, period in sources:
            g[cell] = fdtd.PointSource(period=period)
        return g

    a = build()
    a._provide_step_buffers()
    lib = a._ensure_plan()
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 1))
    a.run(steps, progress_bar=False)
    pairs = _pair_launches(a, lib)[0]
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 0))
    assert pairs == 2 * (steps // 4), (shape, "the two-step kernel did not run")
    b = build()
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    os.environ["FDTDHIP_MARCH2_ROWS"] = "0"
    try:
        c = build()
        c.run(steps, progress_bar=False)
    finally:
        del os.environ["FDTDHIP_MARCH2_ROWS"]
    torch.cuda.synchronize()
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H), (seed, shape, steps, sources)
    assert torch.equal(c.E, b.E) and torch.equal(c.H, b.H), (seed, shape, steps, sources, "independent waves")
    o = orc.OracleGrid(shape, permittivity=eps)
    o.E[...], o.H[...] = E0, H0
    for cell, period in sources:
        o.add_point_source(cell, period=period)
    o.run(steps)
    assert rel_l2(host(a.E), o.E) <= TOL[dtype] and rel_l2(host(a.H), o.H) <= TOL[dtype], (seed, shape)


def test_two_steps_per_launch_on_random_small_grids():
    """FDTDHIP_FUSED=2 gives grids of any size the marching kernels (read once per process,
    hence the child)."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (f"import sys\nsys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
            "import test_march2 as t\n"
            "for seed in range(120):\n"
            "    t._small_case(seed)\n"
            "print('two steps per launch on 120 small grids')\n")
    done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDTDHIP_FUSED="2"),
                          capture_output=True, text=True, timeout=280, cwd=here)
    assert done.returncode == 0 and "two steps per launch" in done.stdout, done.stdout[-2000:] + done.stderr[-3000:]


_FACES = {
    # the order DomainBorderPML registers them in, corners shared by three faces
    "six-faces": [(2, "low", 7), (2, "high", 7), (0, "low", 7), (0, "high", 7), (1, "low", 7), (1, "high", 7)],
    # another order of the axes, unequal thicknesses, faces missing, one a single cell thick
    "four-faces": [(1, "high", 5), (0, "low", 9), (2, "low", 1), (2, "high", 11)],
    "x-faces": [(0, "low", 4), (0, "high", 6)],
    "no-faces": [],
}


def _pml_case(dtype, shape, faces, steps, seed=0, objects=False):
    """Two steps per pass on a plan with PML faces: the two-step kernel over the inner box, the
    cells around it by two steps of the shell kernels.  Bitwise equal to explicit half steps
    (fields and memory variables), the oracle within tolerance.  Sources in the faces, in every
    layer between the faces and the inner box, and inside it.  objects: an anisotropic box with
    off-diagonal entries and an absorbing one that overlaps it inside the core, a plain one that
    reaches into the low x face -- the box that holds them and the layers around it belong to
    the cells that take the steps one after the other; sources next to it and inside it."""
    fdtd.set_backend(f"hip.{dtype}")
    nx, ny, nz = shape
    rs = np.random.RandomState(40 + seed)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    lo = {ax: 0 for ax in range(3)}
    for axis, side, t in _FACES[faces]:
        if side == "low":
            lo[axis] = t
    sources = [((nx // 2, ny // 2, nz // 2), 11), ((1, 2, 3), 7), ((nx - 2, ny - 1, nz - 3), 9)]
    for layer in range(6):      # one source per layer inwards from the low faces
        sources.append(((min(lo[0] + layer, nx - 1), min(lo[1] + layer, ny - 1), min(lo[2] + layer, nz - 1)), 5 + layer))
    sources.append(((min(lo[0] + 3, nx - 1), ny // 2, nz // 2 + 1), 13))
    off = 0.04 * np.random.RandomState(5).randn(8, 10, 30, 3, 3)
    if objects:      # sources: under an object, and one to four cells beside the box of the objects
        sources = sources[:3] + [((16, 20, 50), 6), ((29, 18, 60), 8), ((12, 12, 38), 5), ((31, 43 % ny, 92), 9),
                                 ((30, 30, 91), 12)]

    def build():
        g = fdtd.Grid(shape, permittivity=1.3)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        if objects:
            ani = fdtd.AnisotropicObject(permittivity=1.8)
            g[14:22, 16:26, 40:70] = ani
            with torch.no_grad():
                ani.inverse_permittivity += fdtd.backend.array(off)
            g[20:28, 14:20, 60:90] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=3e-3)
            g[2:12, 30:40, 20:50] = fdtd.Object(permittivity=2.5)
        for axis, side, t in _FACES[faces]:
            loc = [slice(None)] * 3
            loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
            g[tuple(loc)] = fdtd.PML()
        for cell, period in sources[:8]:
            g[cell] = fdtd.PointSource(period=period)
        # detectors through faces, layers, inner box and (with objects) the box of the objects
        g.dets = [fdtd.LineDetector(), fdtd.LineDetector(), fdtd.BlockDetector()]
        g[0:nx, ny // 2, nz // 2 + 1] = g.dets[0]
        g[16, 20, 0:nz] = g.dets[1]
        g[10:24, 12:22, 45:48] = g.dets[2]
        return g

    a = build()
    a._provide_step_buffers()
    lib = a._ensure_plan()
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 1))
    a.run(steps, progress_bar=False)
    pairs = _pair_launches(a, lib)[0]
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 0))
    assert pairs == 2 * (steps // 4), (shape, faces, "the two-step kernel did not run")
    b = build()
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    os.environ["FDTDHIP_MARCH2_ROWS"] = "0"
    try:
        c = build()
        c.run(steps, progress_bar=False)
    finally:
        del os.environ["FDTDHIP_MARCH2_ROWS"]
    torch.cuda.synchronize()
    for other, what in ((a, "a workgroup per tile"), (c, "independent waves")):
        assert torch.equal(other.E, b.E) and torch.equal(other.H, b.H), (dtype, shape, faces, steps, what)
        for po, pb in zip(other.boundaries, b.boundaries):
            assert torch.equal(po._psiE, pb._psiE) and torch.equal(po._psiH, pb._psiH), (shape, faces, what)
        for n, (do, dbb) in enumerate(zip(other.dets, b.dets)):
            assert torch.equal(torch.stack(list(do.E)), torch.stack(list(dbb.E))), (shape, faces, what, "detector E", n)
            assert torch.equal(torch.stack(list(do.H)), torch.stack(list(dbb.H))), (shape, faces, what, "detector H", n)
    o = orc.OracleGrid(shape, permittivity=1.3)
    o.E[...], o.H[...] = E0, H0
    if objects:
        oa = o.add_object((14, 22, 16, 26, 40, 70), 1.8, kind="anisotropic")
        oa.inverse_permittivity += off
        o.add_object((20, 28, 14, 20, 60, 90), 1.5, kind="absorbing", conductivity=3e-3)
        o.add_object((2, 12, 30, 40, 20, 50), 2.5)
    for axis, side, t in _FACES[faces]:
        o.add_pml(axis, side, t)
    for cell, period in sources[:8]:
        o.add_point_source(cell, period=period)
    o.run(steps)
    assert rel_l2(host(a.E), o.E) <= TOL[dtype] and rel_l2(host(a.H), o.H) <= TOL[dtype], (shape, faces)


def test_two_steps_per_pass_with_pml_faces():
    """FDTDHIP_FUSED=2 gives grids of any size the marching kernels (read once per process,
    hence the child)."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (f"import sys\nsys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
            "import test_march2 as t\n"
            "n = 0\n"
            "for dtype, shape in (('float64', (40, 44, 130)), ('float32', (33, 50, 70)), ('float32', (46, 41, 600))):\n"
            "    for faces in ('six-faces', 'four-faces', 'x-faces'):\n"
            "        for steps in (4, 9):\n"
            "            t._pml_case(dtype, shape, faces, steps, seed=n)\n"
            "            n += 1\n"
            "for dtype, shape in (('float64', (40, 44, 130)), ('float32', (46, 50, 600))):\n"
            "    for faces in ('six-faces', 'four-faces', 'no-faces'):\n"
            "        t._pml_case(dtype, shape, faces, 9, seed=n, objects=True)\n"
            "        n += 1\n"
            "print('two steps per pass with faces:', n, 'cases')\n")
    done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDTDHIP_FUSED="2"),
                          capture_output=True, text=True, timeout=280, cwd=here)
    assert done.returncode == 0 and "two steps per pass with faces" in done.stdout, \
        done.stdout[-2000:] + done.stderr[-3000:]


def test_random_plans_take_two_steps_per_pass_and_equal_half_steps_bitwise():
    """Random plans (tools/fuzz_pairs.py: any subset of the faces with PMLs of 1-7 cells, up
    to three objects of the plain kinds anywhere, up to eight source points, line and block
    detectors, runs split in two calls) stepped two steps per pass, one step per pass and by
    explicit half steps: fields, memory variables and detector readings bit for bit."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    tool = os.path.join(os.path.dirname(here), "tools", "fuzz_pairs.py")
    for first, count, dtype in ((100, 30, "float32"), (200, 20, "float64")):
        done = subprocess.run([sys.executable, tool, str(first), str(count), dtype],
                              env=dict(os.environ, FDTDHIP_FUSED="2"), capture_output=True, text=True,
                              timeout=280, cwd=here)
        assert done.returncode == 0 and " 0 differ" in done.stdout, done.stdout[-2000:] + done.stderr[-3000:]
        # (nearly every such plan allows pairs; the count keeps the comparison from being empty)
        took = int(done.stdout.rsplit("differ,", 1)[1].split()[0])
        assert took >= count * 3 // 4, done.stdout[-500:]


def test_line_and_plane_sources_with_whole_steps_at_the_default_policy(monkeypatch):
    """Sources of more than eight points and plane sources (which set E and H on their plane,
    reference fdtd/sources.py:515-525) are applied by launches of their own between the half
    steps; the box that holds their cells is stepped by the shell launches next to the marching
    kernels.  A grid of the default policy's size with three PML faces, a periodic boundary
    along y with source points on both of its planes, a line source that starts in a face, a
    plane source and a point source next to it (their common box is 13 planes of 232): two
    steps per pass, one step per pass and explicit half steps, bit for bit."""
    monkeypatch.delenv("FDTDHIP_FUSED", raising=False)
    fdtd.set_backend("hip.float32")
    shape = (232, 128, 256)
    lib = _lib.load()
    rs = np.random.RandomState(3)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)

    def build():
        g = fdtd.Grid(shape)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        g[0:6, :, :] = fdtd.PML()
        g[:, :, 0:7] = fdtd.PML()
        g[:, :, -7:] = fdtd.PML()
        # (a periodic boundary along y: its two planes are stepped by the shell launches and
        # copied onto each other by a launch after each half step, before the sources)
        g[:, 0, :] = fdtd.PeriodicBoundary()
        # (all sources of this plan share one box: those on the boundary's planes stay near the others)
        g[10, 0, 100] = fdtd.PointSource(period=7)                 # on the first plane of the boundary
        g[8, 127, 105] = fdtd.PointSource(period=11)               # ... and on the last
        g[3:15, 70, 100] = fdtd.LineSource(period=13)              # starts inside the x-low face
        g[12, :, :] = fdtd.PlaneSource(period=17, polarization="z")
        g[14, 75, 110] = fdtd.PointSource(period=9)                # two cells from the plane
        det = fdtd.LineDetector()
        g[0:231, 72, 101] = det
        return g, det

    runs = {}
    for way, env in (("pairs", None), ("single", "0"), ("half", None)):
        if env is not None:
            monkeypatch.setenv("FDTDHIP_MARCH2", env)
        before = lib.fdtdhip_two_step_launches()
        g, det = build()
        if way == "half":
            for _ in range(11):
                g.update_E()
                g.update_H()
                g.time_steps_passed += 1
        else:
            g.run(6, progress_bar=False)
            g.run(5, progress_bar=False)
        torch.cuda.synchronize()
        if env is not None:
            monkeypatch.delenv("FDTDHIP_MARCH2")
        taken = lib.fdtdhip_two_step_launches() - before
        assert taken == (4 if way == "pairs" else 0), (way, taken)
        if way == "single":      # whole steps by the one-step marching kernel
            assert _lib.check(lib.fdtdhip_plan_whole_step_kind(g._plan, 0, None, None)) == 2
        runs[way] = [g.E, g.H, torch.stack(list(det.E)), torch.stack(list(det.H))] + \
            [t for b in g.boundaries if hasattr(b, "thickness") for t in (b._psiE, b._psiH)]
    assert float(runs["half"][0].abs().max()) > 0
    for way in ("pairs", "single"):
        for n, (a, b) in enumerate(zip(runs[way], runs["half"])):
            assert torch.equal(a, b), (way, n)
