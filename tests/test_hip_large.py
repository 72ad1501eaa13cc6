"""The large-grid kernels (the ones bench.py measures) against the oracle and against
each other, on a real MI355X.

The library picks its kernels by plane size: grids whose (y, z) plane has 1024 cells
or more run the unconditional-load kernels (row lane mapping with rotated z segments
when the rows fill their 64-lane segments -- the 128- and 192-cell rows below --, flat
mapping otherwise; non-temporal access), smaller ones the shape-agnostic kernels the golden scenes exercise.  Here
the grids are large enough for the former and small enough for the numpy oracle to
follow in seconds; at the benchmark's full size (512^3) the checks are
size-independent properties: the constant-material and array-material paths agree
bit for bit, and a source moved by whole cells moves its field with it.

Tolerances: float64 relative L2 <= 1e-12, float32 <= 1e-5 (against the float64 oracle).
"""

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd
from fdtd_amd import _lib
from oracle import yee_numpy as orc
from test_hip_parity import TOL, host, rel_l2

pytestmark = pytest.mark.gpu


def _build(shape, rs, dense, periodic_y=False):
    """The same scene in the engine and in the oracle.  dense: random material arrays
    (the kernels must read them); otherwise homogeneous ones (read as constants)."""
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    nx, ny, nz = shape
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    o.E[...], o.H[...] = E0, H0
    g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
    if dense:
        ie, im = 1.0 / (1 + rs.rand(*shape, 3)), 1.0 / (1 + 0.2 * rs.rand(*shape, 3))
        o.inverse_permittivity[...], o.inverse_permeability[...] = ie, im
        g.inverse_permittivity = fdtd.backend.array(ie)
        g.inverse_permeability = fdtd.backend.array(im)
    faces = [(0, "low", 5), (0, "high", 4), (2, "low", 6), (2, "high", 5)]
    if not periodic_y:
        faces += [(1, "low", 4), (1, "high", 6)]
    for axis, side, t in faces:
        key = [slice(None)] * 3
        key[axis] = slice(0, t) if side == "low" else slice(-t, None)
        g[tuple(key)] = fdtd.PML()
        o.add_pml(axis, side, t)
    if periodic_y:
        g[:, 0, :] = fdtd.PeriodicBoundary()
        o.add_periodic(1)
    box = (7, 15, 9, 30, 8, 25)
    g[7:15, 9:30, 8:25] = fdtd.AnisotropicObject(permittivity=2.5)
    oa = o.add_object(box, 2.5, kind="anisotropic")
    off = 0.02 * rs.randn(8, 21, 17, 3, 3)
    oa.inverse_permittivity += off
    with torch.no_grad():
        g.objects[0].inverse_permittivity += fdtd.backend.array(off)
    g[12:20, 25:40, 20:28] = fdtd.AbsorbingObject(permittivity=1.8, conductivity=2e-3)
    o.add_object((12, 20, 25, 40, 20, 28), 1.8, kind="absorbing", conductivity=2e-3)
    src = (nx // 2, ny // 2, nz // 2)
    g[src] = fdtd.PointSource(period=11)
    o.add_point_source(src, period=11)
    return g, o


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("dense", [False, True], ids=["const-materials", "array-materials"])
@pytest.mark.parametrize("shape,periodic_y", [((26, 70, 66), False), ((24, 150, 30), False),
                                              ((25, 66, 67), True), ((22, 72, 128), False),
                                              ((21, 70, 192), True)])
def test_large_plane_kernels_against_oracle(shape, periodic_y, dense, dtype):
    fdtd.set_backend(f"hip.{dtype}")
    assert shape[1] * shape[2] >= 4096
    g, o = _build(shape, np.random.RandomState(5), dense, periodic_y)
    g.run(15, progress_bar=False)
    o.run(15)
    torch.cuda.synchronize()
    found = _lib.load().fdtdhip_plan_scan_materials(g._plan, 1)
    assert found == (0 if dense else 3), "material scan outcome"
    assert rel_l2(host(g.E), o.E) <= TOL[dtype]
    assert rel_l2(host(g.H), o.H) <= TOL[dtype]
    for b, ob in zip(g.boundaries, o.boundaries):
        if isinstance(ob, orc.OraclePML):
            # the two memory variables per field that are ever non-zero
            u, v = (ob.axis + 1) % 3, (ob.axis + 2) % 3
            for mine, fam in ((b._psiE, ob.psi_E), (b._psiH, ob.psi_H)):
                want = np.stack([fam[u][..., ob.axis], fam[v][..., ob.axis]])
                assert rel_l2(host(mine).reshape(want.shape), want) <= TOL[dtype]


def test_material_write_is_noticed():
    """An in-place write to a scanned material array is seen before the next step."""
    fdtd.set_backend("hip.float64")
    shape = (12, 70, 64)
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    g[6, 35, 32] = fdtd.PointSource(period=9)
    o.add_point_source((6, 35, 32), period=9)
    g.run(6, progress_bar=False); o.run(6)
    g.inverse_permittivity[5:8, 30:40, 28:36, :] = 0.4
    o.inverse_permittivity[5:8, 30:40, 28:36, :] = 0.4
    g.inverse_permeability[4:9, 33:37, 30:34, 1] *= 0.9
    o.inverse_permeability[4:9, 33:37, 30:34, 1] *= 0.9
    g.run(6, progress_bar=False); o.run(6)
    assert rel_l2(host(g.E), o.E) <= 1e-12 and rel_l2(host(g.H), o.H) <= 1e-12
    # writes torch does not count need the explicit notice
    g.inverse_permittivity.data[2:4, 10:20, 10:20, :] = 0.7
    o.inverse_permittivity[2:4, 10:20, 10:20, :] = 0.7
    g.materials_changed()
    g.run(4, progress_bar=False); o.run(4)
    assert rel_l2(host(g.E), o.E) <= 1e-12 and rel_l2(host(g.H), o.H) <= 1e-12


def test_material_write_that_torch_cannot_see_is_caught_by_the_next_long_run():
    """A write through ``tensor.data`` bumps no version counter.  While constants are in use
    the grid scans its material arrays again at the start of every call that advances 16
    steps or more (and of shorter calls once 16 steps have gone by since the last scan), so
    such a run gives the reference's fields without ``materials_changed()``."""
    fdtd.set_backend("hip.float64")
    shape = (12, 70, 64)
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    g[6, 35, 32] = fdtd.PointSource(period=9)
    o.add_point_source((6, 35, 32), period=9)
    g.run(20, progress_bar=False); o.run(20)
    assert _lib.load().fdtdhip_plan_scan_materials(g._plan, 1) == 3    # constants in use
    g.inverse_permittivity.data[5:8, 30:40, 28:36, :] = 0.4
    o.inverse_permittivity[5:8, 30:40, 28:36, :] = 0.4
    g.inverse_permeability.data[4:9, 33:37, 30:34, 1] *= 0.9
    o.inverse_permeability[4:9, 33:37, 30:34, 1] *= 0.9
    g.run(16, progress_bar=False); o.run(16)
    assert rel_l2(host(g.E), o.E) <= 1e-12 and rel_l2(host(g.H), o.H) <= 1e-12
    # single steps: noticed once 16 of them have gone by since the last scan (grids of fewer
    # than 2**17 cells, whose scan costs as much as fifteen steps: once 256 have gone by)
    for shape, patience in (((32, 70, 64), 16), (shape, 256)):
        h = fdtd.Grid(shape)
        h[6, 35, 32] = fdtd.PointSource(period=9)
        h.run(3, progress_bar=False)
        assert h._materials_mask == 3
        h.inverse_permittivity.data[5:8, 30:40, 28:36, :] = 0.4
        for _ in range(patience - 3):
            h.run(1, progress_bar=False)
        assert h._materials_mask == 3
        for _ in range(2):
            h.run(1, progress_bar=False)
        assert h._materials_mask == 2      # the inverse permittivity is read from memory again


def _full_size_grid(n, source, t=10):
    g = fdtd.Grid((n, n, n))
    for axis in range(3):
        lo, hi = [slice(None)] * 3, [slice(None)] * 3
        lo[axis], hi[axis] = slice(0, t), slice(-t, None)
        g[tuple(lo)] = fdtd.PML()
        g[tuple(hi)] = fdtd.PML()
    g[source] = fdtd.PointSource(period=14)
    return g


def test_full_size_material_paths_agree_bitwise():
    """512^3 float32 (the benchmark grid): homogeneous materials read as constants and
    the same arrays read from memory give identical fields."""
    fdtd.set_backend("hip.float32")
    n, steps, src = 512, 8, (256, 250, 261)
    a = _full_size_grid(n, src)
    a.run(steps, progress_bar=False)
    lib = _lib.load()
    assert lib.fdtdhip_plan_scan_materials(a._plan, 1) == 3
    b = _full_size_grid(n, src)
    b._ensure_plan()
    b._material_scans = b._MAX_MATERIAL_SCANS + 1   # keep the arrays in use
    assert lib.fdtdhip_plan_scan_materials(b._plan, 0) == 0
    b.run(steps, progress_bar=False)
    torch.cuda.synchronize()
    assert float(a.E.abs().max()) > 0
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_full_size_translation_invariance(dtype):
    """Away from the boundaries the update is the same at every cell: moving the source
    by whole cells moves the field bit for bit.  The 512^3 x 3 arrays hold 4.0e8
    elements; in float64 their byte offsets pass 2^31."""
    fdtd.set_backend(f"hip.{dtype}")
    n = 512
    steps, w = 12, 16
    s1, s2 = (n // 2 - 60, n // 2 - 40, n // 2 - 70), (n // 2 + 90, n // 2 + 75, n // 2 + 101)
    fields = []
    for s in (s1, s2):
        g = _full_size_grid(n, s)
        g.run(steps, progress_bar=False)
        win = tuple(slice(c - w, c + w) for c in s)
        fields.append((g.E[win].clone(), g.H[win].clone()))
        del g
        torch.cuda.empty_cache()
    assert float(fields[0][0].abs().max()) > 0
    assert torch.equal(fields[0][0], fields[1][0]) and torch.equal(fields[0][1], fields[1][1])


def test_offsets_beyond_2_to_31_elements():
    """1024 x 1024 x 768 float32: 8.1e8 cells, so the element offsets of the field arrays
    (3 per cell) pass 2^31 from plane 911 on.  A source, a detector and the update itself are
    exercised beyond that plane: the field around a source at plane 1000 equals, bit for bit,
    the field around a source at plane 40, and the detector's last reading equals the field
    at its cells."""
    fdtd.set_backend("hip.float32")
    shape, steps, w = (1024, 1024, 768), 8, 8
    s1, s2 = (40, 300, 200), (1000, 700, 500)
    assert 3 * (s2[0] - w) * shape[1] * shape[2] > 2 ** 31
    fields = []
    for s in (s1, s2):
        g = fdtd.Grid(shape)
        for axis in range(3):
            lo, hi = [slice(None)] * 3, [slice(None)] * 3
            lo[axis], hi[axis] = slice(0, 10), slice(-10, None)
            g[tuple(lo)] = fdtd.PML()
            g[tuple(hi)] = fdtd.PML()
        g[s] = fdtd.PointSource(period=14)
        det = fdtd.LineDetector()
        g[s[0] - 3:s[0] + 4, s[1] + 1, s[2] + 2] = det
        g.run(steps, progress_bar=False)
        win = tuple(slice(c - w, c + w) for c in s)
        fields.append((g.E[win].clone(), g.H[win].clone()))
        last_E, last_H = det.E[-1], det.H[-1]
        assert torch.equal(last_E, g.E[s[0] - 3:s[0] + 4, s[1] + 1, s[2] + 2])
        assert torch.equal(last_H, g.H[s[0] - 3:s[0] + 4, s[1] + 1, s[2] + 2])
        assert float(last_E.abs().max()) > 0
        # nothing was written elsewhere: the field is zero outside the light cone
        far = g.E[s[0] - 3 * w:s[0] - 2 * w] if s[0] > 500 else g.E[s[0] + 2 * w:s[0] + 3 * w]
        assert float(far.abs().max()) == 0
        del g, det, far
        torch.cuda.empty_cache()
    assert torch.equal(fields[0][0], fields[1][0]) and torch.equal(fields[0][1], fields[1][1])


@pytest.mark.parametrize("shape", [(70001, 3, 5), (2, 262201, 17), (3, 2, 300007), (66000, 70, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_extreme_aspect_ratios_against_oracle(shape):
    """One extent beyond what a launch grid dimension holds (65535 planes, 65535 row quads)
    or far beyond the others: the library has to pick a kernel and a grid that cover such a
    shape.  PMLs on the long axis, an object, sources near both ends, 6 steps, float64
    against the oracle."""
    fdtd.set_backend("hip.float64")
    long = int(np.argmax(shape))
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    for side, key in (("low", slice(0, 6)), ("high", slice(-6, None))):
        loc = [slice(None)] * 3
        loc[long] = key
        g[tuple(loc)] = fdtd.PML()
        o.add_pml(long, side, 6)
    box = [(0, min(2, n)) for n in shape]
    box[long] = (shape[long] - 40, shape[long] - 9)
    g[tuple(slice(a, b) for a, b in box)] = fdtd.Object(permittivity=2.0)
    o.add_object(tuple(v for ab in box for v in ab), 2.0)
    for pos in (8, shape[long] // 2, shape[long] - 8):
        cell = [n // 2 for n in shape]
        cell[long] = pos
        g[tuple(cell)] = fdtd.PointSource(period=7)
        o.add_point_source(tuple(cell), period=7)
    g.run(6, progress_bar=False)
    o.run(6)
    assert np.abs(o.E).max() > 0
    assert rel_l2(host(g.E), o.E) <= 1e-12 and rel_l2(host(g.H), o.H) <= 1e-12


def test_large_grid_kernels_on_small_ragged_shapes():
    """FDTDHIP_FORCE=1 makes the library use its large-grid kernels whatever the plane
    size: every golden scene (2-D, 1-D, ragged extents, all boundary orders) must still
    match the reference.  The switch is read once per process, hence the child process."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FDTDHIP_FORCE="1")
    done = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here, "test_hip_parity.py"), "-q", "-x",
         "-m", "gpu", "-k", "scene_matches_reference", "-p", "no:cacheprovider"],
        env=env, capture_output=True, text=True, timeout=280, cwd=os.path.dirname(here))
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert " passed" in done.stdout


@pytest.mark.parametrize("fused", ["1", "2", "0"])
def test_one_launch_step_forced_on_and_off(fused):
    """By default the library takes a one-launch step for small 2-D grids (one lane per
    cell) and for large 3-D grids (workgroups marching along x) only.  FDTDHIP_FUSED=1
    applies the former wherever the plan allows (3-D scenes too), FDTDHIP_FUSED=2 the latter
    (2-D, 1-D and ragged scenes too), FDTDHIP_FUSED=0
    neither: every golden scene must match the reference either way, and so must the
    differentiable runs (the one-launch step records their curl traces; the paired adjoint
    launches follow the same switch).  Read once per process, hence the child process."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FDTDHIP_FUSED=fused)
    done = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here, "test_hip_parity.py"),
         os.path.join(here, "test_adjoint.py"), "-q", "-x", "-m", "gpu",
         "-k", "scene_matches_reference or gradients_match_reference_autograd or finite_differences",
         "-p", "no:cacheprovider"],
        env=env, capture_output=True, text=True, timeout=280, cwd=os.path.dirname(here))
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert " passed" in done.stdout


@pytest.mark.parametrize("mode", ["0", "2"])
def test_adjoint_with_the_direct_gather_forced_off_and_on(mode):
    """On large grids the second adjoint pass derives the difference adjoints of cells far
    from PMLs and objects from the field adjoint itself, and the first pass runs over boxes
    that cover the other cells only.  FDTDHIP_ADJ_DIRECT=0 keeps the plain two-pass form
    everywhere, =2 applies the direct form to grids of any
    size: the gradients of the reference's autograd and the finite-difference checks must
    hold either way.  Read once per process, hence the child process."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FDTDHIP_ADJ_DIRECT=mode)
    done = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here, "test_fuzz.py"),
         os.path.join(here, "test_adjoint.py"), "-q", "-x", "-m", "gpu",
         "-k", "adjoint_against_finite or gradients_match_reference_autograd or finite_differences",
         "-p", "no:cacheprovider"],
        env=env, capture_output=True, text=True, timeout=280, cwd=os.path.dirname(here))
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert " passed" in done.stdout


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_long_run_against_oracle(dtype):
    """300 steps of a 64 x 96 x 128 grid (row-mapped large-grid kernels, constant materials,
    six PMLs, an object, a line detector) against the oracle: the tolerances hold after
    many steps, not just a few (float64 <= 1e-12, float32 <= 1e-5 relative L2)."""
    fdtd.set_backend(f"hip.{dtype}")
    shape, steps = (64, 96, 128), 300
    g, o = fdtd.Grid(shape), orc.OracleGrid(shape)
    for axis in range(3):
        for side, key in (("low", slice(0, 8)), ("high", slice(-8, None))):
            loc = [slice(None)] * 3
            loc[axis] = key
            g[tuple(loc)] = fdtd.PML()
            o.add_pml(axis, side, 8)
    g[20:40, 30:60, 40:90] = fdtd.Object(permittivity=2.1)
    o.add_object((20, 40, 30, 60, 40, 90), 2.1)
    g[32, 48, 30] = fdtd.PointSource(period=17)
    o.add_point_source((32, 48, 30), period=17)
    det = fdtd.LineDetector()
    g[10:54, 48, 64] = det
    odet = o.add_line_detector(list(range(10, 54)), [48] * 44, [64] * 44)
    g.run(steps, progress_bar=False)
    o.run(steps)
    torch.cuda.synchronize()
    assert rel_l2(host(g.E), o.E) <= TOL[dtype]
    assert rel_l2(host(g.H), o.H) <= TOL[dtype]
    assert rel_l2(host(det.E), np.array(odet.E)) <= TOL[dtype]


_MARCH_CASES = [("float32", (100, 170, 250)), ("float32", (96, 128, 1024)), ("float64", (99, 170, 260))]
_MARCH_IDS = ["f32-ragged", "f32-1024-cell-rows", "f64-ragged"]
# PML faces of the marching-kernel tests as (axis, side, thickness), in registration order
_MARCH_FACES = {
    "no-pml": [],
    # the order DomainBorderPML registers them in, corners shared by three faces
    "six-faces": [(2, "low", 7), (2, "high", 7), (0, "low", 7), (0, "high", 7), (1, "low", 7), (1, "high", 7)],
    # another order of the axes, unequal thicknesses, faces missing, one a single cell thick
    "four-faces": [(1, "high", 5), (0, "low", 9), (2, "low", 1), (2, "high", 11)],
}


@pytest.mark.parametrize("dtype,shape", _MARCH_CASES, ids=_MARCH_IDS)
def test_whole_step_marching_kernel_equals_half_steps_bitwise(dtype, shape):
    """Large 3-D grids take whole steps with one pass over memory (workgroups
    marching along x, out of place).  The same scene advanced by `run` (marching kernel, an odd
    number of steps so that a last step of two half-step launches follows the pairs) and by
    explicit half steps must give identical bits: fields, detector records; and both must
    match the oracle.  Extents that are no multiples of the tile height, the wave width or the
    chunk length; several sources, random start fields and materials that
    the scan proves constant (float32 ragged, float64) or not (1024-cell rows)."""
    _march_case(dtype, shape, objects=False)


def test_whole_step_marching_kernel_with_faces_and_objects_forced():
    """With PML faces and / or objects the marching kernel covers the box no face reaches,
    minus the E half step of the box that holds the objects; the shell around it and that
    box take out-of-place half steps of their own.  The library picks this while the shell
    is a small part of the grid (the 512^3 tests run it that way); FDTDHIP_FUSED=2 applies it
    to the smaller grids here: same bitwise checks as above, the PML memory variables
    included.  The switch is read once per process, hence the child."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (f"import sys\nsys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
            "import test_hip_large as t\n"
            "for dtype, shape in t._MARCH_CASES:\n"
            "    for faces, objects in (('six-faces', False), ('four-faces', False), ('no-pml', True),\n"
            "                           ('six-faces', True), ('four-faces', True)):\n"
            "        t._march_case(dtype, shape, objects=objects, faces=faces)\n"
            "print('marched with faces and objects')\n")
    done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDTDHIP_FUSED="2"),
                          capture_output=True, text=True, timeout=280, cwd=here)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert "marched with faces and objects" in done.stdout


def _march_case(dtype, shape, objects, faces="no-pml"):
    fdtd.set_backend(f"hip.{dtype}")
    nx, ny, nz = shape
    rs = np.random.RandomState(21)
    E0, H0 = 1e-2 * rs.randn(*shape, 3), 1e-2 * rs.randn(*shape, 3)
    dense = nz == 1024
    ie = 1.0 / (1 + rs.rand(*shape, 3)) if dense else None
    steps = 7
    # (sources on the corners of the grid -- inside the faces, if any -- and under objects too)
    sources = (((nx // 2, ny // 2, nz // 2), 11), ((0, 0, 0), 7), ((nx - 1, ny - 1, nz - 1), 9),
               ((33, 64, 63), 13), ((33, 64, 64), 5))

    def build():
        g = fdtd.Grid(shape)
        g.E, g.H = fdtd.backend.array(E0), fdtd.backend.array(H0)
        if dense:
            g.inverse_permittivity = fdtd.backend.array(ie)
        off = 0.03 * np.random.RandomState(4).randn(25, 40, 90, 3, 3)
        if objects:
            g[10:30, 40:90, 60:150] = fdtd.Object(permittivity=2.1)
            ani = fdtd.AnisotropicObject(permittivity=1.8)
            g[25:50, 20:60, 100:190] = ani
            with torch.no_grad():
                ani.inverse_permittivity += fdtd.backend.array(off)
            g[40:60, 70:120, 30:80] = fdtd.AbsorbingObject(permittivity=1.5, conductivity=3e-3)
            g[2:12, ny - 30:ny - 2, 20:60] = fdtd.Object(permittivity=2.5)   # reaches into faces
        for axis, side, t in _MARCH_FACES[faces]:
            loc = [slice(None)] * 3
            loc[axis] = slice(0, t) if side == "low" else slice(-t, None)
            g[tuple(loc)] = fdtd.PML()
        for cell, period in sources:
            g[cell] = fdtd.PointSource(period=period)
        det = fdtd.LineDetector()
        g[5:65, 64, 64] = det
        return g, det, off

    import ctypes
    import os

    a, da, off = build()
    a._provide_step_buffers()
    lib = a._ensure_plan()
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 1))
    # (one step per pass: this case is about the one-step marching kernel; the switch is read per call)
    before = os.environ.get("FDTDHIP_MARCH2")
    os.environ["FDTDHIP_MARCH2"] = "0"
    try:
        a.run(steps, progress_bar=False)
    finally:
        if before is None:
            del os.environ["FDTDHIP_MARCH2"]
        else:
            os.environ["FDTDHIP_MARCH2"] = before
    msE, nE, msH, nH = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.fdtdhip_plan_get_timing(a._plan, ctypes.byref(msE), ctypes.byref(nE),
                                            ctypes.byref(msH), ctypes.byref(nH)))
    _lib.check(lib.fdtdhip_plan_enable_timing(a._plan, 0))
    # six whole-step launches (counted with the E launches) and one step of two half steps
    assert (nE.value, nH.value) == (steps, 1), "the marching kernel did not run"
    b, db, _ = build()
    for _ in range(steps):
        b.update_E()
        b.update_H()
        b.time_steps_passed += 1
    torch.cuda.synchronize()
    assert float(a.E.abs().max()) > 0
    assert torch.equal(a.E, b.E) and torch.equal(a.H, b.H)
    assert torch.equal(torch.stack(list(da.E)), torch.stack(list(db.E)))
    assert torch.equal(torch.stack(list(da.H)), torch.stack(list(db.H)))
    for pa, pb in zip(a.boundaries, b.boundaries):
        assert torch.equal(pa._psiE, pb._psiE) and torch.equal(pa._psiH, pb._psiH)
        if pa.thickness > 1:   # (a face one cell thick never takes a field difference)
            assert float(pa._psiE.abs().max()) > 0 and float(pa._psiH.abs().max()) > 0
    if nz == 1024:
        return      # (the oracle comparison of the other two shapes covers the arithmetic)
    o = orc.OracleGrid(shape)
    o.E[...], o.H[...] = E0, H0
    if objects:
        o.add_object((10, 30, 40, 90, 60, 150), 2.1)
        oa = o.add_object((25, 50, 20, 60, 100, 190), 1.8, kind="anisotropic")
        oa.inverse_permittivity += off
        o.add_object((40, 60, 70, 120, 30, 80), 1.5, kind="absorbing", conductivity=3e-3)
        o.add_object((2, 12, ny - 30, ny - 2, 20, 60), 2.5)
    for axis, side, t in _MARCH_FACES[faces]:
        o.add_pml(axis, side, t)
    for cell, period in sources:
        o.add_point_source(cell, period=period)
    o.run(steps)
    assert rel_l2(host(a.E), o.E) <= TOL[dtype] and rel_l2(host(a.H), o.H) <= TOL[dtype]
