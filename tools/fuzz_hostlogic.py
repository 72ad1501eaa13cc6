"""Coordinate handling of the user-facing API, reference against fdtd_amd, on random scenes
with keys given in metres (tests/fuzz_scenes.py, floats=True).  Runs on the CPU in the
development container (the reference is imported; fdtd_amd is used without a GPU through its
host-staging device):

    python tools/fuzz_hostlogic.py [first_seed] [count] [/root/reference]

For every scene both sides must resolve every component to the same integer cells (boxes,
point lists, PML faces and thicknesses, polarisations, periods in steps), build the same
object tensors, leave the same inverse permittivity on the grid and print the same
``str(grid)`` -- or raise the same exception type.
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import fdtd as ref  # noqa: E402  (the real reference)

import fdtd_amd  # noqa: E402
import fuzz_scenes  # noqa: E402
import scenes  # noqa: E402

ref.set_backend("numpy")
fdtd_amd.set_backend("hip.float64", device="host-staging")


def build(mod, scene):
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            grid, made = scenes.build_api(scene, mod)
            if mod is ref:
                # some mistakes only surface in the reference's first step (an object box
                # beyond the grid: fdtd/objects.py:116-126); fdtd_amd refuses them at
                # registration with an IndexError
                probe, _ = scenes.build_api(scene, mod)
                try:
                    probe.step()
                except ValueError:       # broadcasting: an object box beyond the grid
                    return "IndexError", None, None, None
                except IndexError:
                    # a source or detector cell beyond the grid: fdtd_amd raises the same
                    # IndexError when its plan is built at the first step (Grid._cell_index;
                    # tests/test_hip_parity.py), which this GPU-less comparison never reaches
                    return "first-step IndexError", None, None, None
        except Exception as e:  # noqa: BLE001 -- the exception type is what is compared
            return type(e).__name__, None, None, None
    resolved, arrays = scenes.resolve(scene, grid, made)
    inv = grid.inverse_permittivity
    inv = inv if isinstance(inv, np.ndarray) else inv.detach().cpu().numpy()
    objs = []
    for kind, t, _, _ in made:
        if kind in ("object", "absorbing", "anisotropic"):
            m = t.inverse_permittivity
            objs.append(m if isinstance(m, np.ndarray) else m.detach().cpu().numpy())
    return None, resolved, (inv, objs), (float(grid.time_step), str(grid), [repr(t) for _, t, _, _ in made])


S = slice


def faulty_item(rs, shape):
    """One registration the API has to refuse (or, for some, accept) -- both sides must
    react with the same exception type."""
    nx, ny, nz = shape
    return [
        ("pml", (S(2, 5), S(None), S(None)), {}),                    # not at a border
        ("pml", (S(0, 3), S(None), S(None)), {}),                    # maybe a second one on x-low
        ("pml", (S(None), S(None), S(None)), {}),                    # no face at all
        ("periodic", (S(0, 2), S(None), S(None)), {}),               # a slice, not an index
        ("periodic", (3, S(None), S(None)), {}),                     # mid-grid
        ("periodic", (S(None), -1, S(None)), {}),                    # maybe a second one on y
        ("object", (S(0, nx + 3), S(0, 1), S(0, 1)), dict(permittivity=2.0)),   # beyond the grid
        ("object", (S(1, 2), S(0, 1), S(0, 1)), dict(permittivity=2.0, name="E")),  # name taken
        ("point", (nx + 2, 0, 0), dict(period=9)),                   # outside
        ("line", (S(1, 1), 0, 0), dict(period=9)),                   # a single point
        ("plane", (0, 0, S(None)), dict(period=9, polarization="z")),   # flat in two directions
        ("plane", (0, S(None), S(None)), dict(period=9, polarization="x")),  # perpendicular
        ("linedet", ([0, 1], 0, 0), {}),                             # a list key
        ("blockdet", (S(0, 1), S(0, 1), S(0, 1)), {}),
    ][int(rs.randint(14))]


bad = errors = 0
for seed in range(first, first + count):
    scene = fuzz_scenes.random_scene(seed, extras=True, floats=True)
    rs = np.random.RandomState(seed + 12345)
    item = None
    if rs.rand() < 0.4:
        item = faulty_item(rs, [n if isinstance(n, int) else int(n / 155e-9 + 0.5) for n in scene["grid"]["shape"]])
        scene["items"].insert(int(rs.randint(len(scene["items"]) + 1)), item)
    re_, rres, rarr, rdt = build(ref, scene)
    ae_, ares, aarr, adt = build(fdtd_amd, scene)
    what = None
    if re_ or ae_:
        errors += 1
        if re_ == "first-step IndexError" and ae_ is None:
            pass
        elif re_ != ae_:
            what = f"exceptions differ: reference {re_}, fdtd_amd {ae_}"
    elif scenes.dumps(rres) != scenes.dumps(ares):
        what = f"resolved coordinates differ:\n  ref {scenes.dumps(rres)}\n  amd {scenes.dumps(ares)}"
    elif rdt != adt:
        what = f"time step or printed form differs:\n  ref {rdt}\n  amd {adt}"
    elif not np.array_equal(rarr[0], aarr[0]) or any(
            not np.array_equal(a, b) for a, b in zip(rarr[1], aarr[1])):
        what = "material arrays differ"
    if what:
        bad += 1
        print(f"seed {seed}: {what}\n   inserted: {item}\n   {scene}", flush=True)
print(f"{count} scenes: {bad} differ ({errors} raised on both or either side)")
sys.exit(1 if bad else 0)
