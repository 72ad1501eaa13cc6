"""Seeded random scenes in the format of ``tests/scenes.py``: random shapes (1-D to 3-D,
ragged extents), boundaries on random faces in random registration order, overlapping
objects of every kind, sources and detectors of every kind.  Every scene is valid input
for the reference's API (no construction errors) and short enough to stay finite.

Used by ``tools/fuzz_reference.py`` (reference against the oracle, on the CPU where the
reference checkout exists) and ``tools/fuzz_parity.py`` / ``tests/test_fuzz.py`` (HIP
engine against the oracle, on the GPU).
"""

import numpy as np

S = slice


def _box(rs, shape, max_frac=0.6):
    """Random non-empty box inside the grid as three slices."""
    key = []
    for n in shape:
        if n == 1:
            key.append(S(0, 1))
            continue
        w = int(rs.randint(1, max(2, int(n * max_frac) + 1)))
        a = int(rs.randint(0, n - w + 1))
        key.append(S(a, a + w))
    return tuple(key)


def random_scene(seed, big=False, extras=False, floats=False, crowded=False, thin_x=0):
    """crowded: 8-24 objects, 9-14 sources and 9-13 detectors instead of up to 3 / 3 / 2 (more
    than any fixed-size batch or table inside the engine holds).  thin_x = n: n to n + 2 planes
    along x (slabs of one or two planes when split over n ranks)."""
    rs = np.random.RandomState(seed)
    dims = rs.choice([1, 2, 3], p=[0.1, 0.4, 0.5])
    lo, hi = (9, 30) if not big else (40, 90)
    ext = [int(rs.randint(lo, hi)) for _ in range(3)]
    flat = list(rs.permutation(3)[: 3 - dims])
    for ax in flat:
        ext[ax] = 1
    if thin_x:
        ext[0] = thin_x + seed % 3
    if big and dims == 3:
        ext[2] = int(rs.choice([64, 65, 96, 128, 70]))
    shape = tuple(ext)
    grid = dict(shape=shape)
    if rs.rand() < 0.4:
        grid["permittivity"] = float(1.0 + rs.rand())
    if rs.rand() < 0.3:
        grid["permeability"] = float(1.0 + 0.5 * rs.rand())

    def cell():
        return tuple(int(rs.randint(0, n)) for n in shape)

    # ---- boundaries: per axis nothing / periodic / PML on one or both faces
    bnd = []
    for ax in range(3):
        n = shape[ax]
        full = [S(None)] * 3
        choice = rs.choice(["none", "periodic", "pml_lo", "pml_hi", "pml_both"],
                           p=[0.15, 0.2, 0.15, 0.15, 0.35])
        if n == 1:
            choice = rs.choice(["none", "periodic"])
        if choice == "periodic":
            key = list(full)
            key[ax] = 0
            bnd.append(("periodic", tuple(key), {}))
        elif choice.startswith("pml"):
            tmax = max(1, min(8, n // 2 - 1))
            for side in ("lo", "hi"):
                if choice in ("pml_" + side, "pml_both"):
                    t = int(rs.randint(1, tmax + 1))
                    key = list(full)
                    key[ax] = S(0, t) if side == "lo" else S(-t, None)
                    kw = {"a": float(10 ** rs.uniform(-9, -5))} if rs.rand() < 0.3 else {}
                    bnd.append(("pml", tuple(key), kw))
    order = rs.permutation(len(bnd))
    items = [bnd[i] for i in order]

    # ---- objects (may overlap each other and the PMLs)
    n3 = [n for n in shape if n > 1]
    for _ in range(int(rs.randint(8, 25)) if crowded else int(rs.randint(0, 4))):
        kind = rs.choice(["object", "absorbing", "anisotropic"])
        key = _box(rs, shape, 0.3) if crowded else _box(rs, shape)
        kw = dict(permittivity=float(1.2 + 2.5 * rs.rand()))
        extra = {}
        if kind == "object" and rs.rand() < 0.4:
            bs = tuple(k.stop - k.start for k in key)
            if rs.rand() < 0.5:
                bs = bs + (3,)      # per component; else per cell
            kw["permittivity"] = ("randpos", int(rs.randint(1 << 30)), bs)
        if kind == "absorbing":
            kw["conductivity"] = float(10 ** rs.uniform(-4, 1))
        if kind == "anisotropic" and rs.rand() < 0.7:
            extra = dict(offdiag_seed=int(rs.randint(1 << 30)), offdiag_scale=0.05)
        items.append((kind, key, kw, extra) if extra else (kind, key, kw))

    # ---- sources
    steps = int(rs.randint(4, 22))
    if extras and rs.rand() < 0.5:
        # a source fed back from its own loop-current detector (reference
        # fdtd/sources.py:708-737) and a direction-vector point source (639-700)
        if rs.rand() < 0.6:
            n = steps + int(rs.randint(0, 5))
            items.append(("soft", cell(), dict(
                waveform_array=("gauss", n, float(rs.uniform(3, n)), float(rs.uniform(3, 9)), 1e-3),
                impedance=float(rs.choice([0.0, 0.1, 0.3])))))
        else:
            items.append(("ccpoint", cell(), dict(dir_vec=tuple(float(v) for v in rs.randn(3)))))
    for _ in range(int(rs.randint(9, 15)) if crowded else int(rs.randint(1, 4))):
        kind = rs.choice(["point", "line", "plane"], p=[0.5, 0.3, 0.2])
        kw = dict(period=int(rs.randint(5, 25)), amplitude=float(0.5 + rs.rand()))
        if rs.rand() < 0.3:
            kw["phase_shift"] = float(rs.rand())
        if kind == "point":
            if rs.rand() < 0.3:
                kw.update(pulse=True, cycle=int(rs.randint(1, 4)))
                if rs.rand() < 0.5:
                    kw["hanning_dt"] = float(rs.uniform(2.0, 12.0))
            items.append(("point", cell(), kw))
        elif kind == "line" and len(n3) >= 1:
            a, b = cell(), cell()
            if max(abs(p - q) for p, q in zip(a, b)) < 2:   # at least two grid points
                continue
            key = tuple(S(min(p, q), max(p, q)) if p != q else p for p, q in zip(a, b))
            if rs.rand() < 0.3:
                kw.update(pulse=True, cycle=int(rs.randint(1, 4)))
                if rs.rand() < 0.5:
                    kw["hanning_dt"] = float(rs.uniform(2.0, 12.0))
            items.append(("line", key, kw))
        elif kind == "plane" and len(n3) == 3:
            # a plane source is exactly one cell thick along exactly one axis (reference
            # fdtd/sources.py:491-505): only true 3-D grids have room for one
            axes = [ax for ax in range(3) if shape[ax] > 1]
            ax = int(rs.choice(axes))
            key = [S(None)] * 3
            key[ax] = int(rs.randint(0, shape[ax]))
            pol = [p for p in "xyz" if p != "xyz"[ax]]
            kw["polarization"] = str(rs.choice(pol))
            items.append(("plane", tuple(key), kw))

    # ---- detectors
    for _ in range(int(rs.randint(9, 14)) if crowded else int(rs.randint(0, 3))):
        kind = rs.choice(["linedet", "blockdet", "currentdet"])
        if kind == "linedet":
            a, b = cell(), cell()
            if max(abs(p - q) for p, q in zip(a, b)) == 0:
                continue
            key = tuple(S(min(p, q), max(p, q)) if p != q else p for p, q in zip(a, b))
            items.append(("linedet", key, {}))
        elif kind == "blockdet":
            # the reference reads the block with inclusive stops: keep one cell of margin
            key = []
            for n in shape:
                if n < 3:
                    key.append(0)
                    continue
                a = int(rs.randint(0, n - 2))
                key.append(S(a, int(rs.randint(a + 1, n - 1))))
            items.append(("blockdet", tuple(key), {}))
        else:
            items.append(("currentdet", cell(), {}))
    scene = dict(grid=grid, items=items, steps=steps)
    return with_float_keys(scene, rs) if floats else scene


def with_float_keys(scene, rs, spacing=155e-9):
    """The same scene with some coordinates given in metres (floats), as users of the API
    write them (reference fdtd/grid.py:171-210): grid extents, object / source / detector
    positions and slice ends, source periods in seconds."""
    def f(c, lo=-0.45, hi=0.45):
        return float((c + rs.uniform(lo, hi)) * spacing)

    shape = tuple(f(n) if n > 1 and rs.rand() < 0.3 else n for n in scene["grid"]["shape"])
    dims = max(1, sum(1 for n in scene["grid"]["shape"] if n > 1))
    time_step = 0.99 * dims ** -0.5 * spacing / 299792458.0   # reference fdtd/grid.py:119-132
    items = []
    for item in scene["items"]:
        kind, key, kw = item[0], item[1], dict(item[2])
        if isinstance(kw.get("period"), int) and rs.rand() < 0.3:
            kw["period"] = float((kw["period"] + rs.uniform(-0.3, 0.3)) * time_step)
        if kind not in ("pml", "periodic") and rs.rand() < 0.6:
            new = []
            for k in key:
                if isinstance(k, slice) and k.start is not None and k.stop is not None and rs.rand() < 0.7:
                    new.append(slice(f(k.start), f(k.stop)) if k.stop > k.start + 0 else k)
                elif isinstance(k, int) and not isinstance(k, bool) and rs.rand() < 0.7:
                    new.append(f(k, 0.0, 0.45) if k == 0 else f(k))
                else:
                    new.append(k)
            key = tuple(new)
        items.append((kind, key, kw) + tuple(item[3:]))
    return dict(grid=dict(scene["grid"], shape=shape), items=items, steps=scene["steps"])


def random_substrate_scene(seed):
    """A random 2-D scene (z extent 1) with an energy-modulated substrate object
    (LearnableAnisotropicObject(is_substrate=True), reference fdtd/objects.py:276-349) added:
    the reference's modifier only works for boxes one cell thick along z."""
    for k in range(200):
        scene = random_scene(seed * 200 + k)
        nx, ny, nz = scene["grid"]["shape"]
        if nz == 1 and nx > 3 and ny > 3:
            break
    rs = np.random.RandomState(seed + 31337)
    box = _box(rs, (nx, ny, 1), max_frac=0.8)
    scene["items"].insert(int(rs.randint(len(scene["items"]) + 1)), (
        "substrate", (box[0], box[1], S(None)), dict(permittivity=float(1.3 + 2.0 * rs.rand())),
        dict(offdiag_seed=int(rs.randint(1 << 30)), offdiag_scale=0.04,
             conv_seed=int(rs.randint(1 << 30)), conv_scale=float(10 ** rs.uniform(0, 2.5)))))
    return scene
