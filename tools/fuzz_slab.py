"""Random scenes on x-slabs against the single-grid oracle:

    python tools/fuzz_slab.py <world> <first_seed> <count> [gpu|cpu]

`world` ranks over gloo.  gpu: every rank drives the HIP engine on GPU 0 (ghost planes,
plane ranges, the cross-rank x-periodic copy); cpu: the slab host logic over the oracle
engine of tests/slab_support.py.  The gathered E and H and every detector's readings
(each rank's part, joined in rank order) must equal the oracle's single-grid result to
1e-12 (float64).  Scenes whose x-PML is thicker than the thinnest slab are left out (refused
at registration); the CPU runs also leave out those with a SoftArbitraryPointSource, which
the oracle slab engine does not have.
FUZZ_BIG=1: extents of 40-128 cells; FUZZ_CROWDED=1: many objects, sources and detectors;
FUZZ_EXTRAS=1: direction-vector point sources and soft sources in some scenes;
FUZZ_FLOAT32=1: the engine in float32 (GPU runs), tolerance 1e-5; FUZZ_THIN_X=n: n to n + 2
planes along x (slabs of one or two planes on n ranks).
"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_scenes  # noqa: E402
import scenes  # noqa: E402
import slab_support  # noqa: E402

# FUZZ_BIG=1     # extents of 40-128 cells (planes for the fast kernels)
# FUZZ_CROWDED=1
# FUZZ_EXTRAS=1   # direction-vector point sources too
# FUZZ_FLOATS=1   # some keys, extents and periods in metres / seconds
DETECTORS = {"linedet": ("E", "H"), "blockdet": ("E", "H"), "currentdet": ("I",)}


# FUZZ_SUBSTRATE=1   # 2-D scenes with an energy-modulated object (GPU only)


def flag(name):
    """Scene options come from the environment (read at every call, so that a test can set
    them for itself and the ranks it spawns)."""
    return os.environ.get("FUZZ_" + name) == "1"


def make_scene(seed, crowded=None):
    if flag("SUBSTRATE"):
        return fuzz_scenes.random_substrate_scene(seed)
    return fuzz_scenes.random_scene(seed, big=flag("BIG"), extras=flag("EXTRAS"), floats=flag("FLOATS"),
                                    thin_x=int(os.environ.get("FUZZ_THIN_X", "0")),
                                    crowded=flag("CROWDED") if crowded is None else crowded)


def stacked(readings):
    """A detector's list of readings as one array (steps, cells along x or points, ...)."""
    return np.array([np.asarray(r.detach().cpu()) if torch.is_tensor(r) else np.asarray(r)
                     for r in readings], dtype=np.float64)


def usable(scene, world, on_gpu=True):
    nx = scene["grid"]["shape"][0]
    if nx < world:
        return False
    thinnest = nx // world
    if not on_gpu and thinnest == 1 and scene["grid"]["shape"][1] == scene["grid"]["shape"][2] == 1:
        return False   # a slab of one cell: the oracle grid of the CPU runs has no such shape
    for item in scene["items"]:
        kind, key = item[0], item[1]
        if kind == "soft" and not on_gpu:   # not in the oracle slab engine of the CPU runs
            return False
        if kind == "pml" and isinstance(key[0], slice) and (key[0].start, key[0].stop) != (None, None):
            t = key[0].stop if key[0].stop is not None else -key[0].start
            if t > thinnest:
                return False
    return True


def crosses(scene, world):
    """Whether a current detector sits on the first plane of some rank's slab (its loop
    then reads the plane of the rank below)."""
    nx = scene["grid"]["shape"][0]
    base, extra = divmod(nx, world)
    firsts = {r * base + min(r, extra) for r in range(world)}
    return any(item[0] == "currentdet" and item[1][0] in firsts for item in scene["items"])


def worker(rank, world, port, seeds, on_gpu, out, crowded=None):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        if on_gpu:
            fdtd.set_backend("hip.float32" if flag("FLOAT32") else "hip.float64", device=0)
            factory = None
        else:
            fdtd.set_backend("hip.float64", device="host-staging")
            factory = slab_support.OracleSlabEngine
        results = {}
        for seed in seeds:
            scene = make_scene(seed, crowded)
            grid, made = scenes.build_api(
                scene, fdtd, grid_factory=lambda **kw: SlabGrid(engine_factory=factory, **kw))
            half = scene["steps"] // 2
            grid.run(half)
            with torch.no_grad():   # (half steps are not recorded for a backward pass)
                for _ in range(scene["steps"] - half):  # the half-step entry points too
                    grid.update_E()
                    grid.update_H()
                    grid.time_steps_passed += 1
            if on_gpu:
                torch.cuda.synchronize()
            E, H = grid.gather("E"), grid.gather("H")
            mine = {}
            for n, item in enumerate(made):
                for attr in DETECTORS.get(item[0], ()):
                    mine[f"det{n}_{attr}"] = stacked(getattr(item[1], attr))
                if item[0] == "soft":   # the rank that holds the cell has the two voltage records
                    mine[f"det{n}_I"] = stacked(item[1].current_detector.I)
                    for attr in ("input_voltage", "source_voltage"):
                        got = stacked(getattr(item[1], attr))
                        mine[f"det{n}_{attr}"] = (got.reshape(len(got), -1) if len(got)
                                                   else np.zeros((scene["steps"], 0)))
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            if rank == 0:
                results[f"E{seed}"], results[f"H{seed}"] = E, H
                results[f"whole{seed}"] = np.array(getattr(grid, "whole_steps_taken", 0))
                results[f"pairs{seed}"] = np.array(getattr(grid, "pair_steps_taken", 0))
                for key in mine:   # the ranks hold consecutive cells of the detector
                    results[f"{key}_{seed}"] = np.concatenate([p[key] for p in parts], axis=1)
        if rank == 0:
            np.savez(out, **results)
    finally:
        dist.destroy_process_group()


def oracle_results(scene):
    import fdtd_amd as fdtd

    fdtd.set_backend("hip.float64", device="host-staging")
    g, made = scenes.build_api(scene, fdtd)
    resolved, arrays = scenes.resolve(scene, g, made)
    og, omade = scenes.build_oracle(resolved, arrays)
    scenes.run_oracle(resolved, og)
    want = {"E": og.E, "H": og.H}
    for n, item in enumerate(omade):
        for attr in DETECTORS.get(made[n][0], ()):
            want[f"det{n}_{attr}"] = stacked(getattr(item[1], attr))
        if made[n][0] == "soft":
            want[f"det{n}_I"] = stacked(item[1].current_detector.I)
            for attr in ("input_voltage", "source_voltage"):
                got = np.array(getattr(item[1], attr), dtype=np.float64)
                want[f"det{n}_{attr}"] = got.reshape(len(got), -1)
    return want


WHOLE_STEP_SCENES = 0
PAIR_SCENES = 0


def run(world, first, count, on_gpu, out, crowded=None):
    seeds = [s for s in range(first, first + count) if usable(make_scene(s, crowded), world, on_gpu)]
    mp.spawn(worker, args=(world, slab_support.free_port(), seeds, on_gpu, out, crowded), nprocs=world, join=True)
    z = np.load(out)
    bad = {}
    global WHOLE_STEP_SCENES   # scenes in which rank 0 advanced by whole steps in stages
    WHOLE_STEP_SCENES = sum(int(z[f"whole{seed}"]) > 0 for seed in seeds)
    global PAIR_SCENES          # ... by two steps per pass in stages
    PAIR_SCENES = sum(int(z[f"pairs{seed}"]) > 0 for seed in seeds)
    for seed in seeds:
        for name, want in oracle_results(make_scene(seed, crowded)).items():
            got = z[f"{name}{seed}"] if name in ("E", "H") else z[f"{name}_{seed}"]
            got = got.reshape(want.shape) if got.size == want.size else got
            if got.shape != want.shape:
                bad.setdefault(seed, {})[name] = f"shape {got.shape} for {want.shape}"
                continue
            n = np.linalg.norm(want.ravel())
            err = np.linalg.norm((got - want).ravel()) / (n if n > 0 else 1.0)
            # float32 (GPU runs, FUZZ_FLOAT32=1): 1e-5, loop currents 1e-3 -- differences of
            # nearly equal field values, as in tests/test_fuzz.py
            tol = 1e-12 if not flag("FLOAT32") else (1e-3 if name.endswith("_I") else 1e-5)
            if not err <= tol:
                # (float32: a loop current is a sum of differences of H values; while it is
                # far below their rounding -- the wave has not reached the loop -- it reads 0)
                if flag("FLOAT32") and name.endswith("_I") and \
                        np.abs(got - want).max() <= 1e-6 * np.abs(z[f"H{seed}"]).max():
                    continue
                bad.setdefault(seed, {})[name] = err
    return seeds, bad


if __name__ == "__main__":
    world, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    on_gpu = len(sys.argv) > 4 and sys.argv[4] == "gpu"
    seeds, bad = run(world, first, count, on_gpu, f"/tmp/fuzz_slab_{os.getpid()}.npz")
    for seed, errs in bad.items():
        print(f"seed {seed}: {errs}\n   {make_scene(seed)}")
    print(f"x{world} {'gpu' if on_gpu else 'cpu'}: {len(seeds)} scenes "
          f"({sum(crosses(make_scene(s), world) for s in seeds)} with a current loop across slabs), "
          f"{WHOLE_STEP_SCENES} advanced by whole steps in stages ({PAIR_SCENES} of them two steps per pass), "
          f"{len(bad)} above {'1e-5' if flag('FLOAT32') else '1e-12'}")
    sys.exit(1 if bad else 0)
