"""Random scenes on x-slabs against the single-grid oracle:

    python tools/fuzz_slab.py <world> <first_seed> <count> [gpu|cpu]

`world` ranks over gloo.  gpu: every rank drives the HIP engine on GPU 0 (ghost planes,
plane ranges, the cross-rank x-periodic copy); cpu: the slab host logic over the oracle
engine of tests/slab_support.py.  The gathered E and H must equal the oracle's single-grid
result to 1e-12 (float64).  Scenes whose x-PML is thicker than the thinnest slab, or with
a detector that reads across a slab face (CurrentDetector), are left out.
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

BIG = os.environ.get("FUZZ_BIG") == "1"     # extents of 40-128 cells (planes for the fast kernels)


def usable(scene, world):
    nx = scene["grid"]["shape"][0]
    if nx < world:
        return False
    thinnest = nx // world
    for item in scene["items"]:
        kind, key = item[0], item[1]
        if kind == "currentdet":
            return False
        if kind == "pml" and isinstance(key[0], slice) and (key[0].start, key[0].stop) != (None, None):
            t = key[0].stop if key[0].stop is not None else -key[0].start
            if t > thinnest:
                return False
    return True


def worker(rank, world, port, seeds, on_gpu, out):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        if on_gpu:
            fdtd.set_backend("hip.float64", device=0)
            factory = None
        else:
            fdtd.set_backend("hip.float64", device="host-staging")
            factory = slab_support.OracleSlabEngine
        results = {}
        for seed in seeds:
            scene = fuzz_scenes.random_scene(seed, big=BIG)
            grid, made = scenes.build_api(
                scene, fdtd, grid_factory=lambda **kw: SlabGrid(engine_factory=factory, **kw))
            half = scene["steps"] // 2
            grid.run(half)
            for _ in range(scene["steps"] - half):  # the half-step entry points too
                grid.update_E()
                grid.update_H()
                grid.time_steps_passed += 1
            if on_gpu:
                torch.cuda.synchronize()
            E, H = grid.gather("E"), grid.gather("H")
            if rank == 0:
                results[f"E{seed}"], results[f"H{seed}"] = E, H
        if rank == 0:
            np.savez(out, **results)
    finally:
        dist.destroy_process_group()


def oracle_fields(scene):
    import fdtd_amd as fdtd

    fdtd.set_backend("hip.float64", device="host-staging")
    g, made = scenes.build_api(scene, fdtd)
    resolved, arrays = scenes.resolve(scene, g, made)
    og, _ = scenes.build_oracle(resolved, arrays)
    scenes.run_oracle(resolved, og)
    return og.E, og.H


def run(world, first, count, on_gpu, out):
    seeds = [s for s in range(first, first + count) if usable(fuzz_scenes.random_scene(s, big=BIG), world)]
    mp.spawn(worker, args=(world, slab_support.free_port(), seeds, on_gpu, out), nprocs=world, join=True)
    z = np.load(out)
    bad = {}
    for seed in seeds:
        E, H = oracle_fields(fuzz_scenes.random_scene(seed, big=BIG))
        for name, want in (("E", E), ("H", H)):
            got = z[f"{name}{seed}"]
            n = np.linalg.norm(want.ravel())
            err = np.linalg.norm((got - want).ravel()) / (n if n > 0 else 1.0)
            if not err <= 1e-12:
                bad.setdefault(seed, {})[name] = err
    return seeds, bad


if __name__ == "__main__":
    world, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    on_gpu = len(sys.argv) > 4 and sys.argv[4] == "gpu"
    seeds, bad = run(world, first, count, on_gpu, f"/tmp/fuzz_slab_{os.getpid()}.npz")
    for seed, errs in bad.items():
        print(f"seed {seed}: {errs}\n   {fuzz_scenes.random_scene(seed, big=BIG)}")
    print(f"x{world} {'gpu' if on_gpu else 'cpu'}: {len(seeds)} scenes, {len(bad)} above 1e-12")
    sys.exit(1 if bad else 0)
