"""Differentiable runs on x-slabs against the undivided grid, on random scenes:

    python tools/fuzz_slab_adjoint.py <world> <first_seed> <count>

`world` ranks over gloo share GPU 0.  A learnable anisotropic object is added to every random
scene (tests/fuzz_scenes.py); the loss is a random-weight quadratic form of the final E and H
(each rank its planes).  Loss and the object's tensor gradient, concatenated along x, must
equal what the same engine gives on one undivided grid (relative L2 <= 1e-12).  Scenes with
an x-PML thicker than the thinnest slab are left out.  (An x-periodic boundary crosses ranks,
forward as a plane copy between the end ranks and backward as its transpose.)
"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_scenes  # noqa: E402
import fuzz_slab  # noqa: E402
import scenes  # noqa: E402
import slab_support  # noqa: E402


def scene_for(seed):
    # FUZZ_THIN_X=n: n to n + 2 planes along x (one- and two-plane slabs on n ranks)
    # FUZZ_EXTRAS=1: sources fed back from their loop current (and direction-vector point sources)
    scene = fuzz_scenes.random_scene(seed, thin_x=int(os.environ.get("FUZZ_THIN_X", "0")),
                                     big=bool(os.environ.get("FUZZ_BIG")), extras=bool(os.environ.get("FUZZ_EXTRAS")))
    rs = np.random.RandomState(seed + 77)
    box = fuzz_scenes._box(rs, scene["grid"]["shape"], max_frac=0.6)
    scene["items"].append(("learnable", box, dict(permittivity=float(1.5 + rs.rand())),
                           dict(offdiag_seed=int(rs.randint(1 << 30)), offdiag_scale=0.03)))
    if not any(it[0] in ("point", "line", "plane") for it in scene["items"]):
        scene["items"].append(("point", tuple(n // 2 for n in scene["grid"]["shape"]), dict(period=9)))
    return scene


def usable(scene, world):
    return fuzz_slab.usable(scene, world)


def local_loss(seed, grid):
    rs = np.random.RandomState(seed + 5)
    shape = (grid.Nx_global, grid.Ny, grid.Nz, 3)
    mine = slice(grid.x_offset, grid.x_offset + grid.Nx)
    WE = torch.as_tensor(rs.rand(*shape)[mine], dtype=grid.E.dtype, device=grid.E.device)
    WH = torch.as_tensor(rs.rand(*shape)[mine], dtype=grid.E.dtype, device=grid.E.device)
    return torch.sum(WE * grid.E ** 2) + torch.sum(WH * grid.H ** 2)


def one(seed, fdtd, factory, backward):
    scene = scene_for(seed)
    grid, made = scenes.build_api(scene, fdtd, grid_factory=factory)
    steps = scene["steps"]
    grid.run(steps // 2, progress_bar=False)
    grid.run(steps - steps // 2, progress_bar=False)
    loss = local_loss(seed, grid)
    backward(grid, loss)
    torch.cuda.synchronize()
    obj = [t for k, t, _, _ in made if k == "learnable"][0]
    g = obj.inverse_permittivity.grad if obj.Nx > 0 else None
    global WHOLE_REVERSE_STEPS, WHOLE_FORWARD_STEPS
    WHOLE_REVERSE_STEPS += getattr(grid, "whole_reverse_steps_taken", 0)
    WHOLE_FORWARD_STEPS += getattr(grid, "whole_steps_taken", 0)
    return loss.item(), (g.cpu().numpy() if g is not None else None)


WHOLE_REVERSE_STEPS = 0   # steps this process undid / advanced by whole steps in stages (slabs)
WHOLE_FORWARD_STEPS = 0


def worker(rank, world, port, seeds, out):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float64", device=0)
        res = {}
        for seed in seeds:
            loss, g = one(seed, fdtd, lambda **kw: SlabGrid(**kw), lambda grid, l: grid.backward(l))
            parts = [None] * world
            dist.all_gather_object(parts, (loss, g))
            if rank == 0:
                res[f"loss{seed}"] = sum(p[0] for p in parts)
                res[f"grad{seed}"] = np.concatenate([p[1] for p in parts if p[1] is not None], axis=0)
        if rank == 0:
            res["whole_reverse_steps"] = WHOLE_REVERSE_STEPS
            res["whole_forward_steps"] = WHOLE_FORWARD_STEPS
            np.savez(out, **res)
    finally:
        dist.destroy_process_group()


def run(world, first, count, out):
    seeds = [s for s in range(first, first + count) if usable(scene_for(s), world)]
    mp.spawn(worker, args=(world, slab_support.free_port(), seeds, out), nprocs=world, join=True)
    z = np.load(out)
    global WHOLE_REVERSE_STEPS, WHOLE_FORWARD_STEPS
    WHOLE_REVERSE_STEPS, WHOLE_FORWARD_STEPS = int(z["whole_reverse_steps"]), int(z["whole_forward_steps"])
    taken = (WHOLE_REVERSE_STEPS, WHOLE_FORWARD_STEPS)
    import fdtd_amd as fdtd

    fdtd.set_backend("hip.float64", device=0)
    bad = {}
    for seed in seeds:
        loss, g = one(seed, fdtd, None, lambda grid, l: l.backward())
        got_l, got_g = float(z[f"loss{seed}"]), z[f"grad{seed}"]
        errs = {}
        if abs(got_l - loss) > 1e-12 * max(abs(loss), 1e-300):
            errs["loss"] = (got_l, loss)
        n = np.linalg.norm(g.ravel())
        e = np.linalg.norm((got_g - g).ravel()) / (n if n > 0 else 1.0) if got_g.shape == g.shape else np.inf
        if not e <= 1e-12:
            errs["grad"] = e
        if errs:
            bad[seed] = errs
    WHOLE_REVERSE_STEPS, WHOLE_FORWARD_STEPS = taken    # (rank 0 of the slab run, not the undivided grids)
    return seeds, bad


if __name__ == "__main__":
    world, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    seeds, bad = run(world, first, count, f"/tmp/fuzz_slab_adj_{os.getpid()}.npz")
    for seed, errs in bad.items():
        print(f"seed {seed}: {errs}\n   {scene_for(seed)}")
    print(f"x{world}: {len(seeds)} scenes, {len(bad)} differ from the undivided grid; rank 0 undid "
          f"{WHOLE_REVERSE_STEPS} steps and advanced {WHOLE_FORWARD_STEPS} as whole steps in stages")
    sys.exit(1 if bad else 0)
