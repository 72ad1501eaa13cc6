"""x-slab decomposition (the multi-GPU path) against the single-grid run.

CPU part (world sizes 2 and 3, gloo): everything of ``fdtd_amd.parallel.SlabGrid``
except the HIP kernels, with the numpy oracle as the per-slab engine (test double,
tests/slab_support.py).  GPU part: two ranks sharing the one GPU of the test box,
real kernels, planes staged through the host by gloo.
"""

import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import scenes  # noqa: E402
import slab_support  # noqa: E402


def _worker(rank, world, port, scene_name, on_gpu, out, planes=None):
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
        scene = scenes.SLAB_SCENES[scene_name]
        grid, made = scenes.build_api(
            scene, fdtd, grid_factory=lambda **kw: SlabGrid(engine_factory=factory, planes=planes, **kw))
        if planes is not None:
            assert grid.Nx == planes[rank] and grid.x_offset == sum(planes[:rank])
        half = scene["steps"] // 2
        grid.run(half)
        for _ in range(scene["steps"] - half):  # the half-step entry points too
            grid.update_E()
            grid.update_H()
            grid.time_steps_passed += 1
        if on_gpu:
            torch.cuda.synchronize()
        E, H = grid.gather("E"), grid.gather("H")
        dets = {}
        for n, (kind, t, _, _) in enumerate(made):
            if kind in ("linedet", "blockdet"):
                mine = fdtd.backend.numpy(t.E) if len(t.E) else np.zeros((0,))
                parts = [None] * world
                dist.all_gather_object(parts, (grid.x_offset, mine))
                dets[n] = parts
        if rank == 0:
            np.savez(out, E=E, H=H, **{f"det{n}_{r}": p[1] for n, ps in dets.items() for r, p in enumerate(ps)})
    finally:
        dist.destroy_process_group()


def _single_grid_oracle(scene):
    import fdtd_amd as fdtd

    fdtd.set_backend("hip.float64", device="host-staging")
    g, made = scenes.build_api(scene, fdtd)
    resolved, arrays = scenes.resolve(scene, g, made)
    og, omade = scenes.build_oracle(resolved, arrays)
    scenes.run_oracle(resolved, og)
    return og, omade


def _check(scene_name, world, on_gpu, tmp_path, planes=None):
    out = str(tmp_path / f"{scene_name}_{world}.npz")
    port = slab_support.free_port()
    mp.spawn(_worker, args=(world, port, scene_name, on_gpu, out, planes), nprocs=world, join=True)
    z = np.load(out)
    scene = scenes.SLAB_SCENES[scene_name]
    og, omade = _single_grid_oracle(scene)
    tol = 1e-12
    for name in ("E", "H"):
        want = getattr(og, name)
        err = np.linalg.norm((z[name] - want).ravel()) / np.linalg.norm(want.ravel())
        assert err <= tol, f"{scene_name} x{world}: {name} rel-L2 {err:.3e}"
    for n, (kind, t) in enumerate(omade):
        if kind == "linedet":  # points ordered along x: slabs concatenate
            got = np.concatenate([z[f"det{n}_{r}"] for r in range(world) if z[f"det{n}_{r}"].size], axis=1)
            want = np.array(t.E)
            assert np.linalg.norm((got - want).ravel()) <= tol * max(np.linalg.norm(want.ravel()), 1e-30)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("scene_name", sorted(scenes.SLAB_SCENES))
def test_slabs_equal_single_grid_cpu(scene_name, world, tmp_path):
    _check(scene_name, world, False, tmp_path)


def test_five_slabs_cpu(tmp_path):
    """More ranks than any component is wide: interior ranks with two neighbours, x-PMLs
    that end inside the first and last slab, objects and line sources cut several times."""
    _check("slab_mix", 5, False, tmp_path)


def test_slabs_of_unequal_thickness_cpu(tmp_path):
    """planes=[...]: the caller deals the x planes to the ranks (thin end slabs that just
    hold their PMLs, a thick one in the middle; detector points and the objects cut at
    other planes than the even split would)."""
    _check("slab_mix", 3, False, tmp_path, planes=[4, 19, 3])
    _check("slab_2d", 4, False, tmp_path, planes=[5, 12, 9, 5])


def _bad_planes_worker(rank, world, port, out):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float64", device="host-staging")
        refused = 0
        for planes in ([5, 5], [3, 0, 7], [4, 4, 4]):
            try:
                SlabGrid((10, 4, 4), planes=planes, engine_factory=slab_support.OracleSlabEngine)
            except ValueError:
                refused += 1
        if rank == 0:
            open(out, "w").write(str(refused))
    finally:
        dist.destroy_process_group()


def test_a_partition_that_does_not_cover_the_grid_is_refused(tmp_path):
    out = str(tmp_path / "bad_planes.txt")
    mp.spawn(_bad_planes_worker, args=(3, slab_support.free_port(), out), nprocs=3, join=True)
    assert open(out).read() == "3"


@pytest.mark.gpu
def test_slabs_of_unequal_thickness_gpu(tmp_path):
    _check("slab_mix", 3, True, tmp_path, planes=[4, 19, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", sorted(scenes.SLAB_SCENES))
def test_slabs_equal_single_grid_gpu(scene_name, tmp_path):
    _check(scene_name, 2, True, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 5])
def test_one_plane_slabs_gpu(world, tmp_path):
    """Five planes on three ranks (2 + 2 + 1) and on five (one plane each: every launch
    covers the plane that waits for a ghost, the interior launch is empty)."""
    _check("slab_one_plane", world, True, tmp_path)


def test_one_plane_per_rank_cpu(tmp_path):
    _check("slab_one_plane", 5, False, tmp_path)


def _local_loss(scene, grid):
    """This rank's part of scenes.adjoint_loss (the parts add up to it)."""
    spec = scene["loss"]
    if spec["kind"] == "window":
        wx, wy, wz = spec["window"]
        lo, hi = grid._clip_x(wx.start, wx.stop)
        w = (slice(lo, hi), wy, wz)
        return -1.0 * torch.sum(torch.sum(grid.E[w] ** 2 + grid.H[w] ** 2, -1))
    rs = np.random.RandomState(spec["seed"])
    shape = (grid.Nx_global, grid.Ny, grid.Nz, 3)
    mine = slice(grid.x_offset, grid.x_offset + grid.Nx)
    WE = torch.as_tensor(rs.rand(*shape)[mine], dtype=grid.E.dtype, device=grid.E.device)
    WH = torch.as_tensor(rs.rand(*shape)[mine], dtype=grid.E.dtype, device=grid.E.device)
    return torch.sum(WE * grid.E ** 2) + torch.sum(WH * grid.H ** 2)


def _adjoint_worker(rank, world, port, scene_name, out, by_hand=False):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float64", device=0)
        scene = scenes.ADJOINT_SCENES[scene_name]
        grid, made = scenes.build_api(scene, fdtd, grid_factory=lambda **kw: SlabGrid(**kw))
        for c, n in enumerate(scene["chunks"]):
            if by_hand and c == 0:       # half steps driven by hand are recorded too
                for _ in range(n):
                    grid.update_E()
                    grid.update_H()
                    grid.time_steps_passed += 1
            else:
                grid.run(n)
        loss = _local_loss(scene, grid)
        grid.backward(loss)
        torch.cuda.synchronize()
        mine = {"loss": loss.item()}
        for n, (kind, t, _, _) in enumerate(made):
            if kind in ("learnable", "substrate"):
                g = t.inverse_permittivity.grad if t.Nx > 0 else None
                mine[f"grad{n}"] = g.cpu().numpy() if g is not None else None
            if kind == "substrate":   # every rank holds the whole weights, and a part of their gradient
                for key, p in ((f"gradw{n}", t.nonlin_conv.weight), (f"gradb{n}", t.nonlin_conv.bias)):
                    mine[key] = p.grad.cpu().numpy() if p.grad is not None else None
            if kind == "cortical":    # seed tensors are whole on every rank, their gradients partial
                for key, leaf in t.leaves.items():
                    mine[f"src{n}_{key}"] = leaf.grad.cpu().numpy() if leaf.grad is not None else None
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        if rank == 0:
            res = {"loss": sum(p["loss"] for p in parts)}
            for key in mine:
                have = [p[key] for p in parts if p[key] is not None]
                if key.startswith("gradw") or key.startswith("gradb") or key.startswith("src"):
                    res[key] = sum(have)
                elif key.startswith("grad"):
                    res[key] = np.concatenate(have, axis=0)
            np.savez(out, **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", sorted(scenes.ADJOINT_SCENES))
def test_slab_gradients_match_reference_autograd(scene_name, tmp_path):
    """Differentiable run on two slabs (two ranks sharing the GPU, planes of difference
    adjoints exchanged per reverse half step) against the gradients the reference's
    autograd gave for the undivided grid (tests/golden/adjoint_*.npz)."""
    out = str(tmp_path / f"{scene_name}.npz")
    mp.spawn(_adjoint_worker, args=(2, slab_support.free_port(), scene_name, out), nprocs=2, join=True)
    got = np.load(out)
    want = np.load(os.path.join(HERE, "golden", f"{scene_name}.npz"))
    assert abs(float(got["loss"]) - float(want["loss"])) <= 1e-11 * abs(float(want["loss"]))
    seen = 0
    for key in got.files:
        if key.startswith("grad") or key.startswith("src"):
            assert got[key].shape == want[key].shape
            err = np.linalg.norm((got[key] - want[key]).ravel()) / np.linalg.norm(want[key].ravel())
            assert err <= 1e-10, f"{scene_name}: {key} rel-L2 {err:.3e}"
            seen += 1
    assert seen >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["adjoint_pml3d", "adjoint_xperiodic", "adjoint_cortical2d_b"])
def test_slab_gradients_with_hand_driven_half_steps(scene_name, tmp_path):
    """The first run of the scene advanced by explicit update_E() / update_H() on two slabs:
    every half step is an autograd node on every rank, its backward the adjoint of that half
    step with its plane exchange (and, across an x-periodic boundary, the transposed copy)."""
    out = str(tmp_path / f"{scene_name}.npz")
    mp.spawn(_adjoint_worker, args=(2, slab_support.free_port(), scene_name, out, True), nprocs=2, join=True)
    got = np.load(out)
    want = np.load(os.path.join(HERE, "golden", f"{scene_name}.npz"))
    assert abs(float(got["loss"]) - float(want["loss"])) <= 1e-11 * abs(float(want["loss"]))
    seen = 0
    for key in got.files:
        if key.startswith("grad") or key.startswith("src"):
            err = np.linalg.norm((got[key] - want[key]).ravel()) / np.linalg.norm(want[key].ravel())
            assert err <= 1e-10, f"{scene_name}: {key} rel-L2 {err:.3e}"
            seen += 1
    assert seen >= 1


@pytest.mark.gpu
def test_rccl_point_to_point_on_one_gpu():
    """The transport of the real multi-GPU runs (RCCL send/recv started before the
    interior kernel, waited for on the stream) exercised on one GPU: a one-rank "nccl"
    group whose only rank sends a plane to itself (tools/rccl_self_check.py)."""
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(slab_support.free_port()))
    script = os.path.join(os.path.dirname(HERE), "tools", "rccl_self_check.py")
    done = subprocess.run([sys.executable, "-u", script], env=env, capture_output=True, text=True,
                          timeout=110)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "RCCL self send/recv of a plane: ok" in done.stdout


@pytest.mark.gpu
def test_staged_steps_over_rccl_equal_half_steps_bitwise():
    """The real transport on the one GPU of the box: a slab with both x faces open that is
    its own neighbour (a one-rank RCCL group sending the planes to itself, tools/
    slab_step_cost.py), advanced by whole steps in stages and, in a second process, by half
    steps.  RCCL's send / recv run on its own stream, ordered against the compute stream only
    by the waits of the schedule: a stage that read a ghost plane before it arrived, or a plane
    packed before its kernel finished, would change the checksum of the final state."""
    import subprocess

    tool = os.path.join(os.path.dirname(HERE), "tools", "slab_step_cost.py")
    sums = {}
    for mode, env in (("two steps per pass", {}), ("staged", {"FDTDHIP_MARCH2": "0"}),
                      ("half steps", {"FDTDHIP_FUSED": "0"})):
        port = str(slab_support.free_port())
        base = {k: v for k, v in os.environ.items() if k not in ("FDTDHIP_FUSED", "FDTDHIP_MARCH2")}   # the library's own choice
        done = subprocess.run([sys.executable, tool, "21", "small"], capture_output=True, text=True, timeout=280,
                              env=dict(base, MASTER_PORT=port, **env))
        assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
        line = [ln for ln in done.stdout.splitlines() if ln.startswith("state checksum")][0]
        taken = int(done.stdout.split("whole steps taken")[1].split(",")[0])
        pairs = int(done.stdout.split("of them two per pass")[1].split()[0])
        assert (taken > 0) == (mode != "half steps"), done.stdout[-500:]
        assert (pairs > 0) == (mode == "two steps per pass"), done.stdout[-500:]
        sums[mode] = line
    assert sums["staged"] == sums["half steps"] == sums["two steps per pass"], sums


@pytest.mark.gpu
def test_staged_reverse_steps_over_rccl_equal_the_two_pass_sweep_bitwise():
    """The same for the reverse sweep: a recorded run of that self-neighbouring slab with a
    learnable box, back-propagated by whole reverse steps in stages (planes of difference
    adjoints packed by the library, sent over RCCL around the four stages) and, in a second
    process, by the two-pass sweep that exchanges a plane between its passes: the checksum of
    the gradients' bit patterns (dL/dE0, dL/dtensor) must be the same."""
    import subprocess

    tool = os.path.join(os.path.dirname(HERE), "tools", "slab_step_cost.py")
    sums = {}
    for mode, env in (("staged", {}), ("two passes", {"FDTDHIP_ADJ_MARCH": "0"})):
        port = str(slab_support.free_port())
        base = {k: v for k, v in os.environ.items() if k not in ("FDTDHIP_FUSED", "FDTDHIP_ADJ_MARCH")}
        done = subprocess.run([sys.executable, tool, "4", "small", "adjoint", "12"], capture_output=True,
                              text=True, timeout=280, env=dict(base, MASTER_PORT=port, **env))
        assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
        line = [ln for ln in done.stdout.splitlines() if ln.startswith("gradient checksum")][0]
        taken = int(done.stdout.split("whole reverse steps taken")[1].split()[0])
        assert (taken > 0) == (mode == "staged"), done.stdout[-500:]
        sums[mode] = line
    assert sums["staged"] == sums["two passes"], sums


def test_partition_arithmetic():
    from fdtd_amd.parallel import SlabGrid

    class Fake(SlabGrid):
        def __init__(self, n, world, rank, planes=None):
            self.Nx_global, self._world, self._rank, self._planes = n, world, rank, planes
            self._partition()

    for n, world in ((26, 2), (26, 3), (512, 8), (9, 4), (4, 4)):
        spans = [(f.x_offset, f.x_offset + f.Nx) for f in (Fake(n, world, r) for r in range(world))]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        Fake(3, 4, 0)
    given = [Fake(20, 3, r, [2, 15, 3]) for r in range(3)]
    assert [(f.x_offset, f.Nx) for f in given] == [(0, 2), (2, 15), (17, 3)]
    assert given[1]._planes_of_end_slab("low") == 2 and given[1]._planes_of_end_slab("high") == 3


def _refusal_worker(rank, world, port, out):
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float64", device="host-staging")
        grid = SlabGrid((11, 8, 1), engine_factory=slab_support.OracleSlabEngine)   # 6 + 5 planes
        grid[0:6, :, :] = fdtd.PML()              # fits rank 0's six planes
        seen = "accepted"
        try:
            grid[-6:, :, :] = fdtd.PML()          # thicker than rank 1's five
        except IndexError as e:
            seen = f"IndexError: {e}"
        everyone = [None] * world
        dist.all_gather_object(everyone, seen)
        if rank == 0:
            with open(out, "w") as f:
                f.write("\n".join(everyone))
    finally:
        dist.destroy_process_group()


def test_x_pml_thicker_than_its_slab_is_refused_on_every_rank(tmp_path):
    """The slab at that end of the simulation cannot hold the PML: every rank raises, also
    those that would not hold any of it -- a refusal on one rank only would leave the others
    waiting for its planes at the first step."""
    out = str(tmp_path / "refusal.txt")
    mp.spawn(_refusal_worker, args=(2, slab_support.free_port(), out), nprocs=2, join=True)
    lines = open(out).read().splitlines()
    assert len(lines) == 2 and all(line.startswith("IndexError: PML thicker") for line in lines), lines


def _material_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float64", device="host-staging")
        eps = 1.0 + torch.arange(11 * 8 * 3, dtype=torch.float64).reshape(11, 8, 3) / 50
        grid = SlabGrid((11, 8, 3), permittivity=eps, permeability=eps[:1] + 1,
                        engine_factory=slab_support.OracleSlabEngine)
        lo, hi = grid.x_offset, grid.x_offset + grid.Nx
        ok = bool(torch.equal(grid.inverse_permittivity, (1 / eps[lo:hi])[..., None].expand(-1, -1, -1, 3)))
        ok = ok and bool(torch.equal(grid.inverse_permeability,
                                     (1 / (eps[:1] + 1))[..., None].expand(grid.Nx, -1, -1, 3)))
        seen = "sliced" if ok else "wrong planes"
        try:
            SlabGrid((11, 8, 3), permittivity=eps[:5], engine_factory=slab_support.OracleSlabEngine)
            seen += ", part accepted"
        except ValueError:
            seen += ", part refused"
        try:
            # 2 * 5 cells fit the 11 planes of the simulation, not one slab's 6 or 5
            fdtd.DomainBorderPML(SlabGrid((11, 12, 10), engine_factory=slab_support.OracleSlabEngine), 5)
            seen += ", border sized by the simulation"
        except IndexError as e:
            seen += f", border: {e}"
        everyone = [None] * world
        dist.all_gather_object(everyone, seen)
        if rank == 0:
            with open(out, "w") as f:
                f.write("\n".join(everyone))
    finally:
        dist.destroy_process_group()


def test_array_materials_of_the_whole_simulation_are_cut_to_the_slab(tmp_path):
    out = str(tmp_path / "materials.txt")
    mp.spawn(_material_worker, args=(2, slab_support.free_port(), out), nprocs=2, join=True)
    lines = open(out).read().splitlines()
    assert lines == ["sliced, part refused, border sized by the simulation"] * 2, lines


def _fuzz(world, first, count, on_gpu, tmp_path, crowded=False):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slab

    seeds, bad = fuzz_slab.run(world, first, count, on_gpu, str(tmp_path / "fuzz.npz"), crowded)
    assert len(seeds) >= count // 3
    assert not bad, bad
    return sum(fuzz_slab.crosses(fuzz_slab.make_scene(s, crowded), world) for s in seeds)


def test_random_scenes_on_slabs_cpu(tmp_path):
    """Seeded random scenes (tests/fuzz_scenes.py) split over two ranks: the slab host logic
    (plane ranges, ghost planes, cross-rank periodic copy, split objects and boundaries)
    over the oracle engine must reproduce the single-grid oracle."""
    _fuzz(2, 0, 24, False, tmp_path)


def test_crowded_scenes_on_slabs_cpu(tmp_path):
    """Scenes with many objects, sources and detectors on three ranks; fields and every
    detector's readings.  Several hold a CurrentDetector cell on the first plane of a slab:
    its loop reads the last H plane of the rank below (for plane 0 of the simulation: of
    the last rank, index -1 of reference fdtd/detectors.py:452-459), which the ranks pass
    round once the H half step is complete."""
    assert _fuzz(3, 5000, 14, False, tmp_path, crowded=True) >= 3


@pytest.mark.gpu
def test_crowded_scenes_on_slabs_gpu(tmp_path):
    assert _fuzz(3, 5000, 60, True, tmp_path, crowded=True) >= 10


@pytest.mark.gpu
def test_scenes_with_feedback_sources_on_slabs_gpu(tmp_path, monkeypatch):
    """Scenes with a SoftArbitraryPointSource (run by the rank that holds its cell, fed by its
    own loop-current detector, which may read across a slab face) or a direction-vector
    point source (whose six additions may fall on two ranks): fields, detector readings,
    the source's voltage records."""
    monkeypatch.setenv("FUZZ_EXTRAS", "1")
    _fuzz(3, 7000, 60, True, tmp_path)


@pytest.mark.gpu
def test_random_scenes_on_slabs_float32_gpu(tmp_path, monkeypatch):
    """The float32 engine on three slabs (the precision of the multi-GPU benchmark) against
    the float64 oracle at 1e-5, on small scenes and on extents of 40-128 cells."""
    monkeypatch.setenv("FUZZ_FLOAT32", "1")
    _fuzz(3, 400, 60, True, tmp_path)
    monkeypatch.setenv("FUZZ_BIG", "1")
    _fuzz(2, 1000, 12, True, tmp_path)


@pytest.mark.gpu
def test_random_scenes_on_slabs_gpu(tmp_path):
    """The same on the HIP engine, three ranks sharing the GPU."""
    _fuzz(3, 400, 90, True, tmp_path)


def _large_slab_worker(rank, world, port, out, plain=False):
    """plain: no object and no detector -- nothing between the steps but the source points, so
    that the slabs may take two steps per pass (11 steps: two pairs of passes, a pair of whole
    steps, one step of half steps)."""
    import torch.distributed as dist

    import fdtd_amd as fdtd
    from fdtd_amd.parallel import SlabGrid

    slab_support.init_gloo(rank, world, port)
    try:
        fdtd.set_backend("hip.float32", device=0)
        grid = SlabGrid((64 * world, 128, 512))
        for key in ((slice(0, 4),), (slice(-4, None),), (slice(None), slice(0, 4)),
                    (slice(None), slice(-4, None)), (slice(None), slice(None), slice(0, 4)),
                    (slice(None), slice(None), slice(-4, None))):
            grid[key] = fdtd.PML()
        grid[64 * world // 2, 60, 250] = fdtd.PointSource(period=17)      # first plane of a slab
        grid[64 * world // 2 - 1, 3, 300] = fdtd.PointSource(period=23)   # last plane, inside a PML
        grid[20, 64, 100] = fdtd.PointSource(period=11)
        if plain:      # sources in every layer next to an open face and next to a PML
            for n in range(1, 5):      # (eight source points in all: the whole-step kernels' limit)
                grid[64 + n, 4 + n, 4 + n] = fdtd.PointSource(period=5 + n)
            grid[64 * world // 2 - 3, 90, 507] = fdtd.PointSource(period=13)
        else:
            grid[64 * world - 30:64 * world - 10, 30:50, 40:90] = fdtd.AbsorbingObject(
                permittivity=2.0, conductivity=1e-2)
        det = fdtd.LineDetector()
        if not plain:
            grid[5:64 * world - 5, 64, 256] = det
        gen = torch.Generator(device=grid.E.device).manual_seed(77 + rank)
        with torch.no_grad():
            for f in (grid.E, grid.H):
                f.uniform_(-1e-3, 1e-3, generator=gen)
            # (7: three pairs of whole steps, one step of half steps)
            grid.run(11 if plain else 7, progress_bar=False)
            if not plain:      # (a run of two pairs alone: the detector's record has room for exactly four rows)
                grid.run(4, progress_bar=False)
        torch.cuda.synchronize()
        E, H = grid.gather("E"), grid.gather("H")
        psi = [fdtd.backend.numpy(torch.cat([b._psiE.reshape(-1), b._psiH.reshape(-1)]))
               for b in grid.boundaries if getattr(b, "_active", False)]
        parts = [None] * world
        readings = fdtd.backend.numpy(torch.stack(det.E)) if not plain else np.zeros((1, 0, 3))
        dist.all_gather_object(parts, (readings, grid.whole_steps_taken, grid.pair_steps_taken,
                                       np.concatenate(psi) if psi else np.zeros(0)))
        if rank == 0:
            np.savez(out, E=E, H=H, det=np.concatenate([p[0] for p in parts], axis=1),
                     whole=np.array([p[1] for p in parts]), pairs=np.array([p[2] for p in parts]),
                     psi=np.concatenate([p[3] for p in parts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_large_slabs_take_whole_steps_by_default_and_equal_half_steps_bitwise(tmp_path, monkeypatch):
    """Slabs of the size the default policy gives to the whole-step kernel (64 x 128 x 512
    float32 per rank, thin PMLs on every face of the simulation, an object, sources on the
    first and on the last plane of a slab): three ranks -- an interior slab with two open
    faces between two end slabs -- against the same run with FDTDHIP_FUSED=0 (two half-step
    launches per step), bit for bit, fields and detector readings."""
    runs = {}
    monkeypatch.delenv("FDTDHIP_FUSED", raising=False)   # the library's own choice first
    for mode in ("default", "0"):
        if mode == "0":
            monkeypatch.setenv("FDTDHIP_FUSED", "0")
        out = str(tmp_path / f"large_{mode}.npz")
        mp.spawn(_large_slab_worker, args=(3, slab_support.free_port(), out), nprocs=3, join=True)
        runs[mode] = np.load(out)
    assert list(runs["default"]["whole"]) == [10, 10, 10] and list(runs["0"]["whole"]) == [0, 0, 0]
    # (eight of the ten in four passes of two steps each: the object and the detector's cells allow it)
    assert list(runs["default"]["pairs"]) == [8, 8, 8] and list(runs["0"]["pairs"]) == [0, 0, 0]
    for name in ("E", "H", "det", "psi"):
        assert np.array_equal(runs["default"][name], runs["0"][name]), name
    assert np.abs(runs["default"]["E"]).max() > 0


@pytest.mark.gpu
def test_large_slabs_take_two_steps_per_pass_and_equal_half_steps_bitwise(tmp_path, monkeypatch):
    """The same slabs without object and detector: nothing between their steps but source
    points, so each pair of steps is one pass of the two-step kernel over the cells three or
    more planes away from the open faces and the PMLs, and two steps of the shell kernels
    over the others around the four plane exchanges (fdtdhip_pair_stage) -- against whole steps
    in stages (FDTDHIP_MARCH2=0) and half steps (FDTDHIP_FUSED=0), bit for bit, fields and
    memory variables."""
    runs = {}
    monkeypatch.delenv("FDTDHIP_FUSED", raising=False)
    for mode, env in (("pairs", {}), ("whole", {"FDTDHIP_MARCH2": "0"}), ("half", {"FDTDHIP_FUSED": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = str(tmp_path / f"plain_{mode}.npz")
        mp.spawn(_large_slab_worker, args=(3, slab_support.free_port(), out, True), nprocs=3, join=True)
        runs[mode] = np.load(out)
        for k in env:
            monkeypatch.delenv(k)
    assert list(runs["pairs"]["pairs"]) == [8, 8, 8] and list(runs["pairs"]["whole"]) == [10, 10, 10]
    assert list(runs["whole"]["pairs"]) == [0, 0, 0] and list(runs["whole"]["whole"]) == [10, 10, 10]
    assert list(runs["half"]["whole"]) == [0, 0, 0]
    for name in ("E", "H", "psi"):
        assert np.array_equal(runs["pairs"][name], runs["half"][name]), name
        assert np.array_equal(runs["whole"][name], runs["half"][name]), name
    assert np.abs(runs["pairs"]["E"]).max() > 0 and np.abs(runs["pairs"]["psi"]).max() > 0


@pytest.mark.gpu
def test_random_scenes_on_slabs_whole_steps_gpu(tmp_path, monkeypatch):
    """Slabs advanced by the whole-step kernel in stages around the two plane exchanges of a
    step (fdtdhip_step_stage; FDTDHIP_FUSED=2 applies it at any size the plan allows, the
    default is large 3-D slabs only): plane 0 and the last plane as shell cells that wait
    for a ghost plane, the core in two launches, PMLs and objects in the shell, on two,
    three and five ranks (slabs of one or two planes: no core left, half steps), in both
    precisions and on extents of 40-128 cells."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import fuzz_slab as fz

    monkeypatch.setenv("FDTDHIP_FUSED", "2")
    _fuzz(3, 400, 90, True, tmp_path)
    taken = fz.WHOLE_STEP_SCENES
    _fuzz(2, 9000, 60, True, tmp_path)
    taken += fz.WHOLE_STEP_SCENES
    monkeypatch.setenv("FUZZ_BIG", "1")
    _fuzz(2, 1000, 16, True, tmp_path)
    taken += fz.WHOLE_STEP_SCENES
    monkeypatch.setenv("FUZZ_FLOAT32", "1")
    _fuzz(3, 1100, 16, True, tmp_path)
    taken += fz.WHOLE_STEP_SCENES
    monkeypatch.delenv("FUZZ_BIG")
    monkeypatch.delenv("FUZZ_FLOAT32")
    monkeypatch.setenv("FUZZ_THIN_X", "5")
    _fuzz(5, 100, 20, True, tmp_path)
    assert taken >= 30, taken


@pytest.mark.gpu
def test_random_scenes_slab_gradients_gpu(tmp_path):
    """Differentiable runs of random scenes on three slabs -- the learnable object anywhere,
    also on one slab only, so that some ranks record the run for their neighbours' sake --
    against the same engine on the undivided grid."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slab_adjoint

    seeds, bad = fuzz_slab_adjoint.run(3, 100, 90, str(tmp_path / "adj.npz"))
    assert len(seeds) >= 20
    assert not bad, bad


@pytest.mark.gpu
def test_random_scenes_slab_gradients_with_feedback_sources_gpu(tmp_path, monkeypatch):
    """Differentiable runs with sources fed back from their loop current (reference
    fdtd/sources.py:708-737) on slabs: the adjoint of Ez at the cell reaches the adjoint of H
    around it, also in the last plane of the rank below when the cell sits on the first plane
    of a slab (slabs of one or two planes: nearly every source does), wrapping around from the
    first rank to the last as the detector's index -1 does."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slab_adjoint as fz

    monkeypatch.setenv("FUZZ_EXTRAS", "1")
    seeds, bad = fz.run(3, 8000, 40, str(tmp_path / "adj_soft3.npz"))
    assert len(seeds) >= 15 and not bad, bad
    monkeypatch.setenv("FUZZ_THIN_X", "5")
    seeds, bad = fz.run(5, 8100, 45, str(tmp_path / "adj_soft5.npz"))
    assert len(seeds) >= 15 and not bad, bad
    assert any(it[0] == "soft" and it[2]["impedance"] > 0 for s in seeds for it in fz.scene_for(s)["items"])


@pytest.mark.gpu
def test_random_scenes_slab_gradients_whole_steps_gpu(tmp_path, monkeypatch):
    """The same with recorded runs advanced and reverse sweeps undone by whole steps in stages
    around the plane exchanges (fdtdhip_step_stage, fdtdhip_adjoint_step_stage; the switches
    apply them at any size the plans allow, the default is large 3-D slabs): on two, three and
    five ranks, larger extents too."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slab_adjoint as fz

    for name in ("FDTDHIP_ADJ_MARCH", "FDTDHIP_ADJ_DIRECT", "FDTDHIP_FUSED"):
        monkeypatch.setenv(name, "2")
    undone = advanced = 0
    for world, first, count, extra in ((2, 7000, 40, {}), (3, 7100, 40, {}), (2, 7200, 12, {"FUZZ_BIG": "1"}),
                                       (5, 7300, 30, {"FUZZ_THIN_X": "5"})):
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        seeds, bad = fz.run(world, first, count, str(tmp_path / f"adj_{world}_{first}.npz"))
        for k in extra:
            monkeypatch.delenv(k)
        assert len(seeds) >= 6, (world, first, len(seeds))
        assert not bad, bad
        undone += fz.WHOLE_REVERSE_STEPS
        advanced += fz.WHOLE_FORWARD_STEPS
    assert undone >= 100 and advanced >= 60, (undone, advanced)


@pytest.mark.gpu
def test_random_scenes_slab_gradients_direct_gather_gpu(tmp_path, monkeypatch):
    """The same with the ranks' reverse sweeps in the form large grids take (difference
    adjoints of cells far from PMLs, objects and the open faces of the slab derived from the
    field adjoint; FDTDHIP_ADJ_DIRECT=2 applies it at any size), also on slabs of one or two
    planes, where every plane lies at an open face."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slab_adjoint

    monkeypatch.setenv("FDTDHIP_ADJ_DIRECT", "2")
    seeds, bad = fuzz_slab_adjoint.run(3, 300, 60, str(tmp_path / "adj.npz"))
    assert len(seeds) >= 15
    assert not bad, bad
    monkeypatch.setenv("FUZZ_THIN_X", "5")
    seeds, bad = fuzz_slab_adjoint.run(5, 100, 30, str(tmp_path / "adj_thin.npz"))
    assert len(seeds) >= 8
    assert not bad, bad
