"""`python bench.py --gpus N` as the round driver starts it: without a launcher the script
has to start its N ranks itself and print rank 0's JSON line.  Rehearsed on the one GPU with
two ranks sharing it and the halo planes staged through the host (FDTD_DIST_BACKEND=gloo);
on a multi-GPU node the same command uses one GPU per rank and RCCL."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    env = dict(os.environ, FDTD_DIST_BACKEND="gloo")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                           "--warmup", "1"], env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    # the headline's series: a 512^3 vacuum slab per rank ...
    assert out["config"]["name"] == "m3v" and out["config"]["cells_per_gpu"] == 512 ** 3
    assert out["config"]["workload"].startswith("1024x512x512 f32")
    assert "gloo" in out["config"]["transport"]
    assert out["config"]["halo"].startswith("512x512x2")
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["one_gpu_grid_of_slab_size"]["name"] == "m3v"
    # ... and the slabs of the 8-GPU configuration, 256 x 1024 x 1024 per rank, beside it
    m4 = out["also"]["m4"]
    assert m4["config"]["name"] == "m4" and m4["config"]["workload"].startswith("512x1024x1024 f32")
    assert m4["config"]["halo"].startswith("1024x1024x2") and m4["n_gpus"] == 2
    assert m4["value"] > 0 and m4["one_gpu_grid_of_slab_size"]["name"] == "m4"

def test_bench_help_needs_no_gpu():
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"],
                          capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert done.returncode == 0 and "--gpus" in done.stdout
