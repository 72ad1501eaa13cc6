"""Golden gradients for the differentiable runs, from the real reference.

    PYTHONPATH=/root/repo python tests/golden/make_golden_adjoint.py [/root/reference [scene ...]]

Runs every scene of ``scenes.ADJOINT_SCENES`` with the reference's ``torch`` (CPU,
float64) backend, back-propagates the scene's loss with torch autograd exactly as
reference examples/learnable_focus.py:117-128 does, and stores loss, final fields
and the gradient of every LearnableAnisotropicObject tensor (for a substrate object
also of its modifier's convolution weights) in ``<scene>.npz``.  Numbers only.
"""

import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")

import torch  # noqa: E402
import fdtd as ref  # noqa: E402

import scenes  # noqa: E402


def main():
    ref.set_backend("torch")
    for name, scene in scenes.ADJOINT_SCENES.items():
        if sys.argv[2:] and name not in sys.argv[2:]:
            continue
        with contextlib.redirect_stdout(io.StringIO()):
            grid, made = scenes.build_api(scene, ref)
            for n in scene["chunks"]:
                grid.run(n, progress_bar=False)
        loss = scenes.adjoint_loss(scene, grid, torch)
        loss.backward()
        out = dict(loss=np.array(loss.item()), E=grid.E.detach().numpy(), H=grid.H.detach().numpy())
        for n, (kind, t, _, _) in enumerate(made):
            if kind in ("learnable", "substrate"):
                out[f"grad{n}"] = t.inverse_permittivity.grad.numpy()
                out[f"inv{n}"] = t.inverse_permittivity.detach().numpy()
            if kind == "substrate":
                out[f"gradw{n}"] = t.nonlin_conv.weight.grad.numpy()
                out[f"gradb{n}"] = t.nonlin_conv.bias.grad.numpy()
            if kind == "cortical":
                for key, leaf in t.leaves.items():
                    out[f"src{n}_{key}"] = leaf.grad.numpy()
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        g = [np.linalg.norm(v) for k, v in out.items() if k.startswith("grad")]
        print(f"{name}.npz ok  loss={loss.item():.16e}  |grad|={g}")


if __name__ == "__main__":
    main()
