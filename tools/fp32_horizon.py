"""How many steps a float32 run stays within 1e-5 (relative L2) of float64.

    python tools/fp32_horizon.py [scene: objects | vacuum] [steps] [stride]

CPU only: the numpy oracle on float64 arrays against its float32 twin (the same slice algebra
on float32 arrays, i.e. what the reference would compute if its float32 backends allocated
float32, reference fdtd/backend.py:87,265-268).  `objects`: the 48 x 46 x 44 scene of
tests/test_full_size_parity.py::test_thousand_steps_against_the_oracle (six PMLs, the three
object kinds, a point source); `vacuum`: a 48^3 grid with a point source and nothing else
(the headline benchmark's scene in small).  Prints the deviation of E and H every `stride`
steps and the first checkpoint at which either exceeds 1e-5.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.yee_numpy import OracleGrid  # noqa: E402  (a tool that measures the oracle itself)


def build(scene, dtype):
    if scene == "vacuum":
        o = OracleGrid((48, 48, 48), dtype=dtype)
        o.add_point_source((24, 24, 24), period=20)
        return o
    o = OracleGrid((48, 46, 44), dtype=dtype)
    for axis in range(3):
        o.add_pml(axis, "low", 6)
        o.add_pml(axis, "high", 6)
    o.add_object((10, 22, 12, 30, 9, 20), 2.2)
    oa = o.add_object((26, 36, 14, 26, 24, 38), 1.9, kind="anisotropic")
    off = 0.03 * np.random.RandomState(8).randn(10, 12, 14, 3, 3)
    for c in range(3):
        off[..., c, c] = 0
    oa.inverse_permittivity += off.astype(dtype)
    o.add_object((12, 30, 32, 40, 26, 36), 1.6, kind="absorbing", conductivity=5e-3)
    o.add_point_source((24, 22, 21), period=19)
    return o


def rel(a, b):
    n = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / n) if n > 0 else 0.0


if __name__ == "__main__":
    scene = sys.argv[1] if len(sys.argv) > 1 else "objects"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    stride = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    a, b = build(scene, np.float64), build(scene, np.float32)
    crossed = None
    for done in range(stride, steps + 1, stride):
        a.run(stride)
        b.run(stride)
        eE, eH = rel(b.E, a.E), rel(b.H, a.H)
        print(f"{scene}: step {done:5d}  twin-float64  E {eE:.2e}  H {eH:.2e}", flush=True)
        if crossed is None and max(eE, eH) > 1e-5:
            crossed = done
    print(f"{scene}: first checkpoint above 1e-5: {crossed}")
