import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")]
import torch
import test_adjoint_march as t
cases = [("float32", (100, 170, 250))]
for dtype, shape in cases:
    for faces, objects, steps in (('four-faces', True, 2), ('six-faces', 'inside', 2), ('four-faces', 'inside', 4), ('six-faces', False, 2), ('four-faces', False, 2)):
        want = t._gradients(dtype, shape, faces, objects, steps, False)
        got = t._gradients(dtype, shape, faces, objects, steps, True)
        for name in want:
            d = (got[name] - want[name]).abs()
            bad = (got[name] != want[name])
            nb = int(bad.sum())
            print(dtype, shape, faces, objects, name, "max diff", float(d.max()), "of", float(want[name].abs().max()), "n bad", nb)
            if nb and name != "dtensor":
                idx = bad.any(-1).nonzero()
                print("  bad cells:", idx.shape[0], "min", idx.min(0).values.tolist(), "max", idx.max(0).values.tolist())
                print("  first", idx[:5].tolist())
