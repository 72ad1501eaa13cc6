"""The operation sequences of tests/test_fuzz.py::operation_sequence_mismatch with the REAL
reference (torch backend on the CPU: its reset() needs tensors) in the engine's place, against the oracle: shows that the oracle
mirrors what the reference does when arrays are written or re-bound, components and
boundaries are registered after steps were taken, and the grid is reset between runs.
CPU only, where the reference checkout exists:

    python tools/fuzz_operations_reference.py first count [/root/reference]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import torch  # noqa: E402
import fdtd as ref  # noqa: E402

import test_fuzz  # noqa: E402


class ReferenceAsEngine:
    """The names operation_sequence_mismatch uses of the engine's package."""

    def __getattr__(self, name):
        return getattr(ref, name)

    @staticmethod
    def set_backend(name):
        ref.set_backend("torch")


test_fuzz.fdtd = ReferenceAsEngine()
torch.cuda.synchronize = lambda *a, **k: None
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
for seed in range(first, first + count):
    out = test_fuzz.operation_sequence_mismatch(seed)
    if out is None:
        skipped += 1
    elif out[0]:
        bad += 1
        print(f"seed {seed}: {out}", flush=True)
print(f"reference vs oracle, operation sequences {first}..{first + count - 1}: {bad} above 1e-12, "
      f"{skipped} non-finite (skipped)")
sys.exit(1 if bad else 0)
