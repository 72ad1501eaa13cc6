"""Many seeded random scenes on the HIP engine against the numpy oracle (the checker of
tests/test_fuzz.py over a wider range of seeds):

    python tools/fuzz_parity.py [first_seed] [count] [float64|float32] [big] [extras] [substrate] [crowded]

Kernel choices can be forced from the environment (FDTDHIP_FORCE=1, FDTDHIP_FUSED=0|1).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtype = sys.argv[3] if len(sys.argv) > 3 else "float64"
big = "big" in sys.argv[4:]
extras = "extras" in sys.argv[4:]
substrate = "substrate" in sys.argv[4:]
crowded = "crowded" in sys.argv[4:]
bad = skipped = paired = 0
from fdtd_amd import _lib  # noqa: E402

for seed in range(first, first + count):
    before = _lib.load().fdtdhip_two_step_launches()
    out = test_fuzz.check_scene(seed, dtype, big=big, extras=extras, substrate=substrate,
                                crowded=crowded)
    paired += _lib.load().fdtdhip_two_step_launches() > before
    if out is None:
        skipped += 1
    elif out[0]:
        bad += 1
        print(f"seed {seed}: {out[0]}\n   {out[1]}", flush=True)
print(f"{dtype}{' big' if big else ''} seeds {first}..{first + count - 1}: {bad} above tolerance, "
      f"{skipped} non-finite in the oracle (skipped); {paired} scenes took two steps per pass", flush=True)
sys.exit(1 if bad else 0)
