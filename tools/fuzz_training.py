"""Re-used grids over optimisation iterations against fresh grids (the checker of
tests/test_fuzz.py over a wider range of seeds):  python tools/fuzz_training.py first count [substrate]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
substrate = "substrate" in sys.argv[3:]
bad = 0
for seed in range(first, first + count):
    try:
        out = test_fuzz.training_loop_mismatch(seed, substrate=substrate)
    except Exception as e:  # noqa: BLE001
        traceback.print_exc(limit=4)
        out = {"error": repr(e)[:200]}
    if out:
        bad += 1
        print("seed", seed, out, flush=True)
print(f"training loops {first}..{first + count - 1}: {bad} differ from fresh grids")
sys.exit(1 if bad else 0)
