"""Random sequences of user operations between runs (the checker of tests/test_fuzz.py over a
wider range of seeds):  python tools/fuzz_operations.py first count [big]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
ops = collections.Counter()
for seed in range(first, first + count):
    out = test_fuzz.operation_sequence_mismatch(seed, big="big" in sys.argv[3:])
    if out is None:
        skipped += 1
        continue
    ops.update(out[1])
    if out[0]:
        bad += 1
        print(f"seed {seed}: {out[0]} after {out[1]}", flush=True)
print(f"operation sequences {first}..{first + count - 1}: {bad} above 1e-12, {skipped} non-finite (skipped); "
      f"operations run: {dict(ops)}")
sys.exit(1 if bad else 0)
