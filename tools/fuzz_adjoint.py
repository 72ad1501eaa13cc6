"""Adjoint directional derivatives against central differences on many seeded random scenes
(the checker of tests/test_fuzz.py over a wider range):  python tools/fuzz_adjoint.py first count [big] [crowded]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz
first, count = int(sys.argv[1]), int(sys.argv[2])
big = "big" in sys.argv[3:]
crowded = "crowded" in sys.argv[3:]
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_scenes
bad = 0; worst = 0.0
for seed in range(first, first + count):
    try:
        err, a, n, scene = test_fuzz.adjoint_mismatch(seed, big=big, scene=fuzz_scenes.random_scene(seed, big=big, crowded=True) if crowded else None)
    except Exception as e:
        print("seed", seed, "ERROR", repr(e)[:300], flush=True); bad += 1; continue
    worst = max(worst, err)
    if not err <= 1e-5:
        bad += 1
        print("seed", seed, err, a, n, scene, flush=True)
print(f"adjoint fuzz {first}..{first+count-1}: {bad} bad, worst {worst:.2e}")
