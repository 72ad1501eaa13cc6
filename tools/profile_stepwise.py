"""Where the host time of a differentiable run advanced one step per autograd node goes
(cProfile over the forward loop and over backward; the stepwise workload of bench_adjoint).

    python tools/profile_stepwise.py [top]
"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_adjoint  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for warm in range(2):
    g, steps, window = bench_adjoint.build("stepwise")
    prof_f, prof_b = cProfile.Profile(), cProfile.Profile()
    torch.cuda.synchronize()
    prof_f.enable()
    for _ in range(steps):
        g.run(1, progress_bar=False)
    torch.cuda.synchronize()
    prof_f.disable()
    loss = -(g.E[window] ** 2 + g.H[window] ** 2).sum()
    prof_b.enable()
    loss.backward()
    torch.cuda.synchronize()
    prof_b.disable()
for name, prof in (("forward", prof_f), ("backward", prof_b)):
    print(f"==== {name}: per step, {steps} steps")
    st = pstats.Stats(prof)
    st.sort_stats("tottime")
    rows = []
    for (fn, line, func), (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((tt, ct, nc, f"{os.path.basename(fn)}:{line}:{func}"))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"total {total / steps * 1e6:.1f} us/step")
    for tt, ct, nc, where in rows[:top]:
        print(f"  {tt / steps * 1e6:7.2f} us own  {ct / steps * 1e6:7.2f} us cum  {nc / steps:6.1f} calls  {where}")
