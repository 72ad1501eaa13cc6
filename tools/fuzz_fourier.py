"""FrequencyRoutines.FFT of fdtd_amd against the reference's (fdtd/fourier.py:172-212) on random
records and padding options; CPU, development container (reference imported, fdtd_amd on its
host-staging device).  Same frequencies and spectra, or the same exception type (numpy refuses
the negative padding that a bin resolution alone, or fewer bins than samples, leads to).
A frequency window is left out: the reference's compute_frequencies fails on it
(fdtd/fourier.py:135-137 use names that do not exist there); fdtd_amd implements it.

    python tools/fuzz_fourier.py [trials] [/root/reference]
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sys.path.insert(0, sys.argv[2] if len(sys.argv) > 2 else "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import fdtd as ref  # noqa: E402  (the real reference)
from fdtd.fourier import FrequencyRoutines as RefRoutines  # noqa: E402

import fdtd_amd  # noqa: E402
from fdtd_amd.fourier import FrequencyRoutines  # noqa: E402

ref.set_backend("numpy")
fdtd_amd.set_backend("hip.float64", device="host-staging")
rs = np.random.RandomState(0)


def run(cls, grid, record, opts):
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            f, s = cls(grid, record).FFT(**opts)
        except Exception as e:  # noqa: BLE001 -- the type is what is compared
            return type(e).__name__, None, None
    as_np = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)  # noqa: E731
    return None, as_np(f), as_np(s)


bad = both = 0
for _ in range(trials):
    shape = (int(rs.randint(5, 30)), int(rs.randint(5, 30)), int(rs.choice([1, 7])))
    rg, ag = ref.Grid(shape), fdtd_amd.Grid(shape)
    n = int(rs.randint(3, 200))
    record = rs.randn(n)
    rg.time_steps_passed = ag.time_steps_passed = n
    opts, r = {}, rs.rand()
    if r < 0.4:
        opts["fft_num_bins_in_window"] = int(rs.randint(1, 4 * n))
    elif r < 0.5:
        opts["fft_bin_freq_resolution"] = float(10 ** rs.uniform(9, 13))
    re_, rf, rsp = run(RefRoutines, rg, record, opts)
    ae_, af, asp = run(FrequencyRoutines, ag, fdtd_amd.backend.array(record), opts)
    if re_ != ae_:
        bad += 1
        print(f"exceptions differ: n={n} {opts}: reference {re_}, fdtd_amd {ae_}")
    elif re_:
        both += 1
    elif (rf.shape != af.shape or not np.allclose(rf, af, rtol=1e-13, atol=0)
          or not np.allclose(rsp, asp, rtol=1e-10, atol=1e-12)):
        bad += 1
        print(f"values differ: n={n} {opts}")
print(f"{trials} trials: {bad} differ, {both} raised the same exception on both sides")
sys.exit(1 if bad else 0)
