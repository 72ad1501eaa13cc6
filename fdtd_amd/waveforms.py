"""Scalar waveforms evaluated on the host (counterpart of reference fdtd/waveforms.py).

``hanning`` is the pulse envelope used by the point and line sources; the native
library evaluates the same expression in C with the same libm for the time
stepping itself.  The Gaussian helpers are conveniences for user scripts.
"""

from math import cos, exp, log, sin, sqrt

import numpy as np


def hanning(f, t, n):
    """Sine of angular rate ``f`` under a raised-cosine window of ``n`` cycles
    (reference fdtd/waveforms.py:8-9)."""
    return (1 / 2) * (1 - cos(f * t / n)) * (sin(f * t))


_FWHM_TO_SIGMA = 2.0 * sqrt(2.0 * log(2))


def normalized_gaussian_pulse(x, fwhm, center=0.0):
    """Unit-peak Gaussian of full width at half maximum ``fwhm``
    (reference fdtd/waveforms.py:34-46)."""
    sigma = fwhm / _FWHM_TO_SIGMA
    return np.exp(-(((x - center) ** 2.0) / (2.0 * (sigma ** 2.0))))


def normalized_gaussian_derivative_pulse(x, fwhm, center=0.0):
    """Unit-peak derivative-of-Gaussian pulse (reference fdtd/waveforms.py:49-53)."""
    sigma = fwhm / _FWHM_TO_SIGMA
    return (exp((1.0 / 2.0) - ((x - center) ** 2.0) / (2.0 * sigma ** 2.0)) * (x - center)) / sigma
