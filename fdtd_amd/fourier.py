"""Frequency-domain post-processing of detector and source records (counterpart of
reference fdtd/fourier.py).  The records already live on the device (lists of per-step
tensors); they are stacked and transformed there with ``torch.fft``.
"""

from math import ceil

import torch

from .backend import backend as bd
from .detectors import BlockDetector, CurrentDetector, LineDetector
from .sources import SoftArbitraryPointSource


def _series(readings):
    """1-D time series from a list of per-step readings: the first scalar of each."""
    if len(readings) == 0:
        return bd.zeros((0,))
    return torch.stack([torch.as_tensor(r, device=bd.device).reshape(-1)[0] for r in readings]).to(bd.float)


def _pad_edge(x, n):
    """``n`` copies of the last sample appended (numpy.pad mode 'edge', reference fourier.py:204)."""
    if n < 0:   # numpy.pad refuses it: a bin resolution alone gives a negative bin count
        raise ValueError("index can't contain negative values")
    if n == 0 or x.numel() == 0:
        return x
    return torch.cat([x, x[-1:].expand(n)])


class FrequencyRoutines:
    """``FrequencyRoutines(grid, objs)`` with ``objs`` a 1-D array, a detector, a
    ``SoftArbitraryPointSource`` or a ``(field detector, CurrentDetector)`` pair
    (reference fdtd/fourier.py:18-48).  Frequencies are in Hz."""

    def __init__(self, grid, objs):
        self.grid = grid
        self.objs = objs

    def compute_padding(self, input_data, dt, freq_window_tuple=None, fft_num_bins_in_window=None,
                        fft_bin_freq_resolution=None):
        """Samples to append so that the requested window holds the requested number of
        bins, and the record's duration (reference fdtd/fourier.py:52-112)."""
        input_length = input_data.shape[0]
        if freq_window_tuple is None:
            begin_freq, end_freq = 0, (input_length / 2.0) / (dt * input_length)
        else:
            begin_freq, end_freq = freq_window_tuple
        end_time = input_length * dt
        if fft_num_bins_in_window is None and fft_bin_freq_resolution is not None:
            fft_num_bins_in_window = (begin_freq - end_freq) / fft_bin_freq_resolution
        elif not (fft_num_bins_in_window or fft_bin_freq_resolution):
            fft_num_bins_in_window = input_length
        required_padding = ceil(fft_num_bins_in_window / ((end_freq - begin_freq) * dt)) - input_length
        return required_padding, end_time

    def compute_frequencies(self, length_with_padding, dt, freq_window_tuple=None):
        """FFT bin frequencies and the index span of the window (reference
        fdtd/fourier.py:122-139; without a window: everything but the last bin)."""
        spectrum_freqs = torch.fft.fftfreq(length_with_padding, d=dt, device=bd.device, dtype=bd.float)
        if freq_window_tuple is None:
            return spectrum_freqs, 0, -1
        begin_freq, end_freq = freq_window_tuple
        begin = int(torch.abs(spectrum_freqs - begin_freq).argmin())
        end = int(torch.abs(spectrum_freqs - end_freq).argmin())
        return spectrum_freqs, begin, end

    def _spectrum(self, series, **window):
        pad, _ = self.compute_padding(series, self.grid.time_step, **window)
        return torch.fft.fft(_pad_edge(series, pad))

    def FFT(self, freq_window_tuple=None, fft_num_bins_in_window=None, fft_bin_freq_resolution=None):
        """Spectrum of one record (reference fdtd/fourier.py:172-212): a plain array, the
        first component of a field detector's first point, a soft source's output voltage or
        a current detector's first point."""
        if self.grid.time_steps_passed == 0:
            return [], []
        o = self.objs
        if torch.is_tensor(o) or bd.is_array(o):
            series = torch.as_tensor(o, device=bd.device).to(bd.float).reshape(-1)
        elif isinstance(o, (LineDetector, BlockDetector)):
            series = _series(o.E)
        elif isinstance(o, SoftArbitraryPointSource):
            series = _series(o.source_voltage)
        elif isinstance(o, CurrentDetector):
            series = _series(o.I)
        else:
            raise ValueError("Sorry, FFT can't yet interpret the argument given.")
        window = dict(freq_window_tuple=freq_window_tuple, fft_num_bins_in_window=fft_num_bins_in_window,
                      fft_bin_freq_resolution=fft_bin_freq_resolution)
        spectrum = self._spectrum(series, **window)
        freqs, b, e = self.compute_frequencies(spectrum.shape[0], self.grid.time_step, freq_window_tuple)
        return freqs[b:e], spectrum[b:e]

    def impedance(self, freq_window_tuple=None, fft_num_bins_in_window=None, fft_bin_freq_resolution=None):
        """Voltage spectrum over current spectrum of one node, zero where the current
        spectrum vanishes (reference fdtd/fourier.py:216-260)."""
        if self.grid.time_steps_passed == 0:
            return [], []
        o = self.objs
        if isinstance(o, SoftArbitraryPointSource):
            voltage, current = _series(o.source_voltage), _series(o.current_detector.I)
        elif (isinstance(o, tuple) and isinstance(o[0], (BlockDetector, LineDetector))
              and isinstance(o[1], CurrentDetector)):
            voltage, current = _series(o[0].E), _series(o[1].I)
        else:
            raise ValueError("Sorry, FFT can't yet interpret the argument given.")
        window = dict(freq_window_tuple=freq_window_tuple, fft_num_bins_in_window=fft_num_bins_in_window,
                      fft_bin_freq_resolution=fft_bin_freq_resolution)
        v, i = self._spectrum(voltage, **window), self._spectrum(current, **window)
        z = torch.where(i != 0, v / torch.where(i != 0, i, torch.ones_like(i)), torch.zeros_like(v))
        freqs, b, e = self.compute_frequencies(v.shape[0], self.grid.time_step, freq_window_tuple)
        return freqs[b:e], z[b:e]
