"""The reference's own quick tests that touch the time-stepping path, run against the hip
backend on the GPU with the same statements (reference tests/test_detectors.py:91-107,
tests/test_fft.py:17-58, tests/test_antenna_impedance.py:17-98 with fewer steps).
"""

import numpy as np
import pytest
import torch

import fdtd_amd as fdtd
from fdtd_amd.backend import backend as bd
from fdtd_amd.boundaries import DomainBorderPML
from fdtd_amd.detectors import BlockDetector, CurrentDetector
from fdtd_amd.fourier import FrequencyRoutines
from fdtd_amd.objects import AbsorbingObject, Object
from fdtd_amd.sources import SoftArbitraryPointSource
from fdtd_amd.waveforms import normalized_gaussian_pulse

pytestmark = pytest.mark.gpu


@pytest.fixture
def grid():  # reference tests/fixtures.py:14-23
    fdtd.set_backend("hip.float64")
    return fdtd.Grid(shape=(10, 10, 10), grid_spacing=100e-9, permittivity=1.0, permeability=1.0,
                     courant_number=None)


def test_CurrentDetector_shape(grid):
    edetector = BlockDetector()
    cdetector = CurrentDetector()
    grid[4, 4, 4] = cdetector
    grid[4, 4, 5] = edetector
    grid.run(10, progress_bar=False)
    assert bd.array(cdetector.I).shape[0:2] == bd.array(edetector.E).shape[0:2]
    assert bd.array(cdetector.I).shape[0] == 10


def test_SAPS_detector_register(grid):
    detector = SoftArbitraryPointSource(bd.zeros(1), impedance=50.0)
    grid[4, 4, 4] = detector
    grid.run(100, progress_bar=False)
    assert isinstance(grid.detectors[0], CurrentDetector)
    assert len(detector.input_voltage) == len(detector.source_voltage) == 100
    assert len(detector.current_detector.I) == 100


def test_padding(grid):
    fr = FrequencyRoutines(grid, None)
    required_padding, end_time = fr.compute_padding(bd.zeros((1000,)), dt=1e-9)
    assert end_time == pytest.approx(1e-6) and required_padding >= 1000


def test_FFT_single_detector_empty_time(grid):
    detector = BlockDetector()
    grid[4, 4, 4] = detector
    assert FrequencyRoutines(grid, detector).FFT() == ([], [])


def test_FFT_single_detector(grid):
    detector = BlockDetector()
    grid[4, 4, 4] = detector
    grid.run(100, progress_bar=False)
    spectrum_freqs, spectrum = FrequencyRoutines(grid, detector).FFT()
    assert spectrum_freqs.shape == spectrum.shape


def test_SAPS_impedance(grid):
    detector = SoftArbitraryPointSource(np.zeros(1), impedance=50.0)
    grid[4, 4, 4] = detector
    grid.run(100, progress_bar=False)
    freqs, z = FrequencyRoutines(grid, detector).impedance()
    assert freqs.shape == z.shape and bool(torch.isfinite(z.real).all())


def test_patch_antenna_runs():
    """The probe-fed patch of reference tests/test_antenna_impedance.py:17-98 (float32 as
    there, 50 ohm source), shortened to 60 steps: builds, steps and yields an impedance."""
    fdtd.set_backend("hip.float32")
    border_cells, spacing = 6, 1.27e-3 / 3
    Nx = int(59.4e-3 / spacing) + 2 * border_cells
    Ny = int(40.4e-3 / spacing) + 2 * border_cells
    Nz = int(1.27e-3 / spacing) + 2 * border_cells + 2
    grid = fdtd.Grid(shape=(Nx, Ny, Nz), grid_spacing=spacing)
    steps = 60
    DomainBorderPML(grid, border_cells=border_cells)
    sl = slice(border_cells, -border_cells)
    copper = AbsorbingObject(permittivity=1.0, conductivity=1e8)
    grid[sl, sl, border_cells:border_cells + 1] = copper
    top = border_cells + 1 + 3
    grid[sl, sl, top:top + 1] = AbsorbingObject(permittivity=1.0, conductivity=1e8)
    px, py = int(59.4e-3 / spacing) // 2 + border_cells, border_cells
    times = np.arange(0, steps) * grid.time_step
    waveform = normalized_gaussian_pulse(times, 20 * grid.time_step, center=30 * grid.time_step)
    source = SoftArbitraryPointSource(waveform_array=waveform, impedance=50.0)
    grid[px, py, border_cells + 1] = source
    grid[px:px + 1, py:py + 1, border_cells + 2:top + 1] = AbsorbingObject(permittivity=1.0, conductivity=1e8)
    substrate = Object(permittivity=2.42)
    grid[border_cells:px, sl, border_cells + 1:top] = substrate
    grid[px + 1:-border_cells, sl, border_cells + 1:top] = Object(permittivity=2.42)
    grid.run(steps, progress_bar=False)
    freqs, z = FrequencyRoutines(grid, source).impedance()
    assert len(source.source_voltage) == steps and freqs.shape == z.shape
