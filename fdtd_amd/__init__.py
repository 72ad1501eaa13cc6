"""fdtd_amd -- an MI355X-native Yee time-stepping engine behind the fdtd Python API.

Same surface as simenhu/fdtd (reference fdtd/__init__.py:6-14) for the
time-stepping path: ``Grid``, ``grid[x, y, z] = Object / Source / Detector /
PML / PeriodicBoundary``, ``grid.run``, ``set_backend`` -- with one backend,
``"hip"``, whose half steps run as hand-written HIP kernels for gfx950 in
``fdtd_amd/csrc/libfdtd_hip.so``.
"""

__version__ = "0.1.0"

from .backend import backend, set_backend
from .grid import Grid, curl_E, curl_H
from .sources import (PointSource, CorticalColumnPointSource, LineSource, PlaneSource,
                      CorticalColumnPlaneSource, SoftArbitraryPointSource)
from .detectors import LineDetector, BlockDetector, CurrentDetector
from .objects import Object, AbsorbingObject, AnisotropicObject, LearnableAnisotropicObject
from .boundaries import PeriodicBoundary, PML, DomainBorderPML
from .fourier import FrequencyRoutines

__all__ = [
    "backend", "set_backend", "Grid", "curl_E", "curl_H",
    "PointSource", "CorticalColumnPointSource", "LineSource", "PlaneSource", "CorticalColumnPlaneSource", "SoftArbitraryPointSource",
    "LineDetector", "BlockDetector", "CurrentDetector",
    "Object", "AbsorbingObject", "AnisotropicObject", "LearnableAnisotropicObject",
    "PeriodicBoundary", "PML", "DomainBorderPML", "FrequencyRoutines",
]
