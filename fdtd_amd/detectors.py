"""Detectors of the hip backend (counterpart of reference fdtd/detectors.py).

Readings never leave the device during a run: a detector owns device buffers in
which the native library deposits one reading per half step; ``detector.E`` /
``detector.H`` / ``detector.I`` are Python lists of per-step tensors (views into
those buffers), the same list-of-arrays shape the reference exposes
(reference fdtd/detectors.py:31-32, 118, 124).
"""

import ctypes

import torch

from . import _lib
from .backend import backend as bd
from .grid import Grid
from .sources import _line_indices
from .typing_ import List, ListOrSlice, Tuple


class _Readings:
    """Device storage for a growing list of equally shaped readings."""

    MIN_CHUNK = 64

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.chunk = None
        self.used = 0
        self.pending = 0

    def reserve(self, n):
        """Pointer to room for ``n`` more readings."""
        if n == 0:
            return None
        if self.chunk is None or self.chunk.dtype != bd.float or self.used + n > self.chunk.shape[0]:
            self.chunk = bd.zeros((max(n, self.MIN_CHUNK),) + self.shape)
            self.used = 0
        self.pending = n
        return self.chunk[self.used:].data_ptr()

    def commit(self, n, into):
        for i in range(n):
            into.append(self.chunk[self.used + i])
        self.used += n
        self.pending = 0


class _Detector:
    _kind = _lib.DET_CELLS

    def __init__(self, name=None):
        self.grid = None
        self.name = name
        self._id = None
        self._bufE = self._bufH = None

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        self.grid = grid
        grid.detectors.append(self)
        grid._claim_name(self)
        self.x, self.y, self.z = self._handle_slices(x, y, z)

    def _signature(self):
        return (type(self).__name__, id(self), tuple(self.x), tuple(self.y), tuple(self.z))

    def _points(self):
        raise NotImplementedError

    def _local_points(self):
        """Points of this detector that lie in this grid's x planes, in local coordinates
        (all of them unless the grid is one x-slab of a decomposed simulation)."""
        g = self.grid
        out = []
        for x, y, z in self._points():
            for v, n in ((x, g.Nx_global), (y, g.Ny), (z, g.Nz)):
                if not -n <= v < n:
                    raise IndexError(f"index {v} is out of bounds for axis with size {n}")
            xl = g._local_x(g._cell_index(x, g.Nx_global))
            if xl is not None:
                out.append((xl, g._cell_index(y, g.Ny), g._cell_index(z, g.Nz)))
        return out

    def _native_add(self, lib, plan):
        flat = [v for p in self._local_points() for v in p]
        self._id = _lib.check(lib.fdtdhip_plan_add_detector(
            plan, self._kind, len(flat) // 3, _lib.int32s(flat), self.grid.grid_spacing))

    def detect_E(self):
        pass

    def detect_H(self):
        pass

    def __repr__(self):
        return f"{self.__class__.__name__}(name={repr(self.name)})"

    def __str__(self):
        loc = ", ".join(f"{n}=[{v[0]}, ... , {v[-1]}]" for n, v in
                        (("x", self.x), ("y", self.y), ("z", self.z)))
        return "    " + repr(self) + "\n" + f"        @ {loc}\n"


class LineDetector(_Detector):
    """Reads E and H (all three components) at the cells of a line
    (reference fdtd/detectors.py:20-139)."""

    def __init__(self, name=None):
        super().__init__(name)
        self.E = []
        self.H = []

    def _handle_slices(self, x, y, z) -> Tuple[List, List, List]:
        res = _line_indices(self.grid, x, y, z, convert=False)
        return res if len(res) == 3 else tuple(res[1])

    def _points(self):
        return list(zip(self.x, self.y, self.z))

    def _shape(self):
        return (len(self._local_points()), 3)

    def _reserve(self, lib, plan, nE, nH):
        if self._bufE is None:
            self._bufE, self._bufH = _Readings(self._shape()), _Readings(self._shape())
        pE, pH = self._bufE.reserve(nE), self._bufH.reserve(nH)
        _lib.check(lib.fdtdhip_plan_set_detector_output(
            plan, self._id, ctypes.c_void_p(pE), ctypes.c_void_p(pH), max(nE, nH)))

    def _commit(self, nE, nH):
        self._bufE.commit(nE, self.E)
        self._bufH.commit(nH, self.H)

    def detector_values(self):
        return {"E": self.E, "H": self.H}


def _block_lists(grid, x, y, z):
    """Inclusive index ranges of a block (reference fdtd/detectors.py:209-239)."""
    if isinstance(x, list) and isinstance(y, list) and isinstance(z, list):
        if not (len(x) == len(y) == len(z)):
            raise IndexError(
                "sources require grid to be indexed with slices or equal length list-indices"
            )
        return x, y, z
    out = []
    for key, n in ((x, grid.Nx_global), (y, grid.Ny), (z, grid.Nz)):
        if isinstance(key, list):
            key = slice(key[0], key[-1], None)
        a = key.start if key.start is not None else 0
        b = key.stop if key.stop is not None else n
        out.append(list(range(a, b + 1)))
    return tuple(out)


class BlockDetector(LineDetector):
    """Reads E and H at every cell of the outer product of three index lists; one
    reading has shape (len(x), len(y), len(z), 3) (reference fdtd/detectors.py:146-280)."""

    def _handle_slices(self, x, y, z):
        return _block_lists(self.grid, x, y, z)

    def _points(self):
        return [(i, j, k) for i in self.x for j in self.y for k in self.z]

    def _local_x_count(self):
        g = self.grid
        return sum(1 for x in self.x if g._local_x(g._cell_index(x, g.Nx_global)) is not None)

    def _shape(self):
        return (self._local_x_count(), len(self.y), len(self.z), 3)


class CurrentDetector(_Detector):
    """Loop current around z at every cell of a block, from eight H samples per cell
    (reference fdtd/detectors.py:284-496); one reading has shape (len(x), len(y), len(z))."""

    _kind = _lib.DET_CURRENT

    def __init__(self, name=None):
        super().__init__(name)
        self.orientation = None
        self.I = []

    def _handle_slices(self, x, y, z):
        return _block_lists(self.grid, x, y, z)

    def _points(self):
        return [(i, j, k) for i in self.x for j in self.y for k in self.z]

    def _reserve(self, lib, plan, nE, nH):
        if self._bufH is None:
            g = self.grid
            nxl = sum(1 for x in self.x if g._local_x(g._cell_index(x, g.Nx_global)) is not None)
            self._bufH = _Readings((nxl, len(self.y), len(self.z)))
        pH = self._bufH.reserve(nH)
        _lib.check(lib.fdtdhip_plan_set_detector_output(
            plan, self._id, None, ctypes.c_void_p(pH), nH))

    def _commit(self, nE, nH):
        self._bufH.commit(nH, self.I)

    def single_point_current(self, px, py, pz):
        """The current reading of one cell for the present H field, as a device scalar."""
        H, d = self.grid.H, self.grid.grid_spacing
        v1 = (H[px, py - 1, pz, 0] - H[px, py, pz, 0]) * d
        v2 = (H[px, py, pz, 1] - H[px - 1, py, pz, 1]) * d
        first = v1 + v2
        v1 = (H[px, py - 1, pz - 1, 0] - H[px, py, pz - 1, 0]) * d
        v2 = v2 + (H[px, py, pz - 1, 1] - H[px - 1, py, pz - 1, 1]) * d
        return (first + (v1 + v2)) / 2.0

    def detector_values(self):
        return {"I": self.I}
