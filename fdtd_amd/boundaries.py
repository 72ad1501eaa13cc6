"""Boundaries of the hip backend: ``PeriodicBoundary`` and ``PML``
(counterpart of reference fdtd/boundaries.py).

A boundary only *describes* itself here (placement, profiles, memory arrays);
its arithmetic runs inside the native half-step kernels:

* PML memory-variable recursion and field correction -- fused into the bulk
  kernels (``csrc/yee_generic.hpp``), following reference
  fdtd/boundaries.py:398-476;
* periodic plane copies -- ``periodic_copy`` kernel, reference
  fdtd/boundaries.py:173-208.

Storage: the reference keeps 6 memory arrays of shape (slab, 3) per PML plus
phi_E / phi_H.  Conductivity lives only in the component normal to the face
(reference fdtd/boundaries.py:491-499 ...), so of those 18 components exactly two
per field can ever be non-zero.  This backend stores just those:
``pml._psiE`` and ``pml._psiH`` of shape (2, slab); the reference's attribute
names (``psi_Ex`` ..., ``phi_E``, ``phi_H``, ``bE`` ...) are available as
read-only views computed on access.
"""

import numpy as np
import torch

from . import _lib
from .backend import backend as bd
from .grid import Grid, describe_location
from .typing_ import IntOrSlice, ListOrSlice

_AXES = "XYZ"


class Boundary:
    """Base class: placement bookkeeping (reference fdtd/boundaries.py:19-128)."""

    def __init__(self, name: str = None):
        self.grid = None
        self.name = name

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        self.grid = grid
        grid.boundaries.append(self)
        self.x = self._handle_slice(x)
        self.y = self._handle_slice(y)
        self.z = self._handle_slice(z)
        grid._claim_name(self)

    def _handle_slice(self, s: ListOrSlice) -> IntOrSlice:
        if isinstance(s, list):
            if len(s) > 1:
                raise ValueError("Use slices or single numbers to index the grid for a boundary")
            return s[0]
        if isinstance(s, slice):
            if s.start is not None and s.stop is not None and abs(s.start - s.stop) <= 1:
                return s.start
            return s
        raise ValueError("Invalid grid indexing used for boundary")

    # the four hooks of the reference's step loop; the native plan does the work
    def update_phi_E(self):
        pass

    def update_phi_H(self):
        pass

    def update_E(self):
        pass

    def update_H(self):
        pass

    def __repr__(self):
        return f"PML(name={repr(self.name)})"  # sic: the reference prints PML for every boundary

    def __str__(self):
        return "    " + repr(self) + "\n" + describe_location(self.x, self.y, self.z)


def _claim_faces(grid, axis, sides, what):
    """At most one boundary per face (reference fdtd/boundaries.py:149-164, 296-343)."""
    letter = "xyz"[axis]
    attrs = [f"_{letter}{side}_boundary" for side in sides]
    return attrs, any(hasattr(grid, a) for a in attrs), f"grid already has an {what} boundary!"


class PeriodicBoundary(Boundary):
    """Periodic boundary; on registration the instance becomes ``_PeriodicBoundaryX/Y/Z``
    (reference fdtd/boundaries.py:132-169)."""

    axis = None

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        super()._register_grid(grid=grid, x=x, y=y, z=z)
        for axis, pos in enumerate((self.x, self.y, self.z)):
            if isinstance(pos, int) and pos in (0, -1):
                letter = "xyz"[axis]
                attrs, taken, _ = _claim_faces(grid, axis, ("low", "high"), "")
                if taken:
                    grid.boundaries.remove(self)
                    raise AttributeError(f"grid already has an {letter}low/{letter}high boundary!")
                for a in attrs:
                    setattr(grid, a, self)
                self.__class__ = _PERIODIC_CLASSES[axis]
                return
        grid.boundaries.remove(self)
        raise IndexError(
            "A periodic boundary should be placed at the boundary of the "
            "grid using a single index (either 0 or -1)"
        )

    def _signature(self):
        return ("periodic", id(self), self.axis)

    def _native_add(self, lib, plan):
        axis = self.axis
        if axis == 0 and self.grid.Nx != self.grid.Nx_global:
            # the two planes live on the first and the last slab: the grid copies them
            # between ranks (fdtd_amd/parallel.py), the plan only keeps the place in the list
            axis = _lib.PERIODIC_X_EXTERNAL
        _lib.check(lib.fdtdhip_plan_add_periodic(plan, axis))


class _PeriodicBoundaryX(PeriodicBoundary):
    axis = 0


class _PeriodicBoundaryY(PeriodicBoundary):
    axis = 1


class _PeriodicBoundaryZ(PeriodicBoundary):
    axis = 2


_PERIODIC_CLASSES = (_PeriodicBoundaryX, _PeriodicBoundaryY, _PeriodicBoundaryZ)


class PML(Boundary):
    """Convolutional perfectly matched layer; on registration the instance becomes one
    of ``_PMLXlow`` ... ``_PMLZhigh`` (reference fdtd/boundaries.py:214-347)."""

    axis = None
    side = None  # "low" | "high"

    def __init__(self, a: float = 1e-8, name: str = None):
        super().__init__(name=name)
        self.k = 1.0  # fixed, as in the reference (fdtd/boundaries.py:237-238)
        self.thickness = 0
        self.a = a

    def _handle_slice(self, s: ListOrSlice) -> slice:
        if isinstance(s, list):
            raise ValueError("One can only use slices to index the grid for a PML")
        if isinstance(s, slice):
            return s
        raise ValueError("Invalid grid indexing used for boundary")

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        super()._register_grid(grid=grid, x=x, y=y, z=z)
        # first axis whose slice hugs a face decides the placement (reference order x, y, z)
        for axis, s in enumerate((self.x, self.y, self.z)):
            low = (s.start is None or s.start == 0) and s.stop is not None and s.stop > 0
            high = s.start is not None and s.stop is None and s.start < 0
            if not (low or high):
                continue
            side = "low" if low else "high"
            attrs, taken, msg = _claim_faces(grid, axis, (side,), "xyz"[axis] + side)
            if taken:
                grid.boundaries.remove(self)
                raise AttributeError(msg)
            setattr(grid, attrs[0], self)
            self.__class__ = _PML_CLASSES[(axis, side)]
            self._calculate_parameters(thickness=s.stop if low else -s.start)
            return
        grid.boundaries.remove(self)
        raise IndexError(
            "not a valid slice for a PML. Make sure the slice is at the border of the PML"
        )

    # ------------------------------------------------------------ parameters
    def _sigma(self, vect):
        """Cubic conductivity profile (reference fdtd/boundaries.py:280-282)."""
        return 40 * vect ** 3 / (self.thickness + 1) ** 4

    def _calculate_parameters(self, thickness: int = 10):
        """1-D profiles along the normal axis, computed on the host in float64 with the
        reference's expressions (fdtd/boundaries.py:384-396 and 491-522)."""
        t = self.thickness = int(thickness)
        g = self.grid
        # an x-face PML exists only on the slab that holds that face of the simulation; every
        # rank checks it against that slab, so that all of them refuse it together
        n = g.shape[self.axis] if self.axis else g._planes_of_end_slab(self.side)
        if t > n:
            raise IndexError("PML thicker than the grid (or than the x-slab that holds it)")
        self._active = self.axis != 0 or (g._is_first_slab if self.side == "low" else g._is_last_slab)
        if not self._active:
            self._psiE = self._psiH = None
            return
        span = slice(None, t) if self.side == "low" else slice(-t, None)
        loc = [slice(None)] * 3
        loc[self.axis] = span
        self.loc = tuple(loc) + (slice(None),)
        self.locx, self.locy, self.locz = (tuple(loc) + (c,) for c in range(3))
        shape = list(g.shape)
        shape[self.axis] = t
        self.shape = tuple(shape)
        self._lo = 0 if self.side == "low" else n - t

        if self.side == "low":
            sE = self._sigma(np.arange(t - 0.5, -0.5, -1.0))
            sH = self._sigma(np.arange(t - 1.0, 0, -1.0))
        else:
            sE = self._sigma(np.arange(0.5, t + 0.5, 1.0))
            sH = self._sigma(np.arange(1.0, t, 1.0))
        sH = np.concatenate([sH, np.zeros(1)])  # last plane of the slab carries no magnetic loss
        sc = g.courant_number
        self._sigmaE1d, self._sigmaH1d = sE, sH
        self._bE1d = np.exp(-(sE / self.k + self.a) * sc)
        self._cE1d = (self._bE1d - 1.0) * sE / (sE * self.k + self.a * self.k ** 2)
        self._bH1d = np.exp(-(sH / self.k + self.a) * sc)
        self._cH1d = (self._bH1d - 1.0) * sH / (sH * self.k + self.a * self.k ** 2)

        self._psiE = bd.zeros((2,) + self.shape)
        self._psiH = bd.zeros((2,) + self.shape)

    def _signature(self):
        if not self._active:
            return ("pml-elsewhere", id(self))
        # (the memory-variable arrays themselves are not part of it: re-bound arrays are
        # handed to the plan by _native_refresh without rebuilding it)
        return ("pml", id(self), self.axis, self.side, self.thickness, tuple(self._psiE.shape))

    def _native_refresh(self, lib, plan):
        """Before every run: the plan follows the arrays ``_psiE`` / ``_psiH`` are bound to."""
        if not self._active:
            return
        for name in ("_psiE", "_psiH"):
            t = getattr(self, name)
            if t.dtype != bd.float or t.device != bd.device or not t.is_contiguous():
                setattr(self, name, t.to(device=bd.device, dtype=bd.float).contiguous())
        _lib.check(lib.fdtdhip_plan_set_pml_state(plan, self._native_id, self._psiE.data_ptr(),
                                                   self._psiH.data_ptr()))
        alt = getattr(self, "_psiE_alt", None)
        if alt is not None and (alt.shape != self._psiE.shape or alt.dtype != self._psiE.dtype
                                or alt.device != self._psiE.device):
            alt = self._psiE_alt = None
        if alt is not getattr(self, "_psiE_alt_bound", None):
            _lib.check(lib.fdtdhip_plan_set_pml_step_buffer(
                plan, self._native_id, alt.data_ptr() if alt is not None else None))
            self._psiE_alt_bound = alt

    def _native_add(self, lib, plan):
        if not self._active:
            return
        for name in ("_psiE", "_psiH"):
            t = getattr(self, name)
            if t.dtype != bd.float or t.device != bd.device or not t.is_contiguous():
                setattr(self, name, t.to(device=bd.device, dtype=bd.float).contiguous())
        self._native_id = _lib.check(lib.fdtdhip_plan_add_pml(
            plan, self.axis, 0 if self.side == "low" else 1, self.thickness,
            _lib.doubles(self._bE1d), _lib.doubles(self._cE1d),
            _lib.doubles(self._bH1d), _lib.doubles(self._cH1d),
            self._psiE.data_ptr(), self._psiH.data_ptr()))
        alt = getattr(self, "_psiE_alt", None)
        if alt is not None and (alt.shape != self._psiE.shape or alt.dtype != self._psiE.dtype
                                or alt.device != self._psiE.device):
            alt = self._psiE_alt = None
        if alt is not None:
            _lib.check(lib.fdtdhip_plan_set_pml_step_buffer(plan, self._native_id, alt.data_ptr()))
        self._psiE_alt_bound = alt

    # ------------------------------------- the reference's attribute names
    def _expand(self, profile, pad_value):
        """(t,) profile of the normal component -> the reference's (slab, 3) array."""
        shp = [1, 1, 1]
        shp[self.axis] = -1
        full = torch.full(self.shape + (3,), pad_value, dtype=bd.float, device=bd.device)
        full[..., self.axis] = torch.as_tensor(profile, dtype=bd.float, device=bd.device).reshape(shp)
        return full

    @property
    def sigmaE(self):
        return self._expand(self._sigmaE1d, 0.0)

    @property
    def sigmaH(self):
        return self._expand(self._sigmaH1d, 0.0)

    @property
    def bE(self):
        return self._expand(self._bE1d, float(np.exp(-self.a * self.grid.courant_number)))

    @property
    def bH(self):
        return self._expand(self._bH1d, float(np.exp(-self.a * self.grid.courant_number)))

    @property
    def cE(self):
        return self._expand(self._cE1d, 0.0)

    @property
    def cH(self):
        return self._expand(self._cH1d, 0.0)

    def _psi_full(self, store, comp):
        """The reference's psi_{E,H}{x,y,z} of shape (slab, 3): zero except entry
        ``axis`` of the two components transverse to the normal."""
        out = torch.zeros(self.shape + (3,), dtype=store.dtype, device=store.device)
        u, v = (self.axis + 1) % 3, (self.axis + 2) % 3
        if comp == u:
            out[..., self.axis] = store[0]
        elif comp == v:
            out[..., self.axis] = store[1]
        return out

    psi_Ex = property(lambda self: self._psi_full(self._psiE, 0))
    psi_Ey = property(lambda self: self._psi_full(self._psiE, 1))
    psi_Ez = property(lambda self: self._psi_full(self._psiE, 2))
    psi_Hx = property(lambda self: self._psi_full(self._psiH, 0))
    psi_Hy = property(lambda self: self._psi_full(self._psiH, 1))
    psi_Hz = property(lambda self: self._psi_full(self._psiH, 2))

    def _phi(self, store):
        """phi of reference fdtd/boundaries.py:446-448 / 474-476 from the two live variables."""
        out = torch.zeros(self.shape + (3,), dtype=store.dtype, device=store.device)
        u, v = (self.axis + 1) % 3, (self.axis + 2) % 3
        out[..., u] = -store[0]
        out[..., v] = store[1]
        return out

    phi_E = property(lambda self: self._phi(self._psiE))
    phi_H = property(lambda self: self._phi(self._psiH))


def _make_pml_class(axis, side):
    return type(f"_PML{_AXES[axis]}{side}", (PML,), {"axis": axis, "side": side,
                "__doc__": f"PML on the {side} {'xyz'[axis]} face."})


_PML_CLASSES = {(a, s): _make_pml_class(a, s) for a in range(3) for s in ("low", "high")}
_PMLXlow, _PMLXhigh = _PML_CLASSES[(0, "low")], _PML_CLASSES[(0, "high")]
_PMLYlow, _PMLYhigh = _PML_CLASSES[(1, "low")], _PML_CLASSES[(1, "high")]
_PMLZlow, _PMLZhigh = _PML_CLASSES[(2, "low")], _PML_CLASSES[(2, "high")]


def DomainBorderPML(grid, border_cells=5):
    """Wrap the whole domain in PMLs of ``border_cells`` cells, with the non-overlapping
    slicing of the reference (fdtd/boundaries.py:617-645).  Alters ``grid`` in place."""
    b = border_cells
    # (an x-slab checks the extent of the whole simulation, like the other ranks)
    if min(getattr(grid, "Nx_global", grid.Nx), grid.Ny, grid.Nz) < 2 * b:
        raise IndexError("PML border_cells larger than domain!")
    grid[:, :, 0:b] = PML()
    grid[:, :, -b:] = PML()
    grid[0:b, :, b:-b] = PML()
    grid[-b:, :, b:-b] = PML()
    grid[b:-b, 0:b, b:-b] = PML()
    grid[b:-b, -b:, b:-b] = PML()
