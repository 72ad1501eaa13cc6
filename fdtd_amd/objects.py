"""Objects of the hip backend (counterpart of reference fdtd/objects.py).

An object is a box with its own inverse-permittivity array.  Registration does
what the reference does on the host side -- build the array, copy the grid's
border values onto the far faces, zero the grid's inverse permittivity inside
the box (reference fdtd/objects.py:60-90) -- using device tensors; the per-step
arithmetic (reference fdtd/objects.py:116-126, 204-218, 251-266) runs inside the
native E half-step kernel, which applies the objects in registration order.
"""

import ctypes

import torch

from . import _lib
from . import constants as const
from .backend import backend as bd
from .grid import Grid, describe_location
from .typing_ import ListOrSlice, Tensorlike


class Object:
    """Isotropic (per-component) dielectric box."""

    _native_kind = _lib.OBJ_ISOTROPIC

    def __init__(self, permittivity: Tensorlike, name: str = None):
        self.grid = None
        self.name = name
        self.permittivity = permittivity

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        if self.name is not None and hasattr(grid, self.name):   # as _claim_name, before anything else
            raise ValueError(f"The grid already has an attribute with name {self.name}")
        # x in the coordinates of the whole simulation, then the part this grid owns
        # (everything, unless the grid is one x-slab of a decomposed simulation)
        self.x_global = self._handle_slice(x, max_index=grid.Nx_global)
        yz = (self._handle_slice(y, max_index=grid.Ny), self._handle_slice(z, max_index=grid.Nz))
        for s, n in zip((self.x_global,) + yz, (grid.Nx_global, grid.Ny, grid.Nz)):
            # The reference registers such a box and then fails in its first update_E, when
            # the box-sized tensor meets the clipped field slice (fdtd/objects.py:116-126);
            # refused here, where the mistake is made.
            if s.start < 0 or s.stop > n or s.stop < s.start:
                raise IndexError(f"object box {s.start}:{s.stop} outside the grid (0:{n})")
        self.grid = grid
        grid.objects.append(self)
        grid._claim_name(self)
        lo, hi = grid._clip_x(self.x_global.start, self.x_global.stop)
        self.x = slice(lo, hi, None)
        self.y = self._handle_slice(y, max_index=grid.Ny)
        self.z = self._handle_slice(z, max_index=grid.Nz)
        self.Nx = abs(self.x.stop - self.x.start)
        self.Ny = abs(self.y.stop - self.y.start)
        self.Nz = abs(self.z.stop - self.z.start)
        whole_Nx = abs(self.x_global.stop - self.x_global.start)

        perm = self.permittivity
        if bd.is_array(perm):
            perm = bd.asarray(perm)
            if perm.dim() == 3:
                perm = perm[:, :, :, None]
            self.permittivity = perm
            if perm.dim() == 4 and perm.shape[0] == whole_Nx and whole_Nx != self.Nx:
                first = lo + grid.x_offset - self.x_global.start
                perm = perm[first:first + self.Nx]
        inv = bd.ones((self.Nx, self.Ny, self.Nz, 3)) / perm

        # far faces take the grid's border value so that the object is symmetric
        gi = grid.inverse_permittivity
        border_x = grid._border_inv_eps_x(self.y, self.z)  # collective on a decomposed grid
        owns_far_x = self.Nx > 0 and grid.x_offset + hi == self.x_global.stop
        if whole_Nx > 1 and owns_far_x:
            inv[-1, :, :, 0] = border_x
        if self.Ny > 1:
            inv[:, -1, :, 1] = gi[self.x, -1, self.z, 1]
        if self.Nz > 1:
            inv[:, :, -1, 2] = gi[self.x, self.y, -1, 2]
        self.inverse_permittivity = inv

        gi[self.x, self.y, self.z] = 0  # the grid's own update adds nothing inside the box

    def _handle_slice(self, s: ListOrSlice, max_index: int = None) -> slice:
        if isinstance(s, list):
            if len(s) != 1:
                raise IndexError(
                    "One can only use slices or single indices to index the grid for an Object"
                )
            return slice(s[0], s[0] + 1, None)
        if not isinstance(s, slice):
            raise ValueError("Invalid grid indexing used for object")
        if s.step is not None and s.step != 1:
            raise IndexError("Can only use slices with unit step to index the grid for an Object")
        start = 0 if s.start is None else s.start
        stop = max_index if s.stop is None else s.stop
        if start < 0:
            start += max_index
        if stop < 0:
            stop += max_index
        return slice(start, stop, None)

    # hooks of the reference's step loop; the native plan does the work
    def update_E(self, curl_H=None):
        pass

    def update_H(self, curl_E=None):
        pass

    # -------------------------------------------------------------- native
    def _box(self):
        return [self.x.start, self.x.stop, self.y.start, self.y.stop, self.z.start, self.z.stop]

    def _tensor(self, name):
        t = getattr(self, name)
        # the kernels index these arrays by the box: a re-bound tensor must keep the shape
        shapes = self.__dict__.setdefault("_registered_shapes", {})
        want = shapes.setdefault(name, tuple(t.shape))
        if tuple(t.shape) != want:
            raise ValueError(f"{type(self).__name__}.{name} has shape {tuple(t.shape)}; the object "
                             f"was registered with {want}")
        if t.device != bd.device or t.dtype != bd.float or not t.is_contiguous():
            keep_grad = t.requires_grad
            t = t.detach().to(device=bd.device, dtype=bd.float).contiguous()
            t.requires_grad_(keep_grad)
            setattr(self, name, t)
        return t

    def _signature(self):
        return (type(self).__name__, id(self), tuple(self._box()),
                self._tensor("inverse_permittivity").data_ptr())

    def _native_add(self, lib, plan):
        if self.Nx == 0:  # no cell of the box lies in this x-slab
            return
        self._native_id = _lib.check(lib.fdtdhip_plan_add_object(
            plan, self._native_kind, _lib.int32s(self._box()),
            ctypes.c_void_p(self._tensor("inverse_permittivity").data_ptr()), None))

    def __repr__(self):
        return f"{self.__class__.__name__}(name={repr(self.name)})"

    def __str__(self):
        return "    " + repr(self) + "\n" + describe_location(self.x, self.y, self.z)


class AbsorbingObject(Object):
    """Dielectric box with conductivity (loss)."""

    _native_kind = _lib.OBJ_ABSORBING

    def __init__(self, permittivity: Tensorlike, conductivity: Tensorlike, name: str = None):
        super().__init__(permittivity, name)
        self.conductivity = conductivity

    def _register_grid(self, grid: Grid, x: slice = None, y: slice = None, z: slice = None):
        super()._register_grid(grid=grid, x=x, y=y, z=z)
        cond = bd.asarray(self.conductivity)
        while cond.dim() < self.inverse_permittivity.dim():
            cond = cond[..., None]
        self.conductivity = torch.broadcast_to(cond, self.inverse_permittivity.shape)
        # reference fdtd/objects.py:195-202, same left-to-right products
        self.absorption_factor = (
            0.5
            * grid.courant_number
            * self.inverse_permittivity
            * self.conductivity
            * grid.grid_spacing
            * const.eta0
        )

    def _signature(self):
        return super()._signature() + (self._tensor("absorption_factor").data_ptr(),)

    def _native_add(self, lib, plan):
        if self.Nx == 0:
            return
        self._native_id = _lib.check(lib.fdtdhip_plan_add_object(
            plan, self._native_kind, _lib.int32s(self._box()),
            ctypes.c_void_p(self._tensor("inverse_permittivity").data_ptr()),
            ctypes.c_void_p(self._tensor("absorption_factor").data_ptr())))


class AnisotropicObject(Object):
    """Box with a full 3x3 inverse-permittivity tensor per cell; starts diagonal, the
    user may fill the off-diagonals afterwards (reference fdtd/objects.py:229-266)."""

    _native_kind = _lib.OBJ_ANISOTROPIC

    def _register_grid(self, grid: Grid, x: slice = None, y: slice = None, z: slice = None):
        super()._register_grid(grid=grid, x=x, y=y, z=z)
        self.inverse_permittivity = torch.diag_embed(self.inverse_permittivity).contiguous()


class LearnableAnisotropicObject(AnisotropicObject):
    """Anisotropic box whose tensor is a leaf that collects gradients (reference
    fdtd/objects.py:276-349).

    ``is_substrate=True`` scales the tensor, every E half step, by
    ``sigmoid(nonlin_conv(energies))``: a 1x1 convolution (``self.nonlin_conv``, a
    ``torch.nn.Conv2d(2, 9, 1)`` whose weight and bias are learnable too) from the
    cell's ``sum(E**2)`` and ``sum(H**2)`` to the nine tensor entries (reference
    fdtd/objects.py:328-337).  The native kernels evaluate it per cell, so the box may
    have any extent along z (the reference's reshape only works for ``Nz == 1``)."""

    def __init__(self, permittivity: Tensorlike, is_substrate=False, name: str = None, device=None):
        super().__init__(permittivity, name)
        self.is_substrate = bool(is_substrate)
        self.device = device
        if not bd.staging:
            self.nonlin_conv = torch.nn.Conv2d(2, 9, kernel_size=1, stride=1, device=bd.device,
                                               dtype=bd.float)
        else:
            self.nonlin_conv = torch.nn.Conv2d(2, 9, kernel_size=1, stride=1, dtype=bd.float)
        if self.is_substrate:
            self._native_kind = _lib.OBJ_ANISOTROPIC_MODULATED
        self._weights_key = None
        self._weights = None
        self._mod_buffers = (None, None)  # (energy trace, weight-gradient buffer) of a recorded run

    def _register_grid(self, grid: Grid, x: slice = None, y: slice = None, z: slice = None):
        super()._register_grid(grid=grid, x=x, y=y, z=z)
        self.inverse_permittivity.requires_grad = True
        self.inverse_permittivity.retain_grad()

    # -- the modifier's weights as the kernels take them: (9, 3) = [w_E, w_H, bias] per entry
    def _packed_weights(self):
        w, b = self.nonlin_conv.weight, self.nonlin_conv.bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != self._weights_key:
            self._weights = torch.cat(
                [w.detach().reshape(9, 2), b.detach().reshape(9, 1)], dim=1
            ).to(device=bd.device, dtype=bd.float).contiguous()
            self._weights_key = key
        return self._weights

    def _native_add(self, lib, plan):
        super()._native_add(lib, plan)
        self._weights_key = None
        self._native_refresh(lib, plan)

    def _native_refresh(self, lib, plan):
        """Hand the current modifier weights (and, during a recorded run or its backward
        pass, the trace and gradient buffers) to the plan; called before every run."""
        if not self.is_substrate or self.Nx == 0:
            return
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        trace, grad = self._mod_buffers
        _lib.check(lib.fdtdhip_plan_set_object_modifier(
            plan, self._native_id, ptr(self._packed_weights()), ptr(trace), ptr(grad)))
