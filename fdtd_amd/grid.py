"""The FDTD grid of the hip backend (counterpart of reference fdtd/grid.py).

The grid owns the four field arrays as ordinary device tensors of shape
``(Nx, Ny, Nz, 3)`` and a native *plan* (``libfdtd_hip``) that mirrors the Python
lists ``boundaries / objects / sources / detectors``.  ``update_E`` /
``update_H`` / ``run`` hand the whole half step (or many steps) to the plan;
nothing of the time stepping is computed in Python or by torch operators.
"""

import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from . import constants as const
from .backend import backend as bd
from .typing_ import Number, Tuple


def _is_int(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device):
    """The current HIP stream of ``device`` (asked at every call: the caller may have
    entered a ``torch.cuda.stream`` context)."""
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))   # (a tenth of the cost of the Stream object)
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _curl(field, forward):
    if bd.staging or not torch.is_tensor(field) or not field.is_cuda:
        raise RuntimeError("curl_E/curl_H of the hip backend need a tensor on the ROCm device")
    if field.dim() != 4 or field.shape[-1] != 3:
        raise ValueError("expected a field of shape (Nx, Ny, Nz, 3)")
    if field.dtype not in (torch.float32, torch.float64):
        raise ValueError("expected a float32 or float64 field")
    f = field.contiguous()
    out = torch.empty_like(f)
    lib = _lib.load()
    fn = lib.fdtdhip_curl_E if forward else lib.fdtdhip_curl_H
    with torch.cuda.device(f.device):
        _lib.check(fn(f.data_ptr(), out.data_ptr(), f.shape[0], f.shape[1], f.shape[2],
                      _lib.F32 if f.dtype == torch.float32 else _lib.F64, _stream_ptr(f.device)))
    return out


def curl_E(E):
    """Curl of an E-type field, located on the H points; forward differences, each one
    dropped at the last index of its axis (reference fdtd/grid.py:29-51)."""
    return _curl(E, True)


def curl_H(H):
    """Curl of an H-type field, located on the E points; backward differences, each one
    dropped at index 0 of its axis (reference fdtd/grid.py:54-76)."""
    return _curl(H, False)


class Grid:
    """The FDTD grid: fields, materials and the registered components
    (reference fdtd/grid.py:80-169)."""

    def __init__(
        self,
        shape: Tuple[Number, Number, Number],
        grid_spacing: float = 155e-9,
        permittivity: float = 1.0,
        permeability: float = 1.0,
        courant_number: float = None,
    ):
        self.grid_spacing = float(grid_spacing)
        self.Nx, self.Ny, self.Nz = self._handle_tuple(shape)
        self.D = int(self.Nx > 1) + int(self.Ny > 1) + int(self.Nz > 1)
        # extent of the whole simulation along x and where this grid's planes start in it;
        # a plain Grid is the whole simulation, an x-slab (fdtd_amd.parallel.SlabGrid) a part
        self.Nx_global, self.x_offset = self.Nx, 0
        self._partition()

        limit = float(self.D) ** (-0.5)
        if courant_number is None:
            self.courant_number = 0.99 * limit  # small stability margin
        elif courant_number > limit:
            raise ValueError(
                f"courant_number {courant_number} too high for a {self.D}D simulation"
            )
        else:
            self.courant_number = float(courant_number)
        self.time_step = self.courant_number * self.grid_spacing / const.c

        full = (self.Nx, self.Ny, self.Nz, 3)
        self.E = bd.zeros(full)
        self.H = bd.zeros(full)
        self.inverse_permittivity = bd.ones(full) / self._local_planes(self._material(permittivity))
        self.inverse_permeability = bd.ones(full) / self._local_planes(self._material(permeability))

        self.time_steps_passed = 0
        self.sources = []
        self.boundaries = []
        self.detectors = []
        self.objects = []
        self.folder = None

        self._plan = None
        self._plan_signature = None
        self._halo = (None, None)  # (H plane below, E plane above) of an x-slab

    # ----------------------------------------------------------- x-slab hooks
    def _partition(self):
        """Choose this grid's share of the x planes (whole range for a plain Grid)."""

    def _clip_x(self, start, stop):
        """Global plane range [start, stop) -> local range (lo, hi); hi <= lo if none is local."""
        lo = max(start - self.x_offset, 0)
        hi = min(stop - self.x_offset, self.Nx)
        return lo, max(lo, hi)

    def _local_planes(self, value):
        """An array-valued material given for the whole simulation -> this grid's x planes
        (x-slabs keep their own range; scalars and arrays of one plane pass through)."""
        if torch.is_tensor(value) and value.dim() >= 1 and self.Nx != self.Nx_global:
            if value.shape[0] == self.Nx_global:
                return value[self.x_offset:self.x_offset + self.Nx]
            if value.shape[0] != 1:
                raise ValueError(
                    f"material array with {value.shape[0]} x planes: an x-slab grid takes arrays of the "
                    f"whole simulation ({self.Nx_global} planes) or of one plane")
        return value

    def _planes_of_end_slab(self, side):
        """Number of x planes of the slab that holds the "low" / "high" x face of the
        simulation (this grid itself unless it is one slab of several)."""
        return self.Nx

    def _local_x(self, xg):
        """Global plane index -> local index, or None if another slab owns it."""
        xl = xg - self.x_offset
        return xl if 0 <= xl < self.Nx else None

    @property
    def _is_first_slab(self):
        return self.x_offset == 0

    @property
    def _is_last_slab(self):
        return self.x_offset + self.Nx == self.Nx_global

    def _border_inv_eps_x(self, y, z):
        """Component 0 of the inverse permittivity on the LAST x plane of the whole
        simulation over the (y, z) window: what objects copy onto their far x face
        (reference fdtd/objects.py:77-80)."""
        return self.inverse_permittivity[-1, y, z, 0]

    @staticmethod
    def _material(value):
        if bd.is_array(value):
            value = bd.asarray(value)
            if value.dim() == 3:
                value = value[:, :, :, None]
            return value
        return float(value)

    # ------------------------------------------------------------------ keys
    def _handle_distance(self, distance: Number) -> int:
        """Metres (any non-int) to a whole number of cells (reference fdtd/grid.py:171-175)."""
        if _is_int(distance):
            return int(distance)
        return int(float(distance) / self.grid_spacing + 0.5)

    def _handle_time(self, time: Number) -> int:
        """Seconds (any non-int) to a whole number of time steps (reference fdtd/grid.py:177-181)."""
        if _is_int(time):
            return int(time)
        return int(float(time) / self.time_step + 0.5)

    def _handle_tuple(self, shape) -> Tuple[int, int, int]:
        if len(shape) != 3:
            raise ValueError(
                f"invalid grid shape {shape}\n"
                f"grid shape should be a 3D tuple containing floats or ints"
            )
        return tuple(self._handle_distance(s) for s in shape)

    def _handle_slice(self, s: slice) -> slice:
        conv = lambda v: self._handle_distance(v) if isinstance(v, float) else v
        return slice(conv(s.start), conv(s.stop), conv(s.step))

    def _handle_single_key(self, key):
        """One entry of ``grid[x, y, z]`` to a slice or a list of cell indices
        (reference fdtd/grid.py:213-223)."""
        if isinstance(key, slice):
            return self._handle_slice(key)
        try:
            return [self._handle_distance(k) for k in key]
        except TypeError:
            return [self._handle_distance(key)]

    def __setitem__(self, key, attr):
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) > 3:
            raise KeyError("maximum number of indices for the grid is 3")
        x, y, z = tuple(key) + (slice(None),) * (3 - len(key))
        attr._register_grid(
            grid=self,
            x=self._handle_single_key(x),
            y=self._handle_single_key(y),
            z=self._handle_single_key(z),
        )

    @staticmethod
    def _cell_index(v, n):
        """Python-style index into an axis of ``n`` cells: negative values count from the
        end, anything beyond raises the IndexError that numpy raises in the reference's
        first step (e.g. fdtd/sources.py:111 with a source outside the grid)."""
        if not -n <= v < n:
            raise IndexError(f"index {v} is out of bounds for axis with size {n}")
        return v % n

    def _claim_name(self, thing):
        """``grid.<name>`` access to named components (e.g. reference fdtd/boundaries.py:54-60)."""
        if thing.name is None:
            return
        if hasattr(self, thing.name):
            raise ValueError(f"The grid already has an attribute with name {thing.name}")
        setattr(self, thing.name, thing)

    # ------------------------------------------------------------ properties
    @property
    def x(self) -> float:
        return self.Nx * self.grid_spacing

    @property
    def y(self) -> float:
        return self.Ny * self.grid_spacing

    @property
    def z(self) -> float:
        return self.Nz * self.grid_spacing

    @property
    def shape(self) -> Tuple[int, int, int]:
        return (self.Nx, self.Ny, self.Nz)

    @property
    def time_passed(self) -> float:
        return self.time_steps_passed * self.time_step

    # ------------------------------------------------------------ native plan
    def _fields(self):
        """The four field tensors, made device-resident / contiguous / of the
        backend precision if the user re-bound them to something else."""
        out = []
        full = (self.Nx, self.Ny, self.Nz, 3)
        seen = getattr(self, "_fields_seen", {})
        for name in ("E", "H", "inverse_permittivity", "inverse_permeability"):
            t = getattr(self, name)
            ok = seen.get(name)
            # (a tensor found in order before keeps its shape, precision, device and layout
            # unless the backend changed: one identity test instead of five properties)
            if ok is not None and ok[0]() is t and ok[1] is bd.float and ok[2] == bd.device:
                out.append(t)
                continue
            if not torch.is_tensor(t):
                t = bd.array(t)
            if tuple(t.shape) != full:
                t = torch.broadcast_to(t, full)
            if t.device != bd.device or t.dtype != bd.float or not t.is_contiguous():
                t = t.to(device=bd.device, dtype=bd.float).contiguous()
            if t is not getattr(self, name):
                setattr(self, name, t)
            seen[name] = (weakref.ref(t), bd.float, bd.device)
            out.append(t)
        self._fields_seen = seen
        return out

    def _bind_state(self, E, H):
        """Bind E and H to tensors known to be device-resident, contiguous and of the backend
        precision (results of a step, or detached aliases of validated tensors): ``_fields``
        then recognises them without looking at their properties again."""
        self.E, self.H = E, H
        seen = self.__dict__.setdefault("_fields_seen", {})
        seen["E"] = (weakref.ref(E), bd.float, bd.device)
        seen["H"] = (weakref.ref(H), bd.float, bd.device)

    def _signature(self):
        sig = [bd.float, bd.device]
        for group in (self.boundaries, self.objects, self.sources, self.detectors):
            sig.append(tuple(t._signature() for t in group))
        return tuple(sig)

    def _ensure_plan(self, steps=1):
        """The native plan, bound to the grid's current tensors and components.  ``steps``:
        how many steps the caller is about to take (see ``_scan_materials``)."""
        if bd.staging:
            raise RuntimeError(
                "this grid lives in host staging memory: time stepping needs the hip "
                "backend on a ROCm device (fdtd_amd.set_backend('hip'))"
            )
        lib = _lib.load()
        made_for = (bd.native_dtype, bd.device)
        if self._plan is not None and getattr(self, "_plan_made_for", made_for) != made_for:
            # set_backend changed precision or device since the plan was made: a plan is
            # tied to both, the grid's arrays are converted below
            _lib.check(lib.fdtdhip_plan_destroy(self._plan))
            self._plan = None
            self._step_buffers = None
            self._materials_key = None
        if self._plan is None:
            self._plan_made_for = made_for
            handle = ctypes.c_void_p()
            _lib.check(lib.fdtdhip_plan_create(
                ctypes.byref(handle), self.Nx, self.Ny, self.Nz, bd.native_dtype,
                self.courant_number, bd.device.index or 0))
            self._plan = handle
            self._plan_signature = None
        E, H, ie, im = self._fields()
        sig = self._signature()
        if sig != self._plan_signature:
            _lib.check(lib.fdtdhip_plan_reset(self._plan))
            for group in (self.boundaries, self.objects, self.sources, self.detectors):
                for thing in group:
                    thing._native_add(lib, self._plan)
            for src in self.sources:   # links between components, once all have their ids
                link = getattr(src, "_native_link", None)
                if link is not None:
                    link(lib, self._plan)
            self._plan_signature = sig
            self._materials_key = None
            # (the components that follow re-bound arrays between runs, in the order of old)
            self._refreshed = [o for group in (self.objects, self.boundaries) for o in group
                               if getattr(o, "_native_refresh", None) is not None]
        alt = getattr(self, "_step_buffers", None)
        if alt is not None and (alt[0].shape != E.shape or alt[0].dtype != E.dtype
                                or alt[0].device != E.device):
            alt = self._step_buffers = None
        lo, hi = self._halo
        if lo is not None or hi is not None:
            if any(t is not None and (t.dtype != bd.float or t.device != bd.device) for t in (lo, hi)):
                lo, hi = (None if t is None else t.to(device=bd.device, dtype=bd.float).contiguous()
                          for t in (lo, hi))
                self._halo = (lo, hi)
                if hasattr(self, "_H_below"):
                    self._H_below, self._E_above = lo, hi
        # stream, fields, step buffers and halo planes in one call
        _lib.check(lib.fdtdhip_plan_bind(
            self._plan, _stream_ptr(bd.device), E.data_ptr(), H.data_ptr(), ie.data_ptr(), im.data_ptr(),
            alt[0].data_ptr() if alt else None, alt[1].data_ptr() if alt else None,
            lo.data_ptr() if lo is not None else None, hi.data_ptr() if hi is not None else None))
        self._scan_materials(lib, ie, im, steps)
        for o in self._refreshed:
            o._native_refresh(lib, self._plan)
        return lib

    _MAX_MATERIAL_SCANS = 4
    _MATERIAL_RECHECK_STEPS = 16
    _MATERIAL_RECHECK_STEPS_SMALL = 256    # grids of fewer than 2**17 cells

    def _scan_materials(self, lib, ie, im, steps=1):
        """Let the library find out whether the material arrays are constant outside
        the objects (it then stops reading them).  Every in-place write to a tensor
        bumps its version counter, so a changed array is scanned again before the
        next half step; arrays that keep changing are scanned a few times and then
        simply read by the kernels.

        Writes that bypass the counter (``tensor.data``, DLPack, foreign code holding the
        pointer) are caught by scanning again: while constants are in use, a call that
        advances ``_MATERIAL_RECHECK_STEPS`` steps or more re-verifies the arrays first, and
        shorter calls (single steps, half steps) do so once that many steps have gone by
        since the last scan -- a scan reads both arrays once, half of what a step moves.  A
        run of 16 steps or more therefore never uses stale constants; hand-driven stepping
        may for at most 15 steps, unless ``materials_changed()`` is called (on grids of fewer
        than 2**17 cells, whose step takes microseconds: for at most 255 steps)."""
        # (the tensors themselves are part of the key, held weakly: a re-bound array can land
        # on the address of the one it replaced, with the same version count)
        key = (ie.data_ptr(), ie._version, im.data_ptr(), im._version)
        last = getattr(self, "_materials_key", None)
        same = last is not None and last[0] == key and last[1]() is ie and last[2]() is im
        if same:
            if not getattr(self, "_materials_mask", 0) or \
                    getattr(self, "_material_scans", 0) >= self._MAX_MATERIAL_SCANS:
                return            # no constants in use: the kernels read the arrays
            gone = getattr(self, "_steps_since_scan", 0)
            # (a scan ends in a read-back that waits for the stream, about 0.15 ms whatever the
            # size: on grids whose step takes 10 us it is spread over more steps)
            patience = self._MATERIAL_RECHECK_STEPS if self.Nx * self.Ny * self.Nz >= 1 << 17 \
                else self._MATERIAL_RECHECK_STEPS_SMALL
            if steps < self._MATERIAL_RECHECK_STEPS and gone + steps <= patience:
                self._steps_since_scan = gone + steps
                return
            self._materials_mask = _lib.check(lib.fdtdhip_plan_scan_materials(self._plan, 1))
            self._steps_since_scan = steps
            return
        self._materials_key = (key, weakref.ref(ie), weakref.ref(im))
        scans = getattr(self, "_material_scans", 0)
        if last is not None:  # the arrays changed under an unchanged plan
            self._material_scans = scans = scans + 1
        self._materials_mask = _lib.check(lib.fdtdhip_plan_scan_materials(
            self._plan, 1 if scans < self._MAX_MATERIAL_SCANS else 0))
        self._steps_since_scan = steps

    def materials_changed(self):
        """Tell the grid that inverse_permittivity / inverse_permeability were written
        to in a way torch does not see; they are examined again before the next step
        (without this call: within ``_MATERIAL_RECHECK_STEPS`` steps, see ``_scan_materials``)."""
        self._materials_key = None
        self._material_scans = 0

    # ------------------------------------------------------------------ output
    def energy_plane(self, x=None, y=None, z=None):
        """``(sum(E**2), sum(H**2))`` over the components on one grid plane, as two device
        tensors oriented like the reference's plots (reference fdtd/visualization.py:85-146:
        yz plane for ``x``, transposed zx plane for ``y``, xy plane for ``z``); only that
        plane is read."""
        given = [(a, v) for a, v in enumerate((x, y, z)) if v is not None]
        if not given:
            raise ValueError(
                "at least one projection plane (x, y or z) should be supplied to visualize the grid!")
        axis, index = given[0]
        name = "xyz"[axis]
        if not isinstance(index, int):
            raise ValueError(f"the `{name}`-location supplied should be a single integer")
        if len(given) > 1:
            raise ValueError(f"if an `{name}`-location is supplied, one should not supply another location!")
        n = (self.Nx, self.Ny, self.Nz)[axis]
        if not -n <= index < n:
            raise IndexError(f"index {index} is out of bounds for axis {axis} with size {n}")
        lib = self._ensure_plan()
        rows, cols = [(self.Ny, self.Nz), (self.Nz, self.Nx), (self.Nx, self.Ny)][axis]
        eE, eH = bd.zeros((rows, cols)), bd.zeros((rows, cols))
        _lib.check(lib.fdtdhip_plane_energy(self._plan, axis, index % n,
                                            ctypes.c_void_p(eE.data_ptr()), ctypes.c_void_p(eH.data_ptr())))
        return eE, eH

    def save_simulation(self, sim_name=None):
        """Create ``fdtd_output/fdtd_output_<timestamp>[ (<sim_name>)]`` and remember it as
        the place ``save_data`` writes to (reference fdtd/grid.py:404-436)."""
        import os
        from datetime import datetime

        now = datetime.now()
        full = "-".join(str(v) for v in (now.year, now.month, now.day, now.hour, now.minute, now.second))
        if sim_name is not None:
            full = full + " (" + sim_name + ")"
        self.folder = os.path.abspath(os.path.join("fdtd_output", "fdtd_output_" + full))
        self.full_sim_name = full
        os.makedirs(self.folder, exist_ok=True)
        return self.folder

    def save_data(self):
        """Write every detector's readings to ``<folder>/detector_readings.npz`` as
        ``"<name> (E)"`` and ``"<name> (H)"`` (reference fdtd/grid.py:487-502).  Each record
        is stacked on the device and copied to the host once."""
        import os

        import numpy as np

        if getattr(self, "folder", None) is None:
            raise Exception(
                "Save location not initialized. Please read about 'fdtd.Grid.saveSimulation()' "
                "or try running 'grid.saveSimulation()'.")
        dic = {}
        for detector in self.detectors:
            values = detector.detector_values()
            for f in ("E", "H"):
                rec = values[f]
                dic[detector.name + f" ({f})"] = (
                    torch.stack(list(rec)).cpu().numpy() if len(rec) else np.zeros((0,)))
        np.savez(os.path.join(self.folder, "detector_readings"), **dic)

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan is not None:
            try:
                _lib.load().fdtdhip_plan_destroy(plan)
            except Exception:
                pass

    # --------------------------------------------------------------- stepping
    def update_E(self):
        """One electric half step (reference fdtd/grid.py:275-299)."""
        if self._half_step_recorded(True):
            return
        self._native_half_step(self._ensure_plan(), True)

    def update_H(self):
        """One magnetic half step (reference fdtd/grid.py:301-325)."""
        if self._half_step_recorded(False):
            return
        self._native_half_step(self._ensure_plan(), False)

    def _native_half_step(self, lib, phase_E):
        """One half step at ``time_steps_passed`` on the prepared plan."""
        if phase_E:
            self._prepare_sources(lib, 1)
        nE, nH = (1, 0) if phase_E else (0, 1)
        for det in self.detectors:
            det._reserve(lib, self._plan, nE, nH)
        fn = lib.fdtdhip_update_E if phase_E else lib.fdtdhip_update_H
        _lib.check(fn(self._plan, self.time_steps_passed))
        for det in self.detectors:
            det._commit(nE, nH)
        if phase_E:
            self._finish_sources(1)

    def _half_step_recorded(self, phase_E):
        """When gradients are wanted, record the half step as an autograd node."""
        from . import autograd

        if bd.staging or not autograd.needs_grad(self):
            return False
        autograd.differentiable_half_step(self, phase_E)
        return True

    def step(self):
        """One full step: E half step, H half step (reference fdtd/grid.py:267-273)."""
        self._advance(1)

    def _advance(self, n):
        if n <= 0:
            return
        from . import autograd

        if not bd.staging and autograd.needs_grad(self):
            autograd.differentiable_advance(self, n)
            return
        if n >= 2:
            self._provide_step_buffers()
        lib = self._ensure_plan(steps=n)
        self._native_run(lib, n)
        self.time_steps_passed += n

    def _native_run(self, lib, n, into=False):
        """``n`` full steps from ``time_steps_passed`` on the prepared plan (the counter is
        advanced by the caller).  ``into``: one step, out of place into the step buffers
        (``fdtdhip_step_into``; returns the kind of whole-step kernel that took it)."""
        self._prepare_sources(lib, n)
        for det in self.detectors:
            det._reserve(lib, self._plan, n, n)
        if into:
            kind = _lib.check(lib.fdtdhip_step_into(self._plan, self.time_steps_passed))
        else:
            kind = _lib.check(lib.fdtdhip_run(self._plan, self.time_steps_passed, n))
        for det in self.detectors:
            det._commit(n, n)
        self._finish_sources(n)
        return kind

    def _single_steps_out_of_place(self):
        """Whether a recorded single step may be taken out of place (a whole grid on one
        device; slabs step in stages around their exchanges)."""
        return (type(self)._native_run is Grid._native_run and self.Nx == self.Nx_global
                and self._halo == (None, None))

    def _prepare_sources(self, lib, n):
        """Sources with per-run inputs hand them to the plan (steps time_steps_passed ...)."""
        for src in self.sources:
            prepare = getattr(src, "_native_prepare", None)
            if prepare is not None:
                prepare(lib, self._plan, self.time_steps_passed, n)

    def _finish_sources(self, n):
        """Sources that record per-step values publish the ``n`` new ones."""
        for src in self.sources:
            finish = getattr(src, "_native_finish", None)
            if finish is not None:
                finish(n)

    def _native_adjoint(self, lib, n, lamE, lamH, work, half=None):
        """Reverse sweep over the last ``n`` steps, or (``half`` = "E" / "H") the adjoint of
        that one half step alone: per-cell pass, then the gather (see ``fdtd_amd/autograd.py``)."""
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        if half is None:
            # second adjoint fields let the library undo whole steps in one pass over the
            # cells away from PMLs and objects (large 3-D grids; FDTDHIP_ADJ_MARCH=2: any)
            alt = None
            if self._adjoint_second_buffers and n >= 2 and (
                    self._adjoint_march_forced() or self.Nx * self.Ny * self.Nz >= 1 << 22):
                alt = (torch.empty_like(lamE), torch.empty_like(lamH))
            _lib.check(lib.fdtdhip_plan_set_adjoint_buffers(
                self._plan, p(alt[0]) if alt else None, p(alt[1]) if alt else None))
            pmls = [b for b in self.boundaries if hasattr(b, "thickness") and b._active] if alt else []
            psi_alt = [(torch.empty_like(b._psiE.detach()), torch.empty_like(b._psiH.detach())) for b in pmls]
            for b, (ae, ah) in zip(pmls, psi_alt):
                _lib.check(lib.fdtdhip_plan_set_pml_adjoint_buffers(self._plan, b._native_id, p(ae), p(ah)))
            try:
                _lib.check(lib.fdtdhip_adjoint_run(self._plan, n, p(lamE), p(lamH), p(work)))
            finally:
                if alt:
                    _lib.check(lib.fdtdhip_plan_set_adjoint_buffers(self._plan, None, None))
                for b in pmls:
                    _lib.check(lib.fdtdhip_plan_set_pml_adjoint_buffers(self._plan, b._native_id, None, None))
            return
        phase = 1 if half == "E" else 0
        _lib.check(lib.fdtdhip_adjoint_cells(self._plan, phase, 0, p(lamE), p(lamH), p(work)))
        _lib.check(lib.fdtdhip_adjoint_gather(self._plan, phase, p(lamE), p(lamH), p(work), None))

    _adjoint_second_buffers = True   # (tests switch it off to obtain the two-pass sweep)

    @staticmethod
    def _adjoint_march_forced_env(name):
        """Whether the environment asks for a marching kernel wherever a plan allows it (tests)."""
        import os

        return os.environ.get(name) == "2"

    @staticmethod
    def _adjoint_march_forced():
        import os

        return os.environ.get("FDTDHIP_ADJ_MARCH") == "2"

    def _provide_step_buffers(self, slab=False):
        """Second E/H (and PML psi_E) buffers that let ``run`` use the whole-step kernels;
        allocated once, and only for scenes the native library can take whole steps on (point,
        line and plane sources; on x-slabs and on grids too small to march: at most eight
        additive source points and no periodic boundary other than along an axis of one cell,
        where the copy is the identity)."""
        if getattr(self, "_step_buffers", None) is not None or (self.Nx != self.Nx_global and not slab):
            return
        points = 0
        # (what only the marching kernels of large undivided 3-D grids take: periodic boundaries
        # along an axis of more than one cell, more than eight source points, plane sources)
        marching_only = False
        for b in self.boundaries:
            if not hasattr(b, "thickness") and self.shape[b.axis] != 1:
                marching_only = True
        from .sources import PlaneSource

        plane = False
        for s in self.sources:
            if type(s) is PlaneSource:
                plane = True
                continue
            if not hasattr(s, "_wave_args"):
                return
            points += len(s.x) if isinstance(s.x, list) else 1
        # (more points, or a plane source: applied by launches of their own around the marching
        # kernels, which large undivided 3-D grids take)
        if (points > 8 or plane or marching_only) and (
                slab or (self.Nx * self.Ny * self.Nz < 1 << 22
                         and not self._adjoint_march_forced_env("FDTDHIP_FUSED"))):
            return
        self._step_buffers = (torch.empty_like(self.E.detach()), torch.empty_like(self.H.detach()))
        for b in self.boundaries:
            if hasattr(b, "thickness") and b._active and getattr(b, "_psiE_alt", None) is None:
                b._psiE_alt = torch.empty_like(b._psiE.detach())

    def run(self, total_time: Number, progress_bar: bool = True):
        """Advance by ``total_time`` (int: steps, float: seconds), reference fdtd/grid.py:250-265.

        The steps are enqueued on the device without host synchronisation; with
        ``progress_bar`` the bar advances per enqueued chunk.
        """
        if isinstance(total_time, float):
            total_time /= self.time_step
        steps = int(total_time)
        if steps <= 0:
            return
        if not progress_bar:
            self._advance(steps)
            return
        from tqdm import tqdm

        chunk = max(1, steps // 50)
        with tqdm(total=steps) as bar:
            done = 0
            while done < steps:
                n = min(chunk, steps - done)
                self._advance(n)
                torch.cuda.current_stream(bd.device).synchronize()
                bar.update(n)
                done += n

    def reset(self):
        """Zero the fields and the step counter; boundary memory and detector
        readings are kept, as in the reference (fdtd/grid.py:327-333)."""
        self.H = torch.zeros_like(self.H.detach())
        self.E = torch.zeros_like(self.E.detach())
        self.time_steps_passed *= 0

    # ----------------------------------------------------------------- text
    def __repr__(self):
        return (
            f"{self.__class__.__name__}(shape=({self.Nx},{self.Ny},{self.Nz}), "
            f"grid_spacing={self.grid_spacing:.2e}, courant_number={self.courant_number:.2f})"
        )

    def __str__(self):
        text = repr(self) + "\n"
        for title, group in (("sources", self.sources), ("detectors", self.detectors),
                             ("boundaries", self.boundaries), ("objects", self.objects)):
            if group:
                text += f"\n{title}:\n" + "".join(str(t) for t in group)
        return text


def _fmt_slice(s):
    if not isinstance(s, slice):
        return str(s)
    a = "" if s.start is None else str(s.start)
    b = "" if s.stop is None else str(s.stop)
    return f"{a}:{b}"


def describe_location(x, y, z):
    """`` @ x=..., y=..., z=...`` line used by the components' ``__str__``."""
    return f"        @ x={_fmt_slice(x)}, y={_fmt_slice(y)}, z={_fmt_slice(z)}\n"
