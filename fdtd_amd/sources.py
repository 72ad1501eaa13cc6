"""Sources of the hip backend (counterpart of reference fdtd/sources.py).

A source resolves its location and spatial profile on the host at registration;
the per-step waveform value is evaluated by the native library in C (same libm
as CPython, same expression order as reference fdtd/sources.py:97-110) and the
injection is a small device kernel that runs after the boundaries of the same
half step (reference fdtd/grid.py:293-295, 319-321).
"""

import ctypes

import numpy as np
import torch

from . import _lib
from .backend import backend as bd
from .grid import Grid
from .typing_ import List, ListOrSlice, Number, Tuple


def _line_indices(grid, x, y, z, convert):
    """Cells on the diagonal of the box spanned by the keys: ``max |extent|`` points
    from ``linspace(start, stop, endpoint=False)`` truncated to int, shared by
    line sources and line detectors (reference fdtd/sources.py:265-314,
    fdtd/detectors.py:81-112)."""
    conv = grid._handle_distance if convert else (lambda v: v)
    if isinstance(x, list) and isinstance(y, list) and isinstance(z, list):
        if not (len(x) == len(y) == len(z)):
            raise IndexError(
                "sources require grid to be indexed with slices or equal length list-indices"
            )
        return [conv(v) for v in x], [conv(v) for v in y], [conv(v) for v in z]
    ends = []
    for key, n in ((x, grid.Nx_global), (y, grid.Ny), (z, grid.Nz)):
        if isinstance(key, list):
            key = slice(conv(key[0]), conv(key[-1]), None)
        a = conv(key.start if key.start is not None else 0)
        b = conv(key.stop if key.stop is not None else n)
        ends.append((a, b))
    m = max(abs(b - a) for a, b in ends)
    return m, [[int(v) for v in np.linspace(a, b, m, endpoint=False).astype(np.int64)]
               if m > 0 else [] for a, b in ends]


class _Waveform:
    """Carrier parameters common to point and line sources."""

    def _init_wave(self, period, amplitude, phase_shift, pulse, cycle, hanning_dt):
        self.period = period
        self.amplitude = amplitude
        self.phase_shift = phase_shift
        self.pulse = pulse
        self.cycle = cycle
        self.frequency = 1.0 / period  # from the period as given (reference fdtd/sources.py:60)
        self.hanning_dt = hanning_dt if hanning_dt is not None else 0.5 / self.frequency

    def _wave_signature(self):
        return (self.period, self.phase_shift, bool(self.pulse), self.cycle, self.frequency,
                self.hanning_dt)

    def _wave_args(self):
        return (float(self.period), float(self.phase_shift), int(bool(self.pulse)),
                float(self.frequency), float(self.hanning_dt), float(self.cycle))

    def update_E(self):
        pass

    def update_H(self):
        pass


class PointSource(_Waveform):
    """Soft source adding to Ez of a single cell (reference fdtd/sources.py:26-131)."""

    def __init__(self, period: Number = 15, amplitude: float = 1.0, phase_shift: float = 0.0,
                 name: str = None, pulse: bool = False, cycle: int = 5,
                 hanning_dt: float = 10.0, verbose=False):
        self.grid = None
        self.name = name
        self.verbose = verbose
        self._init_wave(period, amplitude, phase_shift, pulse, cycle, hanning_dt)

    def _register_grid(self, grid: Grid, x: Number, y: Number, z: Number):
        self.grid = grid
        grid.sources.append(self)
        grid._claim_name(self)
        try:
            (x,), (y,), (z,) = x, y, z
        except (TypeError, ValueError):
            raise ValueError("a point source should be placed on a single grid cell.")
        self.x, self.y, self.z = grid._handle_tuple((x, y, z))
        self.period = grid._handle_time(self.period)

    def _signature(self):
        return ("point", id(self), self.x, self.y, self.z, self.amplitude) + self._wave_signature()

    def _native_add(self, lib, plan):
        g = self.grid
        xl = g._local_x(g._cell_index(self.x, g.Nx_global))  # negative indices wrap like Python
        if xl is None:  # the cell belongs to another x-slab
            return
        cell = [xl, g._cell_index(self.y, g.Ny), g._cell_index(self.z, g.Nz)]
        _lib.check(lib.fdtdhip_plan_add_source(
            plan, _lib.SRC_ADD_POINTS, 2, 2, 1, _lib.int32s(cell),
            _lib.doubles([self.amplitude]), *self._wave_args()))

    def __repr__(self):
        return (f"{self.__class__.__name__}(period={self.period}, amplitude={self.amplitude}, "
                f"phase_shift={self.phase_shift}, name={repr(self.name)})")

    def __str__(self):
        return "    " + repr(self) + "\n" + f"        @ x={self.x}, y={self.y}, z={self.z}\n"


class CorticalColumnPointSource(PointSource):
    """Directional point source (reference fdtd/sources.py:133-167): per axis ``a`` it adds
    ``s * dir_vec[a]`` to Ez of its cell and subtracts the same from the cell one step back
    along ``a`` (indices wrap like Python's), ``s = sin(2 pi q / period + phase_shift)``.
    As in the reference every constructor argument except ``dir_vec`` is ignored (its
    ``super().__init__()`` call passes none on)."""

    def __init__(self, period: Number = 15, amplitude: float = 1.0, phase_shift: float = 0.0,
                 name: str = None, pulse: bool = False, cycle: int = 5,
                 hanning_dt: float = 10.0, dir_vec=(1, 1, 1)):
        super().__init__()
        self.dir_vec = bd.array([float(v) for v in dir_vec])

    def _signature(self):
        return super()._signature() + tuple(float(v) for v in self.dir_vec)

    def _native_add(self, lib, plan):
        g = self.grid
        d = [float(v) for v in self.dir_vec]
        # the six additions in the reference's order; the cell itself appears three times,
        # which the library then injects serially
        cells = [(self.x - 1, self.y, self.z), (self.x, self.y, self.z),
                 (self.x, self.y - 1, self.z), (self.x, self.y, self.z),
                 (self.x, self.y, self.z - 1), (self.x, self.y, self.z)]
        weights = [-d[0], d[0], -d[1], d[1], -d[2], d[2]]
        flat, prof = [], []
        for (x, y, z), w in zip(cells, weights):
            xl = g._local_x(g._cell_index(x, g.Nx_global))
            if xl is not None:
                flat += [xl, g._cell_index(y, g.Ny), g._cell_index(z, g.Nz)]
                prof.append(self.amplitude * w)
        if prof:
            _lib.check(lib.fdtdhip_plan_add_source(
                plan, _lib.SRC_ADD_POINTS, 2, 2, len(prof), _lib.int32s(flat),
                _lib.doubles(prof), *self._wave_args()))


class LineSource(_Waveform):
    """Soft source adding to Ez along a line with a Gaussian profile whose sum is the
    amplitude (reference fdtd/sources.py:170-355)."""

    def __init__(self, period: Number = 15, amplitude: float = 1.0, phase_shift: float = 0.0,
                 name: str = None, pulse: bool = False, cycle: int = 5,
                 hanning_dt: float = 10.0):
        self.grid = None
        self.name = name
        self._init_wave(period, amplitude, phase_shift, pulse, cycle, hanning_dt)

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        self.grid = grid
        grid.sources.append(self)
        grid._claim_name(self)
        self.x, self.y, self.z = self._handle_slices(x, y, z)
        self.period = grid._handle_time(self.period)

        mid = len(self.x) // 2
        r2 = np.array(
            (np.array(self.x) - self.x[mid]) ** 2
            + (np.array(self.y) - self.y[mid]) ** 2
            + (np.array(self.z) - self.z[mid]) ** 2,
            np.float64,
        )
        profile = np.exp(-(r2 ** 2) / (2 * (0.5 * r2.max()) ** 2))
        profile /= profile.sum()
        profile *= self.amplitude
        self.profile = profile  # host float64; uploaded with the plan

    def _handle_slices(self, x, y, z) -> Tuple[List, List, List]:
        res = _line_indices(self.grid, x, y, z, convert=True)
        if len(res) == 3:
            return res
        m, (xs, ys, zs) = res
        if m < 2:
            raise ValueError("a LineSource should consist of at least two gridpoints")
        return xs, ys, zs

    def _signature(self):
        return ("line", id(self), tuple(self.x), tuple(self.y), tuple(self.z),
                self.profile.tobytes()) + self._wave_signature()

    def _native_add(self, lib, plan):
        g = self.grid
        cells, profile = [], []
        for x, y, z, p in zip(self.x, self.y, self.z, self.profile):
            xl = g._local_x(g._cell_index(x, g.Nx_global))
            if xl is not None:  # points of other x-slabs are injected there
                cells += [xl, g._cell_index(y, g.Ny), g._cell_index(z, g.Nz)]
                profile.append(p)
        if not profile:
            return
        _lib.check(lib.fdtdhip_plan_add_source(
            plan, _lib.SRC_ADD_POINTS, 2, 2, len(profile), _lib.int32s(cells),
            _lib.doubles(profile), *self._wave_args()))

    def __repr__(self):
        return (f"{self.__class__.__name__}(period={self.period}, amplitude={self.amplitude}, "
                f"phase_shift={self.phase_shift}, name={repr(self.name)})")

    def __str__(self):
        loc = ", ".join(f"{n}=[{v[0]}, ... , {v[-1]}]" for n, v in
                        (("x", self.x), ("y", self.y), ("z", self.z)))
        return "    " + repr(self) + "\n" + f"        @ {loc}\n"


class PlaneSource:
    """Hard source: overwrites one E component and one H component on a plane with
    ``amplitude * sin(...)`` every half step (reference fdtd/sources.py:359-540)."""

    def __init__(self, period: Number = 15, amplitude: float = 1.0, phase_shift: float = 0.0,
                 name: str = None, polarization: str = "z"):
        self.grid = None
        self.period = period
        self.amplitude = amplitude
        self.phase_shift = phase_shift
        self.name = name
        self.polarization = polarization

    def _register_grid(self, grid: Grid, x: ListOrSlice, y: ListOrSlice, z: ListOrSlice):
        self.grid = grid
        grid.sources.append(self)
        grid._claim_name(self)
        try:
            self.x, self.y, self.z = self._handle_slices(x, y, z)
        except ValueError:
            grid.sources.remove(self)
            raise
        self.period = grid._handle_time(self.period)
        shape = (self.x.stop - self.x.start, self.y.stop - self.y.start, self.z.stop - self.z.start)
        self.profile = self.amplitude * np.ones(shape)

    def _handle_slices(self, x, y, z):
        g = self.grid
        spans = []
        for key, n in ((x, g.Nx_global), (y, g.Ny), (z, g.Nz)):
            if not isinstance(key, slice):
                if isinstance(key, list):
                    (key,) = key
                c = g._handle_distance(key)
                key = slice(c, c + 1, None)
            a = g._handle_distance(key.start if key.start is not None else 0)
            b = g._handle_distance(key.stop if key.stop is not None else n)
            spans.append(slice(a, b) if a < b else (slice(b, a) if a > b else slice(a, a + 1)))
        extents = [s.stop - s.start for s in spans]
        if extents.count(0) > 0:
            raise ValueError("Given location for PlaneSource results in slices of length 0!")
        if extents.count(1) == 0:
            raise ValueError("Given location for PlaneSource is not a 2D plane!")
        if extents.count(1) > 1:
            raise ValueError(
                "Given location for PlaneSource should have no more than one dimension in "
                "which it's flat.\nUse a LineSource for lower dimensional sources."
            )
        self._Epol = "xyz".index(self.polarization)
        if extents[self._Epol] == 1:
            raise ValueError(
                "PlaneSource cannot be polarized perpendicular to the orientation of the plane."
            )
        # H component: transverse to E and to the propagation direction
        # (table of reference fdtd/sources.py:507-511)
        probe, first, second = [(2, 1, 2), (2, 0, 2), (1, 0, 1)][self._Epol]
        self._Hpol = first if extents[probe] == 1 else second
        return tuple(spans)

    def update_E(self):
        pass

    def update_H(self):
        pass

    def _signature(self):
        return ("plane", id(self), self.x.start, self.x.stop, self.y.start, self.y.stop,
                self.z.start, self.z.stop, self._Epol, self._Hpol, self.period,
                self.amplitude, self.phase_shift)

    def _native_add(self, lib, plan):
        lo, hi = self.grid._clip_x(self.x.start, self.x.stop)
        if hi <= lo:
            return
        box = [lo, hi, self.y.start, self.y.stop, self.z.start, self.z.stop]
        _lib.check(lib.fdtdhip_plan_add_source(
            plan, _lib.SRC_SET_BOX, self._Epol, self._Hpol, 1, _lib.int32s(box),
            _lib.doubles([self.amplitude]), float(self.period), float(self.phase_shift),
            0, 0.0, 0.0, 1.0))

    def __repr__(self):
        return (f"{self.__class__.__name__}(period={self.period}, amplitude={self.amplitude}, "
                f"phase_shift={self.phase_shift}, name={repr(self.name)}, "
                f"polarization={repr(self.polarization)})")

    def __str__(self):
        loc = ", ".join(f"{n}=[{s.start}, ... , {s.stop}]" for n, s in
                        (("x", self.x), ("y", self.y), ("z", self.z)))
        return "    " + repr(self) + "\n" + f"        @ {loc}\n"


class CorticalColumnPlaneSource(PlaneSource):
    """Additive source over a plane of "cortical columns" (reference
    fdtd/sources.py:542-608): every column ``c`` has an activation image, a 3x3 direction
    kernel, an oscillator (frequency, phase) and a field-energy dependent gate
    ``sigmoid(nonlin_conv(energies))``; each E half step adds

        amp * sum_c sin(2 pi t f_c + phi_c) * footprint_c * activation_c * gate_c

    to ``E[..., 2]`` on the plane (``t`` = physical time).  ``seed`` provides the column
    data (torch tensors, which may require grad); ``nonlin_conv`` is a
    ``torch.nn.Conv2d(2, num_ccs, 1)`` whose parameters learn.  An unseeded source does
    nothing.  The field-independent factors are prepared with torch (so autograd reaches
    every seed tensor), the energy gate and the injection run in the native kernels."""

    def __init__(self, period: Number = 15, amplitude: float = 1.0, phase_shift: float = 0.0,
                 name: str = None, polarization: str = "z", num_cortical_columns=16):
        # the plane source underneath never fires (reference fdtd/sources.py:553)
        super().__init__(period, 0, phase_shift, name, polarization)
        self.num_ccs = int(num_cortical_columns)
        self.cc_dirs = None
        kw = {} if bd.staging else {"device": bd.device}
        self.nonlin_conv = torch.nn.Conv2d(2, self.num_ccs, kernel_size=1, stride=1,
                                           dtype=bd.float, **kw)
        self._native_id = None
        self._held = None        # tensors the plan currently points to
        self._recording = None   # set by the autograd node while it records a run

    def seed(self, cc_activations, cc_dirs, cc_freqs, cc_phases, amp_scaler):
        """cc_activations (1, num_ccs, H, W) on the source's plane, cc_dirs (1, num_ccs, 3, 3),
        cc_freqs and cc_phases (num_ccs,), amp_scaler a number or tensor."""
        self.cc_activations = cc_activations
        self.cc_dirs = cc_dirs
        self.cc_freqs = cc_freqs
        self.cc_phases = cc_phases
        self.amp_scaler = amp_scaler

    def _signature(self):
        return ("cortical", id(self), self.x.start, self.x.stop, self.y.start, self.y.stop,
                self.z.start, self.z.stop, self.num_ccs)

    def _local_box(self):
        lo, hi = self.grid._clip_x(self.x.start, self.x.stop)
        return [lo, hi, self.y.start, self.y.stop, 0, self.grid.Nz]

    def _native_add(self, lib, plan):
        self._native_id = None
        box = self._local_box()
        if box[1] <= box[0]:
            return
        # the reference always adds to the last component (fdtd/sources.py:608)
        self._native_id = _lib.check(lib.fdtdhip_plan_add_modulated_source(
            plan, _lib.int32s(box), 2, self.num_ccs))

    # -- field-independent factors, with torch so that gradients reach the seed tensors
    def _torch_inputs(self, q0, n):
        """(P, osc): P (num_ccs, local columns) = amp * footprint * activation on this
        rank's part of the plane; osc (n, num_ccs) oscillator values of steps q0..q0+n-1."""
        g = self.grid
        ih, iw = self.x.stop - self.x.start, self.y.stop - self.y.start
        dirs = self.cc_dirs.to(device=bd.device, dtype=bd.float)
        unit = dirs / dirs.reshape(1, self.num_ccs, -1).sum(-1)[:, :, None, None]
        ones = torch.ones((1, ih - 2, iw - 2), device=bd.device, dtype=bd.float)
        footprint = torch.conv_transpose2d(ones, unit, bias=None, stride=1)   # (num_ccs, ih, iw)
        acts = self.cc_activations.to(device=bd.device, dtype=bd.float)
        P = (footprint[None] * acts * self.amp_scaler)[0]
        lo, hi = g._clip_x(self.x.start, self.x.stop)          # local planes of the box
        first = lo + g.x_offset - self.x.start
        P = P[:, first:first + (hi - lo)].reshape(self.num_ccs, -1).contiguous()
        t = (q0 + torch.arange(n, device=bd.device, dtype=bd.float)) * g.time_step
        freqs = self.cc_freqs.to(device=bd.device, dtype=bd.float)
        phases = self.cc_phases.to(device=bd.device, dtype=bd.float)
        osc = torch.sin(2 * np.pi * t[:, None] * freqs[None, :] + phases[None, :]).contiguous()
        return P, osc

    def _packed_weights(self):
        w, b = self.nonlin_conv.weight, self.nonlin_conv.bias
        return torch.cat([w.detach().reshape(self.num_ccs, 2), b.detach().reshape(self.num_ccs, 1)],
                         dim=1).to(device=bd.device, dtype=bd.float).contiguous()

    def _native_prepare(self, lib, plan, q0, n):
        """Give the plan this source's inputs for steps q0..q0+n-1 (called before every run)."""
        if self._native_id is None:
            return
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        if self.cc_dirs is None:
            _lib.check(lib.fdtdhip_plan_set_modulated_source(
                plan, self._native_id, None, None, 0, 0, None, None, None, None, None))
            return
        if self._recording is not None:
            P, osc, wb, trace, gP, gosc, gwb, q0 = self._recording
        else:
            with torch.no_grad():
                P, osc = self._torch_inputs(q0, n)
            wb, trace, gP, gosc, gwb = self._packed_weights(), None, None, None, None
        self._held = (P, osc, wb, trace, gP, gosc, gwb)
        _lib.check(lib.fdtdhip_plan_set_modulated_source(
            plan, self._native_id, ptr(P), ptr(osc), osc.shape[0], q0, ptr(wb), ptr(trace),
            ptr(gP), ptr(gosc), ptr(gwb)))


class SoftArbitraryPointSource:
    """Voltage source with a series resistance at one cell (reference
    fdtd/sources.py:612-740): every step adds ``(waveform_array[q] + impedance * I) /
    grid_spacing`` to Ez, ``I`` being the reading its own ``CurrentDetector`` (registered at
    the same cell, ``self.current_detector``) took after the previous step.  The feedback
    runs on the device; ``input_voltage`` and ``source_voltage`` are lists of per-step
    device scalars of shape (1, 1, 1), like the reference's nested lists.  On an x-slab grid
    the rank whose slab holds the cell runs the source and fills the two lists; on the other
    ranks they stay empty."""

    # Differentiable runs carry the feedback (the adjoint of Ez at the cell reaches the adjoint
    # of H around it, as on the reference's tape; on x-slabs also across a slab face, see
    # SlabGrid._native_adjoint); the waveform is a constant.
    _no_autograd = None

    def __init__(self, waveform_array, name: str = None, impedance: float = 0.0):
        self.grid = None
        self.name = name
        self.current_detector = None
        self.waveform_array = waveform_array
        self.impedance = impedance
        self.input_voltage = []
        self.source_voltage = []
        self._native_id = None
        self._vin = self._vout = None

    def _register_grid(self, grid: Grid, x: Number, y: Number, z: Number):
        from .detectors import CurrentDetector, _Readings

        self.grid = grid
        grid.sources.append(self)
        grid._claim_name(self)
        try:
            (x,), (y,), (z,) = x, y, z
        except (TypeError, ValueError):
            raise ValueError("a point source should be placed on a single grid cell.")
        self.x, self.y, self.z = grid._handle_tuple((x, y, z))
        # the reference means to call the detector "<name>_I" (fdtd/sources.py:700-703)
        self.current_detector = CurrentDetector(
            name=None if self.name is None else self.name + "_I")
        grid[self.x, self.y, self.z] = self.current_detector
        self._vin, self._vout = _Readings((1, 1, 1)), _Readings((1, 1, 1))

    def _signature(self):
        w = np.asarray(bd.numpy(self.waveform_array) if torch.is_tensor(self.waveform_array)
                       else self.waveform_array, dtype=np.float64)
        return ("soft", id(self), self.x, self.y, self.z, float(self.impedance), w.tobytes())

    def _native_add(self, lib, plan):
        g = self.grid
        w = np.ascontiguousarray(
            bd.numpy(self.waveform_array) if torch.is_tensor(self.waveform_array)
            else self.waveform_array, dtype=np.float64).reshape(-1)
        # on x-slabs the rank that holds the cell runs the source and keeps its two records
        xl = g._local_x(g._cell_index(self.x, g.Nx_global))
        self._native_id = None
        if xl is None:
            return
        cell = [xl, g._cell_index(self.y, g.Ny), g._cell_index(self.z, g.Nz)]
        self._native_id = _lib.check(lib.fdtdhip_plan_add_soft_source(
            plan, _lib.int32s(cell), w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), w.size,
            float(self.impedance), float(g.grid_spacing)))

    def _native_link(self, lib, plan):
        """After all components are in the plan: name the detector that feeds this source."""
        if self._native_id is not None:
            _lib.check(lib.fdtdhip_plan_link_soft_source(plan, self._native_id, self.current_detector._id))

    def _native_prepare(self, lib, plan, q0, n):
        if self._native_id is None:
            return
        last = self.current_detector.I[-1] if len(self.current_detector.I) else None
        pin, pout = self._vin.reserve(n), self._vout.reserve(n)
        _lib.check(lib.fdtdhip_plan_set_soft_source_io(
            plan, self._native_id, ctypes.c_void_p(last.data_ptr()) if last is not None else None,
            ctypes.c_void_p(pin), ctypes.c_void_p(pout), n))

    def _native_finish(self, n):
        if self._native_id is None:
            return
        self._vin.commit(n, self.input_voltage)
        self._vout.commit(n, self.source_voltage)

    def update_E(self):
        pass

    def update_H(self):
        pass

    def __repr__(self):
        return f"{self.__class__.__name__}()"   # as the reference prints it (fdtd/sources.py:742-743)

    def __str__(self):
        return "    " + repr(self) + "\n" + f"        @ x={self.x}, y={self.y}, z={self.z}\n"
