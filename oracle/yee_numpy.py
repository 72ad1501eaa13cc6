"""CPU restatement (numpy, float64) of the reference's Yee time-stepping path.

TEST INFRASTRUCTURE ONLY.  This module is the parity oracle for the HIP engine:
it may be imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- never by the ``fdtd_amd`` package.

It restates, with the same slice algebra and the same order of floating-point
operations, what simenhu/fdtd's numpy backend computes for

    * the two curls                      (reference fdtd/grid.py:29-76)
    * Grid.update_E / update_H / step    (reference fdtd/grid.py:267-325)
    * PML and periodic boundaries        (reference fdtd/boundaries.py:173-208, 356-614)
    * Object / AbsorbingObject / AnisotropicObject
                                         (reference fdtd/objects.py:60-90, 116-126,
                                          188-218, 244-266)
    * Point / Line / Plane sources       (reference fdtd/sources.py:95-111, 235-245,
                                          316-337, 515-525) and the hanning pulse
                                         (reference fdtd/waveforms.py:8-9)
    * Line / Block / Current detectors   (reference fdtd/detectors.py:114-124,
                                          241-265, 443-480)

Parity pin: ``tests/golden/make_golden.py`` runs every scene through the real
reference (imported from its checkout) and through this module and requires
exact (bitwise) equality before it writes the fixtures that
``tests/test_oracle_golden.py`` replays without the reference.  Beyond those scenes,
``tools/fuzz_reference.py`` runs seeded random scenes (``tests/fuzz_scenes.py``: random
shapes, boundary orders, overlapping objects, sources, detectors) through the reference and
through this module and requires the same equality; 720 scenes, no difference.

All coordinates taken here are already integers (cells / time steps given as
ints, or seconds converted with ``steps_from_seconds``); the float-to-cell
conversions of the user-facing API are host logic of the engine under test.

Two extensions that the reference does not have are clearly marked: ghost
planes for x-slab decomposition (``halo_lo_H`` / ``halo_hi_E``), used only to
check that a slab-decomposed run equals the single-grid run.

float32 twin.  The reference has no float32 trajectory (its ``*.float32`` backends still
allocate float64 arrays, reference fdtd/backend.py:87, 265-268), so there is nothing to pin
one against.  ``OracleGrid(..., dtype=np.float32)`` runs the same slice algebra on float32
arrays -- every array operation rounds to float32; set-up quantities (PML profiles, source
amplitudes) are computed in float64 and rounded once, which is what the engine under test
does -- to tell apart, for a float32 result that is far from the float64 one, inherent
float32 rounding (twin and engine agree) from a difference in the arithmetic (they do
not).  With the default ``dtype=np.float64`` nothing changes: that path stays bitwise
equal to the reference.
"""

from math import pi, sin, cos

import numpy as np

C0 = 299792458.0  # reference fdtd/constants.py:5
MU0 = 4e-7 * pi  # reference fdtd/constants.py:17
ETA0 = MU0 * C0  # reference fdtd/constants.py:23


# ----------------------------------------------------------------------------
# curls
# ----------------------------------------------------------------------------
def curl_E(E, halo_hi=None):
    """Forward-difference curl of an E-type field; every difference is dropped
    at the last index of its own axis (reference fdtd/grid.py:29-51).

    ``halo_hi`` (extension): the E plane that follows the last x plane in a
    slab-decomposed run; with it the x differences of the last plane are kept.
    """
    out = np.zeros(E.shape, E.dtype)
    out[:, :-1, :, 0] += E[:, 1:, :, 2] - E[:, :-1, :, 2]
    out[:, :, :-1, 0] -= E[:, :, 1:, 1] - E[:, :, :-1, 1]
    out[:, :, :-1, 1] += E[:, :, 1:, 0] - E[:, :, :-1, 0]
    out[:-1, :, :, 1] -= E[1:, :, :, 2] - E[:-1, :, :, 2]
    out[:-1, :, :, 2] += E[1:, :, :, 1] - E[:-1, :, :, 1]
    out[:, :-1, :, 2] -= E[:, 1:, :, 0] - E[:, :-1, :, 0]
    if halo_hi is not None:
        # same position in the accumulation order as the in-array x terms
        # cannot be reproduced bit for bit here (y term of comp 2 comes after);
        # the slab test therefore compares at 1e-13, not exactly.
        out[-1, :, :, 1] -= halo_hi[:, :, 2] - E[-1, :, :, 2]
        out[-1, :, :, 2] += halo_hi[:, :, 1] - E[-1, :, :, 1]
    return out


def curl_H(H, halo_lo=None):
    """Backward-difference curl of an H-type field; every difference is dropped
    at index 0 of its own axis (reference fdtd/grid.py:54-76).

    ``halo_lo`` (extension): the H plane that precedes x plane 0 in a
    slab-decomposed run.
    """
    out = np.zeros(H.shape, H.dtype)
    out[:, 1:, :, 0] += H[:, 1:, :, 2] - H[:, :-1, :, 2]
    out[:, :, 1:, 0] -= H[:, :, 1:, 1] - H[:, :, :-1, 1]
    out[:, :, 1:, 1] += H[:, :, 1:, 0] - H[:, :, :-1, 0]
    out[1:, :, :, 1] -= H[1:, :, :, 2] - H[:-1, :, :, 2]
    out[1:, :, :, 2] += H[1:, :, :, 1] - H[:-1, :, :, 1]
    out[:, 1:, :, 2] -= H[:, 1:, :, 0] - H[:, :-1, :, 0]
    if halo_lo is not None:
        out[0, :, :, 1] -= H[0, :, :, 2] - halo_lo[:, :, 2]
        out[0, :, :, 2] += H[0, :, :, 1] - halo_lo[:, :, 1]
    return out


def hanning(f, t, n):
    """Hanning-windowed sine (reference fdtd/waveforms.py:8-9)."""
    return (1 / 2) * (1 - cos(f * t / n)) * (sin(f * t))


# ----------------------------------------------------------------------------
# boundaries
# ----------------------------------------------------------------------------
class OraclePML:
    """One PML slab (reference fdtd/boundaries.py:356-476 and the six placement
    classes 479-614).  ``axis`` in 0..2, ``side`` 'low' or 'high'."""

    def __init__(self, grid, axis, side, thickness, a=1e-8):
        t = int(thickness)
        self.grid, self.axis, self.side, self.thickness, self.a, self.k = grid, axis, side, t, a, 1.0
        n = grid.shape[axis]
        span = slice(None, t) if side == "low" else slice(-t, None)
        cells = [slice(None)] * 3
        cells[axis] = span
        self.cells = tuple(cells)
        shape = list(grid.shape)
        shape[axis] = t
        self.shape = tuple(shape)

        def profile(v):  # reference fdtd/boundaries.py:280-282
            return 40 * v ** 3 / (t + 1) ** 4

        if side == "low":  # reference fdtd/boundaries.py:491-499 (and Y/Z twins)
            sE = profile(np.arange(t - 0.5, -0.5, -1.0))
            sH = profile(np.arange(t - 1.0, 0, -1.0))
        else:  # reference fdtd/boundaries.py:514-522
            sE = profile(np.arange(0.5, t + 0.5, 1.0))
            sH = profile(np.arange(1.0, t, 1.0))
        bshape = [1, 1, 1]
        bshape[axis] = -1
        self.sigmaE = np.zeros(self.shape + (3,))
        self.sigmaE[..., axis] = sE.reshape(bshape)
        self.sigmaH = np.zeros(self.shape + (3,))
        head = [slice(None)] * 3
        head[axis] = slice(None, -1)
        self.sigmaH[tuple(head) + (axis,)] = sH.reshape(bshape)

        dt = grid.dtype
        z = lambda: np.zeros(self.shape + (3,), dt)
        self.phi_E, self.phi_H = z(), z()
        self.psi_E = [z(), z(), z()]  # psi_Ex, psi_Ey, psi_Ez
        self.psi_H = [z(), z(), z()]
        sc = grid.courant_number
        # reference fdtd/boundaries.py:384-396
        self.bE = np.exp(-(self.sigmaE / self.k + self.a) * sc)
        self.cE = (self.bE - 1.0) * self.sigmaE / (self.sigmaE * self.k + self.a * self.k ** 2)
        self.bH = np.exp(-(self.sigmaH / self.k + self.a) * sc)
        self.cH = (self.bH - 1.0) * self.sigmaH / (self.sigmaH * self.k + self.a * self.k ** 2)
        if dt != np.float64:   # float32 twin: profiles in float64, rounded once
            self.bE, self.cE, self.bH, self.cH = (v.astype(dt) for v in (self.bE, self.cE, self.bH, self.cH))

    # (psi index u, difference axis d, field component w), in the reference's order
    _TERMS = ((0, 1, 2), (0, 2, 1), (1, 2, 0), (1, 0, 2), (2, 0, 1), (2, 1, 0))

    def update_phi_E(self):  # reference fdtd/boundaries.py:422-448
        for p in self.psi_E:
            p *= self.bE
        c = self.cE
        F = [self.grid.H[self.cells + (w,)] for w in range(3)]
        for u, d, w in self._TERMS:
            hi = [slice(None)] * 3
            lo = [slice(None)] * 3
            hi[d], lo[d] = slice(1, None), slice(None, -1)
            hi, lo = tuple(hi), tuple(lo)
            self.psi_E[u][hi + (d,)] += (F[w][hi] - F[w][lo]) * c[hi + (d,)]
        self.phi_E[..., 0] = self.psi_E[0][..., 1] - self.psi_E[0][..., 2]
        self.phi_E[..., 1] = self.psi_E[1][..., 2] - self.psi_E[1][..., 0]
        self.phi_E[..., 2] = self.psi_E[2][..., 0] - self.psi_E[2][..., 1]

    def update_phi_H(self):  # reference fdtd/boundaries.py:450-476
        for p in self.psi_H:
            p *= self.bH
        c = self.cH
        F = [self.grid.E[self.cells + (w,)] for w in range(3)]
        for u, d, w in self._TERMS:
            hi = [slice(None)] * 3
            lo = [slice(None)] * 3
            hi[d], lo[d] = slice(1, None), slice(None, -1)
            hi, lo = tuple(hi), tuple(lo)
            self.psi_H[u][lo + (d,)] += (F[w][hi] - F[w][lo]) * c[lo + (d,)]
        self.phi_H[..., 0] = self.psi_H[0][..., 1] - self.psi_H[0][..., 2]
        self.phi_H[..., 1] = self.psi_H[1][..., 2] - self.psi_H[1][..., 0]
        self.phi_H[..., 2] = self.psi_H[2][..., 0] - self.psi_H[2][..., 1]

    def update_E(self):  # reference fdtd/boundaries.py:398-408
        g = self.grid
        g.E[self.cells] += g.courant_number * g.inverse_permittivity[self.cells] * self.phi_E

    def update_H(self):  # reference fdtd/boundaries.py:410-420
        g = self.grid
        g.H[self.cells] -= g.courant_number * g.inverse_permeability[self.cells] * self.phi_H


class OraclePeriodic:
    """Periodic boundary along one axis (reference fdtd/boundaries.py:173-208)."""

    def __init__(self, grid, axis):
        self.grid, self.axis = grid, axis
        first, last = [slice(None)] * 3, [slice(None)] * 3
        first[axis], last[axis] = 0, -1
        self.first, self.last = tuple(first), tuple(last)

    def update_phi_E(self):
        pass

    def update_phi_H(self):
        pass

    def update_E(self):
        self.grid.E[self.first] = self.grid.E[self.last]

    def update_H(self):
        self.grid.H[self.last] = self.grid.H[self.first]


# ----------------------------------------------------------------------------
# objects
# ----------------------------------------------------------------------------
class OracleObject:
    """Dielectric box.  ``kind``: 'object' | 'absorbing' | 'anisotropic' | 'substrate'
    (reference fdtd/objects.py:60-90, 116-126, 188-218, 244-266; 'substrate' is the
    LearnableAnisotropicObject with is_substrate=True of 276-349: an anisotropic tensor whose
    nine entries are scaled, per cell and step, by sigmoid(w_E*sum(E^2) + w_H*sum(H^2) + b);
    ``modifier_weights`` holds (w_E, w_H, b) per entry, shape (9, 3), set by the caller.
    The reference's reshape of the 1x1 convolution's output (334-337) only works for boxes
    one cell thick along z)."""

    def __init__(self, grid, kind, box, permittivity, conductivity=None):
        x0, x1, y0, y1, z0, z1 = box
        self.grid, self.kind = grid, kind
        self.loc = (slice(x0, x1), slice(y0, y1), slice(z0, z1))
        self.Nx, self.Ny, self.Nz = abs(x1 - x0), abs(y1 - y0), abs(z1 - z0)
        dt = grid.dtype
        perm = np.array(permittivity)
        if perm.ndim == 3:
            perm = perm[:, :, :, None]
        inv = np.ones((self.Nx, self.Ny, self.Nz, 3), dt) / perm.astype(dt)
        gi = grid.inverse_permittivity
        x, y, z = self.loc
        if self.Nx > 1:
            inv[-1, :, :, 0] = gi[-1, y, z, 0]
        if self.Ny > 1:
            inv[:, -1, :, 1] = gi[x, -1, z, 1]
        if self.Nz > 1:
            inv[:, :, -1, 2] = gi[x, y, -1, 2]
        gi[x, y, z] = 0
        self.inverse_permittivity = inv
        if kind == "absorbing":
            cond = np.asarray(conductivity).astype(dt)
            while cond.ndim < inv.ndim:
                cond = cond[..., None]
            cond = np.broadcast_to(cond, inv.shape)
            self.absorption_factor = (
                0.5 * grid.courant_number * inv * cond * grid.grid_spacing * ETA0
            )
        elif kind in ("anisotropic", "substrate"):
            if kind == "substrate":
                if self.Nz != 1:
                    raise ValueError("a substrate object has to be one cell thick along z")
                self.modifier_weights = np.zeros((9, 3))
            eye = np.zeros((self.Nx * self.Ny * self.Nz, 3, 3), dt)
            eye[:, range(3), range(3)] = 1.0
            self.inverse_permittivity = np.reshape(
                np.reshape(inv, (-1, 1, 3)) * eye, (self.Nx, self.Ny, self.Nz, 3, 3)
            )
        elif kind != "object":
            raise ValueError(kind)

    def update_E(self, curl):
        g, loc = self.grid, self.loc
        sc = g.courant_number
        if self.kind == "object":
            g.E[loc] += sc * self.inverse_permittivity * curl[loc]
        elif self.kind == "absorbing":
            f = self.absorption_factor
            g.E[loc] *= (1 - f) / (1 + f)
            g.E[loc] += sc * self.inverse_permittivity * curl[loc] / (1 + f)
        else:
            tensor = self.inverse_permittivity
            if self.kind == "substrate":
                # energies of the fields as they are now: E after the grid's own update and
                # the objects registered earlier (reference fdtd/objects.py:328-337)
                eE = np.sum(g.E[loc] ** 2, -1)
                eH = np.sum(g.H[loc] ** 2, -1)
                w = self.modifier_weights
                z = w[:, 0][None, None, None, :] * eE[..., None] + w[:, 1][None, None, None, :] * eH[..., None] \
                    + w[:, 2][None, None, None, :]
                with np.errstate(over="ignore"):     # exp(+big) = inf gives the limit 0
                    mod = 1.0 / (1.0 + np.exp(-z))
                tensor = tensor * np.reshape(mod, (self.Nx, self.Ny, self.Nz, 3, 3))
            prod = np.einsum(
                "ijk,ikl->ijl",
                np.reshape(tensor, (-1, 3, 3)),
                np.reshape(curl[loc], (-1, 3, 1)),
            )
            g.E[loc] += np.reshape(sc * prod, (self.Nx, self.Ny, self.Nz, 3))


# ----------------------------------------------------------------------------
# sources
# ----------------------------------------------------------------------------
class _Waveform:
    """sin carrier or hanning pulse shared by the point and line sources
    (reference fdtd/sources.py:97-110, 318-333)."""

    def __init__(self, grid, period, phase_shift, pulse, cycle, hanning_dt):
        self.frequency = 1.0 / period  # from the period as given, before conversion
        self.period = grid.steps_from_time(period)
        self.phase_shift, self.pulse, self.cycle = phase_shift, pulse, cycle
        self.hanning_dt = hanning_dt if hanning_dt is not None else 0.5 / self.frequency

    def factor(self, q):
        if self.pulse:
            t1 = int(2 * pi / (self.frequency * self.hanning_dt / self.cycle))
            if q < t1:
                return hanning(self.frequency, q * self.hanning_dt, self.cycle)
            return 0
        return sin(2 * pi * q / self.period + self.phase_shift)


class OraclePointSource:
    def __init__(self, grid, cell, period=15, amplitude=1.0, phase_shift=0.0,
                 pulse=False, cycle=5, hanning_dt=10.0):
        self.grid, self.cell, self.amplitude = grid, tuple(cell), amplitude
        self.wave = _Waveform(grid, period, phase_shift, pulse, cycle, hanning_dt)

    def update_E(self):  # reference fdtd/sources.py:95-111
        x, y, z = self.cell
        self.grid.E[x, y, z, 2] += self.amplitude * self.wave.factor(self.grid.time_steps_passed)

    def update_H(self):
        pass


class OracleCCPointSource:
    """Directional point source (reference fdtd/sources.py:133-167); amplitude 1, period 15,
    no phase: its constructor forwards none of its arguments."""

    def __init__(self, grid, cell, dir_vec):
        self.grid, self.cell = grid, tuple(cell)
        self.dir_vec = np.array(dir_vec, dtype=np.float64)
        self.amplitude, self.period, self.phase_shift = 1.0, 15, 0.0

    def update_E(self):
        q = self.grid.time_steps_passed
        src = self.amplitude * sin(2 * pi * q / self.period + self.phase_shift)
        x, y, z = self.cell
        E = self.grid.E
        d = [self.grid.dtype.type(src * v) for v in self.dir_vec]   # (a no-op in float64)
        E[x - 1, y, z, 2] -= d[0]
        E[x, y, z, 2] += d[0]
        E[x, y - 1, z, 2] -= d[1]
        E[x, y, z, 2] += d[1]
        E[x, y, z - 1, 2] -= d[2]
        E[x, y, z, 2] += d[2]

    def update_H(self):
        pass


class OracleSoftSource:
    """Voltage source with a series resistance: the voltage of the step plus impedance x
    the current its own loop detector read after the previous step, divided by the cell
    size, is added to Ez (reference fdtd/sources.py:708-737; the detector is registered
    at the same cell, sources.py:705-706)."""

    def __init__(self, grid, cell, waveform, impedance, detector):
        self.grid, self.cell = grid, tuple(cell)
        self.waveform_array = np.asarray(waveform)
        self.impedance = impedance
        self.current_detector = detector
        self.input_voltage, self.source_voltage = [], []

    def update_E(self):
        q = self.grid.time_steps_passed
        input_voltage = self.waveform_array[q] if q < self.waveform_array.shape[0] else 0.0
        current = float(self.current_detector.I[-1][0][0][0]) if q > 0 else 0.0
        if self.impedance > 0:
            output_voltage = input_voltage + self.impedance * current
        else:
            output_voltage = input_voltage
        x, y, z = self.cell
        self.grid.E[x, y, z, 2] += output_voltage / self.grid.grid_spacing
        self.input_voltage.append([[[input_voltage]]])
        self.source_voltage.append([[[output_voltage]]])

    def update_H(self):
        pass


class OracleLineSource:
    def __init__(self, grid, xs, ys, zs, period=15, amplitude=1.0, phase_shift=0.0,
                 pulse=False, cycle=5, hanning_dt=10.0):
        self.grid = grid
        self.x, self.y, self.z = list(xs), list(ys), list(zs)
        self.wave = _Waveform(grid, period, phase_shift, pulse, cycle, hanning_dt)
        L = len(self.x)  # reference fdtd/sources.py:235-245
        vect = np.array(
            (np.array(self.x) - self.x[L // 2]) ** 2
            + (np.array(self.y) - self.y[L // 2]) ** 2
            + (np.array(self.z) - self.z[L // 2]) ** 2,
            np.float64,
        )
        self.profile = np.exp(-(vect ** 2) / (2 * (0.5 * vect.max()) ** 2))
        self.profile /= self.profile.sum()
        self.profile *= amplitude

    def update_E(self):  # reference fdtd/sources.py:316-337
        vect = self.profile * self.wave.factor(self.grid.time_steps_passed)
        for x, y, z, value in zip(self.x, self.y, self.z, vect):
            self.grid.E[x, y, z, 2] += self.grid.dtype.type(value)

    def update_H(self):
        pass


class OraclePlaneSource:
    """Hard (assigning) plane source (reference fdtd/sources.py:416-427, 500-525)."""

    def __init__(self, grid, box, period=15, amplitude=1.0, phase_shift=0.0, polarization="z"):
        x0, x1, y0, y1, z0, z1 = box
        self.grid = grid
        self.loc = (slice(x0, x1), slice(y0, y1), slice(z0, z1))
        self.period = grid.steps_from_time(period)
        self.phase_shift = phase_shift
        self.Epol = "xyz".index(polarization)
        zflat = (z1 - z0) == 1
        yflat = (y1 - y0) == 1
        flat_first, a, b = [(zflat, 1, 2), (zflat, 0, 2), (yflat, 0, 1)][self.Epol]
        self.Hpol = a if flat_first else b
        self.profile = amplitude * np.ones((x1 - x0, y1 - y0, z1 - z0))

    def _vect(self):
        q = self.grid.time_steps_passed
        return self.profile * sin(2 * pi * q / self.period + self.phase_shift)

    def update_E(self):
        self.grid.E[self.loc + (self.Epol,)] = self._vect()

    def update_H(self):
        self.grid.H[self.loc + (self.Hpol,)] = self._vect()


# ----------------------------------------------------------------------------
# detectors
# ----------------------------------------------------------------------------
class OracleLineDetector:
    def __init__(self, grid, xs, ys, zs):
        self.grid, self.x, self.y, self.z = grid, list(xs), list(ys), list(zs)
        self.E, self.H = [], []

    def detect_E(self):  # reference fdtd/detectors.py:114-118
        self.E.append(self.grid.E[self.x, self.y, self.z])

    def detect_H(self):  # reference fdtd/detectors.py:120-124
        self.H.append(self.grid.H[self.x, self.y, self.z])


class OracleBlockDetector:
    """Outer product of three index lists (reference fdtd/detectors.py:241-265);
    each reading is stored as one (len(x), len(y), len(z), 3) array."""

    def __init__(self, grid, xs, ys, zs):
        self.grid, self.x, self.y, self.z = grid, list(xs), list(ys), list(zs)
        self.E, self.H = [], []

    def _read(self, F):
        out = np.empty((len(self.x), len(self.y), len(self.z), 3), F.dtype)
        for i, row in enumerate(self.x):
            for j, col in enumerate(self.y):
                for k, pillar in enumerate(self.z):
                    out[i, j, k] = F[row, col, [pillar]][0]
        return out

    def detect_E(self):
        self.E.append(self._read(self.grid.E))

    def detect_H(self):
        self.H.append(self._read(self.grid.H))


class OracleCurrentDetector:
    """Loop current around z from eight H samples, two layers averaged
    (reference fdtd/detectors.py:443-480).  Index -1 wraps like Python."""

    def __init__(self, grid, xs, ys, zs):
        self.grid, self.x, self.y, self.z = grid, list(xs), list(ys), list(zs)
        self.I = []

    def _point(self, px, py, pz):
        H, d = self.grid.H, self.grid.grid_spacing
        v1 = ((H[px, py - 1, pz, 0]) - (H[px, py, pz, 0])) * d
        v2 = ((H[px, py, pz, 1]) - (H[px - 1, py, pz, 1])) * d
        current_1 = v1 + v2
        v1 = ((H[px, py - 1, pz - 1, 0]) - (H[px, py, pz - 1, 0])) * d
        v2 += ((H[px, py, pz - 1, 1]) - (H[px - 1, py, pz - 1, 1])) * d
        current_2 = v1 + v2
        return (current_1 + current_2) / 2.0

    def detect_E(self):
        pass

    def detect_H(self):
        out = np.empty((len(self.x), len(self.y), len(self.z)), self.grid.H.dtype)
        for i, row in enumerate(self.x):
            for j, col in enumerate(self.y):
                for k, pillar in enumerate(self.z):
                    out[i, j, k] = self._point(row, col, pillar)
        self.I.append(out)


# ----------------------------------------------------------------------------
# grid
# ----------------------------------------------------------------------------
class OracleGrid:
    """State + half-step drivers (reference fdtd/grid.py:110-154, 250-325)."""

    def __init__(self, shape, grid_spacing=155e-9, permittivity=1.0, permeability=1.0,
                 courant_number=None, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.grid_spacing = float(grid_spacing)
        self.Nx, self.Ny, self.Nz = (int(s) for s in shape)
        self.shape = (self.Nx, self.Ny, self.Nz)
        self.D = int(self.Nx > 1) + int(self.Ny > 1) + int(self.Nz > 1)
        limit = float(self.D) ** (-0.5)
        if courant_number is None:
            self.courant_number = 0.99 * limit
        elif courant_number > limit:
            raise ValueError("courant_number too high")
        else:
            self.courant_number = float(courant_number)
        self.time_step = self.courant_number * self.grid_spacing / C0
        full = (self.Nx, self.Ny, self.Nz, 3)
        self.E = np.zeros(full, self.dtype)
        self.H = np.zeros(full, self.dtype)
        self.inverse_permittivity = np.ones(full, self.dtype) / float(permittivity)
        self.inverse_permeability = np.ones(full, self.dtype) / float(permeability)
        self.time_steps_passed = 0
        self.sources, self.boundaries, self.detectors, self.objects = [], [], [], []
        # extension (not in the reference): x-slab ghost planes
        self.halo_lo_H = None
        self.halo_hi_E = None

    def steps_from_time(self, t):  # reference fdtd/grid.py:177-181
        if not isinstance(t, int):
            return int(float(t) / self.time_step + 0.5)
        return t

    # registration helpers ---------------------------------------------------
    def add_pml(self, axis, side, thickness, a=1e-8):
        b = OraclePML(self, axis, side, thickness, a)
        self.boundaries.append(b)
        return b

    def add_periodic(self, axis):
        b = OraclePeriodic(self, axis)
        self.boundaries.append(b)
        return b

    def add_object(self, box, permittivity, kind="object", conductivity=None):
        o = OracleObject(self, kind, box, permittivity, conductivity)
        self.objects.append(o)
        return o

    def add_point_source(self, cell, **kw):
        s = OraclePointSource(self, cell, **kw)
        self.sources.append(s)
        return s

    def add_line_source(self, xs, ys, zs, **kw):
        s = OracleLineSource(self, xs, ys, zs, **kw)
        self.sources.append(s)
        return s

    def add_plane_source(self, box, **kw):
        s = OraclePlaneSource(self, box, **kw)
        self.sources.append(s)
        return s

    def add_cc_point_source(self, cell, dir_vec):
        s = OracleCCPointSource(self, cell, dir_vec)
        self.sources.append(s)
        return s

    def add_soft_source(self, cell, waveform, impedance, det_xs, det_ys, det_zs):
        d = self.add_current_detector(det_xs, det_ys, det_zs)
        s = OracleSoftSource(self, cell, waveform, impedance, d)
        self.sources.append(s)
        return s

    def add_line_detector(self, xs, ys, zs):
        d = OracleLineDetector(self, xs, ys, zs)
        self.detectors.append(d)
        return d

    def add_block_detector(self, xs, ys, zs):
        d = OracleBlockDetector(self, xs, ys, zs)
        self.detectors.append(d)
        return d

    def add_current_detector(self, xs, ys, zs):
        d = OracleCurrentDetector(self, xs, ys, zs)
        self.detectors.append(d)
        return d

    # stepping -----------------------------------------------------------------
    def update_E(self):  # reference fdtd/grid.py:275-299
        for b in self.boundaries:
            b.update_phi_E()
        curl = curl_H(self.H, self.halo_lo_H)
        self.E += self.courant_number * self.inverse_permittivity * curl
        for o in self.objects:
            o.update_E(curl)
        for b in self.boundaries:
            b.update_E()
        for s in self.sources:
            s.update_E()
        for d in self.detectors:
            d.detect_E()

    def update_H(self):  # reference fdtd/grid.py:301-325
        for b in self.boundaries:
            b.update_phi_H()
        curl = curl_E(self.E, self.halo_hi_E)
        self.H -= self.courant_number * self.inverse_permeability * curl
        for b in self.boundaries:
            b.update_H()
        for s in self.sources:
            s.update_H()
        for d in self.detectors:
            d.detect_H()

    def step(self):  # reference fdtd/grid.py:267-273
        self.update_E()
        self.update_H()
        self.time_steps_passed += 1

    def run(self, steps):
        for _ in range(int(steps)):
            self.step()
