"""Support for the multi-process x-slab tests.

``OracleSlabEngine`` is a TEST DOUBLE for the native engine of
``fdtd_amd.parallel.SlabGrid``: it executes the half steps of one slab with the
numpy oracle, directly on the memory of the slab's (host-staging) tensors.  With it
the CPU suite exercises everything of the multi-GPU path except the HIP kernels:
plane partitioning, clipping of components to a slab, the collective border-value
lookup, the exchange schedule and direction, ghost-plane use -- against the
single-grid oracle run.
"""

import os
import socket

import numpy as np
import torch

from oracle import yee_numpy as orc


class _SlabDirectionalPoint:
    """The six additions of OracleCCPointSource.update_E (reference fdtd/sources.py:150-167),
    each made by the rank whose slab holds its plane (index -1 wraps around the whole
    simulation)."""

    def __init__(self, g, og, s):
        self.g, self.og = g, og
        d = [float(v) for v in s.dir_vec]
        x, y, z = int(s.x), int(s.y), int(s.z)
        self.terms = [((x - 1, y, z), -d[0]), ((x, y, z), d[0]), ((x, y - 1, z), -d[1]),
                      ((x, y, z), d[1]), ((x, y, z - 1), -d[2]), ((x, y, z), d[2])]
        self.period = s.period

    def update_E(self):
        g, og = self.g, self.og
        src = np.sin(2 * np.pi * og.time_steps_passed / self.period)
        for (x, y, z), w in self.terms:
            xl = g._local_x(x % g.Nx_global)
            if xl is not None:
                og.E[xl, y, z, 2] += src * w

    def update_H(self):
        pass


def _plane3(ghost):
    """A slab's ghost buffer -- (Ny, Nz, 2): the y and z components of the neighbour's
    plane -- as the three-component plane the oracle's curls take (x component unused)."""
    if ghost is None:
        return None
    h = ghost.numpy()
    return np.concatenate([np.zeros(h.shape[:-1] + (1,), h.dtype), h], axis=-1)


class OracleSlabEngine:
    def __init__(self, grid):
        self.grid = grid
        self.og = None
        self.sig = None
        self.calls = []

    # -- build the oracle twin of the slab from the engine's resolved components
    def prepare(self, n=1):
        g = self.grid
        sig = g._signature()
        if self.og is not None and sig == self.sig:
            return
        og = orc.OracleGrid((g.Nx, g.Ny, g.Nz), grid_spacing=g.grid_spacing)
        og.courant_number, og.time_step = g.courant_number, g.time_step
        og.E, og.H = g.E.numpy(), g.H.numpy()  # shared memory
        og.inverse_permittivity = g.inverse_permittivity.numpy()
        og.inverse_permeability = g.inverse_permeability.numpy()
        self.dets = []
        self.ext = None   # place of a cross-rank x-periodic boundary in og.boundaries
        for b in g.boundaries:
            if hasattr(b, "thickness"):
                if b._active:
                    og.add_pml(b.axis, b.side, b.thickness, b.a)
            elif b.axis == 0 and g.Nx != g.Nx_global:
                self.ext = len(og.boundaries)
            else:
                og.add_periodic(b.axis)
        for o in g.objects:
            if o.Nx == 0:
                continue
            kind = {0: "object", 1: "absorbing", 2: "anisotropic"}[o._native_kind]
            box = (o.x.start, o.x.stop, o.y.start, o.y.stop, o.z.start, o.z.stop)
            oo = og.add_object(box, 1.0, kind=kind, conductivity=0.0)
            oo.inverse_permittivity = o.inverse_permittivity.detach().numpy()
            if kind == "absorbing":
                oo.absorption_factor = o.absorption_factor.numpy()
        for s in g.sources:
            name = type(s).__name__
            if name == "PointSource":
                xl = g._local_x(s.x)
                if xl is None:
                    continue
                t = og.add_point_source((xl, s.y, s.z), period=s.period, amplitude=s.amplitude,
                                        phase_shift=s.phase_shift, pulse=s.pulse, cycle=s.cycle,
                                        hanning_dt=s.hanning_dt)
                t.wave.frequency = s.frequency
            elif name == "LineSource":
                keep = [n for n, x in enumerate(s.x) if g._local_x(x) is not None]
                if not keep:
                    continue
                t = og.add_line_source([g._local_x(s.x[n]) for n in keep], [s.y[n] for n in keep],
                                       [s.z[n] for n in keep], period=s.period,
                                       phase_shift=s.phase_shift, pulse=s.pulse, cycle=s.cycle,
                                       hanning_dt=s.hanning_dt)
                t.wave.frequency = s.frequency
                t.profile = np.array([s.profile[n] for n in keep])
            elif name == "CorticalColumnPointSource":
                og.sources.append(_SlabDirectionalPoint(g, og, s))
            elif name == "PlaneSource":
                lo, hi = g._clip_x(s.x.start, s.x.stop)
                if hi <= lo:
                    continue
                og.add_plane_source((lo, hi, s.y.start, s.y.stop, s.z.start, s.z.stop),
                                    period=s.period, amplitude=s.amplitude,
                                    phase_shift=s.phase_shift, polarization=s.polarization)
            else:
                raise NotImplementedError(f"{name} in the oracle slab engine")
        for d in g.detectors:
            name = type(d).__name__
            if name == "LineDetector":
                pts = d._local_points()
                od = og.add_line_detector([p[0] for p in pts], [p[1] for p in pts], [p[2] for p in pts])
            elif name == "BlockDetector":
                xs = [g._local_x(x) for x in d.x if g._local_x(x) is not None]
                od = og.add_block_detector(xs, d.y, d.z)
            else:
                xs = [g._local_x(x) for x in d.x if g._local_x(x) is not None]
                od = og.add_current_detector(xs, d.y, d.z)
            self.dets.append((d, od))
        self.og, self.sig = og, sig

    def reserve_detectors(self, nE, nH):
        pass

    # -- current detectors whose loop reads the plane under the slab
    def defer_currents(self, plane):
        self.below = plane

    def _detect_H(self):
        for d in self.og.detectors:
            if not (isinstance(d, orc.OracleCurrentDetector) and getattr(self, "below", None) is not None):
                d.detect_H()

    def read_deferred_currents(self):
        # index -1 of the slab's H with the plane from below appended is that plane
        og = self.og
        stacked = np.concatenate([og.H, self.below.numpy()[None]], axis=0)
        for d in og.detectors:
            if isinstance(d, orc.OracleCurrentDetector):
                assert all(0 <= x < og.Nx for x in d.x)
                d.grid = type("WithPlaneBelow", (), dict(H=stacked, grid_spacing=og.grid_spacing))
                d.detect_H()
                d.grid = og

    def commit_detectors(self, nE, nH):
        for d, od in self.dets:
            for attr in ("E", "H", "I"):
                if hasattr(od, attr):
                    src, dst = getattr(od, attr), getattr(d, attr)
                    while len(dst) < len(src):
                        dst.append(torch.from_numpy(np.array(src[len(dst)])))

    # -- half-step pieces: ranges are recorded, the oracle acts when the step completes
    def bulk_E(self, x0, x1):
        self.calls.append(("E", x0, x1))

    def bulk_H(self, x0, x1):
        self.calls.append(("H", x0, x1))

    def _check_cover(self, phase):
        g = self.grid
        mine = [(a, b) for p, a, b in self.calls if p == phase and b > a]
        covered = sorted(mine)
        pos = 0
        for a, b in covered:
            assert a == pos, f"planes {pos}:{a} of the {phase} half step were skipped"
            pos = b
        assert pos == g.Nx, f"{phase} half step stopped at plane {pos} of {g.Nx}"
        # the plane that needs a ghost must be the last one enqueued
        if phase == "E" and g._H_below is not None and g.Nx > 1:
            assert mine[-1] == (0, 1)
        if phase == "H" and g._E_above is not None and g.Nx > 1:
            assert mine[-1] == (g.Nx - 1, g.Nx)
        self.calls = [c for c in self.calls if c[0] != phase]

    def post_E(self, q):
        self._check_cover("E")
        g, og = self.grid, self.og
        og.time_steps_passed = q
        og.halo_lo_H = _plane3(g._H_below)
        og.update_E()

    def post_H(self, q):
        self._check_cover("H")
        g, og = self.grid, self.og
        og.time_steps_passed = q
        og.halo_hi_E = _plane3(g._E_above)
        if getattr(self, "below", None) is None:
            og.update_H()
        else:   # the statements of OracleGrid.update_H with the current detectors left out
            dets, og.detectors = og.detectors, []
            og.update_H()
            og.detectors = dets
            self._detect_H()

    # -- the same half steps cut at the cross-rank periodic boundary (the statements of
    #    OracleGrid.update_E / update_H, reference fdtd/grid.py:275-325)
    def post_E_stage(self, q, stage):
        g, og, k = self.grid, self.og, self.ext
        if stage == 0:
            self._check_cover("E")
            og.time_steps_passed = q
            for b in og.boundaries:
                b.update_phi_E()
            curl = orc.curl_H(og.H, _plane3(g._H_below))
            og.E += og.courant_number * og.inverse_permittivity * curl
            for o in og.objects:
                o.update_E(curl)
            for b in og.boundaries[:k]:
                b.update_E()
        else:
            for b in og.boundaries[k:]:
                b.update_E()
            for s in og.sources:
                s.update_E()
            for d in og.detectors:
                d.detect_E()

    def post_H_stage(self, q, stage):
        g, og, k = self.grid, self.og, self.ext
        if stage == 0:
            self._check_cover("H")
            og.time_steps_passed = q
            for b in og.boundaries:
                b.update_phi_H()
            curl = orc.curl_E(og.E, _plane3(g._E_above))
            og.H -= og.courant_number * og.inverse_permeability * curl
            for b in og.boundaries[:k]:
                b.update_H()
        else:
            for b in og.boundaries[k:]:
                b.update_H()
            for s in og.sources:
                s.update_H()
            self._detect_H()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_gloo(rank, world, port):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
