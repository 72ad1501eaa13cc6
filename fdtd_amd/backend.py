"""The ``hip`` array backend: the seam through which this engine plugs into the
fdtd API (counterpart of reference fdtd/backend.py:350-426).

As in the reference, one module-level ``backend`` object is shared by every
module of the package and ``set_backend(name)`` re-configures it in place.
Accepted names:

    ``"hip"``            float64 tensors on the current ROCm device
    ``"hip.float64"``
    ``"hip.float32"``    really float32 (the reference's ``*.float32`` names still
                         allocate float64, see reference fdtd/backend.py:87, 265-268)

PyTorch-ROCm tensors are only the array container: every time-stepping
operation runs in ``libfdtd_hip.so`` (see ``fdtd_amd/_lib.py``).  There is no CPU
compute path in this package; ``set_backend`` raises ``RuntimeError`` when no
ROCm device is visible, mirroring reference fdtd/backend.py:378-385.

``device="host-staging"`` allocates the arrays in host memory so that grids can be
*described* (registration, index arithmetic, PML profiles) on a machine without
a GPU; any attempt to time-step such a grid raises.
"""

import numpy

import torch

from . import _lib

# used by test parametrisation, default first (counterpart of reference fdtd/backend.py:36-42)
backend_names = [
    dict(backends="hip"),
    dict(backends="hip.float64"),
    dict(backends="hip.float32"),
]

_FLOATS = {"float32": torch.float32, "float64": torch.float64}


class HipBackend:
    """Array primitives on the ROCm device (the subset of reference
    fdtd/backend.py:196-342 that the fdtd modules use)."""

    pi = numpy.pi
    int = torch.int64
    float = torch.float64
    device = None  # torch.device once configured
    staging = False

    def __repr__(self):
        return self.__class__.__name__

    # -- configuration -----------------------------------------------------
    def _require(self):
        if self.device is None:
            set_backend("hip")
        return self.device

    @property
    def native_dtype(self):
        return _lib.F32 if self.float == torch.float32 else _lib.F64

    # -- constructors ------------------------------------------------------
    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.float, device=self._require())

    def ones(self, shape, dtype=None):
        return torch.ones(shape, dtype=dtype or self.float, device=self._require())

    def zeros_like(self, arr):
        return torch.zeros_like(arr)

    def array(self, arr, dtype=None):
        dtype = dtype or self.float
        dev = self._require()
        if torch.is_tensor(arr):
            return arr.clone().to(device=dev, dtype=dtype)
        if isinstance(arr, (list, tuple)) and arr and all(torch.is_tensor(a) for a in arr):
            return torch.stack([a.to(device=dev, dtype=dtype) for a in arr])
        return torch.as_tensor(numpy.asarray(arr), dtype=dtype).to(dev)

    def asarray(self, arr, dtype=None):
        if torch.is_tensor(arr) and arr.device == self._require() and dtype in (None, arr.dtype):
            return arr
        return self.array(arr, dtype)

    def arange(self, *args, **kw):
        kw.setdefault("device", self._require())
        return torch.arange(*args, **kw)

    def linspace(self, start, stop, num=50, endpoint=True):
        vals = numpy.linspace(start, stop, num, endpoint=endpoint)
        return torch.as_tensor(vals, dtype=self.float).to(self._require())

    # -- math ----------------------------------------------------------------
    exp = staticmethod(torch.exp)
    sin = staticmethod(torch.sin)
    cos = staticmethod(torch.cos)
    sum = staticmethod(torch.sum)
    max = staticmethod(torch.max)
    stack = staticmethod(torch.stack)
    squeeze = staticmethod(torch.squeeze)
    reshape = staticmethod(torch.reshape)
    bmm = staticmethod(torch.bmm)
    broadcast_to = staticmethod(torch.broadcast_to)
    broadcast_arrays = staticmethod(torch.broadcast_tensors)
    divide = staticmethod(torch.div)
    pad = staticmethod(torch.nn.functional.pad)
    fft = staticmethod(torch.fft.fft)
    fftfreq = staticmethod(numpy.fft.fftfreq)

    @staticmethod
    def transpose(arr, axes=None):
        if axes is None:
            axes = tuple(range(arr.dim() - 1, -1, -1))
        return arr.permute(*axes)

    @staticmethod
    def is_array(arr):
        return isinstance(arr, numpy.ndarray) or torch.is_tensor(arr)

    @staticmethod
    def numpy(arr):
        """Host copy as a numpy array."""
        if torch.is_tensor(arr):
            return arr.detach().cpu().numpy()
        if isinstance(arr, (list, tuple)) and arr and torch.is_tensor(arr[0]):
            return torch.stack(list(arr)).detach().cpu().numpy()
        return numpy.asarray(arr)


backend = HipBackend()


def set_backend(name: str = "hip", device=None):
    """Select precision (and optionally the device ordinal) of the hip backend.

    Raises ``RuntimeError`` when the native library or a ROCm device is missing
    and ``ValueError`` for names that are not hip backends.
    """
    parts = name.split(".")
    if parts[0] != "hip" or len(parts) > 2:
        raise ValueError(
            f"Unknown backend '{name}'. This package provides 'hip', 'hip.float32' and "
            "'hip.float64'; CPU backends live in the reference implementation."
        )
    dtype = parts[1] if len(parts) == 2 else "float64"
    if dtype not in _FLOATS:
        raise ValueError(f"Unknown precision '{dtype}'. Available: 'float32', 'float64'")
    _lib.load()  # raises RuntimeError if libfdtd_hip.so is absent
    if device == "host-staging":
        backend.device, backend.staging = torch.device("cpu"), True
    else:
        if not torch.cuda.is_available() or _lib.device_count() == 0:
            raise RuntimeError(
                "hip backend is not available.\n"
                "No ROCm device is visible to this process "
                "(torch.cuda.is_available() is False or hipGetDeviceCount() is 0)."
            )
        if device is None:
            device = torch.cuda.current_device()
        backend.device, backend.staging = torch.device("cuda", int(device)), False
    backend.float = _FLOATS[dtype]
