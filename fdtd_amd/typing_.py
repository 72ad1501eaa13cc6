"""Type aliases used in signatures (counterpart of reference fdtd/typing_.py)."""

from typing import List, Sequence, Tuple, Union

import numpy as _np

try:
    import torch as _torch

    Tensorlike = Union[_np.ndarray, "_torch.Tensor", float, int]
except ImportError:  # pragma: no cover
    Tensorlike = Union[_np.ndarray, float, int]

Number = Union[float, int]
ListOrSlice = Union[List[int], slice]
IntOrSlice = Union[int, slice]
Index = Union[Number, slice, Sequence[Number]]

__all__ = ["Tuple", "List", "Number", "Tensorlike", "ListOrSlice", "IntOrSlice", "Index"]
