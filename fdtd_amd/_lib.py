"""ctypes binding of ``libfdtd_hip.so`` (C ABI declared in ``include/fdtd_hip.h``).

The library is the only compute path of this package: if it cannot be loaded the
package raises instead of falling back to anything else.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FDTDHIP_LIBRARY: another build of the same library, for kernel experiments)
LIB_PATH = os.environ.get("FDTDHIP_LIBRARY") or os.path.join(_HERE, "csrc", "libfdtd_hip.so")

F32, F64 = 0, 1
OBJ_ISOTROPIC, OBJ_ABSORBING, OBJ_ANISOTROPIC, OBJ_ANISOTROPIC_MODULATED = 0, 1, 2, 3
SRC_ADD_POINTS, SRC_SET_BOX = 0, 1
DET_CELLS, DET_CURRENT = 0, 1
PERIODIC_X_EXTERNAL = 3
ABI_VERSION = 1

_p, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pd, _pi32 = C.POINTER(C.c_double), C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/fdtd_hip.h declares
PROTOTYPES = {
    "fdtdhip_abi_version": (_i, []),
    "fdtdhip_last_error": (C.c_char_p, []),
    "fdtdhip_device_count": (_i, []),
    "fdtdhip_plan_create": (_i, [C.POINTER(_p), _i, _i, _i, _i, _d, _i]),
    "fdtdhip_plan_destroy": (_i, [_p]),
    "fdtdhip_plan_reset": (_i, [_p]),
    "fdtdhip_plan_set_stream": (_i, [_p, _p]),
    "fdtdhip_plan_set_fields": (_i, [_p, _p, _p, _p, _p]),
    "fdtdhip_plan_scan_materials": (_i, [_p, _i]),
    "fdtdhip_plan_set_halo": (_i, [_p, _p, _p]),
    "fdtdhip_plan_bind": (_i, [_p] * 10),
    "fdtdhip_plan_set_step_buffers": (_i, [_p, _p, _p]),
    "fdtdhip_plan_set_pml_step_buffer": (_i, [_p, _i, _p]),
    "fdtdhip_plan_add_pml": (_i, [_p, _i, _i, _i, _pd, _pd, _pd, _pd, _p, _p]),
    "fdtdhip_plan_set_pml_state": (_i, [_p, _i, _p, _p]),
    "fdtdhip_plan_add_periodic": (_i, [_p, _i]),
    "fdtdhip_plan_add_object": (_i, [_p, _i, _pi32, _p, _p]),
    "fdtdhip_plan_add_source": (_i, [_p, _i, _i, _i, _i64, _pi32, _pd, _d, _d, _i, _d, _d, _d]),
    "fdtdhip_plan_add_soft_source": (_i, [_p, _pi32, _pd, _i64, _d, _d]),
    "fdtdhip_plan_link_soft_source": (_i, [_p, _i, _i]),
    "fdtdhip_plan_set_soft_source_io": (_i, [_p, _i, _p, _p, _p, _i64]),
    "fdtdhip_plan_set_soft_source_adjoint": (_i, [_p, _i, _i64]),
    "fdtdhip_plan_set_soft_source_adjoint_below": (_i, [_p, _i, _p]),
    "fdtdhip_adjoint_post_stage": (_i, [_p, _i, _i64, _i, _p, _p]),
    "fdtdhip_plan_add_modulated_source": (_i, [_p, _pi32, _i, _i]),
    "fdtdhip_plan_set_modulated_source": (_i, [_p, _i, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "fdtdhip_plan_add_detector": (_i, [_p, _i, _i64, _pi32, _d]),
    "fdtdhip_plan_set_detector_output": (_i, [_p, _i, _p, _p, _i64]),
    "fdtdhip_plan_set_detector_below": (_i, [_p, _i, _p]),
    "fdtdhip_read_deferred_currents": (_i, [_p]),
    "fdtdhip_update_E": (_i, [_p, _i64]),
    "fdtdhip_update_H": (_i, [_p, _i64]),
    "fdtdhip_update_E_bulk": (_i, [_p, _i, _i]),
    "fdtdhip_update_E_post": (_i, [_p, _i64]),
    "fdtdhip_update_H_bulk": (_i, [_p, _i, _i]),
    "fdtdhip_update_H_post": (_i, [_p, _i64]),
    "fdtdhip_update_E_post_stage": (_i, [_p, _i64, _i]),
    "fdtdhip_update_H_post_stage": (_i, [_p, _i64, _i]),
    "fdtdhip_run": (_i, [_p, _i64, _i64]),
    "fdtdhip_plan_whole_step_kind": (_i, [_p, _i, _pi32, _pi32]),
    "fdtdhip_step_stage": (_i, [_p, _i64, _i]),
    "fdtdhip_step_into": (_i, [_p, _i64]),
    "fdtdhip_plan_two_step_box": (_i, [_p, _i, C.POINTER(C.c_int32)]),
    "fdtdhip_two_step_launches": (C.c_int64, []),
    "fdtdhip_pair_stage": (_i, [_p, _i64, _i]),
    "fdtdhip_plan_set_object_trace": (_i, [_p, _i, _p, _i64]),
    "fdtdhip_plan_set_object_grad": (_i, [_p, _i, _p]),
    "fdtdhip_plan_set_object_modifier": (_i, [_p, _i, _p, _p, _p]),
    "fdtdhip_plan_set_pml_adjoint": (_i, [_p, _i, _p, _p]),
    "fdtdhip_plan_set_adjoint_buffers": (_i, [_p, _p, _p]),
    "fdtdhip_plan_set_pml_adjoint_buffers": (_i, [_p, _i, _p, _p]),
    "fdtdhip_plan_adjoint_whole_steps": (_i, [_p]),
    "fdtdhip_adjoint_pack_plane": (_i, [_p, _i, _i64, _i, _p, _p, _p]),
    "fdtdhip_adjoint_step_stage": (_i, [_p, _i64, _i, _i, _p, _p, _p, _p, _p, _p]),
    "fdtdhip_adjoint_run": (_i, [_p, _i64, _p, _p, _p]),
    "fdtdhip_adjoint_run_noalloc": (_i, [_p, _i64, _p, _p]),
    "fdtdhip_adjoint_cells": (_i, [_p, _i, _i64, _p, _p, _p]),
    "fdtdhip_adjoint_gather": (_i, [_p, _i, _p, _p, _p, _p]),
    "fdtdhip_plane_energy": (_i, [_p, _i, _i, _p, _p]),
    "fdtdhip_plan_enable_timing": (_i, [_p, _i]),
    "fdtdhip_plan_get_timing": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "fdtdhip_plan_get_timing_parts": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "fdtdhip_curl_E": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fdtdhip_curl_H": (_i, [_p, _p, _i, _i, _i, _i, _p]),
}

_lib = None


class NativeError(RuntimeError):
    """A call into libfdtd_hip failed."""


def load():
    """Load the shared library once; raise RuntimeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C fdtd_amd/csrc` "
            "(or __graft_entry__.build()); this package has no other compute path"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    got = lib.fdtdhip_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libfdtd_hip ABI {got} does not match the binding ({ABI_VERSION})")
    _lib = lib
    return lib


def check(rc):
    """Turn a negative status into an exception carrying the library's message."""
    if rc < 0:
        msg = load().fdtdhip_last_error()
        raise NativeError(msg.decode() if msg else f"libfdtd_hip error {rc}")
    return rc


def device_count():
    return load().fdtdhip_device_count()


def doubles(values):
    arr = (C.c_double * len(values))(*[float(v) for v in values])
    return arr


def int32s(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr
