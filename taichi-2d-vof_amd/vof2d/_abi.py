"""ctypes prototypes of the C ABI declared in include/vof2d.h.

`bind(lib, prefix)` attaches argtypes/restype to every entry point of a loaded
shared library.  The product always binds ``libvof2d_hip.so`` with prefix
``vof_`` (see `_lib.py`); the parity tests additionally bind the CPU oracle,
which exports the same signatures with prefix ``ovof_``.
"""
import ctypes as C

VOF_ABI_VERSION = 2
VOF_F64, VOF_F32 = 0, 1
VOF_OK, VOF_EINVAL, VOF_EHIP, VOF_ENOMEM, VOF_ESTATE = 0, -1, -2, -3, -4
VOF_FLAG_NO_GRAPH = 1
VOF_COMM_ID_BYTES, VOF_COMM_LOOPBACK = 128, 1
VOF_XCHG_F, VOF_XCHG_U, VOF_XCHG_V, VOF_XCHG_P = 1, 2, 4, 8
VOF_XCHG_US, VOF_XCHG_VS, VOF_XCHG_RHS = 16, 32, 64
VOF_RESID_ABS, VOF_RESID_REL, VOF_RESID_TINY = 0, 1, 1e-300

ERRNAMES = {VOF_EINVAL: "VOF_EINVAL", VOF_EHIP: "VOF_EHIP", VOF_ENOMEM: "VOF_ENOMEM",
            VOF_ESTATE: "VOF_ESTATE"}


def halo_rows(jacobi_iters):
    """VOF_HALO_ROWS of include/vof2d.h."""
    return int(jacobi_iters) + 8


class Desc(C.Structure):
    """struct vof2d_desc"""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("nx", C.c_int32), ("ny", C.c_int32),
        ("dtype", C.c_int32),
        ("coord_cast_f32", C.c_int32),
        ("row_lo", C.c_int32), ("row_hi", C.c_int32),
        ("own_lo", C.c_int32), ("own_hi", C.c_int32),
        ("jacobi_iters", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_int32),
        ("Lx", C.c_double), ("Ly", C.c_double),
        ("rho_l", C.c_double), ("rho_g", C.c_double),
        ("nu_l", C.c_double), ("nu_g", C.c_double),
        ("sigma", C.c_double),
        ("gx", C.c_double), ("gy", C.c_double),
        ("dt", C.c_double),
    ]


H = C.c_void_p
_i32, _i64, _dbl, _str = C.c_int32, C.c_int64, C.c_double, C.c_char_p

# name -> (restype, argtypes); the complete symbol list of include/vof2d.h
SIGNATURES = {
    "desc_default": (C.c_int, [C.POINTER(Desc), _i32, _i32, _i32]),
    "create": (C.c_int, [C.POINTER(Desc), C.c_void_p, C.POINTER(H)]),
    "destroy": (C.c_int, [H]),
    "set_init_F": (C.c_int, [H, _i32]),
    "set_BC": (C.c_int, [H]),
    "cal_nu_rho": (C.c_int, [H]),
    "get_normal_young": (C.c_int, [H]),
    "advect_upwind": (C.c_int, [H]),
    "solve_p_jacobi": (C.c_int, [H, _i32]),
    "update_uv": (C.c_int, [H]),
    "fct_x_sweep": (C.c_int, [H]),
    "fct_y_sweep": (C.c_int, [H]),
    "solve_VOF_rudman": (C.c_int, [H, _i64]),
    "post_process_f": (C.c_int, [H]),
    "step": (C.c_int, [H, _i64]),
    "step_phase": (C.c_int, [H, _i32]),
    "get_istep": (C.c_int, [H, C.POINTER(_i64)]),
    "set_istep": (C.c_int, [H, _i64]),
    "solve_p_residual": (C.c_int, [H, _dbl, _i32, _i32, C.POINTER(_i32), C.POINTER(_dbl)]),
    "jacobi_sweeps_residual": (C.c_int, [H, _i32, _i32, C.POINTER(_dbl)]),
    "jacobi_sweeps_norms": (C.c_int, [H, _i32, _i32, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "residual_value": (_dbl, [_dbl, _dbl, _i32]),
    "solve_p": (C.c_int, [H, _dbl, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_dbl)]),
    "get_field": (C.c_int, [H, _str, C.c_void_p, C.c_size_t]),
    "set_field": (C.c_int, [H, _str, C.c_void_p, C.c_size_t]),
    "get_rows": (C.c_int, [H, _str, _i32, _i32, C.c_void_p, C.c_size_t]),
    "set_rows": (C.c_int, [H, _str, _i32, _i32, C.c_void_p, C.c_size_t]),
    "field_view": (C.c_int, [H, _str, C.POINTER(C.c_void_p), C.POINTER(_i64), C.POINTER(_i64),
                             C.POINTER(_i64)]),
    "copy_rows": (C.c_int, [H, H, _str, _i32, _i32]),
    "get_vis_field": (C.c_int, [H, _str, C.c_void_p, C.c_size_t]),
    "interp_velocity": (C.c_int, [H, C.c_void_p, C.c_size_t]),
    "set_param": (C.c_int, [H, _str, _dbl]),
    "get_param": (C.c_int, [H, _str, C.POINTER(_dbl)]),
    "get_counter": (C.c_int, [H, _str, C.POINTER(_i64)]),
    "sync": (C.c_int, [H]),
    "timer_start": (C.c_int, [H]),
    "timer_stop": (C.c_int, [H, C.POINTER(C.c_float)]),
    "time_jacobi": (C.c_int, [H, _i32, C.POINTER(C.c_float)]),
    "profile_steps": (C.c_int, [H, _i64]),
    "get_profile": (C.c_int, [H, _str, C.POINTER(_dbl), C.POINTER(_i64)]),
    "reset_profile": (C.c_int, [H]),
    "comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "comm_init": (C.c_int, [H, C.c_void_p, _i32, _i32, _i32]),
    "comm_exchange": (C.c_int, [H, C.c_uint32]),
    "step_exchange": (C.c_int, [H, _i64, _i32]),
    "step_tm_piece": (C.c_int, [H, _i32]),
    "comm_allreduce_max": (C.c_int, [H, C.POINTER(_dbl)]),
    "comm_info": (C.c_int, [H, C.POINTER(_i32), C.POINTER(_i32)]),
    "comm_destroy": (C.c_int, [H]),
    "selftest_division": (C.c_int, [_i32, _i64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "last_error": (C.c_char_p, [H]),
    "backend": (C.c_char_p, []),
}


# entry points that only the GPU library implements (timing / profiling on a HIP stream)
GPU_ONLY = ("timer_start", "timer_stop", "time_jacobi", "profile_steps", "get_profile", "reset_profile",
            "selftest_division", "comm_get_unique_id", "comm_init", "comm_exchange", "step_exchange", "step_tm_piece", "comm_destroy",
            "comm_allreduce_max", "comm_info")


class Api:
    """Bound entry points of one library: api.step(h, n) -> int."""

    def __init__(self, lib, prefix, optional=()):
        self.lib, self.prefix = lib, prefix
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, prefix + name)
            except AttributeError:
                if name not in optional:
                    missing.append(prefix + name)
                continue
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        if missing:
            raise ImportError("library %r lacks symbols: %s" % (getattr(lib, "_name", lib), ", ".join(missing)))


def bind(lib, prefix="vof_", optional=()):
    return Api(lib, prefix, optional)
