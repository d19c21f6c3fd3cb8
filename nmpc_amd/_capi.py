"""ctypes binding of the C-ABI declared in include/nmpc_hip_ddp.h (libnmpc_hip_ddp.so).

This is the same stub a maintainer of a Python front end would write (INTEGRATION.md shows the C++ one);
it contains no numerics — every number comes from the HIP library.  If the library is missing it is built
with hipcc; if that is impossible the import fails loudly (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

MAX_ALPHA = 32
NTRACE = 12

# nmpc_hip_status
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_RUNTIME = -2
ERR_UNKNOWN_MODEL = -3
ERR_HIP = -4
ERR_NO_DEVICE = -5
ERR_NOT_SOLVED = -6

# nmpc_hip_field
FIELD_X, FIELD_U, FIELD_COST, FIELD_KFF, FIELD_KFB, FIELD_TRACE, FIELD_STATUS, FIELD_ITERS, FIELD_TRACE_LAST, \
    FIELD_DV, FIELD_QP_RETVAL, FIELD_QP_FREE_MASK, FIELD_INPUT_DIM = range(13)

TRACE_COLUMNS = (
    "iter", "cost", "lambda", "dlambda", "alpha", "k_rel_norm", "cost_update_actual", "cost_update_expected",
    "cost_update_ratio", "alpha_idx", "n_backward", "n_forward",
)


class Config(C.Structure):
    """nmpc_hip_ddp_config = DDPSolver::Configuration (DDPSolver.h:47-110) + BoxQP::Configuration."""

    _fields_ = [
        ("with_input_constraint", C.c_int),
        ("max_iter", C.c_int),
        ("horizon_steps", C.c_int),
        ("reg_type", C.c_int),
        ("initial_lambda", C.c_double),
        ("initial_dlambda", C.c_double),
        ("lambda_factor", C.c_double),
        ("lambda_min", C.c_double),
        ("lambda_max", C.c_double),
        ("k_rel_norm_thre", C.c_double),
        ("lambda_thre", C.c_double),
        ("cost_update_ratio_thre", C.c_double),
        ("cost_update_thre", C.c_double),
        ("n_alpha", C.c_int),
        ("alpha_list", C.c_double * MAX_ALPHA),
        ("use_state_eq_second_derivative", C.c_int),
        ("qp_max_iter", C.c_int),
        ("qp_grad_thre", C.c_double),
        ("qp_rel_improve_thre", C.c_double),
        ("qp_step_factor", C.c_double),
        ("qp_min_step", C.c_double),
        ("qp_armijo_param", C.c_double),
        ("trace_level", C.c_int),
        ("line_search_fan_out", C.c_int),
        ("ragged_schedule", C.c_int),
    ]


class MpcOptions(C.Structure):
    """nmpc_hip_ddp_mpc_options (include/nmpc_hip_ddp.h)."""

    _fields_ = [
        ("n_ticks", C.c_int),
        ("shift_warm_start", C.c_int),
        ("max_iter_after_first", C.c_int),
        ("sim_substeps", C.c_int),
        ("sim_dt", C.c_double),
        ("clamp_u0", C.c_int),
    ]


_lib = None

# every symbol include/nmpc_hip_ddp.h declares
EXPORTS = (
    "nmpc_hip_ddp_default_config", "nmpc_hip_ddp_model_count", "nmpc_hip_ddp_model_name",
    "nmpc_hip_ddp_model_info", "nmpc_hip_ddp_model_scalar_bytes", "nmpc_hip_ddp_model_default_params", "nmpc_hip_ddp_create",
    "nmpc_hip_ddp_destroy", "nmpc_hip_ddp_set_config", "nmpc_hip_ddp_get_config",
    "nmpc_hip_ddp_set_model_params", "nmpc_hip_ddp_set_model_params_batch", "nmpc_hip_ddp_set_input_limits_batch",
    "nmpc_hip_ddp_input_dims", "nmpc_hip_ddp_set_input_limits", "nmpc_hip_ddp_set_input_limits_horizon",
    "nmpc_hip_ddp_set_input_limits_schedule", "nmpc_hip_ddp_solve", "nmpc_hip_ddp_solve_async", "nmpc_hip_ddp_request_hw_queues", "nmpc_hip_ddp_solve_stream", "nmpc_hip_ddp_stream_get", "nmpc_hip_ddp_last_stream_stats",
    "nmpc_hip_ddp_solve_device", "nmpc_hip_ddp_synchronize", "nmpc_hip_ddp_get", "nmpc_hip_ddp_get_device",
    "nmpc_hip_ddp_field_bytes", "nmpc_hip_ddp_last_solve_ms", "nmpc_hip_ddp_last_solve_phases", "nmpc_hip_ddp_timing_stats",
    "nmpc_hip_ddp_kernel_name", "nmpc_hip_ddp_kernel_name_for_batch", "nmpc_hip_ddp_set_kernel", "nmpc_hip_ddp_set_dispatch_batch",
    "nmpc_hip_ddp_last_solve_launches", "nmpc_hip_ddp_mpc_default_options", "nmpc_hip_ddp_mpc_run",
    "nmpc_hip_ddp_last_error",
)


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building first if needed) libnmpc_hip_ddp.so and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NMPC_HIP_DDP_LIB")  # developer override: A/B-time two builds on the same GPU box
    if not path:
        path = _build.LIB_PATH
        if _build.needs_build():
            path = _build.build()
    if not os.path.exists(path):
        raise RuntimeError("libnmpc_hip_ddp.so is missing and could not be built; there is no CPU fallback")
    L = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.nmpc_hip_ddp_default_config.argtypes = [C.POINTER(Config)]
    L.nmpc_hip_ddp_model_count.argtypes = []
    L.nmpc_hip_ddp_model_name.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.nmpc_hip_ddp_model_info.argtypes = [C.c_char_p, ip, ip, ip, C.POINTER(C.c_size_t)]
    L.nmpc_hip_ddp_model_default_params.argtypes = [C.c_char_p, vp, C.c_size_t]
    L.nmpc_hip_ddp_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.nmpc_hip_ddp_destroy.argtypes = [vp]
    L.nmpc_hip_ddp_set_config.argtypes = [vp, C.POINTER(Config)]
    L.nmpc_hip_ddp_get_config.argtypes = [vp, C.POINTER(Config)]
    L.nmpc_hip_ddp_set_model_params.argtypes = [vp, vp, C.c_size_t]
    L.nmpc_hip_ddp_set_model_params_batch.argtypes = [vp, vp, C.c_size_t]
    L.nmpc_hip_ddp_set_input_limits_batch.argtypes = [vp, dp, dp]
    L.nmpc_hip_ddp_input_dims.argtypes = [vp, C.c_double, ip]
    L.nmpc_hip_ddp_set_input_limits.argtypes = [vp, dp, dp]
    L.nmpc_hip_ddp_set_input_limits_horizon.argtypes = [vp, dp, dp, C.c_int]
    L.nmpc_hip_ddp_set_input_limits_schedule.argtypes = [vp, dp, dp, C.c_int, C.c_int]
    L.nmpc_hip_ddp_solve.argtypes = [vp, dp, dp, dp]
    L.nmpc_hip_ddp_solve_async.argtypes = [vp, dp, dp, dp]
    L.nmpc_hip_ddp_solve_device.argtypes = [vp, vp, vp, vp, vp]
    L.nmpc_hip_ddp_synchronize.argtypes = [vp]
    L.nmpc_hip_ddp_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.nmpc_hip_ddp_get_device.argtypes = [vp, C.c_int, vp, C.c_size_t, vp]
    L.nmpc_hip_ddp_field_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.nmpc_hip_ddp_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.nmpc_hip_ddp_last_solve_phases.argtypes = [vp, dp, dp, dp]
    L.nmpc_hip_ddp_kernel_name.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.nmpc_hip_ddp_last_solve_launches.argtypes = [vp, C.POINTER(C.c_int)]
    L.nmpc_hip_ddp_kernel_name_for_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p)]
    L.nmpc_hip_ddp_set_kernel.argtypes = [vp, C.c_char_p]
    L.nmpc_hip_ddp_set_dispatch_batch.argtypes = [vp, C.c_int]
    L.nmpc_hip_ddp_mpc_default_options.argtypes = [C.POINTER(MpcOptions)]
    L.nmpc_hip_ddp_mpc_run.argtypes = [vp, dp, dp, dp, C.POINTER(MpcOptions), dp, dp, dp, ip, ip, ip, dp, dp]
    L.nmpc_hip_ddp_timing_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong), dp, dp]
    L.nmpc_hip_ddp_request_hw_queues.argtypes = [C.c_int, ip]
    L.nmpc_hip_ddp_solve_stream.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    L.nmpc_hip_ddp_stream_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.nmpc_hip_ddp_last_stream_stats.argtypes = [vp, ip, C.POINTER(C.c_float)]
    L.nmpc_hip_ddp_last_error.argtypes = []
    L.nmpc_hip_ddp_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name != "nmpc_hip_ddp_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def last_error() -> str:
    return load().nmpc_hip_ddp_last_error().decode(errors="replace")


def check(rc: int) -> None:
    """Turn a status code back into the exception type the reference throws for the same misuse
    (std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError)."""
    if rc == OK:
        return
    msg = last_error()
    if rc in (ERR_INVALID_ARGUMENT, ERR_UNKNOWN_MODEL):
        raise ValueError(msg)
    raise RuntimeError(f"[nmpc_hip_ddp {rc}] {msg}")
