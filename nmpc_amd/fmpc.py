"""Python mirror of the reference's nmpc_fmpc::FmpcSolver interface for a BATCH of problem instances.

Same member names, argument meaning and error behaviour as /root/reference/nmpc_fmpc/include/nmpc_fmpc/FmpcSolver.h:17-427
(`config()`, `solve()`, `variable()`, `coeffList()`, `traceDataList()`, `computationDuration()`, `dumpTraceDataList()`,
`Variable::reset()`), with a leading batch axis.  Everything numeric happens in libnmpc_hip_ddp.so through the C-ABI
(include/nmpc_hip_fmpc.h); this file marshals arrays and re-raises status codes as the exception types the reference throws.
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _capi

NTRACE = 6
TRACE_COLUMNS = ("iter", "kkt_error", "barrier_eps", "alpha_s_max", "alpha_nu_max", "alpha_s")
(FIELD_X, FIELD_U, FIELD_LAMBDA, FIELD_S, FIELD_NU, FIELD_STATUS, FIELD_ITERS, FIELD_TRACE, FIELD_GAIN_K, FIELD_GAIN_k,
 FIELD_GAIN_S, FIELD_GAIN_P, FIELD_BARRIER_EPS, FIELD_DELTA_X, FIELD_DELTA_U, FIELD_DELTA_LAMBDA, FIELD_DELTA_S, FIELD_DELTA_NU,
 FIELD_MERIT, FIELD_PARTIALS) = range(20)
STATUS_INVALID_VARIABLE = -2


class Status:
    """FmpcSolver::Status (FmpcSolver.h:92-114)."""
    Uninitialized = 0
    Succeeded = 1
    ErrorInForward = 2
    ErrorInBackward = 3
    ErrorInUpdate = 4
    MaxIterationReached = 5
    IterationContinued = 6


class CConfig(C.Structure):
    """nmpc_hip_fmpc_config (include/nmpc_hip_fmpc.h)."""

    _fields_ = [
        ("horizon_steps", C.c_int),
        ("max_iter", C.c_int),
        ("kkt_error_thre", C.c_double),
        ("check_nan", C.c_int),
        ("init_complementary_variable", C.c_int),
        ("update_barrier_eps", C.c_int),
        ("break_if_llt_fails", C.c_int),
        ("enable_line_search", C.c_int),
        ("merit_const_scale_from_lagrange_multipliers", C.c_int),
        ("use_graph", C.c_int),
        ("time_kernels", C.c_int),
    ]


# every symbol include/nmpc_hip_fmpc.h declares
EXPORTS = (
    "nmpc_hip_fmpc_default_config", "nmpc_hip_fmpc_model_count", "nmpc_hip_fmpc_model_name", "nmpc_hip_fmpc_model_info",
    "nmpc_hip_fmpc_model_default_params", "nmpc_hip_fmpc_create", "nmpc_hip_fmpc_destroy", "nmpc_hip_fmpc_set_config",
    "nmpc_hip_fmpc_get_config", "nmpc_hip_fmpc_set_problem", "nmpc_hip_fmpc_set_variable", "nmpc_hip_fmpc_reset_variable",
    "nmpc_hip_fmpc_solve", "nmpc_hip_fmpc_solve_device", "nmpc_hip_fmpc_synchronize", "nmpc_hip_fmpc_get",
    "nmpc_hip_fmpc_field_bytes", "nmpc_hip_fmpc_last_solve_ms", "nmpc_hip_fmpc_last_solve_kernel_ms", "nmpc_hip_fmpc_mpc_run", "nmpc_hip_fmpc_kernel_names",
    "nmpc_hip_fmpc_last_error",
)

_declared = False


def load():
    """The library of nmpc_amd._capi with the FMPC prototypes declared."""
    global _declared
    L = _capi.load()
    if _declared:
        return L
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.nmpc_hip_fmpc_default_config.argtypes = [C.POINTER(CConfig)]
    L.nmpc_hip_fmpc_model_count.argtypes = []
    L.nmpc_hip_fmpc_model_name.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.nmpc_hip_fmpc_model_info.argtypes = [C.c_char_p, ip, ip, ip, C.POINTER(C.c_size_t)]
    L.nmpc_hip_fmpc_model_default_params.argtypes = [C.c_char_p, vp, C.c_size_t]
    L.nmpc_hip_fmpc_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.nmpc_hip_fmpc_destroy.argtypes = [vp]
    L.nmpc_hip_fmpc_set_config.argtypes = [vp, C.POINTER(CConfig)]
    L.nmpc_hip_fmpc_get_config.argtypes = [vp, C.POINTER(CConfig)]
    L.nmpc_hip_fmpc_set_problem.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.nmpc_hip_fmpc_set_variable.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.nmpc_hip_fmpc_reset_variable.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    L.nmpc_hip_fmpc_solve.argtypes = [vp, dp, dp]
    L.nmpc_hip_fmpc_solve_device.argtypes = [vp, vp, vp, vp]
    L.nmpc_hip_fmpc_synchronize.argtypes = [vp]
    L.nmpc_hip_fmpc_get.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int]
    L.nmpc_hip_fmpc_field_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.nmpc_hip_fmpc_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.nmpc_hip_fmpc_last_solve_kernel_ms.argtypes = [vp, dp, ip]
    L.nmpc_hip_fmpc_mpc_run.argtypes = [vp, dp, dp, C.c_int, C.c_double, C.c_int, C.c_int, dp, dp, ip, ip, dp, dp, dp]
    L.nmpc_hip_fmpc_kernel_names.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.nmpc_hip_fmpc_last_error.argtypes = []
    L.nmpc_hip_fmpc_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name != "nmpc_hip_fmpc_last_error":
            getattr(L, name).restype = C.c_int
    _declared = True
    return L


def check(rc: int) -> None:
    """std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError (FmpcSolver.hpp:285-354)."""
    if rc == _capi.OK:
        return
    msg = load().nmpc_hip_fmpc_last_error().decode(errors="replace")
    if rc in (_capi.ERR_INVALID_ARGUMENT, _capi.ERR_UNKNOWN_MODEL):
        raise ValueError(msg)
    raise RuntimeError(f"[nmpc_hip_fmpc {rc}] {msg}")


def model_names():
    L = load()
    out = []
    for i in range(L.nmpc_hip_fmpc_model_count()):
        p = C.c_char_p()
        check(L.nmpc_hip_fmpc_model_name(i, C.byref(p)))
        out.append(p.value.decode())
    return out


def model_info(model: str):
    """(state_dim, input_dim, ineq_dim, param_bytes)."""
    n, m, g, pb = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    check(load().nmpc_hip_fmpc_model_info(model.encode(), C.byref(n), C.byref(m), C.byref(g), C.byref(pb)))
    return n.value, m.value, g.value, pb.value


class FmpcProblem:
    """A problem object as the library sees it: the name of a registered problem type and the memory image of the C++ object
    (include/nmpc_amd/models/Fmpc*.hpp).  Every shipped FMPC problem type is a struct of doubles whose first member is dt, so
    the image is exposed as a float64 array `p` (p[0] = dt) and named views where the layout is fixed."""

    def __init__(self, model: str, dt: Optional[float] = None):
        self.model = model
        n, m, g, pb = model_info(model)
        self.state_dim, self.input_dim, self.ineq_dim = n, m, g
        blob = (C.c_ubyte * pb)()
        check(load().nmpc_hip_fmpc_model_default_params(model.encode(), blob, pb))
        self.p = np.frombuffer(bytes(blob), dtype=np.float64).copy()
        if dt is not None:
            self.p[0] = dt

    def dt(self) -> float:
        return float(self.p[0])

    def stateDim(self) -> int:
        return self.state_dim

    def inputDim(self) -> int:
        return self.input_dim

    def ineqDim(self) -> int:
        return self.ineq_dim

    def blob(self) -> bytes:
        return np.ascontiguousarray(self.p, dtype=np.float64).tobytes()


class FmpcProblemOscillator(FmpcProblem):
    """nmpc_amd::FmpcProblemOscillator = the reference's FmpcProblemOscillator (TestFmpcOscillator.cpp:18-135)."""

    def __init__(self, dt: float = 0.01):
        super().__init__("fmpc_oscillator", dt)


class FmpcProblemCartPole(FmpcProblem):
    """nmpc_amd::FmpcProblemCartPole = the reference's FmpcProblemCartPole (TestFmpcCartPole.cpp:32-267).  Image: dt, cart_mass,
    pole_mass, pole_length, running_x[4], running_u[1], terminal_x[4], ref_pos, u_max, x_max."""

    def __init__(self, dt: float = 0.01, ref_pos: float = 0.0):
        super().__init__("fmpc_cartpole", dt)
        self.p[13] = ref_pos

    @property
    def ref_pos(self) -> float:
        return float(self.p[13])


class FmpcProblemPointMass(FmpcProblem):
    """nmpc_amd::FmpcProblemPointMass (two inputs; no reference counterpart).  Image: dt, mass, drag, target[2], w_pos, w_vel,
    w_u, w_u_cross, w_term, u_max[2]."""

    def __init__(self, dt: float = 0.02):
        super().__init__("fmpc_pointmass", dt)


class Configuration:
    """FmpcSolver::Configuration (FmpcSolver.h:57-89).  Defaults come from the library."""

    _FIELDS = ("horizon_steps", "max_iter", "kkt_error_thre", "check_nan", "init_complementary_variable", "update_barrier_eps",
               "break_if_llt_fails", "enable_line_search", "merit_const_scale_from_lagrange_multipliers", "use_graph",
               "time_kernels")
    _BOOL = ("check_nan", "init_complementary_variable", "update_barrier_eps", "break_if_llt_fails", "enable_line_search",
             "merit_const_scale_from_lagrange_multipliers", "use_graph", "time_kernels")

    def __init__(self):
        c = CConfig()
        check(load().nmpc_hip_fmpc_default_config(C.byref(c)))
        self.print_level = 1  # host-side only (FmpcSolver.h:60)
        for k in self._FIELDS:
            v = getattr(c, k)
            setattr(self, k, bool(v) if k in self._BOOL else v)

    def to_c(self) -> CConfig:
        c = CConfig()
        for k in self._FIELDS:
            v = getattr(self, k)
            setattr(c, k, float(v) if k == "kkt_error_thre" else int(v))
        return c


@dataclass
class Variable:
    """FmpcSolver::Variable (FmpcSolver.h:117-158) of every instance: x_list [B][T+1][N], u_list [B][T][M], lambda_list
    [B][T+1][N], s_list [B][T][G], nu_list [B][T][G]."""
    x_list: np.ndarray
    u_list: np.ndarray
    lambda_list: np.ndarray
    s_list: np.ndarray
    nu_list: np.ndarray

    @staticmethod
    def make(problem: FmpcProblem, horizon_steps: int, batch: int) -> "Variable":
        n, m, g = problem.state_dim, problem.input_dim, problem.ineq_dim
        T = horizon_steps
        return Variable(np.zeros((batch, T + 1, n)), np.zeros((batch, T, m)), np.zeros((batch, T + 1, n)),
                        np.zeros((batch, T, g)), np.zeros((batch, T, g)))

    def reset(self, x: float, u: float, lam: float, s: float, nu: float) -> None:
        """Variable::reset (FmpcSolver.hpp:42-69)."""
        self.x_list[:] = x
        self.u_list[:] = u
        self.lambda_list[:] = lam
        self.s_list[:] = s
        self.nu_list[:] = nu

    @property
    def horizon_steps(self) -> int:
        return self.u_list.shape[1]

    def arrays(self):
        return self.x_list, self.u_list, self.lambda_list, self.s_list, self.nu_list


KERNEL_CLASSES = ("barrier", "coeff", "riccati", "delta", "step_length", "line_search", "update", "other")


@dataclass
class ComputationDuration:
    """FmpcSolver::ComputationDuration (FmpcSolver.h:252-287) for the whole batch [ms]: `solve` is the HIP-event time of the
    last solve; with config().time_kernels the per-kernel times fill the reference's split — coeff (coefficient kernel),
    backward + forward (the Riccati kernel runs both recursions: reported under backward), update (delta + step-length +
    line-search + update kernels) — and `kernels` / `launches` hold every kernel class (KERNEL_CLASSES)."""
    solve: float = 0.0
    coeff: float = 0.0
    backward: float = 0.0
    forward: float = 0.0
    update: float = 0.0
    kernels: Optional[dict] = None
    launches: Optional[dict] = None


class FmpcSolverBatch:
    """nmpc_fmpc::FmpcSolver<N, M, G> for `batch` independent instances on one MI355X."""

    def __init__(self, problem: FmpcProblem, batch: int, horizon_steps: int = 100, device: int = 0):
        self._L = load()
        self._h = C.c_void_p()
        self.problem = problem
        self.batch = int(batch)
        check(self._L.nmpc_hip_fmpc_create(problem.model.encode(), int(horizon_steps), int(batch), int(device), C.byref(self._h)))
        self._config = Configuration()
        self._config.horizon_steps = int(horizon_steps)
        self._pushed = None
        self.setProblem(problem)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._L.nmpc_hip_fmpc_destroy(h)
            h.value = None

    def config(self) -> Configuration:
        """FmpcSolver::config() (FmpcSolver.h:272-281): mutate the returned object; it is pushed to the library at solve()."""
        return self._config

    def _push_config(self) -> None:
        c = self._config.to_c()
        key = bytes(c)
        if key != self._pushed:
            check(self._L.nmpc_hip_fmpc_set_config(self._h, C.byref(c)))
            self._pushed = key

    def setProblem(self, problem, per_instance: bool = False) -> None:
        """The problem object the solver was constructed with (FmpcSolver.h:270,393); per_instance: a sequence of `batch`
        problem objects of the same type, one per instance."""
        if per_instance:
            blob = b"".join(p.blob() for p in problem)
        else:
            blob = problem.blob()
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        check(self._L.nmpc_hip_fmpc_set_problem(self._h, buf, len(blob), int(per_instance)))

    def setVariable(self, variable: Variable, barrier_eps=None) -> None:
        """Upload `initial_variable` (FmpcSolver.h:283)."""
        n, m, g = self.problem.state_dim, self.problem.input_dim, self.problem.ineq_dim
        T, B = self._config.horizon_steps, self.batch
        # FmpcSolver::checkVariable (FmpcSolver.hpp:287-311): sequence lengths -> std::invalid_argument
        for name, a, steps, e in (("x_list", variable.x_list, T + 1, n), ("u_list", variable.u_list, T, m),
                                  ("lambda_list", variable.lambda_list, T + 1, n), ("s_list", variable.s_list, T, g),
                                  ("nu_list", variable.nu_list, T, g)):
            if a.shape != (B, steps, e):
                raise ValueError(f"[FMPC] {name} length should be {steps} (shape {(B, steps, e)}) but {a.shape}.")
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in variable.arrays()]
        be = None if barrier_eps is None else np.ascontiguousarray(np.broadcast_to(barrier_eps, (B,)), dtype=np.float64)
        ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        check(self._L.nmpc_hip_fmpc_set_variable(self._h, *[ptr(a) for a in arrs], ptr(be), 0))

    def solve(self, current_t, current_x, initial_variable: Optional[Variable] = None) -> np.ndarray:
        """FmpcSolver::solve (FmpcSolver.h:283) for every instance; returns the Status of each [B].  initial_variable = None
        continues from the resident variable (the callers' `variable = solver.variable()`)."""
        B, n = self.batch, self.problem.state_dim
        self._push_config()
        if initial_variable is not None:
            self.setVariable(initial_variable)
        x0 = np.ascontiguousarray(current_x, dtype=np.float64)
        if x0.shape != (B, n):
            raise ValueError(f"current_x must have shape {(B, n)}, got {x0.shape}")
        t = None if current_t is None else np.ascontiguousarray(np.broadcast_to(current_t, (B,)), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        check(self._L.nmpc_hip_fmpc_solve(self._h, None if t is None else t.ctypes.data_as(dp), x0.ctypes.data_as(dp)))
        return self.status()

    def _get(self, field: int, shape, dtype=np.float64) -> np.ndarray:
        out = np.zeros(shape, dtype=dtype)
        if out.size:
            check(self._L.nmpc_hip_fmpc_get(self._h, field, out.ctypes.data_as(C.c_void_p), out.nbytes, 0))
        return out

    def status(self) -> np.ndarray:
        return self._get(FIELD_STATUS, (self.batch,), np.int32)

    def iters(self) -> np.ndarray:
        return self._get(FIELD_ITERS, (self.batch,), np.int32)

    def variable(self) -> Variable:
        """FmpcSolver::variable() (FmpcSolver.h:286-289)."""
        n, m, g = self.problem.state_dim, self.problem.input_dim, self.problem.ineq_dim
        T, B = self._config.horizon_steps, self.batch
        return Variable(self._get(FIELD_X, (B, T + 1, n)), self._get(FIELD_U, (B, T, m)), self._get(FIELD_LAMBDA, (B, T + 1, n)),
                        self._get(FIELD_S, (B, T, g)), self._get(FIELD_NU, (B, T, g)))

    def deltaVariable(self) -> Variable:
        """delta_variable_ (FmpcSolver.h:402) of the last iteration that reached the forward pass."""
        n, m, g = self.problem.state_dim, self.problem.input_dim, self.problem.ineq_dim
        T, B = self._config.horizon_steps, self.batch
        return Variable(self._get(FIELD_DELTA_X, (B, T + 1, n)), self._get(FIELD_DELTA_U, (B, T, m)),
                        self._get(FIELD_DELTA_LAMBDA, (B, T + 1, n)), self._get(FIELD_DELTA_S, (B, T, g)),
                        self._get(FIELD_DELTA_NU, (B, T, g)))

    def coeffList(self) -> dict:
        """The gains of coeffList() (FmpcSolver.h:292-295): k [B][T][M], K [B][T][M][N], s [B][T+1][N], P [B][T+1][N][N]."""
        n, m = self.problem.state_dim, self.problem.input_dim
        T, B = self._config.horizon_steps, self.batch
        K = self._get(FIELD_GAIN_K, (B, T, n, m))  # column-major M x N per step
        P = self._get(FIELD_GAIN_P, (B, T + 1, n, n))
        return dict(k=self._get(FIELD_GAIN_k, (B, T, m)), K=np.transpose(K, (0, 1, 3, 2)).copy(),
                    s=self._get(FIELD_GAIN_S, (B, T + 1, n)), P=np.transpose(P, (0, 1, 3, 2)).copy())

    def barrierEps(self) -> np.ndarray:
        return self._get(FIELD_BARRIER_EPS, (self.batch,))

    def meritFunc(self) -> np.ndarray:
        """[B][3]: merit_func_, merit_deriv_, merit_const_scale_ of the last line search (FmpcSolver.h:417-423)."""
        return self._get(FIELD_MERIT, (self.batch, 3))

    def partials(self) -> np.ndarray:
        """Diagnostic: [B][T+1][4] per-timestep terms of the horizon reductions as the last kernels left them (KKT-error terms,
        alpha_s / alpha_nu candidates, s . nu)."""
        return self._get(FIELD_PARTIALS, (self.batch, self._config.horizon_steps + 1, 4))

    def traceDataList(self) -> np.ndarray:
        """traceDataList() (FmpcSolver.h:298-301): [B][max_iter][NTRACE] (TRACE_COLUMNS); rows beyond iters() are zero."""
        return self._get(FIELD_TRACE, (self.batch, self._config.max_iter, NTRACE))

    def computationDuration(self) -> ComputationDuration:
        ms = C.c_float()
        check(self._L.nmpc_hip_fmpc_last_solve_ms(self._h, C.byref(ms)))
        d = ComputationDuration(solve=float(ms.value))
        if self._config.time_kernels:
            k = np.zeros(len(KERNEL_CLASSES))
            n = np.zeros(len(KERNEL_CLASSES), dtype=np.int32)
            check(self._L.nmpc_hip_fmpc_last_solve_kernel_ms(self._h, k.ctypes.data_as(C.POINTER(C.c_double)),
                                                             n.ctypes.data_as(C.POINTER(C.c_int))))
            d.kernels = dict(zip(KERNEL_CLASSES, (float(v) for v in k)))
            d.launches = dict(zip(KERNEL_CLASSES, (int(v) for v in n)))
            d.coeff = d.kernels["coeff"] + d.kernels["barrier"]
            d.backward = d.kernels["riccati"]
            d.update = d.kernels["delta"] + d.kernels["step_length"] + d.kernels["line_search"] + d.kernels["update"]
        return d

    def dumpTraceDataList(self, file_path: str, instance: int = 0) -> None:
        """FmpcSolver::dumpTraceDataList (FmpcSolver.hpp:257-283) for one instance; the four duration columns of the reference
        carry barrier_eps / alpha_s_max / alpha_nu_max / alpha_s here."""
        tr = self.traceDataList()[instance]
        it = int(self.iters()[instance])
        with open(file_path, "w") as f:
            f.write(" ".join(TRACE_COLUMNS) + "\n")
            for row in tr[:it]:
                f.write(" ".join([str(int(row[0]))] + [repr(float(v)) for v in row[1:]]) + "\n")

    def mpcRun(self, current_t, current_x, n_ticks: int, sim_dt: float, sim_substeps: int = 1, use_feedback: bool = False) -> dict:
        """The reference's closed-loop caller patterns, device-resident (nmpc_hip_fmpc_mpc_run)."""
        B, n, m = self.batch, self.problem.state_dim, self.problem.input_dim
        self._push_config()
        x0 = np.ascontiguousarray(current_x, dtype=np.float64)
        if x0.shape != (B, n):
            raise ValueError(f"current_x must have shape {(B, n)}, got {x0.shape}")
        t = np.ascontiguousarray(np.broadcast_to(0.0 if current_t is None else current_t, (B,)), dtype=np.float64)
        out = dict(x=np.zeros((B, n_ticks, n)), u0=np.zeros((B, n_ticks, m)), status=np.zeros((B, n_ticks), np.int32),
                   iters=np.zeros((B, n_ticks), np.int32), kkt_error=np.zeros((B, n_ticks)), x_final=np.zeros((B, n)),
                   t_final=np.zeros(B))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        check(self._L.nmpc_hip_fmpc_mpc_run(self._h, t.ctypes.data_as(dp), x0.ctypes.data_as(dp), int(n_ticks), float(sim_dt),
                                            int(sim_substeps), int(use_feedback), out["x"].ctypes.data_as(dp),
                                            out["u0"].ctypes.data_as(dp), out["status"].ctypes.data_as(ip),
                                            out["iters"].ctypes.data_as(ip), out["kkt_error"].ctypes.data_as(dp),
                                            out["x_final"].ctypes.data_as(dp), out["t_final"].ctypes.data_as(dp)))
        return out

    def kernelNames(self):
        p = C.c_char_p()
        check(self._L.nmpc_hip_fmpc_kernel_names(self._h, C.byref(p)))
        return p.value.decode().split(",")
