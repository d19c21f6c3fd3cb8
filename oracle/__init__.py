"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front end of the CPU oracle (oracle/ddp_oracle.hpp: a restatement of the reference's
nmpc_ddp::DDPSolver / BoxQP on plain arrays).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker / reported baseline.  Nothing under
nmpc_amd/ imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
NTRACE = 12
TRACE_FIELDS = (
    "iter", "cost", "lambda", "dlambda", "alpha", "k_rel_norm", "cost_update_actual",
    "cost_update_expected", "cost_update_ratio", "alpha_idx", "n_backward", "n_forward",
)


class OracleConfig(C.Structure):
    """Mirror of `oracle_config` (oracle_capi.cpp) = DDPSolver::Configuration (DDPSolver.h:47-110)."""

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
        ("alpha_list", C.c_double * 32),
    ]


def build(native: bool = False, out_dir: Optional[str] = None) -> str:
    """Compile the oracle with g++ (a few seconds).  native=True builds the -O3 -march=native variant that
    bench.py times as cpu_baseline; it must be built on the machine that runs it."""
    out_dir = out_dir or _BUILD
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target, f"OUT={out_dir}"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(out_dir, "liboracle_ddp_native.so" if native else "liboracle_ddp.so")


_libs: dict = {}


def lib(native: bool = False, out_dir: Optional[str] = None):
    key = (native, out_dir)
    if key in _libs:
        return _libs[key]
    path = os.path.join(out_dir or _BUILD, "liboracle_ddp_native.so" if native else "liboracle_ddp.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("oracle_capi.cpp", "ddp_oracle.hpp", "models.hpp", "model_cartpole.hpp", "models_builder.hpp")]
    if (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        path = build(native, out_dir)
    L = C.CDLL(path)
    L.oracle_default_config.argtypes = [C.POINTER(OracleConfig)]
    L.oracle_default_config.restype = None
    _libs[key] = L
    return L


def default_config(**kw) -> OracleConfig:
    c = OracleConfig()
    lib().oracle_default_config(C.byref(c))
    for k, v in kw.items():
        if k == "alpha_list":
            v = list(v)
            c.n_alpha = len(v)
            for i, a in enumerate(v):
                c.alpha_list[i] = a
        else:
            if not hasattr(c, k):
                raise AttributeError(k)
            setattr(c, k, v)
    return c


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def model_dims(model: str):
    n, m, p = C.c_int(), C.c_int(), C.c_int()
    rc = lib().oracle_model_dims(model.encode(), C.byref(n), C.byref(m), C.byref(p))
    if rc != 0:
        raise ValueError(f"unknown oracle model {model!r}")
    return n.value, m.value, p.value


def input_dims(model: str, params, t0: float, T: int) -> np.ndarray:
    out = np.zeros(T, dtype=np.int32)
    p = _f64(params)
    rc = lib().oracle_input_dims(model.encode(), _dp(p), C.c_double(t0), T, _ip(out))
    assert rc == 0
    return out


@dataclass
class ModelEval:
    m: int
    xn: np.ndarray
    running_cost: float
    terminal_cost: float
    Fx: np.ndarray
    Fu: np.ndarray
    Lx: np.ndarray
    Lu: np.ndarray
    Lxx: np.ndarray
    Luu: np.ndarray
    Lxu: np.ndarray
    Vx: np.ndarray
    Vxx: np.ndarray


def model_eval(model: str, params, t: float, x, u) -> ModelEval:
    """All nine DDPProblem methods at (t, x, u).  Matrices are returned as (rows, cols) numpy arrays."""
    n, mmax, _ = model_dims(model)
    mm = max(mmax, 1)
    x = _f64(x)
    ubuf = np.zeros(mm)
    u = np.asarray(u, dtype=np.float64).ravel()
    ubuf[: u.size] = u
    p = _f64(params)
    xn = np.zeros(n)
    rc_, tc_ = C.c_double(), C.c_double()
    Fx = np.zeros(n * n)
    Fu = np.zeros(n * mm)
    Lx = np.zeros(n)
    Lu = np.zeros(mm)
    Lxx = np.zeros(n * n)
    Luu = np.zeros(mm * mm)
    Lxu = np.zeros(n * mm)
    Vx = np.zeros(n)
    Vxx = np.zeros(n * n)
    m = C.c_int()
    L = lib()
    L.oracle_model_eval.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_double] + [C.POINTER(C.c_double)] * 3 + [
        C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_double)] * 9 + [C.POINTER(C.c_int)]
    rc = L.oracle_model_eval(model.encode(), _dp(p), t, _dp(x), _dp(ubuf), _dp(xn), C.byref(rc_), C.byref(tc_),
                             _dp(Fx), _dp(Fu), _dp(Lx), _dp(Lu), _dp(Lxx), _dp(Luu), _dp(Lxu), _dp(Vx), _dp(Vxx),
                             C.byref(m))
    assert rc == 0
    mi = m.value
    return ModelEval(
        m=mi, xn=xn, running_cost=rc_.value, terminal_cost=tc_.value,
        Fx=Fx.reshape(n, n).T.copy(), Fu=Fu[: n * mi].reshape(mi, n).T.copy(),
        Lx=Lx, Lu=Lu[:mi].copy(), Lxx=Lxx.reshape(n, n).T.copy(),
        Luu=Luu[: mi * mi].reshape(mi, mi).T.copy(), Lxu=Lxu[: n * mi].reshape(mi, n).T.copy(),
        Vx=Vx, Vxx=Vxx.reshape(n, n).T.copy(),
    )


@dataclass
class BoxQPResult:
    x: np.ndarray
    retval: int
    free_idxs: list
    iter: int
    factorization_num: int


def boxqp_solve(H, g, lower, upper, initial_x=None) -> BoxQPResult:
    H = np.asarray(H, dtype=np.float64)
    m = H.shape[0]
    Hc = np.ascontiguousarray(H.T).ravel()  # column-major
    g, lower, upper = _f64(g), _f64(lower), _f64(upper)
    x0 = _f64(initial_x)
    x = np.zeros(m)
    free = np.zeros(m, dtype=np.int32)
    retval, nfree, it, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().oracle_boxqp_solve(m, _dp(Hc), _dp(g), _dp(lower), _dp(upper), _dp(x0), _dp(x), C.byref(retval),
                                  _ip(free), C.byref(nfree), C.byref(it), C.byref(nf))
    assert rc == 0
    return BoxQPResult(x, retval.value, list(free[: nfree.value]), it.value, nf.value)


@dataclass
class SolveResult:
    """Reference-layout results of one solve: X (T+1, n), U (T, mm), cost (T+1,), k (T, mm),
    K (T, mm, n) [K[t] is the m x n gain], trace (n_trace, 12)."""
    X: np.ndarray
    U: np.ndarray
    cost: np.ndarray
    k: np.ndarray
    K: np.ndarray
    trace: np.ndarray
    status: int
    dV: np.ndarray
    qp_retval: np.ndarray
    qp_free_mask: np.ndarray
    m_list: np.ndarray = field(default=None)

    @property
    def iters(self) -> int:
        return int(self.trace[-1, 0])


def solve(model: str, cfg: OracleConfig, x0, u_init, t0: float = 0.0, params=None,
          lower=None, upper=None) -> SolveResult:
    """lower / upper: (MM,) constant limits, or (T, MM) tables of time-varying limits (input_limits_func_(t0 + i dt))."""
    n, mmax, _ = model_dims(model)
    mm = max(mmax, 1)
    T = cfg.horizon_steps
    x0 = _f64(x0)
    u_init = _f64(np.asarray(u_init, dtype=np.float64).reshape(T, mm))
    p = _f64(params)
    lo, up = _f64(lower), _f64(upper)
    X = np.zeros((T + 1, n))
    U = np.zeros((T, mm))
    cost = np.zeros(T + 1)
    k = np.zeros((T, mm))
    K = np.zeros((T, n, mm))
    trace = np.zeros((cfg.max_iter + 1, NTRACE))
    ntr, status = C.c_int(), C.c_int()
    dV = np.zeros(2)
    qret = np.zeros(T, dtype=np.int32)
    qmask = np.zeros(T, dtype=np.uint32)
    L = lib()
    L.oracle_ddp_solve.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(OracleConfig), C.c_double] + [
        C.POINTER(C.c_double)] * 10 + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int), C.POINTER(C.c_uint), C.c_int]
    per_step = 1 if (lo is not None and lo.ndim == 2) else 0
    rc = L.oracle_ddp_solve(model.encode(), _dp(p), C.byref(cfg), t0, _dp(x0), _dp(u_init), _dp(lo), _dp(up),
                            _dp(X), _dp(U), _dp(cost), _dp(k), _dp(K), _dp(trace), C.byref(ntr), C.byref(status),
                            _dp(dV), _ip(qret), qmask.ctypes.data_as(C.POINTER(C.c_uint)), per_step)
    if rc != 0:
        raise RuntimeError(f"oracle_ddp_solve failed: {rc}")
    return SolveResult(X, U, cost, k, K.transpose(0, 2, 1).copy(), trace[: ntr.value].copy(), status.value, dV,
                       qret, qmask, input_dims(model, params, t0, T))


@dataclass
class BatchResult:
    X: np.ndarray
    U: np.ndarray
    cost: np.ndarray
    k: np.ndarray
    K: np.ndarray
    status: np.ndarray
    iters: np.ndarray
    trace_last: np.ndarray
    alpha_idx_hist: Optional[np.ndarray]
    total_iters: int
    seconds: float


def solve_batch(model: str, cfg: OracleConfig, x0, u_init, t0=None, params=None, lower=None, upper=None,
                n_threads: int = 1, want_gains: bool = True, want_alpha_hist: bool = False,
                native: bool = False, native_dir: Optional[str] = None) -> BatchResult:
    n, mmax, _ = model_dims(model)
    mm = max(mmax, 1)
    T = cfg.horizon_steps
    x0 = _f64(x0)
    B = x0.shape[0]
    u_init = _f64(np.asarray(u_init, dtype=np.float64).reshape(B, T, mm))
    t0 = _f64(t0)
    p = _f64(params)
    lo, up = _f64(lower), _f64(upper)
    X = np.zeros((B, T + 1, n))
    U = np.zeros((B, T, mm))
    cost = np.zeros((B, T + 1))
    k = np.zeros((B, T, mm)) if want_gains else None
    K = np.zeros((B, T, n, mm)) if want_gains else None
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    trl = np.zeros((B, NTRACE))
    hist = np.zeros((B, cfg.max_iter), dtype=np.int32) if want_alpha_hist else None
    tot = C.c_longlong()
    sec = C.c_double()
    L = lib(native, native_dir) if native else lib()
    L.oracle_ddp_solve_batch.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(OracleConfig), C.c_int] + [
        C.POINTER(C.c_double)] * 5 + [C.c_int] + [C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_int)] * 2 + [
        C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_int]
    # limits: (MM,) constant, (T, MM) one time-varying table for all, (B, T, MM) one per instance
    per_step = 0 if lo is None else {1: 0, 2: 1, 3: 2}[lo.ndim]
    rc = L.oracle_ddp_solve_batch(model.encode(), _dp(p), C.byref(cfg), B, _dp(t0), _dp(x0), _dp(u_init), _dp(lo),
                                  _dp(up), n_threads, _dp(X), _dp(U), _dp(cost), _dp(k), _dp(K), _ip(status),
                                  _ip(iters), _dp(trl), _ip(hist), C.byref(tot), C.byref(sec), per_step)
    if rc != 0:
        raise RuntimeError(f"oracle_ddp_solve_batch failed: {rc}")
    return BatchResult(X, U, cost, k, None if K is None else K.transpose(0, 1, 3, 2).copy(), status, iters, trl,
                       hist, tot.value, sec.value)


@dataclass
class MpcResult:
    t: np.ndarray
    x: np.ndarray
    u0: np.ndarray
    iters: np.ndarray
    m0: np.ndarray
    x_final: np.ndarray
    t_final: float


def mpc_run(model: str, cfg: OracleConfig, x0, n_ticks: int, t0: float = 0.0, params=None,
            max_iter_after_first: int = 0, shift_warm_start: bool = True, sim_substeps: int = 0,
            sim_dt: float = 0.0, lower=None, upper=None) -> MpcResult:
    n, mmax, _ = model_dims(model)
    mm = max(mmax, 1)
    x0 = _f64(x0)
    p = _f64(params)
    lo, up = _f64(lower), _f64(upper)
    t_log = np.zeros(n_ticks)
    x_log = np.zeros((n_ticks, n))
    u_log = np.zeros((n_ticks, mm))
    it_log = np.zeros(n_ticks, dtype=np.int32)
    m_log = np.zeros(n_ticks, dtype=np.int32)
    xf = np.zeros(n)
    tf = C.c_double()
    cfg_copy = OracleConfig.from_buffer_copy(cfg)
    L = lib()
    L.oracle_mpc_run.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(OracleConfig), C.c_int, C.c_double,
                                 C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_double] + [
        C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_double)] * 2
    rc = L.oracle_mpc_run(model.encode(), _dp(p), C.byref(cfg_copy), max_iter_after_first, t0, _dp(x0), n_ticks,
                          1 if shift_warm_start else 0, sim_substeps, sim_dt, _dp(lo), _dp(up), _dp(t_log),
                          _dp(x_log), _dp(u_log), _ip(it_log), _ip(m_log), _dp(xf), C.byref(tf))
    if rc != 0:
        raise RuntimeError(f"oracle_mpc_run failed: {rc}")
    return MpcResult(t_log, x_log, u_log, it_log, m_log, xf, tf.value)


# default parameter vectors (same order as each model's setParams in oracle/models*.hpp)
def default_params(model: str, **over) -> np.ndarray:
    model = model[:-4] if model.endswith("_f32") else model  # the fp32 instantiations take the same parameter vector
    if model == "cartpole":
        d = dict(dt=0.01, cart_mass=1.0, pole_mass=0.5, pole_length=2.0, running_x=(0.1, 1.0, 0.01, 0.1),
                 running_u=0.001, terminal_x=(0.1, 1.0, 0.01, 0.1), ref_pos=0.0)
        d.update(over)
        return np.array([d["dt"], d["cart_mass"], d["pole_mass"], d["pole_length"], *d["running_x"],
                         d["running_u"], *d["terminal_x"], d["ref_pos"]], dtype=np.float64)
    if model == "bipedal":
        d = dict(dt=0.01, running_vel=1e-14, running_zmp=1e-1, terminal_pos=1e2, terminal_vel=1.0, end_t=20.0)
        d.update(over)
        return np.array([d[k] for k in ("dt", "running_vel", "running_zmp", "terminal_pos", "terminal_vel",
                                        "end_t")], dtype=np.float64)
    if model == "vertical":
        d = dict(dt=0.01, running_x=(1.0, 1e-3), running_u=1e-4, terminal_x=(1.0, 1e-3), mass=1.0,
                 ref_switch_t=8.0)
        d.update(over)
        return np.array([d["dt"], *d["running_x"], d["running_u"], *d["terminal_x"], d["mass"],
                         d["ref_switch_t"]], dtype=np.float64)
    if model == "centroidal":
        d = dict(dt=0.03, running_u=1e-6, mass=100.0, flight_t0=1.4, flight_t1=1.6, ref_switch_t=1.5,
                 w_pos_ang=1.0, w_lin=0.0, rect2=(0.4, -0.1, 0.6, 0.1))
        d.update(over)
        return np.array([d["dt"], d["running_u"], d["mass"], d["flight_t0"], d["flight_t1"], d["ref_switch_t"],
                         d["w_pos_ang"], d["w_lin"], *d["rect2"]], dtype=np.float64)
    if model == "quadrotor":
        d = dict(dt=0.02, mass=1.0, J=(0.01, 0.01, 0.02), arm=0.2, yaw_coef=0.05, w_pos=1.0, w_rpy=0.5,
                 w_vel=0.1, w_omega=0.05, w_u=0.01, wt_scale=10.0, ref_pos=(0.0, 0.0, 1.0))
        d.update(over)
        return np.array([d["dt"], d["mass"], *d["J"], d["arm"], d["yaw_coef"], d["w_pos"], d["w_rpy"], d["w_vel"],
                         d["w_omega"], d["w_u"], d["wt_scale"], *d["ref_pos"], 0.0, 0.0], dtype=np.float64)
    if model == "manipulator":
        d = dict(dt=0.01, w_diag=2.0, w_off=0.15, damping=0.5, grav_scale=4.0, wq=1.0, wv=0.05, wu=0.002,
                 wt_scale=20.0, q_ref_scale=0.3)
        d.update(over)
        return np.array([d[k] for k in ("dt", "w_diag", "w_off", "damping", "grav_scale", "wq", "wv", "wu",
                                        "wt_scale", "q_ref_scale")] + [0.0, 0.0], dtype=np.float64)
    if model == "planar_vtol":
        d = dict(dt=0.02, mass=1.0, inertia=0.02, arm=0.25, w_pos=1.0, w_ang=0.5, w_vel=0.1, w_omega=0.05, w_u=0.01,
                 wt_scale=10.0, ref_pos=(0.0, 1.0))
        d.update(over)
        return np.array([d["dt"], d["mass"], d["inertia"], d["arm"], d["w_pos"], d["w_ang"], d["w_vel"], d["w_omega"],
                         d["w_u"], d["wt_scale"], *d["ref_pos"]], dtype=np.float64)
    raise ValueError(model)
