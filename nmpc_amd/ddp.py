"""Python mirror of the reference's nmpc_ddp::DDPSolver interface for a BATCH of problem instances.

Same member names, argument meaning and error behaviour as
/root/reference/nmpc_ddp/include/nmpc_ddp/DDPSolver.h:255-308 (`config()`, `solve()`, `setInputLimitsFunc()`,
`controlData()`, `traceDataList()`, `computationDuration()`, `dumpTraceDataList()`), with a leading batch
axis.  Everything numeric happens in libnmpc_hip_ddp.so through the C-ABI (include/nmpc_hip_ddp.h); this file
only marshals arrays and re-raises status codes as the exception types the reference throws.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from .models import _Problem


class Configuration:
    """DDPSolver::Configuration (DDPSolver.h:47-110).  Defaults come from the library
    (nmpc_hip_ddp_default_config), not from Python."""

    _PLAIN = ("with_input_constraint", "max_iter", "horizon_steps", "reg_type", "initial_lambda",
              "initial_dlambda", "lambda_factor", "lambda_min", "lambda_max", "k_rel_norm_thre", "lambda_thre",
              "cost_update_ratio_thre", "cost_update_thre", "use_state_eq_second_derivative", "qp_max_iter",
              "qp_grad_thre", "qp_rel_improve_thre", "qp_step_factor", "qp_min_step", "qp_armijo_param",
              "trace_level", "line_search_fan_out", "ragged_schedule")

    def __init__(self):
        c = _capi.Config()
        _capi.check(_capi.load().nmpc_hip_ddp_default_config(C.byref(c)))
        self.print_level = 1  # host-side only (DDPSolver.h:62-63)
        for k in self._PLAIN:
            v = getattr(c, k)
            setattr(self, k, bool(v) if k in ("with_input_constraint", "use_state_eq_second_derivative") else v)
        self.alpha_list = np.array(c.alpha_list[: c.n_alpha], dtype=np.float64)

    def to_c(self) -> _capi.Config:
        c = _capi.Config()
        for k in self._PLAIN:
            setattr(c, k, int(getattr(self, k)) if isinstance(getattr(c, k), int) else float(getattr(self, k)))
        a = np.asarray(self.alpha_list, dtype=np.float64).ravel()
        if a.size > _capi.MAX_ALPHA:
            raise ValueError(f"alpha_list supports at most {_capi.MAX_ALPHA} entries")
        c.n_alpha = a.size
        for i, v in enumerate(a):
            c.alpha_list[i] = v
        return c


@dataclass
class ControlData:
    """DDPSolver::ControlData (DDPSolver.h:113-123) of one instance."""
    x_list: np.ndarray  # (T+1, n)
    u_list: List[np.ndarray]  # T vectors of size inputDim(t)
    cost_list: np.ndarray  # (T+1,)


@dataclass
class TraceData:
    """DDPSolver::TraceData (DDPSolver.h:179-216).  duration_* have no per-instance meaning on the GPU and are
    0; alpha_idx / n_backward / n_forward are the discrete decisions of the iteration."""
    iter: int = 0
    cost: float = 0.0
    lambda_: float = 0.0
    dlambda: float = 0.0
    alpha: float = 0.0
    k_rel_norm: float = 0.0
    cost_update_actual: float = 0.0
    cost_update_expected: float = 0.0
    cost_update_ratio: float = 0.0
    duration_derivative: float = 0.0
    duration_backward: float = 0.0
    duration_forward: float = 0.0
    alpha_idx: int = -1
    n_backward: int = 0
    n_forward: int = 0


@dataclass
class MpcLog:
    """Per-tick log of DDPSolverBatch.mpcRun (the columns the reference's MPC tests dump, e.g.
    TestDDPBipedal.cpp:251-262): state handed to the solve of the tick, first input of its solution, iterations, status."""
    t: np.ndarray  # (B, n_ticks)
    x: np.ndarray  # (B, n_ticks, n)
    u0: np.ndarray  # (B, n_ticks, MM)
    iters: np.ndarray  # (B, n_ticks)
    status: np.ndarray  # (B, n_ticks)
    m0: np.ndarray  # (B, n_ticks) input dimension of the first timestep
    x_final: np.ndarray  # (B, n) state after the last advance
    t_final: np.ndarray  # (B,)
    duration: "ComputationDuration" = None  # of the run's last solve (batch-wide)

    def ticks(self, b: int):
        """The rows of instance b as result_tables.Tick objects."""
        from .result_tables import Tick
        return [Tick(float(self.t[b, k]), self.x[b, k], self.u0[b, k], int(self.m0[b, k]), int(self.iters[b, k]), self.duration)
                for k in range(self.t.shape[1])]

    def dump(self, file_path: str, b: int, columns) -> None:
        """Write instance b's closed-loop run as one of the reference's per-test result tables (TestDDPBipedal.cpp:242,259-262
        and the three others; makers in nmpc_amd.result_tables): a header of column names, one line per tick — the format the
        reference's plot scripts load with np.genfromtxt(path, names=True)."""
        from .result_tables import write_table
        write_table(file_path, self.ticks(b), columns)


@dataclass
class StreamResult:
    """Results of DDPSolverBatch.solveStream: one row per instance of the queue (controlData() / the last traceDataList() row of each)."""
    X: np.ndarray  # (N, T+1, n)
    U: np.ndarray  # (N, T, MM)
    cost: np.ndarray  # (N, T+1)
    status: np.ndarray  # (N,)  1 converged, 0 max_iter, -1 failed
    iters: np.ndarray  # (N,)
    trace_last: np.ndarray  # (N, 12)
    dV: np.ndarray  # (N, 2)
    rounds: int = 0
    device_ms: float = 0.0


@dataclass
class ComputationDuration:
    """DDPSolver::ComputationDuration (DDPSolver.h:219-247) for the whole batch [msec]: `solve` is the HIP-event
    time of ingest + solve kernel, `opt` the solve kernel alone, `setup` their difference.  `backward` and `forward`
    split `opt` by the shader-clock shares of the kernel's phases (nmpc_hip_ddp_last_solve_phases).  The linearisation is
    fused into the backward sweep and the sweep's sub-steps are not timed separately: `derivative`, `Q`, `reg`, `gain` (parts
    of `backward` here) stay 0."""
    solve: float = 0.0
    setup: float = 0.0
    opt: float = 0.0
    derivative: float = 0.0
    backward: float = 0.0
    forward: float = 0.0
    Q: float = 0.0
    reg: float = 0.0
    gain: float = 0.0


class DDPSolverBatch:
    """Batched DDP solver on one MI355X.

    problem: a nmpc_amd.models problem handle (the reference passes std::shared_ptr<DDPProblem>, DDPSolver.h:255)
    batch_size: number of independent instances solved per `solve()` call
    """

    def __init__(self, problem: _Problem, batch_size: int, device: int = 0):
        self._L = _capi.load()
        self.problem = problem
        self.batch_size = int(batch_size)
        self.device = int(device)
        self.n, self.m_max, self.dynamic_input, _ = problem.dims()
        self.mm = max(self.m_max, 1)
        self._config = Configuration()
        self._h = C.c_void_p()
        self._h_T = None
        self._limits = None
        self._cache = {}

    # ---- DDPSolver::config()  (DDPSolver.h:258-267) ----
    def config(self) -> Configuration:
        return self._config

    # ---- DDPSolver::setInputLimitsFunc  (DDPSolver.h:282-285) ----
    def setInputLimitsFunc(self, input_limits_func) -> None:
        """input_limits_func(t) -> (lower, upper).  The reference evaluates it at every timestep of the backward pass,
        input_limits_func_(current_t + i * dt) (DDPSolver.hpp:470-472); solve() samples it there, for every instance's own
        current_t, and hands the device the constant pair when the samples agree and the sampled table otherwise
        (nmpc_hip_ddp_set_input_limits_horizon)."""
        self._limits_func = input_limits_func

    def _sample_limits_func(self, t0: np.ndarray, extra_rows: int = 0) -> None:
        """Samples the limits function at current_t + i dt for the timesteps of the next solve (+ extra_rows further ones: a
        device-resident shift loop of extra_rows + 1 ticks, each of which starts one timestep later)."""
        if getattr(self, "_limits_func", None) is None:
            return
        T = int(self._config.horizon_steps) + int(extra_rows)
        dt = float(self.problem.dt())
        same_t0 = bool(np.all(t0 == t0[0]))
        starts = t0[:1] if same_t0 else t0
        lo = np.full((len(starts), T, self.mm), -np.inf)
        up = np.full((len(starts), T, self.mm), np.inf)
        for tb, ts in enumerate(starts):
            for i in range(T):
                l, u = self._limits_func(float(ts) + i * dt)
                l = np.asarray(l, dtype=np.float64).ravel()
                u = np.asarray(u, dtype=np.float64).ravel()
                if l.size != u.size or l.size > self.mm:
                    raise ValueError("input_limits_func should return vectors of the input dimension")
                lo[tb, i, : l.size] = l
                up[tb, i, : u.size] = u
        self._limits = (lo[0, 0].copy(), up[0, 0].copy())
        constant = bool(np.all(lo == lo[0, 0]) and np.all(up == up[0, 0]))
        self._limits_horizon = None if constant else (np.ascontiguousarray(lo), np.ascontiguousarray(up), not same_t0)
        self._limits_horizon_dirty = True

    def setInputLimitsHorizon(self, lower, upper) -> None:
        """Time-varying limits as tables: (T, MM) shared by every instance or (B, T, MM); None, None removes them."""
        if lower is None and upper is None:
            self._limits_horizon = None
        else:
            lo = np.ascontiguousarray(np.asarray(lower, dtype=np.float64))
            up = np.ascontiguousarray(np.asarray(upper, dtype=np.float64))
            T = int(self._config.horizon_steps)
            if lo.shape != up.shape or lo.shape[-2:] != (T, self.mm) or lo.ndim not in (2, 3):
                raise ValueError(f"limits tables should have shape ({T}, {self.mm}) or (B, {T}, {self.mm})")
            self._limits_horizon = (lo, up, lo.ndim == 3)
        self._limits_func = None
        self._limits_horizon_dirty = True

    def setInputLimits(self, lower, upper) -> None:
        lo = np.full(self.mm, -np.inf)
        up = np.full(self.mm, np.inf)
        lower = np.asarray(lower, dtype=np.float64).ravel()
        upper = np.asarray(upper, dtype=np.float64).ravel()
        lo[: lower.size] = lower
        up[: upper.size] = upper
        if lower.size == 1 and self.mm > 1:  # scalar limits apply to every input
            lo[:] = lower[0]
            up[:] = upper[0]
        self._limits = (lo, up)
        # constant limits replace a limits function / table given earlier (DDPSolverBatch.hpp setInputLimits does the same)
        self._limits_func = None
        if getattr(self, "_limits_horizon", None) is not None:
            self._limits_horizon = None
            self._limits_horizon_dirty = True

    def setInputLimitsBatch(self, lower, upper) -> None:
        """Per-instance limits: lower, upper of shape (B, MM) (constant in time), or None, None to go back to the shared
        ones.  A batch of DDPSolver objects each with its own setInputLimitsFunc."""
        if lower is None and upper is None:
            self._limits_batch = None
        else:
            lo = np.ascontiguousarray(np.asarray(lower, dtype=np.float64).reshape(self.batch_size, self.mm))
            up = np.ascontiguousarray(np.asarray(upper, dtype=np.float64).reshape(self.batch_size, self.mm))
            self._limits_batch = (lo, up)
        self._limits_batch_dirty = True

    # ---- handle management ----
    def _ensure_handle(self):
        T = int(self._config.horizon_steps)
        if self._h and self._h_T == T:
            return
        self.close()
        _capi.check(self._L.nmpc_hip_ddp_create(self.problem.name.encode(), T, self.batch_size, self.device,
                                                C.byref(self._h)))
        self._h_T = T
        if getattr(self, "_kernel", None) is not None:  # choices made before the (lazily created) handle existed
            _capi.check(self._L.nmpc_hip_ddp_set_kernel(self._h, self._kernel.encode()))
        if getattr(self, "_dispatch_batch", 0):
            _capi.check(self._L.nmpc_hip_ddp_set_dispatch_batch(self._h, self._dispatch_batch))
        self._problem_batch_dirty = getattr(self, "_problem_batch", None) is not None  # a new handle starts shared
        self._limits_batch_dirty = getattr(self, "_limits_batch", None) is not None
        self._limits_horizon_dirty = getattr(self, "_limits_horizon", None) is not None  # a table given to the old handle

    def close(self):
        if getattr(self, "_h", None):
            self._L.nmpc_hip_ddp_destroy(self._h)
            self._h = C.c_void_p()
            self._h_T = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _push_state(self):
        self._ensure_handle()
        _capi.check(self._L.nmpc_hip_ddp_set_model_params(self._h, C.byref(self.problem.blob),
                                                          C.sizeof(self.problem.blob)))
        if getattr(self, "_problem_batch_dirty", False):
            if self._problem_batch is None:
                _capi.check(self._L.nmpc_hip_ddp_set_model_params_batch(self._h, None, 0))
            else:
                nb = C.sizeof(self.problem.blob)
                raw = b"".join(bytes(p.blob) for p in self._problem_batch)
                _capi.check(self._L.nmpc_hip_ddp_set_model_params_batch(self._h, raw, nb))
            self._problem_batch_dirty = False
        c = self._config.to_c()
        if getattr(self, "_pooled", False) and c.ragged_schedule == 0:
            c.ragged_schedule = 1  # a pool's handle has other batches queued behind this one: "automatic" means on (where supported)
        _capi.check(self._L.nmpc_hip_ddp_set_config(self._h, C.byref(c)))
        if self._limits is not None:
            lo, up = self._limits
            _capi.check(self._L.nmpc_hip_ddp_set_input_limits(
                self._h, lo.ctypes.data_as(C.POINTER(C.c_double)), up.ctypes.data_as(C.POINTER(C.c_double))))
        if getattr(self, "_limits_horizon_dirty", False):
            dp = C.POINTER(C.c_double)
            hz = getattr(self, "_limits_horizon", None)
            if hz is None:
                _capi.check(self._L.nmpc_hip_ddp_set_input_limits_horizon(self._h, None, None, 0))
            else:
                lo, up, per_instance = hz
                if lo.shape[-2] < self._h_T:  # (validated against the horizon_steps of the time it was given)
                    raise ValueError(f"limits tables have {lo.shape[-2]} rows but horizon_steps is {self._h_T}")
                _capi.check(self._L.nmpc_hip_ddp_set_input_limits_schedule(self._h, lo.ctypes.data_as(dp), up.ctypes.data_as(dp),
                                                                           lo.shape[-2], 1 if per_instance else 0))
            self._limits_horizon_dirty = False
        if getattr(self, "_limits_batch_dirty", False):
            dp = C.POINTER(C.c_double)
            if self._limits_batch is None:
                _capi.check(self._L.nmpc_hip_ddp_set_input_limits_batch(self._h, None, None))
            else:
                lo, up = self._limits_batch
                _capi.check(self._L.nmpc_hip_ddp_set_input_limits_batch(self._h, lo.ctypes.data_as(dp),
                                                                        up.ctypes.data_as(dp)))
            self._limits_batch_dirty = False

    def setProblemBatch(self, problems: Optional[Sequence[_Problem]]) -> None:
        """One problem object per instance (a batch of DDPSolver objects each built with its own problem,
        DDPSolver.hpp:20-24), or None to go back to the shared one.  dt and inputDim(t) must agree with the shared
        problem."""
        self._problem_batch = None if problems is None else list(problems)
        if self._problem_batch is not None and len(self._problem_batch) != self.batch_size:
            raise ValueError(f"problem batch should be {self.batch_size} but {len(self._problem_batch)}.")
        self._problem_batch_dirty = True

    def inputDims(self, t0: float) -> np.ndarray:
        """problem->inputDim(t0 + i dt) for every step of the horizon."""
        self._push_state()
        out = np.zeros(self._h_T, dtype=np.int32)
        _capi.check(self._L.nmpc_hip_ddp_input_dims(self._h, float(t0), out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    def _pack_u(self, initial_u_list, current_t) -> np.ndarray:
        """Accept either a padded array (B, T, MM) or nested sequences [b][i] -> vector of size inputDim(t).
        Size checks reproduce DDPSolver.hpp:41-58."""
        B, T, MM = self.batch_size, int(self._config.horizon_steps), self.mm
        if isinstance(initial_u_list, np.ndarray) and initial_u_list.dtype != object:
            u = np.asarray(initial_u_list, dtype=np.float64)
            if u.ndim == 2 and MM == 1:
                u = u[:, :, None]
            if u.ndim != 3 or u.shape[0] != B:
                raise ValueError(f"initial_u_list batch should be {B} but {u.shape[0] if u.ndim else 0}.")
            if u.shape[1] != T:
                raise ValueError(f"initial_u_list length should be {T} but {u.shape[1]}.")
            if u.shape[2] != MM:
                raise RuntimeError(f"initial_u dimension should be {MM} but {u.shape[2]}.")
            return np.ascontiguousarray(u)
        if len(initial_u_list) != B:
            raise ValueError(f"initial_u_list batch should be {B} but {len(initial_u_list)}.")
        u = np.zeros((B, T, MM))
        for b, ul in enumerate(initial_u_list):
            if len(ul) != T:
                raise ValueError(f"initial_u_list length should be {T} but {len(ul)}.")
            dims = self.inputDims(current_t[b]) if self.dynamic_input else None
            for i, ui in enumerate(ul):
                ui = np.asarray(ui, dtype=np.float64).ravel()
                want = int(dims[i]) if dims is not None else self.m_max
                if ui.size != want:
                    raise RuntimeError(f"initial_u dimension should be {want} but {ui.size}. i: {i}, "
                                       f"time: {current_t[b] + i * self.problem.dt()}")
                u[b, i, : ui.size] = ui
        return u

    # ---- DDPSolver::solve  (DDPSolver.h:275, DDPSolver.hpp:26-141) ----
    def solve(self, current_t, current_x, initial_u_list) -> np.ndarray:
        """current_t: scalar or (B,), current_x: (B, n), initial_u_list: (B, T, MM) padded or nested lists.
        Returns a bool array: True where the reference's solve() would return true (retval == 1)."""
        B = self.batch_size
        t0 = np.broadcast_to(np.asarray(current_t, dtype=np.float64), (B,)).copy()
        x0 = np.ascontiguousarray(np.asarray(current_x, dtype=np.float64).reshape(B, self.n))
        self._sample_limits_func(t0)
        self._push_state()
        u = self._pack_u(initial_u_list, t0)
        dp = C.POINTER(C.c_double)
        _capi.check(self._L.nmpc_hip_ddp_solve(self._h, t0.ctypes.data_as(dp), x0.ctypes.data_as(dp),
                                               u.ctypes.data_as(dp)))
        self._cache = {}
        status = self.status()
        if self._config.print_level >= 1:
            n_fail = int((status < 0).sum())
            if n_fail:
                print(f"[DDP] Failure due to large lambda in {n_fail} of {B} instances.")
        return status == 1

    # ---- a queue of instances through the handle's slots (nmpc_hip_ddp_solve_stream; round 6) ----
    def solveStream(self, current_t, current_x, initial_u_list, span: int = 0) -> "StreamResult":
        """N >> batch_size instances (current_t: scalar or (N,), current_x: (N, n), initial_u_list: (N, T, MM)), each solved to ITS
        convergence as DDPSolver::solve would (DDPSolver.hpp:26-141): batch_size slots, and the slot of an instance that has finished
        takes the next one of the queue after at most `span` (0: 8) further iterations of its neighbours
        (include/nmpc_amd/hip/stream_schedule.hpp).  Every instance returns the bits of its lone solve on the same kernel family.
        Kernel families with resumable launches only (n <= 4, one input, fp64; shared problem object and limits)."""
        x0 = np.ascontiguousarray(np.asarray(current_x, dtype=np.float64))
        if x0.ndim != 2 or x0.shape[1] != self.n:
            raise ValueError("current_x should be (N, %d)" % self.n)
        N = x0.shape[0]
        t0 = np.broadcast_to(np.asarray(current_t, dtype=np.float64), (N,)).copy()
        self._sample_limits_func(t0[:1])
        self._push_state()
        T = int(self._config.horizon_steps)
        u = np.ascontiguousarray(np.asarray(initial_u_list, dtype=np.float64).reshape(N, T, self.mm))
        dp = C.POINTER(C.c_double)
        _capi.check(self._L.nmpc_hip_ddp_solve_stream(self._h, N, t0.ctypes.data_as(dp), x0.ctypes.data_as(dp), u.ctypes.data_as(dp), int(span)))
        self._cache = {}

        def get(field, dtype, shape):
            out = np.zeros(shape, dtype=dtype)
            _capi.check(self._L.nmpc_hip_ddp_stream_get(self._h, field, out.ctypes.data_as(C.c_void_p), out.nbytes))
            return out

        rounds, ms = C.c_int(), C.c_float()
        _capi.check(self._L.nmpc_hip_ddp_last_stream_stats(self._h, C.byref(rounds), C.byref(ms)))
        return StreamResult(X=get(_capi.FIELD_X, np.float64, (N, T + 1, self.n)), U=get(_capi.FIELD_U, np.float64, (N, T, self.mm)),
                            cost=get(_capi.FIELD_COST, np.float64, (N, T + 1)), status=get(_capi.FIELD_STATUS, np.int32, (N,)),
                            iters=get(_capi.FIELD_ITERS, np.int32, (N,)),
                            trace_last=get(_capi.FIELD_TRACE_LAST, np.float64, (N, len(_capi.TRACE_COLUMNS))),
                            dV=get(_capi.FIELD_DV, np.float64, (N, 2)), rounds=rounds.value, device_ms=ms.value)

    # ---- the reference's receding-horizon caller loops, device-resident   ----
    def mpcRun(self, current_t, current_x, initial_u_list, n_ticks: int, shift_warm_start: bool = True,
               max_iter_after_first: int = 0, sim_substeps: int = 0, sim_dt: float = 0.0,
               clamp_u0: bool = True) -> "MpcLog":
        """n_ticks x { solve; advance (t, x, u_list) on the device }.  shift_warm_start=True is the loop of
        TestDDPBipedal.cpp:243-268 / TestDDPVerticalMotion.cpp:290-326 / TestDDPCentroidalMotion.cpp:307-347,
        False the plant loop of TestDDPCartPole.cpp:323-346,388-403 (see nmpc_hip_ddp_mpc_options)."""
        B = self.batch_size
        t0 = np.broadcast_to(np.asarray(current_t, dtype=np.float64), (B,)).copy()
        x0 = np.ascontiguousarray(np.asarray(current_x, dtype=np.float64).reshape(B, self.n))
        # the shift loop starts every tick one timestep later: a limits function is sampled over horizon + n_ticks - 1 steps
        self._sample_limits_func(t0, extra_rows=(int(n_ticks) - 1) if shift_warm_start else 0)
        self._push_state()
        u = self._pack_u(initial_u_list, t0)
        opt = _capi.MpcOptions()
        _capi.check(self._L.nmpc_hip_ddp_mpc_default_options(C.byref(opt)))
        opt.n_ticks = int(n_ticks)
        opt.shift_warm_start = 1 if shift_warm_start else 0
        opt.max_iter_after_first = int(max_iter_after_first)
        opt.sim_substeps = int(sim_substeps)
        opt.sim_dt = float(sim_dt)
        opt.clamp_u0 = 1 if clamp_u0 else 0
        log = MpcLog(t=np.zeros((B, n_ticks)), x=np.zeros((B, n_ticks, self.n)), u0=np.zeros((B, n_ticks, self.mm)),
                     iters=np.zeros((B, n_ticks), np.int32), status=np.zeros((B, n_ticks), np.int32),
                     m0=np.zeros((B, n_ticks), np.int32), x_final=np.zeros((B, self.n)), t_final=np.zeros(B))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _capi.check(self._L.nmpc_hip_ddp_mpc_run(
            self._h, t0.ctypes.data_as(dp), x0.ctypes.data_as(dp), u.ctypes.data_as(dp), C.byref(opt),
            log.t.ctypes.data_as(dp), log.x.ctypes.data_as(dp), log.u0.ctypes.data_as(dp), log.iters.ctypes.data_as(ip),
            log.status.ctypes.data_as(ip), log.m0.ctypes.data_as(ip), log.x_final.ctypes.data_as(dp),
            log.t_final.ctypes.data_as(dp)))
        self._cache = {}
        log.duration = self.computationDuration()
        return log

    def solveDevice(self, d_t0: Optional[int], d_x0: int, d_u_init: int, stream: Optional[int] = None) -> None:
        """Asynchronous solve from DEVICE pointers (integers, e.g. torch.Tensor.data_ptr()) in the reference
        layouts; nothing crosses PCIe.  Call synchronize() before reading results."""
        self._push_state()
        _capi.check(self._L.nmpc_hip_ddp_solve_device(self._h, C.c_void_p(d_t0 or 0), C.c_void_p(d_x0),
                                                      C.c_void_p(d_u_init), C.c_void_p(stream or 0)))
        self._cache = {}

    def synchronize(self) -> None:
        if self._h:  # (a solver that has not solved yet has no handle and nothing in flight: a pool's idle handles)
            _capi.check(self._L.nmpc_hip_ddp_synchronize(self._h))

    # ---- results ----
    def _field(self, field: int, dtype, shape) -> np.ndarray:
        if field in self._cache:
            return self._cache[field]
        out = np.zeros(shape, dtype=dtype)
        _capi.check(self._L.nmpc_hip_ddp_get(self._h, field, out.ctypes.data_as(C.c_void_p), out.nbytes))
        self._cache[field] = out
        return out

    def getDevice(self, field: int, d_out: int, nbytes: int, stream: Optional[int] = None) -> None:
        """Pack one result field (reference layout) into DEVICE memory, e.g. an RCCL send buffer."""
        _capi.check(self._L.nmpc_hip_ddp_get_device(self._h, field, C.c_void_p(d_out), nbytes,
                                                    C.c_void_p(stream or 0)))

    def X(self) -> np.ndarray:
        return self._field(_capi.FIELD_X, np.float64, (self.batch_size, self._h_T + 1, self.n))

    def U(self) -> np.ndarray:
        return self._field(_capi.FIELD_U, np.float64, (self.batch_size, self._h_T, self.mm))

    def cost(self) -> np.ndarray:
        return self._field(_capi.FIELD_COST, np.float64, (self.batch_size, self._h_T + 1))

    def kff(self) -> np.ndarray:
        return self._field(_capi.FIELD_KFF, np.float64, (self.batch_size, self._h_T, self.mm))

    def Kfb(self) -> np.ndarray:
        """(B, T, MM, n): Kfb()[b, t] is the m x n feedback gain of step t."""
        raw = self._field(_capi.FIELD_KFB, np.float64, (self.batch_size, self._h_T, self.n, self.mm))
        return raw.transpose(0, 1, 3, 2)

    def status(self) -> np.ndarray:
        return self._field(_capi.FIELD_STATUS, np.int32, (self.batch_size,))

    def iters(self) -> np.ndarray:
        return self._field(_capi.FIELD_ITERS, np.int32, (self.batch_size,))

    def trace(self) -> np.ndarray:
        """(B, max_iter+1, 12); rows beyond iters[b] are zero.  Needs config().trace_level >= 1."""
        return self._field(_capi.FIELD_TRACE, np.float64,
                           (self.batch_size, int(self._config.max_iter) + 1, _capi.NTRACE))

    def traceLast(self) -> np.ndarray:
        return self._field(_capi.FIELD_TRACE_LAST, np.float64, (self.batch_size, _capi.NTRACE))

    def dV(self) -> np.ndarray:
        return self._field(_capi.FIELD_DV, np.float64, (self.batch_size, 2))

    def qpRetval(self) -> np.ndarray:
        return self._field(_capi.FIELD_QP_RETVAL, np.int32, (self.batch_size, self._h_T))

    def qpFreeMask(self) -> np.ndarray:
        return self._field(_capi.FIELD_QP_FREE_MASK, np.uint32, (self.batch_size, self._h_T))

    def inputDimList(self) -> np.ndarray:
        return self._field(_capi.FIELD_INPUT_DIM, np.int32, (self.batch_size, self._h_T))

    # ---- DDPSolver::controlData()  (DDPSolver.h:288-291) ----
    def controlData(self, b: int) -> ControlData:
        dims = self.inputDimList()[b]
        U = self.U()[b]
        return ControlData(self.X()[b], [U[i, : dims[i]].copy() for i in range(self._h_T)], self.cost()[b])

    # ---- DDPSolver::traceDataList()  (DDPSolver.h:294-297) ----
    def traceDataList(self, b: int) -> List[TraceData]:
        tr = self.trace()[b]
        n = int(self.iters()[b]) + 1
        out = []
        for r in tr[:n]:
            out.append(TraceData(iter=int(r[0]), cost=r[1], lambda_=r[2], dlambda=r[3], alpha=r[4], k_rel_norm=r[5],
                                 cost_update_actual=r[6], cost_update_expected=r[7], cost_update_ratio=r[8],
                                 alpha_idx=int(r[9]), n_backward=int(r[10]), n_forward=int(r[11])))
        return out

    # ---- DDPSolver::computationDuration()  (DDPSolver.h:300-303) ----
    def computationDuration(self) -> ComputationDuration:
        tot, ker = C.c_float(), C.c_float()
        _capi.check(self._L.nmpc_hip_ddp_last_solve_ms(self._h, C.byref(tot), C.byref(ker)))
        bw, fw, other = C.c_double(), C.c_double(), C.c_double()
        _capi.check(self._L.nmpc_hip_ddp_last_solve_phases(self._h, C.byref(bw), C.byref(fw), C.byref(other)))
        return ComputationDuration(solve=tot.value, setup=tot.value - ker.value, opt=ker.value, backward=bw.value,
                                   forward=fw.value)

    def kernelName(self) -> str:
        """gfx950 kernel the next solve launches (lane mapping), as rocprofv3 lists it."""
        self._push_state()  # the handle is created lazily (horizon_steps may still change before the first solve)
        name = C.c_char_p()
        _capi.check(self._L.nmpc_hip_ddp_kernel_name(self._h, C.byref(name)))
        return name.value.decode()

    def setKernel(self, name: str = "auto") -> None:
        """Pin the kernel family ("auto", "1w", "2w", "quad", "wpi", "tile64", "tile32"): nmpc_hip_ddp_set_kernel."""
        self._kernel = name
        if self._h:
            _capi.check(self._L.nmpc_hip_ddp_set_kernel(self._h, name.encode()))

    def setDispatchBatch(self, batch: int = 0) -> None:
        """The batch size the kernel family is chosen for (a shard of a larger solve: the whole batch's size)."""
        self._dispatch_batch = int(batch)
        if self._h:
            _capi.check(self._L.nmpc_hip_ddp_set_dispatch_batch(self._h, int(batch)))

    def kernelNameForBatch(self, batch: int) -> str:
        self._push_state()
        name = C.c_char_p()
        _capi.check(self._L.nmpc_hip_ddp_kernel_name_for_batch(self._h, int(batch), C.byref(name)))
        return name.value.decode()

    def lastSolveLaunches(self) -> int:
        """Kernel launches the last solve was cut into (1: one whole-solve launch; more: the ragged-convergence schedule)."""
        n = C.c_int()
        _capi.check(self._L.nmpc_hip_ddp_last_solve_launches(self._h, C.byref(n)))
        return n.value

    def timingStats(self, reset: bool = False):
        """(number of device solves, sum of ingest+kernel ms, sum of solve-kernel ms) since the last reset."""
        n, tot, ker = C.c_longlong(), C.c_double(), C.c_double()
        _capi.check(self._L.nmpc_hip_ddp_timing_stats(self._h, 1 if reset else 0, C.byref(n), C.byref(tot),
                                                      C.byref(ker)))
        return n.value, tot.value, ker.value

    # ---- DDPSolver::dumpTraceDataList  (DDPSolver.hpp:562-598): same 12 columns, same separator ----
    def dumpTraceDataList(self, b: int, file_path: str) -> None:
        with open(file_path, "w") as f:
            f.write("iter cost lambda dlambda alpha k_rel_norm cost_update_actual cost_update_expected "
                    "cost_update_ratio duration_derivative duration_backward duration_forward\n")
            for t in self.traceDataList(b):
                f.write(f"{t.iter} {t.cost:g} {t.lambda_:g} {t.dlambda:g} {t.alpha:g} {t.k_rel_norm:g} "
                        f"{t.cost_update_actual:g} {t.cost_update_expected:g} {t.cost_update_ratio:g} "
                        f"{t.duration_derivative:g} {t.duration_backward:g} {t.duration_forward:g}\n")


def request_hw_queues(n: int = 16) -> bool:
    """Ask the HIP runtime for at least n hardware queues (nmpc_hip_ddp_request_hw_queues: GPU_MAX_HW_QUEUES, read when the runtime
    initialises; streams that share a queue do not overlap).  True when the request can still take effect or the runtime already
    has that many; False when the runtime of this process is up with fewer — call it (or construct the pool) before anything touches
    the device, or export GPU_MAX_HW_QUEUES yourself.  Importing nmpc_amd does not change the environment."""
    ok = C.c_int(0)
    _capi.check(_capi.load().nmpc_hip_ddp_request_hw_queues(int(n), C.byref(ok)))
    return bool(ok.value)


class DDPSolverPool:
    """Several DDPSolverBatch handles of the same problem and batch size, each with its own stream: consecutive batches are
    queued round-robin and overlap on the device.  A batch that is solved to convergence ends with a tail — a few instances
    that run for hundreds of iterations on a few CUs (every DDPSolver object of the reference runs its own loop to the end,
    DDPSolver.hpp:115-123); with the next batches already queued on other streams their workgroups take the CUs the finished
    instances have vacated.  Results of a batch are those of the handle it ran on (bit-identical to a lone handle)."""

    def __init__(self, problem: _Problem, batch_size: int, n_handles: int = 4, device: int = 0):
        if n_handles < 1:
            raise ValueError("n_handles should be positive")
        #: whether the runtime has (or will have) one hardware queue per handle; False: fewer batches overlap than there are handles
        self.hw_queues_ok = request_hw_queues(min(max(n_handles, 4), 64))
        if not self.hw_queues_ok:
            import warnings
            warnings.warn("DDPSolverPool: the HIP runtime is already initialised with GPU_MAX_HW_QUEUES=%s (< %d handles): streams that "
                          "share a hardware queue do not overlap; call nmpc_amd.request_hw_queues() or export the variable before the first "
                          "HIP call of the process" % (__import__("os").environ.get("GPU_MAX_HW_QUEUES", "unset (4)"), n_handles))
        self.solvers = [DDPSolverBatch(problem, batch_size, device=device) for _ in range(n_handles)]
        for s in self.solvers:
            # Configuration::ragged_schedule 0 (automatic) is off for a lone handle's solve() / solveDevice() — max_iter is only a cap,
            # and a lone stream gains nothing from the schedule — and on for a pool: the freed CUs go to the batches queued behind
            s._pooled = True
        self._next = 0

    def config(self) -> Configuration:
        """DDPSolver::config() of every handle (the first one's object: call applyConfig() after changing it)."""
        return self.solvers[0].config()

    def applyConfig(self) -> None:
        c0 = self.solvers[0].config()
        for s in self.solvers[1:]:
            c = s.config()
            for key, val in vars(c0).items():
                setattr(c, key, val.copy() if isinstance(val, np.ndarray) else val)

    def submit(self, d_t0: Optional[int], d_x0: int, d_u_init: int) -> DDPSolverBatch:
        """Queue one batch (DEVICE pointers, reference layouts) on the next handle's stream; returns that handle.  Submitting
        again to the same handle (after n_handles further batches) overwrites its results: read them first."""
        s = self.solvers[self._next]
        self._next = (self._next + 1) % len(self.solvers)
        s.solveDevice(d_t0, d_x0, d_u_init)
        return s

    def synchronize(self) -> None:
        for s in self.solvers:
            s.synchronize()
