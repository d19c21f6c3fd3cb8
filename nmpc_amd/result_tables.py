"""The per-test result tables of the reference's closed-loop tests, written from an MpcLog so that the reference's plot scripts
(nmpc_ddp/tests/scripts/plotTestDDP*.py: np.genfromtxt(path, names=True)) read batched GPU runs unchanged.

A table is a list of (column name, function of one tick) pairs; `MpcLog.dump(path, b, columns)` writes one line per tick of
instance b under the space-separated header.  The makers below reproduce the four headers of the reference:
    TestDDPBipedal.cpp:242,259-262            time com_pos com_vel planned_zmp ref_zmp omega^2 iter
    TestDDPVerticalMotion.cpp:289,307-312     time pos vel force ref_pos num_contact iter
    TestDDPCentroidalMotion.cpp:302-305,326-  time pos_x .. angular_momentum_z force_x force_y force_z ref_pos_x .. iter duration_*
    TestDDPCartPole.cpp:316,339-340           time pos theta vel omega force ref_pos disturbance
The reference schedules (ref_zmp, omega^2, ref_pos, the stance's ridges) are test-harness functions of t there
(std::function members); here the caller passes them as Python callables — no model math lives in this package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple

import numpy as np


@dataclass
class Tick:
    """One row of an MpcLog for one instance: what the reference's loops have at hand when they dump a line."""
    t: float
    x: np.ndarray  # state handed to the solve of the tick (controlData().x_list[0])
    u0: np.ndarray  # first input of the solution, padded to MM
    m0: int  # its dimension
    iter: int  # traceDataList().back().iter
    duration: "object"  # ComputationDuration of the run (batch-wide, the same in every row)


Column = Tuple[str, Callable[[Tick], float]]


def bipedal_table(ref_zmp_func: Callable[[float], float], omega2_func: Callable[[float], float]) -> List[Column]:
    return [("time", lambda k: k.t), ("com_pos", lambda k: k.x[0]), ("com_vel", lambda k: k.x[1]),
            ("planned_zmp", lambda k: k.u0[0]), ("ref_zmp", lambda k: ref_zmp_func(k.t)),
            ("omega^2", lambda k: omega2_func(k.t)), ("iter", lambda k: k.iter)]


def vertical_motion_table(ref_pos_func: Callable[[float], float]) -> List[Column]:
    return [("time", lambda k: k.t), ("pos", lambda k: k.x[0]), ("vel", lambda k: k.x[1]),
            ("force", lambda k: float(np.sum(k.u0[: k.m0]))), ("ref_pos", lambda k: ref_pos_func(k.t)),
            ("num_contact", lambda k: k.m0), ("iter", lambda k: k.iter)]


def centroidal_motion_table(ridges_func: Callable[[float], np.ndarray], ref_pos_func: Callable[[float], Sequence[float]]) -> List[Column]:
    """ridges_func(t): the stance's 3 x m ridge matrix (ref_stance_func(t).ridges_mat)."""
    cols: List[Column] = [("time", lambda k: k.t)]
    for i, name in enumerate(("pos_x", "pos_y", "pos_z", "linear_momentum_x", "linear_momentum_y", "linear_momentum_z",
                              "angular_momentum_x", "angular_momentum_y", "angular_momentum_z")):
        cols.append((name, (lambda i: lambda k: k.x[i])(i)))

    def force(k, c):
        r = np.asarray(ridges_func(k.t), dtype=np.float64).reshape(3, -1)
        return float(r[c, : k.m0] @ k.u0[: k.m0]) if k.m0 > 0 else 0.0

    for c, name in enumerate(("force_x", "force_y", "force_z")):
        cols.append((name, (lambda c: lambda k: force(k, c))(c)))
    for c, name in enumerate(("ref_pos_x", "ref_pos_y", "ref_pos_z")):
        cols.append((name, (lambda c: lambda k: ref_pos_func(k.t)[c])(c)))
    cols.append(("iter", lambda k: k.iter))
    for f in ("setup", "opt", "derivative", "backward", "forward", "Q", "reg", "gain"):
        cols.append(("duration_" + f, (lambda f: lambda k: getattr(k.duration, f, 0.0))(f)))
    return cols


def cart_pole_table(ref_pos_func: Callable[[float], float]) -> List[Column]:
    """One line per MPC tick (the reference writes one per simulation step): the plant state at the tick, the clamped input."""
    return [("time", lambda k: k.t), ("pos", lambda k: k.x[0]), ("theta", lambda k: k.x[1]), ("vel", lambda k: k.x[2]),
            ("omega", lambda k: k.x[3]), ("force", lambda k: k.u0[0]), ("ref_pos", lambda k: ref_pos_func(k.t)),
            ("disturbance", lambda k: 0.0)]


def fmpc_oscillator_table() -> List[Column]:
    """TestFmpcOscillator.cpp:168,186-188: time x[0] x[1] u[0] mpc_iter computation_time kkt_error (from an FMPC mpcRun log:
    Tick.duration carries (solve ms of the run, kkt_error of the tick))."""
    return [("time", lambda k: k.t), ("x[0]", lambda k: k.x[0]), ("x[1]", lambda k: k.x[1]), ("u[0]", lambda k: k.u0[0]),
            ("mpc_iter", lambda k: k.iter), ("computation_time", lambda k: k.duration[0]), ("kkt_error", lambda k: k.duration[1])]


def fmpc_cart_pole_table(ref_pos_func: Callable[[float], float]) -> List[Column]:
    """TestFmpcCartPole.cpp:337,364-365: time pos theta vel omega force ref_pos disturbance (one line per MPC tick)."""
    return cart_pole_table(ref_pos_func)


def fmpc_ticks(log: dict, instance: int, t0: float, tick_dt: float, solve_ms: float = 0.0) -> List[Tick]:
    """Rows of instance `instance` of an nmpc_amd.fmpc.FmpcSolverBatch.mpcRun log (x, u0, iters, kkt_error per tick)."""
    n_ticks = log["x"].shape[1]
    return [Tick(t=t0 + k * tick_dt, x=log["x"][instance, k], u0=log["u0"][instance, k], m0=log["u0"].shape[2],
                 iter=int(log["iters"][instance, k]), duration=(solve_ms, float(log["kkt_error"][instance, k])))
            for k in range(n_ticks)]


def write_table(file_path: str, ticks: Sequence[Tick], columns: Sequence[Column]) -> None:
    with open(file_path, "w") as f:
        f.write(" ".join(name for name, _ in columns) + "\n")
        for k in ticks:
            f.write(" ".join("%g" % fn(k) if not isinstance(fn(k), (int, np.integer)) else "%d" % fn(k) for _, fn in columns) + "\n")
