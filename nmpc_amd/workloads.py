"""Synthetic problem batches of BASELINE.json's configs (SURVEY.md §8 d), shared by bench.py and tests/.

All randomness is splitmix64 -> U[0,1) so that any host language regenerates identical bits from the seed:
    state += 0x9E3779B97F4A7C15; z = state; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;
    z = (z ^ z>>27) * 0x94D049BB133111EB; z ^= z>>31; u = (z >> 11) * 2^-53
Draw order: instance-major, i.e. all numbers of instance 0, then instance 1, ...
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_MASK = (1 << 64) - 1


def splitmix64_uniform(seed: int, count: int) -> np.ndarray:
    """`count` doubles in [0, 1) from splitmix64 seeded with `seed`."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class Workload:
    name: str
    model: str  # registry name
    n: int
    m: int
    T: int
    B: int
    dt: float
    x0: np.ndarray  # (B, n)
    u_init: np.ndarray  # (B, T, max(m,1))
    t0: np.ndarray  # (B,)
    params: dict = field(default_factory=dict)  # overrides of the problem's default parameters
    limits: Optional[tuple] = None  # (lower, upper) or None


def cartpole_batch(B: int = 4096, T: int = 100, seed: int = 1234, constrained: bool = False, fp32: bool = False) -> Workload:
    """C2: x0 ~ U([-1,1] x [-pi,pi] x [-1,1] x [-1,1]), u_init = 0, class-default weights
    (TestDDPCartPole.cpp:44-46), dt = 0.01; optional +-15 N box (TestDDPCartPole.cpp:379-386).  fp32: the problem type
    instantiated in float ("cartpole_f32": the fp32 tile kernel's n = 4, m = 1 shape)."""
    u = splitmix64_uniform(seed, 4 * B).reshape(B, 4)
    lo = np.array([-1.0, -np.pi, -1.0, -1.0])
    hi = np.array([1.0, np.pi, 1.0, 1.0])
    x0 = lo + (hi - lo) * u
    return Workload("cartpole_batch", "cartpole_f32" if fp32 else "cartpole", 4, 1, T, B, 0.01, x0, np.zeros((B, T, 1)), np.zeros(B),
                    limits=(np.array([-15.0]), np.array([15.0])) if constrained else None)


def cartpole_single(T: int = 100) -> Workload:
    """C1: the reference's swing-up start x0 = (0, pi, 0, 0) (TestDDPCartPole.cpp:306)."""
    return Workload("cartpole_single", "cartpole", 4, 1, T, 1, 0.01, np.array([[0.0, np.pi, 0.0, 0.0]]),
                    np.zeros((1, T, 1)), np.zeros(1))


def bipedal_batch(B: int = 1024, T: int = 300, seed: int = 1234) -> Workload:
    """C3: per-instance start time t0 ~ U[0,17] s on the reference's ref_zmp / omega^2 schedule
    (TestDDPBipedal.cpp:171-225), x0 ~ (U[-0.05,0.05], U[-0.1,0.1]), u_init = 0."""
    u = splitmix64_uniform(seed, 3 * B).reshape(B, 3)
    t0 = 17.0 * u[:, 0]
    x0 = np.stack([-0.05 + 0.1 * u[:, 1], -0.1 + 0.2 * u[:, 2]], axis=1)
    return Workload("bipedal_batch", "bipedal", 2, 1, T, B, 0.01, x0, np.zeros((B, T, 1)), t0)


def vertical_batch(B: int = 256, T: int = 300, seed: int = 1234, constrained: bool = True) -> Workload:
    """Variable input dimension (TestDDPVerticalMotion.cpp): t0 ~ U[0,7] s so horizons straddle the nu changes
    at t = 2, 3, 4.5, 5; x0 ~ (U[0.8,1.4], U[-0.5,0.5]); u_init = 0; optional [0, 30] N box (:262-270)."""
    u = splitmix64_uniform(seed, 3 * B).reshape(B, 3)
    t0 = 7.0 * u[:, 0]
    x0 = np.stack([0.8 + 0.6 * u[:, 1], -0.5 + 1.0 * u[:, 2]], axis=1)
    return Workload("vertical_batch", "vertical", 2, 2, T, B, 0.01, x0, np.zeros((B, T, 2)), t0,
                    limits=(np.array([0.0, 0.0]), np.array([30.0, 30.0])) if constrained else None)


def centroidal_batch(B: int = 64, T: int = 100, seed: int = 1234) -> Workload:
    """Centroidal motion (TestDDPCentroidalMotion.cpp:239-365): t0 ~ U[0,1] s, CoM perturbed by +-2 cm,
    u_init = 0."""
    u = splitmix64_uniform(seed, 4 * B).reshape(B, 4)
    t0 = 1.0 * u[:, 0]
    x0 = np.zeros((B, 9))
    x0[:, 0] = -0.02 + 0.04 * u[:, 1]
    x0[:, 1] = -0.02 + 0.04 * u[:, 2]
    x0[:, 2] = 1.0 - 0.02 + 0.04 * u[:, 3]
    return Workload("centroidal_batch", "centroidal", 9, 16, T, B, 0.03, x0, np.zeros((B, T, 16)), t0)


def _normal_from_uniform(u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def quadrotor_batch(B: int = 8192, T: int = 50, seed: int = 1234, constrained: bool = False,
                    fp32: bool = False) -> Workload:
    """C4: hover perturbation x0 ~ N(0, 0.3^2) around (0,0,1) (Box-Muller on the splitmix stream), u_init = hover
    thrust m g / 4, dt = 0.02.  fp32: the problem type instantiated in float ("quadrotor_f32"), what BASELINE.json names."""
    u = splitmix64_uniform(seed, 24 * B).reshape(B, 24)
    x0 = 0.3 * _normal_from_uniform(u[:, :12], u[:, 12:])
    x0[:, 2] += 1.0
    hover = 1.0 * 9.80665 / 4
    limits = (np.full(4, 0.7 * hover), np.full(4, 1.3 * hover)) if constrained else None  # rotor thrust box
    return Workload("quadrotor_batch", "quadrotor_f32" if fp32 else "quadrotor", 12, 4, T, B, 0.02, x0,
                    np.full((B, T, 4), hover), np.zeros(B), limits=limits)


def planar_vtol_batch(B: int = 4096, T: int = 60, seed: int = 1234, constrained: bool = False) -> Workload:
    """Builder-defined n = 6, m = 2 shape (planar VTOL): hover perturbation x0 ~ N(0, 0.3^2) around (0, 1), attitude
    ~ N(0, 0.2^2), u_init = hover thrust m g / 2, dt = 0.02."""
    u = splitmix64_uniform(seed, 12 * B).reshape(B, 12)
    x0 = 0.3 * _normal_from_uniform(u[:, :6], u[:, 6:])
    x0[:, 2] *= 2.0 / 3.0
    x0[:, 1] += 1.0
    hover = 1.0 * 9.80665 / 2
    limits = (np.full(2, 0.6 * hover), np.full(2, 1.4 * hover)) if constrained else None  # rotor thrust box
    return Workload("planar_vtol_batch", "planar_vtol", 6, 2, T, B, 0.02, x0, np.full((B, T, 2), hover), np.zeros(B), limits=limits)


def manipulator_batch(B: int = 8192, T: int = 30, seed: int = 1234, constrained: bool = False, fp32: bool = False) -> Workload:
    """C5 (per-GPU shard): q0 ~ U[-1,1]^7, qd0 ~ U[-0.5,0.5]^7, u_init = gravity compensation at q0, dt = 0.01.  fp32: the
    problem type instantiated in float ("manipulator_f32")."""
    u = splitmix64_uniform(seed, 14 * B).reshape(B, 14)
    q0 = -1.0 + 2.0 * u[:, :7]
    qd0 = -0.5 + 1.0 * u[:, 7:]
    x0 = np.concatenate([q0, qd0], axis=1)
    grav = 4.0 * (7 - np.arange(7)) / 7.0
    u_gc = grav[None, :] * np.sin(np.cumsum(q0, axis=1))
    limits = (np.full(7, -3.0), np.full(7, 3.0)) if constrained else None  # joint torque box
    return Workload("manipulator_batch", "manipulator_f32" if fp32 else "manipulator", 14, 7, T, B, 0.01, x0,
                    np.repeat(u_gc[:, None, :], T, axis=1).copy(), np.zeros(B), limits=limits)


def algorithmic_words_per_instance_iteration(n: int, m: int, T: int, n_bw: float = 1.0, n_fw: float = 1.0) -> float:
    """words of the reference's materialised dataflow per instance-iteration."""
    D = 2 * n * n + 2 * n * m + n + m + m * m
    lin = T * ((n + m) + D) + (2 * n + n * n)
    bw = T * (D + m + m * n) + (n + n * n)
    fw = T * ((n + 2 * m + m * n) + (n + m + 1)) + (2 * n + 1)
    return lin + n_bw * bw + n_fw * fw


def fused_words_per_instance_iteration(n: int, m: int, T: int, n_bw: float = 1.0, n_fw: float = 1.0) -> float:
    """Words the fused TPI kernel actually has to move: derivatives never leave the registers."""
    bw = T * ((n + m) + m + m * n) + n
    fw = T * ((n + 2 * m + m * n) + (n + m + 1)) + (2 * n + 1)
    return n_bw * bw + n_fw * fw
