"""Pins the CPU oracle (oracle/, the restatement of the reference's Eigen path) to everything the reference's
own tests hold for the DDP hot path (SURVEY.md §8 c):

  * the BoxQP known-answer vectors              nmpc_ddp/tests/src/TestBoxQP.cpp:35-98
  * the finite-difference Jacobian checks        TestDDPCartPole.cpp:609-649, TestDDPCentroidalMotion.cpp:367-411
  * every EXPECT_LT of the closed-loop MPC tests TestDDPBipedal.cpp:162-279, TestDDPVerticalMotion.cpp:236-347,
                                                 TestDDPCentroidalMotion.cpp:239-365, TestDDPCartPole.cpp:336,351-354
  * an independent NumPy/SciPy restatement       oracle/ddp_numpy.py
"""
import numpy as np
import pytest

import oracle
from oracle import ddp_numpy as dn

# ---------------------------------------------------------------------------------------------------
# BoxQP known answers (TestBoxQP.cpp:35-98; problems from qpOASES example1b).  The reference runs the five
# problems once with fixed-size and once with dynamic-size matrices; the oracle has one code path, so the second
# sweep here passes an explicit zero initial guess (the overload BoxQP.h:126-132 forwards to).
# ---------------------------------------------------------------------------------------------------
H_QP = np.array([[1.0, 0.0], [0.0, 0.5]])
QP_CASES = [
    ((1.5, 1.0), (-10, -10), (10, 10), (-1.5, -2.0)),
    ((1.5, 1.0), (0.5, -2.0), (5.0, 2.0), (0.5, -2.0)),
    ((1.0, 1.5), (0.0, -1.0), (5.0, -0.5), (0.0, -1.0)),
    ((1.5, 1.0), (-5.0, -1.0), (-2.0, 2.0), (-2.0, -1.0)),
    ((1.0, 1.5), (-5.0, -10.0), (-2.0, 10.0), (-2.0, -3.0)),
]


@pytest.mark.parametrize("explicit_x0", [False, True])
@pytest.mark.parametrize("g,lower,upper,x_gt", QP_CASES)
def test_boxqp_known_answers(g, lower, upper, x_gt, explicit_x0):
    r = oracle.boxqp_solve(H_QP, g, lower, upper, np.zeros(2) if explicit_x0 else None)
    assert np.linalg.norm(r.x - np.array(x_gt)) < 1e-6  # TestBoxQP.cpp:29
    assert r.retval > 0


@pytest.mark.parametrize("g,lower,upper,x_gt", QP_CASES)
def test_boxqp_numpy_restatement_agrees(g, lower, upper, x_gt):
    a = oracle.boxqp_solve(H_QP, g, lower, upper)
    b = dn.boxqp(H_QP, g, lower, upper)
    assert a.retval == b.retval and list(a.free_idxs) == list(b.free_idxs) and a.iter == b.iters
    np.testing.assert_allclose(a.x, b.x, rtol=0, atol=1e-14)


def test_boxqp_random_spd_against_numpy():
    rng = np.random.default_rng(7)
    for m in (1, 2, 3, 5, 8):
        for _ in range(20):
            A = rng.normal(size=(m, m))
            H = A @ A.T + 0.5 * np.eye(m)
            g = rng.normal(size=m) * 3
            lo = -rng.uniform(0.1, 2.0, size=m)
            up = rng.uniform(0.1, 2.0, size=m)
            x0 = rng.uniform(-3, 3, size=m)
            a = oracle.boxqp_solve(H, g, lo, up, x0)
            b = dn.boxqp(H, g, lo, up, x0)
            assert a.retval == b.retval
            assert list(a.free_idxs) == list(b.free_idxs)
            np.testing.assert_allclose(a.x, b.x, rtol=1e-10, atol=1e-12)
            assert np.all(a.x >= lo) and np.all(a.x <= up)


# ---------------------------------------------------------------------------------------------------
# Jacobian checks, same method and tolerance as the reference (central differences, eps = 1e-6, Frobenius < 1e-6)
# ---------------------------------------------------------------------------------------------------
def fd_jacobians(model, params, t, x, u, eps=1e-6):
    n, m = x.size, u.size
    Fx = np.zeros((n, n))
    Fu = np.zeros((n, m))
    for i in range(n):
        e = np.zeros(n)
        e[i] = eps
        Fx[:, i] = (oracle.model_eval(model, params, t, x + e, u).xn - oracle.model_eval(model, params, t, x - e, u).xn) / (2 * eps)
    for i in range(m):
        e = np.zeros(m)
        e[i] = eps
        Fu[:, i] = (oracle.model_eval(model, params, t, x, u + e).xn - oracle.model_eval(model, params, t, x, u - e).xn) / (2 * eps)
    return Fx, Fu


def fd_cost(model, params, t, x, u, eps=1e-5):
    n, m = x.size, u.size
    z0 = np.concatenate([x, u])

    def L(z):
        return oracle.model_eval(model, params, t, z[:n], z[n:]).running_cost

    g = np.zeros(n + m)
    Hh = np.zeros((n + m, n + m))
    for i in range(n + m):
        e = np.zeros(n + m)
        e[i] = eps
        g[i] = (L(z0 + e) - L(z0 - e)) / (2 * eps)
        for j in range(n + m):
            f = np.zeros(n + m)
            f[j] = eps
            Hh[i, j] = (L(z0 + e + f) - L(z0 + e - f) - L(z0 - e + f) + L(z0 - e - f)) / (4 * eps * eps)
    return g, Hh


def test_cartpole_check_derivative():
    # TestDDPCartPole.cpp:609-649: x = (1, -2, 3, -4), u = 10, dt = 0.01
    x = np.array([1.0, -2.0, 3.0, -4.0])
    u = np.array([10.0])
    ev = oracle.model_eval("cartpole", None, 0.0, x, u)
    Fx, Fu = fd_jacobians("cartpole", None, 0.0, x, u)
    assert np.linalg.norm(ev.Fx - Fx) < 1e-6
    assert np.linalg.norm(ev.Fu - Fu) < 1e-6


def test_centroidal_check_derivative():
    # TestDDPCentroidalMotion.cpp:367-411: constant stance, dt = 0.01, random x, u in [-1, 1]
    rng = np.random.default_rng(3)
    p = oracle.default_params("centroidal", dt=0.01, flight_t0=1e9, flight_t1=2e9, ref_switch_t=1e9)
    for _ in range(5):
        x = rng.uniform(-1, 1, 9)
        u = rng.uniform(-1, 1, 16)
        ev = oracle.model_eval("centroidal", p, 0.0, x, u)
        assert ev.m == 16
        Fx, Fu = fd_jacobians("centroidal", p, 0.0, x, u)
        assert np.linalg.norm(ev.Fx - Fx) < 1e-6
        assert np.linalg.norm(ev.Fu - Fu) < 1e-6


@pytest.mark.parametrize("model", ["cartpole", "bipedal", "vertical", "centroidal", "quadrotor", "manipulator", "planar_vtol"])
def test_all_models_jacobians_and_cost_derivatives(model):
    """Same check for every model the build ships (the builder-defined quadrotor / manipulator have no reference
    counterpart: this is what pins their analytic derivatives), plus the cost gradient / Hessian blocks."""
    rng = np.random.default_rng(11)
    n, mmax, _ = oracle.model_dims(model)
    t = {"vertical": 2.5, "bipedal": 7.4}.get(model, 0.3)  # vertical: two inputs; bipedal: inside the omega^2 ramp
    for _ in range(3):
        x = rng.uniform(-0.7, 0.7, n)
        m = int(oracle.input_dims(model, None, t, 1)[0])
        u = rng.uniform(-1, 1, m) * (5.0 if model in ("cartpole", "quadrotor", "planar_vtol") else 1.0)
        ev = oracle.model_eval(model, None, t, x, u)
        Fx, Fu = fd_jacobians(model, None, t, x, u)
        assert np.linalg.norm(ev.Fx - Fx) < 1e-6
        assert np.linalg.norm(ev.Fu - Fu) < 1e-6
        g, Hh = fd_cost(model, None, t, x, u)
        scale = 1.0 + np.abs(Hh).max()
        np.testing.assert_allclose(np.concatenate([ev.Lx, ev.Lu]), g, rtol=1e-6, atol=1e-7 * scale)
        np.testing.assert_allclose(ev.Lxx, Hh[:n, :n], rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(ev.Luu, Hh[n:, n:], rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(ev.Lxu, Hh[:n, n:], rtol=1e-4, atol=2e-5 * scale)


# ---------------------------------------------------------------------------------------------------
# closed-loop MPC tests of the reference, ROS-free, with every EXPECT_LT
# ---------------------------------------------------------------------------------------------------
def test_mpc_bipedal():
    """TestDDPBipedal.cpp:162-279: dt 0.01, horizon 3 s (T = 300), 20 s, default max_iter."""
    cfg = oracle.default_config(horizon_steps=300)
    n_ticks = 2000
    r = oracle.mpc_run("bipedal", cfg, [0.0, 0.0], n_ticks, shift_warm_start=True)
    m = oracle.lib()  # noqa: F841
    ref = np.array([_bipedal_ref_zmp(t) for t in r.t])
    assert np.all(np.abs(r.u0[:, 0] - ref) < 1e-2)  # :254 planned ZMP tracks the reference
    assert abs(r.x_final[0] - _bipedal_ref_zmp(r.t_final)) < 1e-2  # :272
    assert abs(r.x_final[1]) < 1e-2  # :273


def _bipedal_ref_zmp(t, end_t=20.0):
    t += 1e-6
    if t <= 1.5 or t >= end_t - 1.5:
        return 0.0
    return 0.15 if int(np.floor((t - 1.0) / 1.0)) % 2 == 0 else -0.15


@pytest.mark.parametrize("with_constraint", [True, False])
def test_mpc_vertical_motion(with_constraint):
    """TestDDPVerticalMotion.cpp:236-347: T = 300, initial_lambda 1e-6, max_iter 3 from the second tick, limits
    [0, 30] N per contact, input dimension 1 / 2 / 0 depending on t."""
    cfg = oracle.default_config(horizon_steps=300, initial_lambda=1e-6, with_input_constraint=int(with_constraint))
    n_ticks = 1000
    r = oracle.mpc_run("vertical", cfg, [1.2, 0.0], n_ticks, max_iter_after_first=3, shift_warm_start=True,
                       lower=[0.0, 0.0], upper=[30.0, 30.0])
    ref = np.where(r.t + 1e-6 < 8.0, 1.0, 0.0)
    assert np.all(np.abs(r.x[:, 0] - ref) < 1.0)  # :305
    assert abs(r.x_final[0] - 0.0) < 1e-2  # :334
    assert abs(r.x_final[1]) < 1e-2  # :335
    assert set(np.unique(r.m0)) == {0, 1, 2}  # the input dimension really changes along the run
    if with_constraint:
        assert r.u0.min() >= -1e-9 and r.u0.max() <= 30.0 + 1e-9


def test_mpc_centroidal_motion():
    """TestDDPCentroidalMotion.cpp:239-365: dt 0.03, T = 100, 3 s (100 ticks), max_iter 3 after the first solve."""
    cfg = oracle.default_config(horizon_steps=100)
    x0 = np.array([0, 0, 1.0, 0, 0, 0, 0, 0, 0])
    r = oracle.mpc_run("centroidal", cfg, x0, 100, max_iter_after_first=3, shift_warm_start=True)
    ref = np.array([[0.0 if t + 1e-6 < 1.5 else 0.5, 0.0, 1.0] for t in r.t])
    assert np.all(np.linalg.norm(r.x[:, :3] - ref, axis=1) < 1.0)  # :320
    ref_end = np.array([0.5, 0.0, 1.0])
    assert np.linalg.norm(r.x_final[:3] - ref_end) < 1e-2  # :350
    assert np.linalg.norm(r.x_final[3:]) < 1.0  # :351
    assert set(np.unique(r.m0)) == {0, 16}


def test_mpc_cartpole_swing_up():
    """Deterministic restatement of TestDDPCartPole.cpp:236-403 with the launch-file parameters
    (tests/test/TestDDPCartPole.test:14-26): T = 200, +-15 N box, max_iter 3, MPC every 4 ms, plant stepped at
    2 ms, 10 s; running_u weight 0.01."""
    cfg = oracle.default_config(horizon_steps=200, max_iter=3, with_input_constraint=1)
    p = oracle.default_params("cartpole", running_u=0.01)
    r = oracle.mpc_run("cartpole", cfg, [0.0, np.pi, 0.0, 0.0], 2500, params=p, shift_warm_start=False,
                       sim_substeps=2, sim_dt=0.002, lower=[-15.0], upper=[15.0])
    assert np.all(np.abs(r.x[:, 0]) < 1e2)  # :336
    assert abs(r.x_final[0]) < 1.0  # :351
    assert abs(r.x_final[1]) < 1e-1  # :352  (pole upright)
    assert abs(r.x_final[2]) < 1.0  # :353
    assert abs(r.x_final[3]) < 1e-1  # :354
    assert np.all(np.abs(r.u0) <= 15.0 + 1e-12)


# ---------------------------------------------------------------------------------------------------
# C++ oracle vs the independent NumPy restatement: whole solves
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("constrained", [False, True])
def test_ddp_solve_against_numpy_restatement(constrained):
    T = 60
    lo, up = np.array([-15.0]), np.array([15.0])
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        x0 = np.array([rng.uniform(-1, 1), rng.uniform(-np.pi, np.pi), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        cfg = oracle.default_config(horizon_steps=T, max_iter=40, with_input_constraint=int(constrained))
        r = oracle.solve("cartpole", cfg, x0, np.zeros((T, 1)), lower=lo, upper=up)
        d = dn.DDP(dn.CartPole(), dn.Config(horizon_steps=T, max_iter=40, with_input_constraint=constrained),
                   limits=(lo, up))
        d.solve(0.0, x0, np.zeros((T, 1)))
        assert r.status == d.status
        assert r.iters == d.trace[-1]["iter"]
        assert [int(t[9]) for t in r.trace] == [t["alpha_idx"] for t in d.trace]
        assert [int(t[10]) for t in r.trace] == [t["n_bw"] for t in d.trace]
        if constrained:
            assert list(r.qp_retval) == d.qp_ret
            assert [int(mk) for mk in r.qp_free_mask] == [sum(1 << i for i in f) for f in d.qp_free]
        Xn, Un = np.array(d.X), np.array(d.U)
        assert np.abs(Xn - r.X).max() <= 1e-10 * (1 + np.abs(r.X).max())
        assert np.abs(Un - r.U).max() <= 1e-10 * (1 + np.abs(r.U).max())
        Kn = np.array(d.K)
        assert np.abs(Kn - r.K).max() <= 1e-9 * (1 + np.abs(r.K).max())


def test_reference_probe_statistics():
    """Facts recorded in SURVEY.md §6 / BASELINE.md §2: x0 = (0, pi, 0, 0) converges in 17 iterations at T = 100
    and 24 at T = 200; alpha_list is 10^linspace(0, -3, 11)."""
    cfg = oracle.default_config()
    alphas = np.array([cfg.alpha_list[i] for i in range(cfg.n_alpha)])
    np.testing.assert_allclose(alphas, 10.0 ** np.linspace(0, -3, 11), rtol=1e-15)
    assert alphas[0] == 1.0 and alphas[-1] == 1e-3
    for T, iters in ((100, 17), (200, 24)):
        cfg = oracle.default_config(horizon_steps=T)
        r = oracle.solve("cartpole", cfg, [0, np.pi, 0, 0], np.zeros((T, 1)))
        assert r.status == 1 and r.iters == iters


def test_misuse_and_edge_cases():
    # lambda > lambda_max => failure status -1 (DDPSolver.hpp:196-204): make every backward pass fail
    cfg = oracle.default_config(horizon_steps=10, lambda_max=1e-3)
    p = oracle.default_params("cartpole", running_u=-1.0)  # negative input weight: Quu is not positive definite
    r = oracle.solve("cartpole", cfg, [0, 0.1, 0, 0], np.zeros((10, 1)), params=p)
    assert r.status == -1
    # max_iter exhaustion => status 0, solve() returns false (DDPSolver.hpp:115-123,140)
    cfg = oracle.default_config(horizon_steps=50, max_iter=2)
    r = oracle.solve("cartpole", cfg, [0, np.pi, 0, 0], np.zeros((50, 1)))
    assert r.status == 0 and r.iters == 2 and r.trace.shape[0] == 3
    # horizon of one step
    cfg = oracle.default_config(horizon_steps=1)
    r = oracle.solve("cartpole", cfg, [0.1, 0.2, 0, 0], np.zeros((1, 1)))
    assert r.status == 1 and r.X.shape == (2, 4)
