"""Pins of the CPU FMPC oracle (oracle/fmpc_oracle.hpp) against what the reference's own tests assert:
TestMathUtils.cpp:7-70, TestFmpcOscillator.cpp:137-266, TestFmpcCartPole.cpp:318-384,625-693 — plus independent NumPy checks
of the pieces that have no golden vector there (the LDLT restatement, the Newton step of one iteration)."""
import numpy as np
import pytest

from oracle import fmpc as O

MODELS = ("fmpc_oscillator", "fmpc_cartpole", "fmpc_pointmass")


def test_l1_norm_directional_deriv_identity_function():
    """TestMathUtils.cpp:11-30: identity Jacobian, 1000 random draws (the first one at func = 0), |analytical - numerical| < 1e-5."""
    rng = np.random.default_rng(0)
    eps = 1e-6
    jac = np.eye(4)
    for i in range(1000):
        func = 100.0 * rng.uniform(-1, 1, 4)
        d = rng.uniform(-1, 1, 4)
        if i == 0:
            func[:] = 0
        ana = O.l1_norm_directional_deriv(func, jac, d)
        num = (np.abs(func + eps * d).sum() - np.abs(func).sum()) / eps
        assert abs(ana - num) < 1e-5


def test_l1_norm_directional_deriv_nonlinear_function():
    """TestMathUtils.cpp:33-69: 3 functions of 4 variables, |analytical - numerical| < 1e-3."""
    rng = np.random.default_rng(1)
    eps = 1e-6

    def fn(x):
        return np.array([x @ x - 10.0, x[1] ** 3 + -5 * x[2] ** 2 + 10 * x[3] + -20, np.sin(x[0]) + np.cos(x[1])])

    def jac(x):
        return np.array([2 * x, [0, 3 * x[1] ** 2, -10 * x[2], 10], [np.cos(x[0]), -np.sin(x[1]), 0, 0]])

    for i in range(1000):
        x = 100.0 * rng.uniform(-1, 1, 4)
        d = rng.uniform(-1, 1, 4)
        if i == 0:
            x[:] = 0
        ana = O.l1_norm_directional_deriv(fn(x), jac(x), d)
        num = (np.abs(fn(x + eps * d)).sum() - np.abs(fn(x)).sum()) / eps
        assert abs(ana - num) < 1e-3 * max(1.0, abs(num) * 1e-2)  # the cubic term reaches 1e6: same relative bar


@pytest.mark.parametrize("model", MODELS)
def test_derivatives_match_central_differences(model):
    """TestFmpcOscillator.cpp:207-266 / TestFmpcCartPole.cpp:625-693 (dt 0.1, eps 1e-6, norm of the difference < 1e-6), extended
    to the cost derivatives."""
    n, m, g, _ = O.model_info(model)
    p = O.default_params(model)
    p[0] = 0.1  # horizon_dt of the reference's derivative tests
    rng = np.random.default_rng(2)
    points = [(np.array([0.1, -0.2, 0.3, -0.4])[:n], np.array([0.3, -0.7])[:m])]
    points += [(rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)) for _ in range(5)]
    eps = 1e-6
    for x, u in points:
        o = O.evaluate(model, p, 0.0, x, u)
        A = np.zeros((n, n)); B = np.zeros((n, m)); Cm = np.zeros((g, n)); D = np.zeros((g, m))
        Lx = np.zeros(n); Lu = np.zeros(m); Lxx = np.zeros((n, n)); Luu = np.zeros((m, m)); Lxu = np.zeros((n, m))
        Vx = np.zeros(n); Vxx = np.zeros((n, n))
        for i in range(n):
            e = np.zeros(n); e[i] = eps
            hi, lo = O.evaluate(model, p, 0.0, x + e, u), O.evaluate(model, p, 0.0, x - e, u)
            A[:, i] = (hi["f"] - lo["f"]) / (2 * eps)
            Cm[:, i] = (hi["g"] - lo["g"]) / (2 * eps)
            Lx[i] = (hi["costs"][0] - lo["costs"][0]) / (2 * eps)
            Vx[i] = (hi["costs"][1] - lo["costs"][1]) / (2 * eps)
            Lxx[:, i] = (hi["Lx"] - lo["Lx"]) / (2 * eps)
            Vxx[:, i] = (hi["Vx"] - lo["Vx"]) / (2 * eps)
        for i in range(m):
            e = np.zeros(m); e[i] = eps
            hi, lo = O.evaluate(model, p, 0.0, x, u + e), O.evaluate(model, p, 0.0, x, u - e)
            B[:, i] = (hi["f"] - lo["f"]) / (2 * eps)
            D[:, i] = (hi["g"] - lo["g"]) / (2 * eps)
            Lu[i] = (hi["costs"][0] - lo["costs"][0]) / (2 * eps)
            Luu[:, i] = (hi["Lu"] - lo["Lu"]) / (2 * eps)
            Lxu[:, i] = (hi["Lx"] - lo["Lx"]) / (2 * eps)
        for name, num in (("A", A), ("B", B), ("C", Cm), ("D", D), ("Lx", Lx), ("Lu", Lu), ("Lxx", Lxx), ("Luu", Luu),
                          ("Lxu", Lxu), ("Vx", Vx), ("Vxx", Vxx)):
            assert np.linalg.norm(o[name] - num) < 1e-6, (model, name)


def test_ldlt_restatement_solves_symmetric_systems():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 6):
        for trial in range(50):
            Q = rng.standard_normal((n, n))
            Gm = Q @ Q.T + 0.1 * np.eye(n) if trial % 2 == 0 else Q + Q.T  # positive definite / indefinite
            b = rng.standard_normal((n, 3))
            x, ok = O.ldlt_solve(Gm, b)
            assert ok
            assert np.allclose(Gm @ x, b, rtol=0, atol=1e-9 * max(1.0, np.linalg.cond(Gm)))
            x2, _ = O.ldlt_solve(Gm, b, use_lu=True)
            assert np.allclose(Gm @ x2, b, rtol=0, atol=1e-9 * max(1.0, np.linalg.cond(Gm)))
    # Eigen's pseudo-inverse of D: an all-zero matrix factorises (info() == Success) and solves to zero
    x, ok = O.ldlt_solve(np.zeros((2, 2)), np.ones(2))
    assert ok and np.all(x == 0)
    # a zero pivot followed by a non-zero one is what info() reports as NumericalIssue
    x, ok = O.ldlt_solve(np.array([[0.0, 1.0], [1.0, 0.0]]), np.ones(2))
    assert not ok


def newton_residuals(model, p, cfg, t0, x0, var, res):
    """Residuals of the linearised KKT system (the equations the Riccati recursion solves, FmpcSolver.hpp:522-708 citing
    (2.26)-(2.27)) evaluated with NumPy at the step `res.delta` the oracle computed from `var`."""
    n, m, g, _ = O.model_info(model)
    T = cfg.horizon_steps
    dt = p[0]
    d = res.delta
    eps_b = res.trace[0][2]
    out = [np.abs(d.x[0] - (x0 - var.x[0])).max()]
    for i in range(T):
        o = O.evaluate(model, p, t0 + i * dt, var.x[i], var.u[i])
        x_bar = o["f"] - var.x[i + 1]
        g_bar = o["g"] + var.s[i]
        Lx_bar = -var.lam[i] + dt * o["Lx"] + o["A"].T @ var.lam[i + 1] + o["C"].T @ var.nu[i]
        Lu_bar = dt * o["Lu"] + o["B"].T @ var.lam[i + 1] + o["D"].T @ var.nu[i]
        r = [d.x[i + 1] - (o["A"] @ d.x[i] + o["B"] @ d.u[i] + x_bar),
             Lx_bar + dt * o["Lxx"] @ d.x[i] + dt * o["Lxu"] @ d.u[i] - d.lam[i] + o["A"].T @ d.lam[i + 1] + o["C"].T @ d.nu[i],
             Lu_bar + dt * o["Lxu"].T @ d.x[i] + dt * o["Luu"] @ d.u[i] + o["B"].T @ d.lam[i + 1] + o["D"].T @ d.nu[i],
             o["C"] @ d.x[i] + o["D"] @ d.u[i] + d.s[i] + g_bar,
             var.nu[i] * d.s[i] + var.s[i] * d.nu[i] + (var.s[i] * var.nu[i] - eps_b)]
        out.append(max(np.abs(v).max() for v in r))
    oT = O.evaluate(model, p, t0 + T * dt, var.x[T], np.zeros(m))
    out.append(np.abs((oT["Vx"] - var.lam[T]) + oT["Vxx"] @ d.x[T] - d.lam[T]).max())
    return np.array(out)


@pytest.mark.parametrize("model", MODELS)
def test_one_iteration_solves_the_linearised_kkt_system(model):
    """No golden vectors for k, K, s, P in the reference: the step they produce is checked against the linear system it must
    solve, with an independent NumPy evaluation."""
    n, m, g, _ = O.model_info(model)
    p = O.default_params(model)
    T = 30
    cfg = O.default_config(horizon_steps=T, max_iter=1)
    rng = np.random.default_rng(4)
    var = O.Variable(0.3 * rng.standard_normal((T + 1, n)), 0.3 * rng.standard_normal((T, m)),
                     0.3 * rng.standard_normal((T + 1, n)), rng.uniform(0.5, 2.0, (T, g)), rng.uniform(0.5, 2.0, (T, g)))
    x0 = 0.3 * rng.standard_normal(n)
    res = O.solve(model, cfg, p, 0.2, x0, var)
    assert res.status == 5 and res.iters == 1
    scale = max(1.0, max(np.abs(a).max() for a in res.delta.arrays()))
    assert newton_residuals(model, p, cfg, 0.2, x0, var, res).max() < 1e-9 * scale
    # the update (FmpcSolver.hpp:801-835) moves along that step with the two step lengths of the trace
    a_s, a_nu = res.trace[0][5], res.trace[0][4]
    assert np.allclose(res.variable.x, var.x + a_s * res.delta.x, rtol=0, atol=1e-14)
    assert np.allclose(res.variable.s, var.s + a_s * res.delta.s, rtol=0, atol=1e-14)
    assert np.allclose(res.variable.lam, var.lam + a_nu * res.delta.lam, rtol=0, atol=1e-14)
    assert np.allclose(res.variable.nu, var.nu + a_nu * res.delta.nu, rtol=0, atol=1e-14)
    assert res.variable.s.min() > 0 and res.variable.nu.min() > 0  # fraction-to-boundary rule (:713-742)
    # P of the backward pass is symmetrised (:627-629)
    assert np.abs(res.P - np.transpose(res.P, (0, 2, 1))).max() == 0


def run_oscillator_loop(enable_line_search=False, end_t=10.0):
    """TestFmpcOscillator.cpp:137-205 without the file dumps."""
    model = "fmpc_oscillator"
    horizon_dt, horizon_duration = 0.01, 4.0
    T = int(horizon_duration / horizon_dt)
    p = O.default_params(model)
    p[0] = horizon_dt
    cfg = O.default_config(horizon_steps=T, max_iter=3, enable_line_search=enable_line_search)
    var = O.Variable.reset(model, T, 0.0, 0.0, 0.0, 1e0, 1e0)
    sim_dt, t = 0.005, 0.0
    x = np.array([0.0, 1.0])
    be = 1e-4
    log = []
    while t < end_t:
        r = O.solve(model, cfg, p, t, x, var, be)
        be = r.barrier_eps  # barrier_eps_ is a member of the solver object the loop reuses (FmpcSolver.h:414)
        u = r.variable.u[0].copy()
        gval = O.evaluate(model, p, t, x, u)["g"]
        log.append((t, x.copy(), u, r.status, r.iters, r.trace[r.iters - 1][1], gval))
        x = O.evaluate(model, p, t, x, u, step_dt=sim_dt)["f"]
        t += sim_dt
        var = r.variable
    return x, log


def test_oscillator_closed_loop_meets_the_reference_bounds():
    x, log = run_oscillator_loop()
    assert all(row[3] in (1, 5) for row in log)  # Succeeded or MaxIterationReached (:173)
    assert all((row[6] <= 0).all() for row in log)  # inequality constraints hold on the applied input (:181-183)
    assert abs(x[0]) < 1e-2 and abs(x[1]) < 1e-2  # final convergence (:197-198)
    # the state constraint is active on the way: x[1] >= -0.05 binds (casadi's example, :16)
    assert min(row[1][1] for row in log) < -0.04
    assert any(row[3] == 1 for row in log[-200:])  # near the origin the KKT error falls below the threshold


def test_oscillator_closed_loop_with_merit_line_search():
    """enable_line_search (FmpcSolver.h:85, FmpcSolver.hpp:748-792) is off in the reference's tests; the loop must still meet
    their bounds with it on."""
    x, log = run_oscillator_loop(enable_line_search=True, end_t=10.0)
    assert all(row[3] in (1, 5) for row in log)
    assert all((row[6] <= 1e-9).all() for row in log)
    assert abs(x[0]) < 1e-2 and abs(x[1]) < 1e-2


def test_cartpole_closed_loop_meets_the_reference_bounds():
    """TestFmpcCartPole.cpp:318-384 with the ROS timer replaced by its nominal schedule: one solve every mpc_dt = 2 sim steps,
    u = u_list[0] + K_0 (x_list[0] - x) in between (:347-351)."""
    model = "fmpc_cartpole"
    horizon_dt, horizon_duration = 0.01, 2.0
    T = int(horizon_duration / horizon_dt)
    p = O.default_params(model)
    p[0] = horizon_dt
    cfg = O.default_config(horizon_steps=T, max_iter=5)
    var = O.Variable.reset(model, T, 0.0, 0.0, 0.0, 1e0, 1e0)
    sim_dt, t = 0.002, 0.0
    x = np.array([0.0, np.pi, 0.0, 0.0])
    be = 1e-4
    u_cur = np.zeros(1)
    r = None
    step = 0
    while t < 10.0:
        if step % 2 == 0:
            r = O.solve(model, cfg, p, t, x, var, be)
            assert r.status in (1, 5)
            be = r.barrier_eps
            u_cur = r.variable.u[0].copy()
            var = r.variable
        u = u_cur + r.K[0] @ (r.variable.x[0] - x)
        x = O.evaluate(model, p, t, x, u, step_dt=sim_dt)["f"]
        t += sim_dt
        step += 1
        assert abs(x[0] - 0.0) < 1e2  # :362
    assert abs(x[0]) < 1.0 and abs(x[1]) < 1e-1 and abs(x[2]) < 1.0 and abs(x[3]) < 1e-1  # :377-380


def test_check_variable_rejects_negative_slacks():
    """checkVariable (FmpcSolver.hpp:338-353): std::runtime_error for a negative s or nu."""
    model = "fmpc_oscillator"
    cfg = O.default_config(horizon_steps=5, max_iter=2)
    var = O.Variable.reset(model, 5)
    var.s[2, 1] = -1e-3
    assert O.solve(model, cfg, None, 0.0, np.zeros(2), var).status == -2
    var = O.Variable.reset(model, 5)
    var.nu[4, 0] = -1.0
    assert O.solve(model, cfg, None, 0.0, np.zeros(2), var).status == -2


def test_init_complementary_variable_and_batch_entry():
    """init_complementary_variable (FmpcSolver.hpp:170-187) resets barrier_eps to 1e-4 and s, nu from the constraint values; the
    threaded batch entry returns what the single entry returns."""
    model = "fmpc_cartpole"
    T = 40
    p = O.default_params(model)
    cfg = O.default_config(horizon_steps=T, max_iter=4, init_complementary_variable=True)
    rng = np.random.default_rng(5)
    B = 6
    var = O.Variable.reset(model, T, batch=B)
    var.x[:] = 0.1 * rng.standard_normal(var.x.shape)
    x0 = 0.2 * rng.standard_normal((B, 4))
    x0[:, 1] += np.pi
    rb = O.solve_batch(model, cfg, p, 0.0, x0, var, n_threads=3)
    for b in range(B):
        r1 = O.solve(model, cfg, p, 0.0, x0[b], O.Variable(*(a[b] for a in var.arrays())), barrier_eps=0.7)
        assert r1.status == rb.status[b] and r1.iters == rb.iters[b]
        for a, c in zip(r1.variable.arrays(), rb.variable.arrays()):
            assert np.array_equal(a, c[b])
        assert np.array_equal(r1.K[0], rb.K0[b])
        # first barrier parameter of the run: 0.5 * mean(s nu) of the re-initialised s, nu (:370-392)
        g0 = np.array([O.evaluate(model, p, 0.0, var.x[b, i], var.u[b, i])["g"] for i in range(T)])
        s0 = 1.01 * np.maximum(-g0, 1e-2)
        nu0 = 1.01 * np.maximum(1e-4 / s0, 1e-2)
        assert np.isclose(r1.trace[0][2], 0.5 * (s0 * nu0).mean(), rtol=1e-13)
