#!/usr/bin/env python3
"""Generates tests/golden/fmpc_golden.npz: input -> output vectors of the FMPC path.

The reference holds no golden vectors for solver internals and its Eigen build cannot be run in this image, so these vectors
come from the CPU oracle (oracle/fmpc_oracle.hpp) AFTER it has been pinned by tests/test_fmpc_oracle_pins.py (the reference's
MathUtils / derivative / closed-loop assertions and the NumPy check of the Newton step).  Data only; regenerate with

    python tests/golden/make_fmpc_golden.py

Cases: for each of the three problem types, solves from Variable::reset(0, 0, 0, 1, 1) and from a perturbed start with 1, 3 and
10 iterations; the first ticks of the reference's two closed loops (warm-started variable, carried barrier parameter); one solve
with init_complementary_variable and one with the merit line search.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import fmpc as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fmpc_golden.npz")
CFG_KEYS = ("horizon_steps", "max_iter", "kkt_error_thre", "check_nan", "init_complementary_variable", "update_barrier_eps",
            "break_if_llt_fails", "enable_line_search", "merit_const_scale_from_lagrange_multipliers")


def run_case(store, names, name, model, params, cfg_kw, t0, x0, var, barrier_eps=1e-4):
    cfg = O.default_config(**cfg_kw)
    r = O.solve(model, cfg, params, t0, x0, var, barrier_eps)
    store[name + "/model"] = np.array(model)
    store[name + "/params"] = np.asarray(params, float)
    store[name + "/cfg"] = np.array([float(getattr(cfg, k)) for k in CFG_KEYS])
    store[name + "/t0"] = np.array(float(t0))
    store[name + "/x0"] = np.asarray(x0, float)
    store[name + "/barrier_eps_in"] = np.array(float(barrier_eps))
    for k, a in zip(("x", "u", "lam", "s", "nu"), var.arrays()):
        store[name + "/in_" + k] = np.asarray(a, float)
    for k, a in zip(("x", "u", "lam", "s", "nu"), r.variable.arrays()):
        store[name + "/out_" + k] = a
    store[name + "/status"] = np.array(r.status)
    store[name + "/iters"] = np.array(r.iters)
    store[name + "/barrier_eps_out"] = np.array(r.barrier_eps)
    store[name + "/trace"] = r.trace
    store[name + "/k"] = r.k
    store[name + "/K"] = r.K
    store[name + "/s_gain"] = r.s
    store[name + "/P"] = r.P
    names.append(name)
    return r


def main():
    store, names = {}, []
    rng = np.random.default_rng(20260929)
    starts = {"fmpc_oscillator": np.array([0.0, 1.0]), "fmpc_cartpole": np.array([0.0, np.pi, 0.0, 0.0]),
              "fmpc_pointmass": np.array([0.0, 0.0, 0.3, -0.2])}
    for model, T in (("fmpc_oscillator", 60), ("fmpc_cartpole", 50), ("fmpc_pointmass", 40)):
        n, m, g, _ = O.model_info(model)
        p = O.default_params(model)
        for it in (1, 3, 10):
            run_case(store, names, f"{model}_reset_it{it}", model, p, dict(horizon_steps=T, max_iter=it), 0.0, starts[model],
                     O.Variable.reset(model, T))
            var = O.Variable(0.05 * rng.standard_normal((T + 1, n)), 0.05 * rng.standard_normal((T, m)),
                             0.05 * rng.standard_normal((T + 1, n)), rng.uniform(0.8, 1.5, (T, g)), rng.uniform(0.8, 1.5, (T, g)))
            run_case(store, names, f"{model}_perturbed_it{it}", model, p, dict(horizon_steps=T, max_iter=it), 0.3,
                     starts[model] + 0.05 * rng.standard_normal(n), var, barrier_eps=0.02)
    # closed loops: the first ticks of TestFmpcOscillator.cpp:164-194 and TestFmpcCartPole.cpp:340-366
    for model, T, max_iter, sim_dt, sub in (("fmpc_oscillator", 400, 3, 0.005, 1), ("fmpc_cartpole", 200, 5, 0.002, 2)):
        p = O.default_params(model)
        var, x, t, be = O.Variable.reset(model, T), starts[model].copy(), 0.0, 1e-4
        for tick in range(6):
            r = run_case(store, names, f"{model}_loop_tick{tick}", model, p, dict(horizon_steps=T, max_iter=max_iter), t, x, var, be)
            var, be = r.variable, r.barrier_eps
            for _ in range(sub):
                u = r.variable.u[0] + (r.K[0] @ (r.variable.x[0] - x) if sub > 1 else 0.0)
                x = O.evaluate(model, p, t, x, u, step_dt=sim_dt)["f"]
                t += sim_dt
    p = O.default_params("fmpc_cartpole")
    run_case(store, names, "fmpc_cartpole_init_complementary", "fmpc_cartpole", p,
             dict(horizon_steps=50, max_iter=4, init_complementary_variable=1), 0.0, starts["fmpc_cartpole"],
             O.Variable.reset("fmpc_cartpole", 50, s=0.5, nu=2.0), barrier_eps=0.3)
    for scale in (0, 1):
        run_case(store, names, f"fmpc_pointmass_line_search{scale}", "fmpc_pointmass", O.default_params("fmpc_pointmass"),
                 dict(horizon_steps=40, max_iter=5, enable_line_search=1, merit_const_scale_from_lagrange_multipliers=scale), 0.0,
                 starts["fmpc_pointmass"], O.Variable.reset("fmpc_pointmass", 40))
    store["__names__"] = np.array(names)
    store["__cfg_keys__"] = np.array(CFG_KEYS)
    np.savez_compressed(OUT, **store)
    print(len(names), "cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
