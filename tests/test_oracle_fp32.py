"""The CPU oracle instantiated in float (namespace oracle_f32: oracle/ddp_oracle.hpp and oracle/models_builder.hpp compiled a
second time with Real = float) — what the fp32 HIP path of BASELINE.json config 4 is compared with (SURVEY.md §8 c).
These tests pin it to the double instantiation: same statements, so model evaluations agree to float resolution and early
iterations take the same decisions; they also document where fp32 leaves the reference's behaviour (the noise regime)."""
import numpy as np

import oracle
from nmpc_amd import workloads


def test_fp32_models_are_registered_with_the_same_shapes():
    assert oracle.model_dims("quadrotor_f32") == oracle.model_dims("quadrotor")
    assert oracle.model_dims("manipulator_f32") == oracle.model_dims("manipulator")


def test_fp32_model_evaluation_matches_fp64_to_float_resolution():
    rng = np.random.default_rng(0)
    for model, n, m in (("quadrotor", 12, 4), ("manipulator", 14, 7)):
        for _ in range(8):
            x = rng.normal(0, 0.3, n)
            u = rng.normal(2.0, 0.5, m)
            a = oracle.model_eval(model, None, 0.0, x, u)
            b = oracle.model_eval(model + "_f32", None, 0.0, x, u)
            for f in ("xn", "Fx", "Fu", "Lx", "Lu", "Lxx", "Luu", "Lxu", "Vx", "Vxx"):
                va, vb = getattr(a, f), getattr(b, f)
                assert np.abs(va - vb).max() <= 2e-6 * (1 + np.abs(va).max()), (model, f)
            assert abs(a.running_cost - b.running_cost) <= 2e-6 * (1 + abs(a.running_cost))
            assert abs(a.terminal_cost - b.terminal_cost) <= 2e-6 * (1 + abs(a.terminal_cost))


def test_fp32_first_iterations_follow_the_fp64_oracle():
    wl = workloads.quadrotor_batch(B=128, T=50, seed=1234)
    cfg = oracle.default_config(horizon_steps=50, max_iter=3)
    a = oracle.solve_batch("quadrotor", cfg, wl.x0, wl.u_init, n_threads=4, want_alpha_hist=True)
    b = oracle.solve_batch("quadrotor_f32", cfg, wl.x0, wl.u_init, n_threads=4, want_alpha_hist=True)
    same = (a.alpha_idx_hist == b.alpha_idx_hist).all(axis=1) & (a.iters == b.iters)
    assert same.mean() >= 0.95
    err = (np.abs(a.X - b.X) / (1 + np.abs(a.X))).reshape(128, -1).max(1)
    assert np.median(err) <= 1e-5 and err[same].max() <= 1e-3
    Ja, Jb = a.cost.sum(1), b.cost.sum(1)
    assert (np.abs(Ja - Jb) / Ja)[same].max() <= 1e-4


def test_fp32_with_default_thresholds_enters_the_noise_regime():
    """cost_update_thre = 1e-7 (DDPSolver.h:109) is below the resolution of a float cost of ~10: the fp32 oracle rejects steps
    on rounding noise and many solves end in status -1 where the fp64 oracle converges; with a threshold float can resolve it
    converges like the fp64 one.  (This is why bench.py reports the fp32 workload in both configurations.)"""
    wl = workloads.quadrotor_batch(B=96, T=50, seed=1234)
    cfg = oracle.default_config(horizon_steps=50, max_iter=60)
    a = oracle.solve_batch("quadrotor", cfg, wl.x0, wl.u_init, n_threads=4)
    b = oracle.solve_batch("quadrotor_f32", cfg, wl.x0, wl.u_init, n_threads=4)
    assert (a.status == 1).all()
    assert (b.status != 1).mean() > 0.3
    cfg2 = oracle.default_config(horizon_steps=50, max_iter=60, cost_update_thre=1e-3)
    c = oracle.solve_batch("quadrotor_f32", cfg2, wl.x0, wl.u_init, n_threads=4)
    assert (c.status == 1).mean() > 0.98
    Ja, Jc = a.cost.sum(1), c.cost.sum(1)
    assert (np.abs(Ja - Jc) / Ja).max() <= 1e-4
