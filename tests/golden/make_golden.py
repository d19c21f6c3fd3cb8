#!/usr/bin/env python3
"""Generates tests/golden/ddp_golden.npz: input -> output vectors of the DDP hot path.

The reference holds no golden vectors for solver internals (SURVEY.md §4/§8 c), and its Eigen build cannot be
run in this image, so these vectors come from the CPU oracle (oracle/ddp_oracle.hpp) AFTER it has been pinned by
tests/test_oracle_pins.py (reference known answers, derivative checks, closed-loop assertions, independent
NumPy restatement).  They are data only (inputs and expected outputs); rerun this script to regenerate:

    python tests/golden/make_golden.py

Cases (SURVEY.md §8 c "golden fixtures to commit"):
  cartpole T=100, 8 splitmix64 seeds x {1 iteration, 10 iterations, converged}, unconstrained
  cartpole T=100, 4 seeds, +-15 N BoxQP, converged
  bipedal  T=300 first solve at t0 = 0 and t0 = 7.5 s (inside the omega^2 transient)
  vertical T=300 solves straddling the input-dimension changes (t0 = 1.9, 4.4), with and without constraints
  centroidal T=100 at t0 = 0 (3 iterations), t0 = 1.0 (horizon crosses the flight phase)
  quadrotor T=50, manipulator T=30: 2 seeds each, 10 iterations
  planar_vtol T=60 (n 6, m 2): 2 seeds, 10 iterations; one box-constrained
  quadrotor_f32 T=50: 2 seeds, 3 iterations, cost_update_thre 1e-3 (the oracle instantiated in float); + 1 box-constrained, 1 iteration
  manipulator_f32 T=30: 1 seed, 2 iterations (fp32 with seven inputs)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from nmpc_amd import workloads  # noqa: E402  (input generators only; no solver code)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_golden.npz")


def run_case(store, name, model, cfg_kw, x0, u_init, t0=0.0, limits=None):
    cfg = oracle.default_config(**cfg_kw)
    lo, up = limits if limits is not None else (None, None)
    r = oracle.solve(model, cfg, x0, u_init, t0=t0, lower=lo, upper=up)
    store[name + "/model"] = np.array(model)
    store[name + "/cfg_keys"] = np.array(list(cfg_kw.keys()))
    store[name + "/cfg_vals"] = np.array([float(v) for v in cfg_kw.values()])
    store[name + "/x0"] = np.asarray(x0, float)
    store[name + "/u_init"] = np.asarray(u_init, float)
    store[name + "/t0"] = np.array(float(t0))
    if limits is not None:
        store[name + "/lower"] = np.asarray(lo, float)
        store[name + "/upper"] = np.asarray(up, float)
    for k in ("X", "U", "cost", "k", "K", "trace", "dV", "qp_retval", "qp_free_mask", "m_list"):
        store[name + "/" + k] = getattr(r, k)
    store[name + "/status"] = np.array(r.status)
    return r


def main():
    store = {}
    names = []
    wl = workloads.cartpole_batch(B=8, T=100, seed=1234)
    for b in range(8):
        for tag, kw in (("it1", dict(max_iter=1)), ("it10", dict(max_iter=10)), ("conv", dict())):
            nm = f"cartpole_s{b}_{tag}"
            run_case(store, nm, "cartpole", dict(horizon_steps=100, **kw), wl.x0[b], wl.u_init[b])
            names.append(nm)
    for b in range(4):
        nm = f"cartpole_box_s{b}"
        run_case(store, nm, "cartpole", dict(horizon_steps=100, with_input_constraint=1), wl.x0[b], wl.u_init[b],
                 limits=([-15.0], [15.0]))
        names.append(nm)
    for t0 in (0.0, 7.5):
        nm = f"bipedal_t{t0:g}"
        run_case(store, nm, "bipedal", dict(horizon_steps=300), [0.01, -0.02], np.zeros((300, 1)), t0=t0)
        names.append(nm)
    for t0 in (1.9, 4.4):
        for con in (0, 1):
            nm = f"vertical_t{t0:g}_c{con}"
            run_case(store, nm, "vertical", dict(horizon_steps=300, initial_lambda=1e-6, with_input_constraint=con),
                     [1.2, 0.0], np.zeros((300, 2)), t0=t0, limits=([0.0, 0.0], [30.0, 30.0]))
            names.append(nm)
    x0c = np.array([0.01, -0.01, 1.0, 0, 0, 0, 0, 0, 0])
    for t0, kw in ((0.0, dict(max_iter=3)), (1.0, dict(max_iter=5))):
        nm = f"centroidal_t{t0:g}"
        run_case(store, nm, "centroidal", dict(horizon_steps=100, **kw), x0c, np.zeros((100, 16)), t0=t0)
        names.append(nm)
    wq = workloads.quadrotor_batch(B=2, T=50, seed=1234)
    wm = workloads.manipulator_batch(B=2, T=30, seed=1234)
    for b in range(2):
        nm = f"quadrotor_s{b}"
        run_case(store, nm, "quadrotor", dict(horizon_steps=50, max_iter=10), wq.x0[b], wq.u_init[b])
        names.append(nm)
        nm = f"manipulator_s{b}"
        run_case(store, nm, "manipulator", dict(horizon_steps=30, max_iter=10), wm.x0[b], wm.u_init[b])
        names.append(nm)
    # round 3: the builder-defined n = 6, m = 2 shape (5 <= n <= 8: fp64 tile kernel), unconstrained and with a rotor-thrust box
    wp = workloads.planar_vtol_batch(B=2, T=60, seed=1234)
    for b in range(2):
        nm = f"planar_vtol_s{b}"
        run_case(store, nm, "planar_vtol", dict(horizon_steps=60, max_iter=10), wp.x0[b], wp.u_init[b])
        names.append(nm)
    wpc = workloads.planar_vtol_batch(B=1, T=60, seed=77, constrained=True)
    run_case(store, "planar_vtol_box", "planar_vtol", dict(horizon_steps=60, max_iter=10, with_input_constraint=1), wpc.x0[0], wpc.u_init[0],
             limits=wpc.limits)
    names.append("planar_vtol_box")
    # fp32 (BASELINE config 4's arithmetic): the oracle instantiated in float, the threshold an fp32 cost resolves
    wf = workloads.quadrotor_batch(B=2, T=50, seed=1234, fp32=True)
    for b in range(2):
        nm = f"quadrotor_f32_s{b}"
        run_case(store, nm, "quadrotor_f32", dict(horizon_steps=50, max_iter=3, cost_update_thre=1e-3), wf.x0[b], wf.u_init[b])
        names.append(nm)
    # the fp32 tile kernel's smallest shape: cart-pole in float (n = 4, m = 1), three iterations
    wcf = workloads.cartpole_batch(B=1, T=100, seed=17, fp32=True)
    run_case(store, "cartpole_f32_s0", "cartpole_f32", dict(horizon_steps=100, max_iter=3, cost_update_thre=1e-3), wcf.x0[0], wcf.u_init[0])
    names.append("cartpole_f32_s0")
    # ... and with the rotor-thrust box, one iteration (BoxQP's termination tests are below float resolution: DESIGN.md §3a)
    wfc = workloads.quadrotor_batch(B=1, T=50, seed=31, constrained=True, fp32=True)
    run_case(store, "quadrotor_f32_box", "quadrotor_f32", dict(horizon_steps=50, max_iter=1, cost_update_thre=1e-3, with_input_constraint=1),
             wfc.x0[0], wfc.u_init[0], limits=wfc.limits)
    names.append("quadrotor_f32_box")
    # an fp32 shape with seven inputs (the fp64 tile kernel's float instantiation): the manipulator in float, two iterations (from
    # the third on its cost differences are below what a float cost resolves: tests/test_gpu_fp32.py)
    wmf = workloads.manipulator_batch(B=1, T=30, seed=21, fp32=True)
    run_case(store, "manipulator_f32_s0", "manipulator_f32", dict(horizon_steps=30, max_iter=2, cost_update_thre=1e-3), wmf.x0[0], wmf.u_init[0])
    names.append("manipulator_f32_s0")
    store["__names__"] = np.array(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
