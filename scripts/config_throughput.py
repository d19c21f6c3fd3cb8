"""Throughput of the secondary BASELINE.json configurations (parity-test cases, not bench lines) on one MI355X:
C3 bipedal B=1024 T=300, C4-shape quadrotor n=12 m=4 T=50 (fp64 here), C5-shape manipulator n=14 m=7 T=30.
Usage: python scripts/config_throughput.py [scale]   (scale divides the C4 / C5 batch, default 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1

def run(name, wl, max_iter, **cfg):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = max_iter
    for k, v in cfg.items():
        setattr(c, k, v)
    if wl.limits is not None:
        s.setInputLimits(*wl.limits)
    s.solve(wl.t0, wl.x0, wl.u_init)
    t0 = time.perf_counter()
    s.solve(wl.t0, wl.x0, wl.u_init)
    wall = time.perf_counter() - t0
    k_ms = s.computationDuration().opt
    its = int(s.iters().sum())
    st = dict(zip(*np.unique(s.status(), return_counts=True)))
    print(f"{name}: B={wl.B} T={wl.T} n={wl.n} m={wl.m} kernel {s.kernelName()} {k_ms:.2f} ms (wall {1e3 * wall:.1f} ms), "
          f"{its} instance-iterations -> {its / k_ms / 1e3:.3f} M instance-it/s = {its / wl.B / (k_ms * 1e-3):.1f} batch-it/s; "
          f"status {st}", flush=True)

run("C3 bipedal", workloads.bipedal_batch(B=1024, T=300, seed=1234), 8)
run("C2 cart-pole +-15 N box", workloads.cartpole_batch(B=4096, T=100, seed=1234, constrained=True), 8, with_input_constraint=True)
run("vertical motion (variable nu, box)", workloads.vertical_batch(B=1024, T=300, seed=1234, constrained=True), 8,
    with_input_constraint=True, initial_lambda=1e-6)
run("centroidal", workloads.centroidal_batch(B=256, T=100, seed=1234), 4)
run("C4-shape quadrotor (fp64)", workloads.quadrotor_batch(B=8192 // scale, T=50, seed=1234), 4)
run("C5-shape manipulator", workloads.manipulator_batch(B=8192 // scale, T=30, seed=1234), 4)
