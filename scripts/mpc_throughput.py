"""Closed-loop (device-resident MPC) throughput on one MI355X: nmpc_hip_ddp_mpc_run on the reference's cart-pole MPC
configuration (TestDDPCartPole.test: T = 200, +-15 N, max_iter 3, MPC every 4 ms) and on the bipedal loop (C3 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmpc_amd

def run(name, make, B, T, ticks, **kw):
    s, x0, mm = make(B, T)
    u = np.zeros((B, T, mm))
    s.mpcRun(0.0, x0, u, 5, **kw)  # warm-up (allocations, code load)
    t0 = time.perf_counter()
    log = s.mpcRun(0.0, x0, u, ticks, **kw)
    dt = time.perf_counter() - t0
    its = log.iters.sum()
    print(f"{name}: B={B} T={T} {ticks} ticks in {dt:.3f} s -> {ticks / dt:.1f} batch-ticks/s, {B * ticks / dt / 1e6:.3f} M "
          f"instance-solves/s, {its / dt / 1e6:.2f} M instance-iterations/s, {1e3 * dt / ticks:.3f} ms per tick "
          f"(host-inclusive wall time; kernel: {s.kernelName()})")

def cartpole(B, T):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(running_u=[0.01]), B)
    c = s.config(); c.print_level = 0; c.horizon_steps = T; c.max_iter = 3; c.with_input_constraint = True
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    rng = np.random.default_rng(1)
    x0 = np.tile(np.array([0.0, np.pi, 0.0, 0.0]), (B, 1)); x0[:, :2] += rng.uniform(-0.2, 0.2, (B, 2))
    return s, x0, 1

def bipedal(B, T):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemBipedal(), B)
    c = s.config(); c.print_level = 0; c.horizon_steps = T
    rng = np.random.default_rng(2)
    x0 = np.stack([rng.uniform(-0.02, 0.02, B), rng.uniform(-0.05, 0.05, B)], 1)
    return s, x0, 1

run("cart-pole swing-up MPC (plant pattern)", cartpole, 4096, 200, 250, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
run("bipedal MPC (shift pattern)", bipedal, 1024, 300, 200, shift_warm_start=True)
