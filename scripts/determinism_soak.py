"""Every kernel family, the same solve N times on fresh handles and on one reused handle: every result bit-identical to the
first (a race between the waves of a workgroup shows up as a run that differs).   python scripts/determinism_soak.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CASES = [
    ("quad c2", lambda: workloads.cartpole_batch(B=4096, T=100, seed=1234), dict(max_iter=8), None),
    ("quad c2 box", lambda: workloads.cartpole_batch(B=4096, T=100, seed=1234, constrained=True), dict(max_iter=8, with_input_constraint=True), None),
    ("quad bipedal", lambda: workloads.bipedal_batch(B=1024, T=300, seed=7), dict(max_iter=4), None),
    ("two-wave 8192", lambda: workloads.cartpole_batch(B=8192, T=100, seed=99), dict(max_iter=6), None),
    ("tile64<float> c4", lambda: workloads.quadrotor_batch(B=8192, T=50, seed=5, fp32=True), dict(max_iter=4, cost_update_thre=1e-3), None),
    ("tile32 c4", lambda: workloads.quadrotor_batch(B=8192, T=50, seed=5, fp32=True), dict(max_iter=4, cost_update_thre=1e-3), "tile32"),
    ("tile32 c4 box", lambda: workloads.quadrotor_batch(B=2048, T=50, seed=5, fp32=True, constrained=True), dict(max_iter=3, cost_update_thre=1e-3, with_input_constraint=True), None),
    ("tile64 c5", lambda: workloads.manipulator_batch(B=8192, T=30, seed=5), dict(max_iter=4), None),
    ("tile64 quadrotor box", lambda: workloads.quadrotor_batch(B=4096, T=50, seed=5, constrained=True), dict(max_iter=3, with_input_constraint=True), None),
    ("tile64 manipulator 512", lambda: workloads.manipulator_batch(B=512, T=30, seed=5), dict(max_iter=4), None),
    ("wpi manipulator 512", lambda: workloads.manipulator_batch(B=512, T=30, seed=5), dict(max_iter=4), "wpi"),
    ("wpi quadrotor box 256", lambda: workloads.quadrotor_batch(B=256, T=50, seed=5, constrained=True), dict(max_iter=3, with_input_constraint=True), "wpi"),
    ("lane vtol 2048", lambda: workloads.planar_vtol_batch(B=2048, T=60, seed=5), dict(max_iter=4), "1w"),
]


EXPECT = {"tile32": "ddp_solve_tile32_kernel", "wpi": "ddp_solve_wpi_kernel", "1w": "ddp_solve_tpi_kernel", "2w": "ddp_solve_tpi2w_kernel",
          "quad": "ddp_solve_quad_kernel", "tile64": "ddp_solve_tile64_kernel"}


def make(wl, cfg, kernel=None):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    if kernel:
        s.setKernel(kernel)  # nmpc_hip_ddp_set_kernel: the case repeats THIS family whatever the dispatch would pick
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    for k, v in cfg.items():
        setattr(c, k, v)
    if wl.limits is not None and cfg.get("with_input_constraint"):
        s.setInputLimits(*wl.limits)
    return s


def digest(s):
    return (s.X().tobytes(), s.U().tobytes(), s.iters().tobytes(), s.status().tobytes(), s.kff().tobytes())


bad_total = 0
for label, mk, cfg, kernel in CASES:
    try:
        wl = mk()
    except AttributeError as e:
        print(f"{label}: skipped ({e})")
        continue
    t = time.time()
    s = make(wl, cfg, kernel)
    s.solve(wl.t0, wl.x0, wl.u_init)
    ref = digest(s)
    name = s.kernelName()
    if kernel and name != EXPECT[kernel]:
        print(f"{label}: pinned {kernel} but ran {name}")
        bad_total += 1
    bad = 0
    for r in range(N):
        h = s if r % 2 else make(wl, cfg, kernel)  # alternately the reused handle and a fresh one
        h.solve(wl.t0, wl.x0, wl.u_init)
        bad += int(digest(h) != ref)
    bad_total += bad
    print(f"{label:24s} {name:26s} {N} repetitions: {bad} differ from the first   ({time.time() - t:.1f} s)", flush=True)
print("TOTAL differing runs:", bad_total)
