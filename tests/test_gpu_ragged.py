"""The ragged-convergence schedule (nmpc_hip_ddp_config::ragged_schedule; include/nmpc_amd/hip/ragged_schedule.hpp): a long solve cut
into resumable launches with a device-side compaction between them returns the BITS of the single whole-solve launch — every output
field, every trace row — and the oracle's decisions (reference: each DDPSolver object runs its own loop to ITS end,
DDPSolver.hpp:115-123).  Also covered: batches that are no multiple of a workgroup, box constraints, the two-wave kernel, reuse of a
handle, shapes without resumable kernels (falling back to one launch), and the solver pools (Python and, in tests/cpp, C++)."""
import numpy as np
import pytest

import oracle
from nmpc_amd import workloads

from test_gpu_parity import make_solver, oracle_batch

pytestmark = pytest.mark.gpu

FIELDS = ("X", "U", "cost", "kff", "Kfb", "trace", "iters", "status", "dV")


def schedule_launches(max_iter, B=1):
    """Launches of the schedule (capi.hip raggedRounds): the last iterations are 16, 32, 48, 64 (batches beyond 4096: 128, 256 too),
    then max_iter."""
    return len([c for c in (16, 32, 48, 64, 128, 256) if c < max_iter and (c <= 64 or B > 4096)]) + 1


def outputs(s):
    return {f: np.array(getattr(s, f)()) for f in FIELDS}


def assert_same_bits(a, b, what):
    for f in FIELDS:
        assert np.array_equal(a[f], b[f], equal_nan=True), (what, f, int((a[f] != b[f]).sum()))


@pytest.mark.parametrize("B,max_iter,constrained", [(1024, 120, False), (300, 70, False), (1000, 500, False), (520, 90, True), (17, 40, False)])
def test_ragged_solve_returns_the_bits_of_one_launch(B, max_iter, constrained):
    wl = workloads.cartpole_batch(B=B, T=100, seed=B + max_iter, constrained=constrained)
    cfg = dict(max_iter=max_iter, with_input_constraint=constrained)
    whole = make_solver(wl, ragged_schedule=-1, **cfg)
    whole.solve(wl.t0, wl.x0, wl.u_init)
    assert whole.lastSolveLaunches() == 1 and whole.kernelName() == "ddp_solve_quad_kernel"
    want = outputs(whole)
    for mode in (0, 1):
        s = make_solver(wl, ragged_schedule=mode, **cfg)
        s.solve(wl.t0, wl.x0, wl.u_init)
        # automatic (0): a lone handle's synchronous solve() takes one launch whatever the max_iter cap is (a warm-started
        # solve that converges at once must not pay the schedule's boundaries); queued solves and pools switch it on (tests below)
        assert s.lastSolveLaunches() == (1 if mode == 0 else schedule_launches(max_iter))
        assert s.kernelName() == "ddp_solve_quad_kernel"
        assert_same_bits(outputs(s), want, f"ragged_schedule {mode}")
        s.solve(wl.t0, wl.x0, wl.u_init)  # the handle again: nothing of the first solve's schedule may linger
        assert_same_bits(outputs(s), want, f"ragged_schedule {mode}, second solve")
    assert want["iters"].max() > (16 if constrained else 32) and (want["iters"] < 16).sum() > 0.3 * B  # the case IS ragged
    if not constrained:
        ref = oracle_batch(wl, **cfg)
        assert np.array_equal(want["iters"], ref.iters) and np.array_equal(want["status"], ref.status)


def test_ragged_schedule_on_the_two_wave_kernel_and_on_bipedal(monkeypatch):
    monkeypatch.setenv("NMPC_HIP_DDP_KERNEL", "2w")
    wl = workloads.cartpole_batch(B=1500, T=100, seed=77)
    whole = make_solver(wl, max_iter=100, ragged_schedule=-1)
    whole.solve(wl.t0, wl.x0, wl.u_init)
    s = make_solver(wl, max_iter=100, ragged_schedule=1)
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.kernelName() == "ddp_solve_tpi2w_kernel" and s.lastSolveLaunches() == schedule_launches(100) == 5 and whole.lastSolveLaunches() == 1
    assert_same_bits(outputs(s), outputs(whole), "two-wave kernel")
    monkeypatch.delenv("NMPC_HIP_DDP_KERNEL")
    wl = workloads.bipedal_batch(B=260, T=300, seed=5)
    whole = make_solver(wl, max_iter=60, ragged_schedule=-1)
    whole.solve(wl.t0, wl.x0, wl.u_init)
    s = make_solver(wl, max_iter=60, ragged_schedule=1)
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.lastSolveLaunches() == 4
    assert_same_bits(outputs(s), outputs(whole), "bipedal")


def test_short_solves_and_unsupported_shapes_stay_one_launch():
    wl = workloads.cartpole_batch(B=256, T=100, seed=3)
    s = make_solver(wl, max_iter=8)
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.lastSolveLaunches() == 1  # automatic: a synchronous solve
    s = make_solver(wl, max_iter=8, ragged_schedule=1)
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.lastSolveLaunches() == 1  # eight iterations are one launch of the schedule anyway
    wl = workloads.manipulator_batch(B=96, T=20, seed=3)
    s = make_solver(wl, max_iter=40, ragged_schedule=1)  # the tile / wave-per-instance kernels have no resumable instantiation
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.lastSolveLaunches() == 1


def test_automatic_mode_follows_the_entry_point():
    """ragged_schedule 0: one launch for solve(), solveDevice() and the ticks of mpcRun() on a lone handle, whatever max_iter says (the
    reference's default is 500 and only a cap); the schedule for a handle of a pool and for nmpc_hip_ddp_solve_async — same bits."""
    import ctypes as C

    import torch

    import nmpc_amd
    from nmpc_amd import _capi

    wl = workloads.cartpole_batch(B=512, T=100, seed=41)
    s = make_solver(wl, max_iter=500)
    s.solve(wl.t0, wl.x0, wl.u_init)
    assert s.lastSolveLaunches() == 1
    want = outputs(s)
    d = [torch.from_numpy(a).cuda() for a in (wl.t0, wl.x0, wl.u_init)]
    s.solveDevice(*[t.data_ptr() for t in d])
    s.synchronize()
    assert s.lastSolveLaunches() == 1
    assert_same_bits(outputs(s), want, "solveDevice")
    dp = C.POINTER(C.c_double)
    _capi.check(s._L.nmpc_hip_ddp_solve_async(s._h, wl.t0.ctypes.data_as(dp), wl.x0.ctypes.data_as(dp), wl.u_init.ctypes.data_as(dp)))
    s.synchronize()
    s._cache = {}
    assert s.lastSolveLaunches() == schedule_launches(500)
    assert_same_bits(outputs(s), want, "solve_async")
    pool = nmpc_amd.DDPSolverPool(nmpc_amd.make_problem(wl.model), wl.B, n_handles=2)
    c = pool.config()
    c.print_level, c.horizon_steps, c.max_iter = 0, wl.T, 500
    pool.applyConfig()
    h = pool.submit(*[t.data_ptr() for t in d])
    pool.synchronize()
    assert h.lastSolveLaunches() == schedule_launches(500)
    assert_same_bits(outputs(h), want, "pool handle")


def test_solver_pool_with_the_ragged_schedule_overlaps_more():
    """32 batches to convergence on eight handles: per-batch results are a lone handle's, and the schedule frees the CUs the converged
    instances held, so the pool's rate is well above the whole-solve launches'."""
    import time

    import torch

    import nmpc_amd

    wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
    prob = nmpc_amd.make_problem(wl.model)
    d = [torch.from_numpy(a).cuda() for a in (wl.t0, wl.x0, wl.u_init)]
    lone = make_solver(wl, max_iter=500, ragged_schedule=-1)
    lone.solve(wl.t0, wl.x0, wl.u_init)
    want = outputs(lone)
    rates = {}
    for mode in (-1, 0):
        pool = nmpc_amd.DDPSolverPool(prob, wl.B, n_handles=8)
        c = pool.config()
        c.print_level, c.horizon_steps, c.max_iter, c.ragged_schedule = 0, wl.T, 500, mode
        pool.applyConfig()
        for _ in range(8):
            pool.submit(*[t.data_ptr() for t in d])
        pool.synchronize()
        t0 = time.perf_counter()
        for _ in range(32):
            pool.submit(*[t.data_ptr() for t in d])
        pool.synchronize()
        rates[mode] = 32 / (time.perf_counter() - t0)
        for h in pool.solvers:
            assert_same_bits(outputs(h), want, f"pool handle, ragged_schedule {mode}")
    print(f"32 batches of 4096 on eight handles (GPU_MAX_HW_QUEUES {__import__('os').environ.get('GPU_MAX_HW_QUEUES')}): {rates[-1]:.1f} batches/s with whole-solve launches, {rates[0]:.1f} with the ragged schedule")
    assert rates[0] > 1.25 * rates[-1]  # (measured 1.4 - 1.6 x from box to box)


def test_cpp_solver_pool_matches_lone_solvers(tmp_path):
    """include/nmpc_amd/DDPSolverBatch.hpp: DDPSolverPool (solveAsync / wait over nmpc_hip_ddp_solve_async) — examples/cartpole_pool.cpp
    solves six batches on a lone solver and on pools of three handles (whole-solve launches, then the ragged schedule), compares every
    batch bit for bit and returns non-zero on a difference."""
    import os
    import re
    import subprocess

    from nmpc_amd import build as hip_build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cartpole_pool")
    libdir = os.path.dirname(hip_build.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", f"-I{root}/include", f"{root}/examples/cartpole_pool.cpp", f"-L{libdir}", "-lnmpc_hip_ddp",
           f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "1024", "6", "3"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = re.findall(r"ragged_schedule\s+(-?\d+): 6 batches in (\S+) ms \(launches per solve: (\d+)\), batches differing from the lone solver: (\d+)",
                      r.stdout)
    assert len(rows) == 2 and all(int(x[3]) == 0 for x in rows)
    assert int(rows[0][2]) == 1 and int(rows[1][2]) > 2  # whole-solve launches, then the schedule's resumable launches


def test_ragged_schedule_moves_the_per_instance_limits_with_their_instances():
    """Per-instance boxes (setInputLimitsBatch) and per-instance time-varying limit tables (setInputLimitsHorizon with (B, T, MM)) are
    PERSISTENT inputs of a handle: the compaction swaps them with their instances and the replay puts them back (a schedule that
    returned early would leave later solves pairing instances with their neighbours' limits).  Same bits as whole-solve launches, and the
    same again on a second solve of the same handle."""
    wl = workloads.cartpole_batch(B=520, T=100, seed=77, constrained=True)
    rng = np.random.default_rng(3)
    lo = -(8.0 + 12.0 * rng.random((wl.B, 1)))
    up = 8.0 + 12.0 * rng.random((wl.B, 1))
    cfg = dict(max_iter=90, with_input_constraint=True)
    for kind in ("batch", "horizon"):
        def prepare(s):
            if kind == "batch":
                s.setInputLimitsBatch(lo, up)
            else:
                ramp = 1.0 + 0.3 * np.linspace(0.0, 1.0, wl.T)[None, :, None]
                s.setInputLimitsHorizon(lo[:, None, :] * ramp, up[:, None, :] * ramp)
        whole = make_solver(wl, ragged_schedule=-1, **cfg)
        prepare(whole)
        whole.solve(wl.t0, wl.x0, wl.u_init)
        want = outputs(whole)
        s = make_solver(wl, ragged_schedule=1, **cfg)
        prepare(s)
        s.solve(wl.t0, wl.x0, wl.u_init)
        assert s.lastSolveLaunches() == schedule_launches(90) and whole.lastSolveLaunches() == 1
        assert_same_bits(outputs(s), want, f"per-instance limits ({kind})")
        s.solve(wl.t0, wl.x0, wl.u_init)  # the limits are where the first solve found them
        assert_same_bits(outputs(s), want, f"per-instance limits ({kind}), second solve")
        assert want["iters"].max() > 16 and len(set(want["iters"].tolist())) > 3
