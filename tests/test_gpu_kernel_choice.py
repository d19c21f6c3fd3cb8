"""Kernel choice through the API: nmpc_hip_ddp_set_kernel / set_dispatch_batch / kernel_name_for_batch and their
mirrors.  The environment variables are developer overrides read ONCE, when a handle is created — not on the launch path."""
import numpy as np
import pytest

import nmpc_amd
from nmpc_amd import workloads

from test_gpu_parity import make_solver, oracle_batch, scaled_err

pytestmark = pytest.mark.gpu


def test_set_kernel_pins_the_family_and_environment_is_read_at_create(monkeypatch):
    wl = workloads.cartpole_batch(B=512, T=60, seed=1)
    s = make_solver(wl, max_iter=6)
    assert s.kernelName() == "ddp_solve_quad_kernel"
    monkeypatch.setenv("NMPC_HIP_DDP_KERNEL", "1w")  # after the handle exists: no effect on it
    assert s.kernelName() == "ddp_solve_quad_kernel"
    s.solve(wl.t0, wl.x0, wl.u_init)
    quad = (s.X().copy(), s.iters().copy(), s.status().copy())
    t = make_solver(wl, max_iter=6)  # a handle created under the override takes it
    assert t.kernelName() == "ddp_solve_tpi_kernel"
    monkeypatch.delenv("NMPC_HIP_DDP_KERNEL")
    assert t.kernelName() == "ddp_solve_tpi_kernel"
    t.setKernel("auto")
    assert t.kernelName() == "ddp_solve_quad_kernel"
    for name, kernel in (("2w", "ddp_solve_tpi2w_kernel"), ("1w", "ddp_solve_tpi_kernel"), ("ddp_solve_tpi2w_kernel", "ddp_solve_tpi2w_kernel"),
                         ("tile64", "ddp_solve_quad_kernel"), ("auto", "ddp_solve_quad_kernel")):  # (no tile kernel for n = 4: ignored)
        s.setKernel(name)
        assert s.kernelName() == kernel, name
        s.solve(wl.t0, wl.x0, wl.u_init)
        # across families: the same decisions, values to rounding (INTEGRATION.md "what is reproducible")
        assert np.array_equal(s.iters(), quad[1]) and np.array_equal(s.status(), quad[2]) and scaled_err(s.X(), quad[0]) < 1e-11
    assert np.array_equal(s.X(), quad[0])  # back on the quad kernel: the same bits
    with pytest.raises((ValueError, RuntimeError)):
        s.setKernel("warp32")
    # a choice made before the (lazily created) handle exists is applied when it is created
    u = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    u.config().horizon_steps = wl.T
    u.setKernel("2w")
    assert u.kernelName() == "ddp_solve_tpi2w_kernel"


def test_dispatch_batch_gives_a_shard_the_whole_batchs_family():
    wl = workloads.cartpole_batch(B=2048, T=40, seed=3)
    s = make_solver(wl, max_iter=5)
    assert s.kernelName() == "ddp_solve_quad_kernel" and s.kernelNameForBatch(8192) == "ddp_solve_tpi2w_kernel"
    s.setDispatchBatch(8192)
    assert s.kernelName() == "ddp_solve_tpi2w_kernel"
    s.setDispatchBatch(0)
    assert s.kernelName() == "ddp_solve_quad_kernel"
    # fp32: the family depends on the batch size at the reference's default threshold (tile64<float> below 8192, tile32 from there)
    wl = workloads.quadrotor_batch(B=256, T=20, seed=3, fp32=True)
    s = make_solver(wl, max_iter=2)
    assert s.kernelName() == "ddp_solve_tile64_kernel" and s.kernelNameForBatch(8192) == "ddp_solve_tile32_kernel"
    whole = workloads.quadrotor_batch(B=8192, T=20, seed=3, fp32=True)
    w = make_solver(whole, max_iter=2)
    w.solve(whole.t0, whole.x0, whole.u_init)
    assert w.kernelName() == "ddp_solve_tile32_kernel"
    for lo, hi in ((0, 4096), (4096, 8192)):
        sh = workloads.quadrotor_batch(B=8192, T=20, seed=3, fp32=True)
        sh.B, sh.x0, sh.u_init, sh.t0 = hi - lo, sh.x0[lo:hi], sh.u_init[lo:hi], sh.t0[lo:hi]
        h = make_solver(sh, max_iter=2)
        h.setDispatchBatch(8192)
        h.solve(sh.t0, sh.x0, sh.u_init)
        assert h.kernelName() == "ddp_solve_tile32_kernel"
        assert np.array_equal(h.X(), w.X()[lo:hi]) and np.array_equal(h.U(), w.U()[lo:hi]) and np.array_equal(h.iters(), w.iters()[lo:hi])


def test_without_the_workspace_the_lane_kernels_take_the_solve(monkeypatch):
    """centroidal / manipulator batches go to the tile kernel, whose gain records live in the per-instance workspace; when
    that allocation fails at create the handle has to fall back to a kernel that needs none — and say so in kernelName()."""
    wl = workloads.manipulator_batch(B=96, T=20, seed=4)
    monkeypatch.setenv("NMPC_HIP_DDP_NO_WORKSPACE", "1")
    s = make_solver(wl, max_iter=4)
    assert s.kernelName() == "ddp_solve_tpi_kernel"  # (the Python mirror creates its handle lazily: here)
    monkeypatch.delenv("NMPC_HIP_DDP_NO_WORKSPACE")
    assert s.kernelName() == "ddp_solve_tpi_kernel"
    s.solve(wl.t0, wl.x0, wl.u_init)
    ref = oracle_batch(wl, max_iter=4)
    assert np.array_equal(s.iters(), ref.iters) and np.array_equal(s.status(), ref.status)
    assert scaled_err(s.X(), ref.X) < 1e-9 and scaled_err(s.Kfb(), ref.K) < 1e-8
    t = make_solver(wl, max_iter=4)
    assert t.kernelName() == "ddp_solve_tile64_kernel"
