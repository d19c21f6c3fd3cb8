"""Streamed solves (nmpc_hip_ddp_solve_stream; include/nmpc_amd/hip/stream_schedule.hpp): a queue of N >> B instances through the B
slots of one handle, a finished instance's slot refilled at the next round boundary.  Reference: every DDPSolver object runs its own
loop to ITS end (DDPSolver.hpp:115-123) — so every instance of the queue must return the bits of its lone solve (same kernel family:
an instance's iterations do not depend on its slot or its neighbours), stop at its own convergence or its own max_iter-th iteration,
and agree with the oracle's decisions."""
import numpy as np
import pytest

from nmpc_amd import workloads

from test_gpu_parity import make_solver, oracle_batch

pytestmark = pytest.mark.gpu

FIELDS = ("X", "U", "cost", "status", "iters", "trace_last", "dV")


def lone_solves(wl, slots, kernel, **cfg):
    """The queue solved in chunks of `slots` instances on a plain handle, one whole-solve launch each."""
    import dataclasses
    out = {f: [] for f in FIELDS}
    for lo in range(0, wl.B, slots):
        hi = min(lo + slots, wl.B)
        part = dataclasses.replace(wl, B=hi - lo, x0=wl.x0[lo:hi], u_init=wl.u_init[lo:hi], t0=wl.t0[lo:hi])
        s = make_solver(part, ragged_schedule=-1, **cfg)
        s.setKernel(kernel)
        s.setDispatchBatch(slots)
        s.solve(part.t0, part.x0, part.u_init)
        got = {"X": s.X(), "U": s.U(), "cost": s.cost(), "status": s.status(), "iters": s.iters(), "trace_last": s.traceLast(), "dV": s.dV()}
        for f in FIELDS:
            out[f].append(np.array(got[f]))
    return {f: np.concatenate(v) for f, v in out.items()}


def stream(wl, slots, kernel, span=0, **cfg):
    import dataclasses
    s = make_solver(dataclasses.replace(wl, B=slots), **cfg)
    s.setKernel(kernel)
    r = s.solveStream(wl.t0, wl.x0, wl.u_init, span=span)
    return s, r


@pytest.mark.parametrize("N,slots,kernel,max_iter,span,constrained", [
    (5000, 1024, "quad", 500, 0, False),     # five rounds of refills, to convergence (five of 4096 such instances never converge: 500)
    (3000, 512, "quad", 40, 16, False),      # an instance's own max_iter-th iteration inside a round (40 = 2 x 16 + 8)
    (2500, 512, "quad", 90, 0, True),        # box-constrained
    (2000, 256, "2w", 100, 8, False),        # the two-wave kernel, shorter rounds
    (700, 1024, "quad", 60, 0, False),       # fewer instances than slots
    (1025, 1024, "quad", 30, 5, False),      # one instance left for the second fill
])
def test_every_instance_of_the_queue_returns_the_bits_of_its_lone_solve(N, slots, kernel, max_iter, span, constrained):
    wl = workloads.cartpole_batch(B=N, T=100, seed=N + max_iter, constrained=constrained)
    cfg = dict(max_iter=max_iter, with_input_constraint=constrained, trace_level=0)
    want = lone_solves(wl, slots, kernel, **cfg)
    s, r = stream(wl, slots, kernel, span=span, **cfg)
    assert s.kernelName() == {"quad": "ddp_solve_quad_kernel", "2w": "ddp_solve_tpi2w_kernel"}[kernel]
    got = {f: getattr(r, f) for f in FIELDS}
    for f in FIELDS:
        assert np.array_equal(got[f], want[f], equal_nan=True), (f, int((got[f] != want[f]).sum()), np.flatnonzero((got[f] != want[f]).reshape(N, -1).any(axis=1))[:8])
    assert r.rounds >= 1 and r.device_ms > 0
    if max_iter == 40:
        assert want["iters"].max() == 40 and (want["status"][want["iters"] == 40] == 0).any()  # (the cap IS exercised in this case)
    assert (want["iters"] < 16).sum() > 0.2 * N and want["iters"].max() > 16  # ragged: slots are refilled in mid-queue
    if not constrained and N <= 3000:
        ref = oracle_batch(wl, **{k: v for k, v in cfg.items() if k != "trace_level"})
        assert np.array_equal(got["iters"], ref.iters) and np.array_equal(got["status"], ref.status)
        assert (np.abs(got["X"] - ref.X) / (1.0 + np.abs(ref.X))).max() < 1e-9


def test_stream_on_bipedal_and_reuse_of_the_handle():
    wl = workloads.bipedal_batch(B=900, T=300, seed=5)
    cfg = dict(max_iter=60, trace_level=0)
    want = lone_solves(wl, 256, "quad", **cfg)
    s, r = stream(wl, 256, "quad", **cfg)
    for f in FIELDS:
        assert np.array_equal(getattr(r, f), want[f], equal_nan=True), f
    r2 = s.solveStream(wl.t0, wl.x0, wl.u_init)  # the handle again: nothing of the first queue may linger
    for f in FIELDS:
        assert np.array_equal(getattr(r2, f), want[f], equal_nan=True), f
    # ... and an ordinary batched solve on the same handle afterwards
    import dataclasses
    part = dataclasses.replace(wl, B=256, x0=wl.x0[:256], u_init=wl.u_init[:256], t0=wl.t0[:256])
    s.solve(part.t0, part.x0, part.u_init)
    assert np.array_equal(s.X(), want["X"][:256]) and np.array_equal(s.iters(), want["iters"][:256])


def test_stream_refuses_shapes_without_resumable_kernels():
    wl = workloads.manipulator_batch(B=64, T=20, seed=3)
    s = make_solver(wl, max_iter=10)
    with pytest.raises(RuntimeError, match="resumable"):
        s.solveStream(wl.t0, wl.x0, wl.u_init)


def test_cpp_stream_matches_lone_solves(tmp_path):
    """include/nmpc_amd/DDPSolverBatch.hpp: solveStream over nmpc_hip_ddp_solve_stream — examples/cartpole_stream.cpp runs 3000 problems
    through 512 slots and compares every one bit for bit with solve() on chunks."""
    import os
    import subprocess

    from nmpc_amd import build as hip_build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cartpole_stream")
    libdir = os.path.dirname(hip_build.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", f"-I{root}/include", f"{root}/examples/cartpole_stream.cpp", f"-L{libdir}", "-lnmpc_hip_ddp",
           f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "3000", "512"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and "STREAM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
