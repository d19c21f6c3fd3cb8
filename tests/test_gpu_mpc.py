"""GPU tests of the device-resident receding-horizon driver (nmpc_hip_ddp_mpc_run): the reference's
four closed-loop tests run as batches on the MI355X — instance 0 is the reference's own initial condition and has to
satisfy every EXPECT_LT of the C++ test; a few instances (0 and perturbed starts) are compared tick by tick with the CPU
oracle's restatement of the same loop (oracle.mpc_run, pinned in tests/test_oracle_pins.py).

Closed-loop tolerance: the loops are stable tracking problems, so rounding differences do not grow; |dx|, |du0| <=
1e-7 (1 + |ref|) over the whole run (thousands of solves), iteration counts equal in >= 99 % of the ticks (a tick
whose termination test sits within rounding of its threshold may stop one iteration apart).
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

CL_TOL = 1e-7


def scaled_err(got, want):
    return float((np.abs(got - want) / (1.0 + np.abs(want))).max())


def check_against_oracle(log, b, ref):
    assert np.array_equal(log.m0[b], ref.m0)
    assert scaled_err(log.t[b], ref.t) <= 1e-12
    assert scaled_err(log.x[b], ref.x) <= CL_TOL
    assert scaled_err(log.u0[b], ref.u0) <= CL_TOL
    assert scaled_err(log.x_final[b], ref.x_final) <= CL_TOL
    assert abs(log.t_final[b] - ref.t_final) <= 1e-9
    assert float((log.iters[b] == ref.iters).mean()) >= 0.99


def bipedal_ref_zmp(t, end_t=20.0):
    t = t + 1e-6
    if t <= 1.5 or t >= end_t - 1.5:
        return 0.0
    return 0.15 if int(np.floor((t - 1.0) / 1.0)) % 2 == 0 else -0.15


def test_bipedal_closed_loop_on_device():
    """TestDDPBipedal.cpp:162-279 — dt 0.01, T = 300, 2000 ticks (20 s)."""
    import nmpc_amd

    B, T, ticks = 64, 300, 2000
    rng = np.random.default_rng(21)
    x0 = np.stack([rng.uniform(-0.02, 0.02, B), rng.uniform(-0.05, 0.05, B)], 1)
    x0[0] = 0.0
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemBipedal(), B)
    s.config().print_level = 0
    s.config().horizon_steps = T
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 1)), ticks, shift_warm_start=True)
    assert np.all(log.status >= 0)
    ref_zmp = np.array([bipedal_ref_zmp(t) for t in log.t[0]])
    assert np.all(np.abs(log.u0[0, :, 0] - ref_zmp) < 1e-2)  # :254
    assert abs(log.x_final[0, 0] - bipedal_ref_zmp(log.t_final[0])) < 1e-2  # :272
    assert abs(log.x_final[0, 1]) < 1e-2  # :273
    cfg = oracle.default_config(horizon_steps=T)
    for b in (0, 17, 63):
        check_against_oracle(log, b, oracle.mpc_run("bipedal", cfg, x0[b], ticks, shift_warm_start=True))
    # the handle holds the last solve
    assert scaled_err(s.X()[:, 1], log.x_final) <= 1e-12


@pytest.mark.parametrize("with_constraint", [False, True])
def test_vertical_motion_closed_loop_on_device(with_constraint):
    """TestDDPVerticalMotion.cpp:236-347 — T = 300, initial_lambda 1e-6, max_iter 3 from the second tick, [0, 30] N box,
    input dimension 1 / 2 / 0 along the run, 1000 ticks."""
    import nmpc_amd

    B, T, ticks = 64, 300, 1000
    rng = np.random.default_rng(22)
    x0 = np.stack([1.2 + rng.uniform(-0.05, 0.05, B), rng.uniform(-0.05, 0.05, B)], 1)
    x0[0] = (1.2, 0.0)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemVerticalMotion(), B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.initial_lambda = 1e-6
    c.with_input_constraint = with_constraint
    s.setInputLimits(np.array([0.0, 0.0]), np.array([30.0, 30.0]))
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 2)), ticks, shift_warm_start=True, max_iter_after_first=3)
    ref_h = np.where(log.t[0] + 1e-6 < 8.0, 1.0, 0.0)
    assert np.all(np.abs(log.x[0, :, 0] - ref_h) < 1.0)  # :305
    assert abs(log.x_final[0, 0]) < 1e-2 and abs(log.x_final[0, 1]) < 1e-2  # :334-335
    assert set(np.unique(log.m0[0])) == {0, 1, 2}
    if with_constraint:
        assert log.u0.min() >= -1e-9 and log.u0.max() <= 30.0 + 1e-9
    cfg = oracle.default_config(horizon_steps=T, initial_lambda=1e-6, with_input_constraint=int(with_constraint))
    for b in (0, 31):
        ref = oracle.mpc_run("vertical", cfg, x0[b], ticks, max_iter_after_first=3, shift_warm_start=True,
                             lower=[0.0, 0.0], upper=[30.0, 30.0])
        if with_constraint:
            # two identical actuators: the constrained QP is degenerate and the split of the total force between them
            # is not decision-stable even in the oracle — compare what is: the state and the total force
            assert scaled_err(log.x[b], ref.x) <= 1e-5
            assert scaled_err(log.u0[b].sum(-1), ref.u0.sum(-1)) <= 1e-4
            assert np.array_equal(log.m0[b], ref.m0)
        else:
            check_against_oracle(log, b, ref)


def test_centroidal_closed_loop_on_device():
    """TestDDPCentroidalMotion.cpp:239-365 — dt 0.03, T = 100, 100 ticks, max_iter 3 after the first solve, nu 16 / 0."""
    import nmpc_amd

    B, T, ticks = 8, 100, 100
    x0 = np.tile(np.array([0, 0, 1.0, 0, 0, 0, 0, 0, 0]), (B, 1))
    x0[1:, :3] += np.random.default_rng(23).uniform(-0.02, 0.02, (B - 1, 3))
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCentroidalMotion(), B)
    s.config().print_level = 0
    s.config().horizon_steps = T
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 16)), ticks, shift_warm_start=True, max_iter_after_first=3)
    ref_pos = np.array([[0.0 if t + 1e-6 < 1.5 else 0.5, 0.0, 1.0] for t in log.t[0]])
    assert np.all(np.linalg.norm(log.x[0, :, :3] - ref_pos, axis=1) < 1.0)  # :320
    assert np.linalg.norm(log.x_final[0, :3] - np.array([0.5, 0.0, 1.0])) < 1e-2  # :350
    assert np.linalg.norm(log.x_final[0, 3:]) < 1.0  # :351
    assert set(np.unique(log.m0[0])) == {0, 16}
    cfg = oracle.default_config(horizon_steps=T)
    for b in (0, 5):
        check_against_oracle(log, b, oracle.mpc_run("centroidal", cfg, x0[b], ticks, max_iter_after_first=3,
                                                    shift_warm_start=True))


def test_cartpole_swing_up_closed_loop_on_device():
    """TestDDPCartPole.cpp:236-403 with the launch-file parameters (tests/test/TestDDPCartPole.test:14-26): T = 200,
    +-15 N box, max_iter 3, MPC every 4 ms, plant stepped at 2 ms, 10 s (2500 ticks), running_u weight 0.01."""
    import nmpc_amd

    B, T, ticks = 64, 200, 2500
    rng = np.random.default_rng(24)
    x0 = np.tile(np.array([0.0, np.pi, 0.0, 0.0]), (B, 1))
    x0[1:, 0] += rng.uniform(-0.2, 0.2, B - 1)
    x0[1:, 1] += rng.uniform(-0.2, 0.2, B - 1)
    prob = nmpc_amd.DDPProblemCartPole(running_u=[0.01])
    s = nmpc_amd.DDPSolverBatch(prob, B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.max_iter = 3
    c.with_input_constraint = True
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 1)), ticks, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
    assert np.all(np.abs(log.x[0, :, 0]) < 1e2)  # :336
    xf = log.x_final[0]
    assert abs(xf[0]) < 1.0 and abs(xf[1]) < 1e-1 and abs(xf[2]) < 1.0 and abs(xf[3]) < 1e-1  # :351-354
    assert np.all(np.abs(log.u0) <= 15.0 + 1e-12)
    cfg = oracle.default_config(horizon_steps=T, max_iter=3, with_input_constraint=1)
    p = oracle.default_params("cartpole", running_u=0.01)
    for b in (0, 40):
        ref = oracle.mpc_run("cartpole", cfg, x0[b], ticks, params=p, shift_warm_start=False, sim_substeps=2,
                             sim_dt=0.002, lower=[-15.0], upper=[15.0])
        # a swing-up is a sensitive trajectory (the pole passes through the unstable region): rounding differences
        # are amplified during the swing and contracted again once balanced
        assert scaled_err(log.x[b], ref.x) <= 1e-4
        assert scaled_err(log.u0[b], ref.u0) <= 1e-3
        assert np.all(np.abs(log.x_final[b] - ref.x_final) <= 1e-6)


def test_mpc_run_misuse():
    import nmpc_amd

    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemBipedal(), 4)
    s.config().print_level = 0
    s.config().horizon_steps = 20
    x0, u = np.zeros((4, 2)), np.zeros((4, 20, 1))
    with pytest.raises(ValueError):  # the plant pattern needs stateEq(t, x, u, dt): bipedal has none
        s.mpcRun(0.0, x0, u, 2, shift_warm_start=False, sim_substeps=2, sim_dt=0.002, clamp_u0=False)
    with pytest.raises(ValueError):
        s.mpcRun(0.0, x0, u, 0)
    c = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(), 4)
    c.config().print_level = 0
    c.config().horizon_steps = 20
    with pytest.raises(RuntimeError):  # clamp_u0 without limits
        c.mpcRun(0.0, np.zeros((4, 4)), np.zeros((4, 20, 1)), 2, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
    with pytest.raises(ValueError):  # plant pattern without a plant step size
        c.mpcRun(0.0, np.zeros((4, 4)), np.zeros((4, 20, 1)), 2, shift_warm_start=False, clamp_u0=False)


def test_cpp_mirror_mpc_example(tmp_path):
    """examples/cartpole_mpc.cpp: DDPSolverBatch<Problem>::mpcRun (plain C++ over the C-ABI, g++) — instance 0 meets the
    reference's end-state bounds (TestDDPCartPole.cpp:351-354) and the batch equals the Python mirror's run."""
    import os
    import re
    import subprocess
    import nmpc_amd
    from nmpc_amd import build as hip_build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cartpole_mpc")
    libdir = os.path.dirname(hip_build.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", f"-I{root}/include", f"{root}/examples/cartpole_mpc.cpp", f"-L{libdir}",
           "-lnmpc_hip_ddp", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    B, ticks = 8, 2500
    r = subprocess.run([exe, str(B), str(ticks)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = re.findall(r"instance (\d+) t_final (\S+) x_final (\S+) (\S+) (\S+) (\S+) max\|u0\| (\S+) max\|pos\| (\S+)", r.stdout)
    assert len(rows) == B
    xf0 = [float(v) for v in rows[0][2:6]]
    assert abs(xf0[0]) < 1.0 and abs(xf0[1]) < 1e-1 and abs(xf0[2]) < 1.0 and abs(xf0[3]) < 1e-1
    assert all(float(row[6]) <= 15.0 + 1e-9 for row in rows)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(running_u=[0.01]), B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = 200
    c.max_iter = 3
    c.with_input_constraint = True
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    x0 = np.array([[0.05 * b, np.pi - 0.01 * b, 0.0, 0.0] for b in range(B)])
    log = s.mpcRun(0.0, x0, np.zeros((B, 200, 1)), ticks, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
    for b, row in enumerate(rows):
        assert abs(float(row[1]) - log.t_final[b]) <= 1e-9
        assert np.all(np.abs(np.array([float(v) for v in row[2:6]]) - log.x_final[b]) <= 1e-8)


def test_closed_loop_with_per_instance_problems():
    """Eight cart-poles with different masses / pole lengths / input boxes in one device-resident MPC run (plant pattern):
    the solver kernel AND the plant step of the advance kernel use each instance's own problem object and limits."""
    import nmpc_amd

    B, T, ticks = 8, 100, 300
    rng = np.random.default_rng(31)
    probs, oparams, lo, up = [], [], [], []
    for b in range(B):
        kw = dict(cart_mass=float(rng.uniform(0.8, 1.4)), pole_mass=float(rng.uniform(0.3, 0.7)),
                  pole_length=float(rng.uniform(1.5, 2.5)))
        probs.append(nmpc_amd.DDPProblemCartPole(running_u=[0.01], **kw))
        oparams.append(oracle.default_params("cartpole", running_u=0.01, **kw))
        lim = float(rng.uniform(10.0, 20.0))
        lo.append([-lim])
        up.append([lim])
    lo, up = np.array(lo), np.array(up)
    x0 = np.tile(np.array([0.0, np.pi, 0.0, 0.0]), (B, 1))
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(running_u=[0.01]), B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.max_iter = 3
    c.with_input_constraint = True
    s.setProblemBatch(probs)
    s.setInputLimitsBatch(lo, up)
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 1)), ticks, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
    cfg = oracle.default_config(horizon_steps=T, max_iter=3, with_input_constraint=1)
    for b in range(B):
        ref = oracle.mpc_run("cartpole", cfg, x0[b], ticks, params=oparams[b], shift_warm_start=False, sim_substeps=2,
                             sim_dt=0.002, lower=lo[b], upper=up[b])
        assert scaled_err(log.x[b], ref.x) <= 1e-4 and scaled_err(log.u0[b], ref.u0) <= 1e-3
        assert np.all(np.abs(log.u0[b]) <= up[b, 0] + 1e-12)
    # the instances really behave differently
    assert np.abs(log.x_final - log.x_final[0]).max() > 1e-2


def test_result_table_dump_and_phase_durations(tmp_path):
    """the reference's per-test result table (TestDDPBipedal.cpp:242,259-262: "time com_pos com_vel
    planned_zmp ref_zmp omega^2 iter") written from a batched run, in the format its plot script loads
    (tests/scripts/plotTestDDPBipedal.py:8: np.genfromtxt(path, names=True)); and computationDuration() with the backward /
    forward split of DDPSolver.h:219-247."""
    import nmpc_amd
    from nmpc_amd import result_tables

    B, T, n_ticks = 4, 300, 30
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem("bipedal"), B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.max_iter = 20
    t0 = np.array([0.0, 2.0, 7.4, 11.0])
    x0 = np.zeros((B, 2))
    log = s.mpcRun(t0, x0, np.zeros((B, T, 1)), n_ticks=n_ticks)

    def ref_zmp(t):  # the harness schedule of TestDDPBipedal.cpp:171-185 (only the columns' plumbing is tested here)
        return 0.1 * np.floor(t + 1e-6)

    def omega2(t):
        return 9.80665 / 1.0

    path = str(tmp_path / "TestDDPBipedalResult.txt")
    log.dump(path, 2, result_tables.bipedal_table(ref_zmp, omega2))
    header = open(path).readline().split()
    assert header == "time com_pos com_vel planned_zmp ref_zmp omega^2 iter".split()
    data = np.genfromtxt(path, dtype=None, delimiter=None, names=True)  # as the reference's plot script does
    assert data.shape == (n_ticks,)
    np.testing.assert_allclose(data["time"], log.t[2], rtol=1e-5)
    np.testing.assert_allclose(data["com_pos"], log.x[2, :, 0], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(data["planned_zmp"], log.u0[2, :, 0], rtol=1e-5, atol=1e-9)
    np.testing.assert_array_equal(data["iter"], log.iters[2])
    # the other three headers
    assert [n for n, _ in result_tables.vertical_motion_table(lambda t: 1.0)] == "time pos vel force ref_pos num_contact iter".split()
    assert [n for n, _ in result_tables.cart_pole_table(lambda t: 0.0)] == "time pos theta vel omega force ref_pos disturbance".split()
    cen = [n for n, _ in result_tables.centroidal_motion_table(lambda t: np.zeros((3, 16)), lambda t: (0, 0, 1))]
    assert cen[:4] == ["time", "pos_x", "pos_y", "pos_z"] and cen[-8:] == ["duration_" + f for f in (
        "setup", "opt", "derivative", "backward", "forward", "Q", "reg", "gain")] and len(cen) == 25
    # computationDuration(): the phases split the kernel time
    d = s.computationDuration()
    assert d.opt > 0 and d.backward > 0 and d.forward > 0
    assert d.backward + d.forward <= d.opt * 1.0001 and d.backward + d.forward >= 0.5 * d.opt
    assert log.duration is not None and log.duration.opt == d.opt


@pytest.mark.parametrize("model,B,T", [("cartpole", 4096, 100), ("cartpole", 8192, 100), ("quadrotor", 256, 50), ("quadrotor_f32", 256, 50)])
def test_phase_durations_on_every_kernel_family(model, B, T):
    """quad, two-wave, wave-per-instance and fp32 tile kernels all report the backward / forward split."""
    import nmpc_amd
    from nmpc_amd import workloads

    wl = (workloads.cartpole_batch(B=B, T=T, seed=2) if model == "cartpole" else
          workloads.quadrotor_batch(B=B, T=T, seed=2, fp32=model.endswith("f32")))
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.max_iter = 4
    s.solve(wl.t0, wl.x0, wl.u_init)
    d = s.computationDuration()
    print(f"{s.kernelName()}: opt {d.opt:.3f} ms = backward {d.backward:.3f} + forward {d.forward:.3f} + other {d.opt - d.backward - d.forward:.3f}")
    assert d.backward > 0.15 * d.opt and d.forward > 0.1 * d.opt and d.backward + d.forward <= 1.0001 * d.opt
