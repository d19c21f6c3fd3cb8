"""Development check of the fp64 tile kernel (ddp_kernels_tile64.hpp) against the oracle and the wave-per-instance kernel:
prints the discrepancies instead of asserting, over group sizes, models and configurations.
    python scripts/tile64_check.py [quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
import oracle  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def scaled_err(got, want):
    return float((np.abs(got - want) / (1.0 + np.abs(want))).max())


def run(model, B, T, seed, group, kernel=None, constrained=False, **cfg):
    if group:
        os.environ["NMPC_HIP_DDP_TILE64_GROUP"] = str(group)
    else:
        os.environ.pop("NMPC_HIP_DDP_TILE64_GROUP", None)
    if kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    else:
        os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    wl = (workloads.quadrotor_batch(B=B, T=T, seed=seed, constrained=constrained) if model == "quadrotor" else
          workloads.manipulator_batch(B=B, T=T, seed=seed, constrained=constrained))
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    for k, v in cfg.items():
        setattr(c, k, v)
    if wl.limits is not None:
        s.setInputLimits(*wl.limits)
    t = time.time()
    s.solve(wl.t0, wl.x0, wl.u_init)
    wall = time.time() - t
    return wl, s, wall


def compare(label, wl, s, ref):
    st_ok = np.array_equal(s.status(), ref.status)
    it_ok = np.array_equal(s.iters(), ref.iters)
    ex = scaled_err(s.X(), ref.X)
    eu = scaled_err(s.U(), ref.U)
    ok = ref.status >= 0
    ek = scaled_err(s.kff()[ok], ref.k[ok]) if ok.any() else 0.0
    eK = scaled_err(s.Kfb()[ok], ref.K[ok]) if ok.any() else 0.0
    Jg, Jr = s.cost().sum(axis=1), ref.cost.sum(axis=1)
    ej = float((np.abs(Jg - Jr) / np.abs(Jr)).max())
    tl = np.array_equal(s.traceLast()[:, (0, 9, 10, 11)], ref.trace_last[:, (0, 9, 10, 11)])
    flag = "OK " if (st_ok and it_ok and tl and max(ex, eu, ek, eK) <= 1e-9 and ej <= 1e-10) else "BAD"
    print(f"{flag} {label:60s} {s.kernelName():24s} status {st_ok} iters {it_ok} trace_ints {tl} X {ex:.1e} U {eu:.1e} k {ek:.1e} "
          f"K {eK:.1e} J {ej:.1e}  kernel {s.computationDuration().opt:.2f} ms", flush=True)
    if not (st_ok and it_ok):
        bad = np.flatnonzero((s.status() != ref.status) | (s.iters() != ref.iters))
        print("    mismatching instances", bad[:10], "gpu", s.status()[bad[:10]], s.iters()[bad[:10]], "ref", ref.status[bad[:10]],
              ref.iters[bad[:10]])


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = []
    for model, T in (("quadrotor", 50), ("manipulator", 30)):
        for group in (0, 32, 5):
            cases.append((model, 96, T, 77, group, dict(max_iter=10)))
        cases.append((model, 96, T, 77, 32, dict(max_iter=10, reg_type=2)))
        cases.append((model, 96, T, 77, 7, dict(max_iter=6, alpha_list=np.array([1.0, 0.3, 0.1, 0.03]))))
        cases.append((model, 65, 7, 78, 32, dict(max_iter=4)))
        cases.append((model, 3, 1, 79, 32, dict(max_iter=3)))
        cases.append((model, 200, T, 80, 32, dict(max_iter=500)))
    if quick:
        cases = cases[:2] + cases[8:10]
    for model, B, T, seed, group, cfg in cases:
        wl, s, _ = run(model, B, T, seed, group, **cfg)
        ocfg = oracle.default_config(horizon_steps=wl.T, **cfg)
        ref = oracle.solve_batch(wl.model, ocfg, wl.x0, wl.u_init, t0=wl.t0, n_threads=8, want_alpha_hist=True)
        compare(f"{model} B {B} T {T} group {group} {({k: (v if np.isscalar(v) else 'list') for k, v in cfg.items()})}", wl, s, ref)
    # box constraints: against the wave-per-instance kernel (the oracle comparison needs the decision-stable mask: pytest)
    for model, T in (("quadrotor", 50), ("manipulator", 30)):
        cfg = dict(with_input_constraint=True, max_iter=10)
        wl, s, _ = run(model, 64, T, 31, 32, constrained=True, **cfg)
        wl2, s2, _ = run(model, 64, T, 31, 0, kernel="wpi", constrained=True, **cfg)
        same = (np.array_equal(s.status(), s2.status()), np.array_equal(s.iters(), s2.iters()),
                np.array_equal(s.qpRetval(), s2.qpRetval()), np.array_equal(s.qpFreeMask(), s2.qpFreeMask()))
        print(f"box {model}: {s.kernelName()} vs {s2.kernelName()}: status/iters/qp_ret/qp_free equal {same}, X {scaled_err(s.X(), s2.X()):.1e} "
              f"U {scaled_err(s.U(), s2.U()):.1e} K {scaled_err(s.Kfb(), s2.Kfb()):.1e}; kernel ms {s.computationDuration().opt:.2f} vs "
              f"{s2.computationDuration().opt:.2f}", flush=True)
    # throughput at the bench shapes
    for model, B, T in (("quadrotor", 8192, 50), ("manipulator", 8192, 30)):
        for kernel in (None, "wpi"):
            wl, s, _ = run(model, B, T, 1234, 0, kernel=kernel, max_iter=8)
            s.solve(wl.t0, wl.x0, wl.u_init)
            ms = s.computationDuration().opt
            its = int(s.iters().sum())
            print(f"{model} B {B} T {T} max_iter 8: {s.kernelName()} {ms:.2f} ms, {its} instance-iterations -> {its / B / ms * 1e3:.0f} batch-it/s",
                  flush=True)


if __name__ == "__main__":
    main()
