"""Development check of the fp64 tile kernel on the reference's centroidal-motion problem (n 9, inputDim(t) in {16, 0}: gains in
natural layout, TileSolver64::stepGainsNatural) against the oracle and the wave-per-instance kernel; prints, does not assert.
    python scripts/tile64_centroidal_check.py [quick]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
import oracle  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def scaled_err(got, want):
    return float((np.abs(got - want) / (1.0 + np.abs(want))).max())


def run(B, T, seed, group=0, kernel=None, reps=1, **cfg):
    for k, v in (("NMPC_HIP_DDP_TILE64_GROUP", group), ("NMPC_HIP_DDP_KERNEL", kernel)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    wl = workloads.centroidal_batch(B=B, T=T, seed=seed)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    for k, v in cfg.items():
        setattr(c, k, v)
    ms = []
    for _ in range(reps):
        s.solve(wl.t0, wl.x0, wl.u_init)
        ms.append(s.computationDuration().opt)
    return wl, s, min(ms)


def compare(label, s, ref):
    st_ok = np.array_equal(s.status(), ref.status)
    it_ok = np.array_equal(s.iters(), ref.iters)
    ex, eu = scaled_err(s.X(), ref.X), scaled_err(s.U(), ref.U)
    ok = ref.status >= 0
    ek = scaled_err(s.kff()[ok], ref.k[ok]) if ok.any() else 0.0
    eK = scaled_err(s.Kfb()[ok], ref.K[ok]) if ok.any() else 0.0
    Jg, Jr = s.cost().sum(axis=1), ref.cost.sum(axis=1)
    ej = float((np.abs(Jg - Jr) / np.abs(Jr)).max())
    tl = np.array_equal(s.traceLast()[:, (0, 9, 10, 11)], ref.trace_last[:, (0, 9, 10, 11)])
    dims = np.array_equal(s.inputDimList(), ref.input_dim) if hasattr(ref, "input_dim") else None
    flag = "OK " if (st_ok and it_ok and tl and max(ex, eu, ek, eK) <= 1e-9 and ej <= 1e-10) else "BAD"
    print(f"{flag} {label:52s} {s.kernelName():24s} status {st_ok} iters {it_ok} trace_ints {tl} dims {dims} X {ex:.1e} U {eu:.1e} "
          f"k {ek:.1e} K {eK:.1e} J {ej:.1e}  kernel {s.computationDuration().opt:.2f} ms", flush=True)
    if not (st_ok and it_ok):
        bad = np.flatnonzero((s.status() != ref.status) | (s.iters() != ref.iters))
        print("    mismatching instances", bad[:10], "gpu", s.status()[bad[:10]], s.iters()[bad[:10]], "ref", ref.status[bad[:10]],
              ref.iters[bad[:10]])


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = [(64, 100, 1234, 0, dict(max_iter=4)), (64, 100, 1234, 32, dict(max_iter=4)), (96, 100, 7, 5, dict(max_iter=10)),
             (96, 60, 8, 32, dict(max_iter=6, reg_type=2)), (65, 7, 78, 32, dict(max_iter=4)), (3, 1, 79, 32, dict(max_iter=3)),
             (40, 100, 9, 32, dict(max_iter=6, alpha_list=np.array([1.0, 0.3, 0.1, 0.03]))), (128, 100, 80, 0, dict(max_iter=60))]
    if quick:
        cases = cases[:3]
    for B, T, seed, group, cfg in cases:
        wl, s, _ = run(B, T, seed, group, **cfg)
        ref = oracle.solve_batch(wl.model, oracle.default_config(horizon_steps=wl.T, **cfg), wl.x0, wl.u_init, t0=wl.t0, n_threads=8)
        compare(f"B {B} T {T} group {group} {({k: (v if np.isscalar(v) else 'list') for k, v in cfg.items()})}", s, ref)
        wl2, s2, _ = run(B, T, seed, 0, kernel="wpi", **cfg)
        print(f"      vs {s2.kernelName()}: status/iters equal {np.array_equal(s.status(), s2.status())} "
              f"{np.array_equal(s.iters(), s2.iters())}, X {scaled_err(s.X(), s2.X()):.1e} K {scaled_err(s.Kfb(), s2.Kfb()):.1e}; "
              f"that kernel against the oracle: k {scaled_err(s2.kff(), ref.k):.1e} K {scaled_err(s2.Kfb(), ref.K):.1e}", flush=True)
        os.environ["NMPC_HIP_DDP_TILE64_WIDE"] = "0"
        wl3, s3, _ = run(B, T, seed, group, **cfg)
        os.environ.pop("NMPC_HIP_DDP_TILE64_WIDE")
        same = all(np.array_equal(a, b) for a, b in ((s.status(), s3.status()), (s.iters(), s3.iters()), (s.X(), s3.X()), (s.U(), s3.U()),
                                                      (s.Kfb(), s3.Kfb()), (s.trace(), s3.trace())))
        print(f"      wide first pass == separate passes, bit for bit: {same}", flush=True)
    for B in (16, 256, 1024, 4096):
        for kernel in (None, "wpi"):
            wl, s, ms = run(B, 100, 1234, 0, kernel=kernel, reps=3, max_iter=8)
            its = int(s.iters().sum())
            print(f"centroidal B {B} T 100 max_iter 8: {s.kernelName()} {ms:.2f} ms, {its} instance-iterations -> "
                  f"{its / B / ms * 1e3:.0f} batch-it/s", flush=True)


if __name__ == "__main__":
    main()
