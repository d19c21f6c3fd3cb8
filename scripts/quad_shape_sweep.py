"""Boundary sweep of the quad kernel against the CPU oracle: batch sizes around its 16-instance workgroups, horizons
around its 16-timestep chunks and 4-timestep forward groups, with and without BoxQP / reg_type 2.  Test infrastructure
(imports oracle/): prints one line per case and a verdict."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import nmpc_amd, oracle
from nmpc_amd import workloads
import test_gpu_parity as tp

os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
bad = 0
n = 0
for model in ("cartpole", "bipedal"):
    for B in (1, 15, 16, 17, 31, 33, 64, 65):
        for T in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 100):
            for variant in ("plain", "reg2", "box"):
                if variant == "box" and model != "cartpole":
                    continue
                if (B + T) % 3 != 0 and not (B in (1, 17) or T in (1, 16, 17)):
                    continue  # thin the grid
                cfg = dict(max_iter=12)
                if model == "cartpole":
                    wl = workloads.cartpole_batch(B=B, T=T, seed=1000 + 7 * B + T, constrained=(variant == "box"))
                else:
                    wl = workloads.bipedal_batch(B=B, T=T, seed=2000 + 7 * B + T)
                if variant == "reg2":
                    cfg["reg_type"] = 2
                if variant == "box":
                    cfg["with_input_constraint"] = True
                s = tp.make_solver(wl, **cfg)
                s.solve(wl.t0, wl.x0, wl.u_init)
                assert s.kernelName() == "ddp_solve_quad_kernel"
                ref = tp.oracle_batch(wl, **cfg)
                n += 1
                try:
                    tp.check_against_oracle(wl, s, ref)
                except AssertionError as e:
                    mask = tp.decision_stable_mask(wl, ref, **cfg)
                    try:
                        tp.check_against_oracle(wl, s, ref, mask=mask)
                        print(f"{model} B={B} T={T} {variant}: agrees on the {int(mask.sum())}/{B} decision-stable instances")
                    except AssertionError as e2:
                        bad += 1
                        print(f"{model} B={B} T={T} {variant}: MISMATCH {str(e2)[:200]}")
print(f"{n} cases, {bad} mismatches")
