#!/usr/bin/env python3
"""Benchmark of the MI355X DDP hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): DDP iterations/s (whole node), batch = 4096 cart-pole instances (nx=4, nu=1, T=100,
fp64) per GPU.  One STEP = one batched `solve()` through the C-ABI with the reference's default
DDPSolver::Configuration except max_iter = --iters-per-solve (default 8, the pre-convergence regime in which an
iteration is nominal: 1 backward + ~1.1 forward passes; the MPC callers of the reference run max_iter = 3).  A DDP
iteration is one DDPSolver::procOnce (DDPSolver.hpp:143-340: linearise + regularised backward pass + line-search
forward pass); instances that meet the reference's termination tests earlier stop earlier, and only iterations that
were actually executed are counted (sum of traceDataList().back().iter over the batch / 4096).  Inputs are already
resident in HBM.  value = n_gpus * K * (executed instance-iterations per solve / 4096) / t, t = max over ranks of the
wall time between two barrier + device-synchronise brackets.  Weak scaling: every rank owns its own 4096 instances; the
only collective is ONE all_gather of the final trajectories (RCCL over xGMI) at the end of the timed job.

The JSON line also carries
  roofline     HBM-roofline accounting of the solve kernel: algorithmic bytes (SURVEY.md §8 d formula with the
               measured backward / forward pass counts) / HIP-event kernel time measured on the launch stream;
  cpu_baseline the CPU oracle ("port" of the reference's Eigen path, oracle/) timed on this host's cores on a
               bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
LANE_MAPPINGS = {
    "ddp_solve_quad_kernel": "quad: 16 instances per workgroup of 4 wavefronts; backward pass on the fp64 matrix cores "
                             "(16 lanes per instance, v_mfma_f64_4x4x4), linearisation parallel over the horizon, "
                             "forward pass one lane per instance (master + helper wavefront)",
    "ddp_solve_tpi2w_kernel": "one lane per instance, 64 instances per workgroup, master + helper wavefront",
    "ddp_solve_tpi_kernel": "one lane per instance, 64 instances per single-wavefront workgroup",
    "ddp_solve_wpi_kernel": "one wavefront per instance: lane = timestep / matrix entry (v_mfma_f64_16x16x4) / step size",
}


# name -> (generator in nmpc_amd.workloads, per-GPU batch, horizon, description)
WORKLOADS = {
    "c2": ("cartpole_batch", 4096, 100,
           "C2 batched cart-pole swing-up: nx=4, nu=1, T=%d, batch=%d per GPU, fp64, "
           "x0~U([-1,1]x[-pi,pi]x[-1,1]x[-1,1]) splitmix64 seed %d, u_init=0, unconstrained"),
    "c3": ("bipedal_batch", 1024, 300,
           "C3 bipedal CoM-ZMP: nx=2, nu=1, T=%d, batch=%d per GPU, fp64, t0~U[0,17] s on the reference's ref_zmp / "
           "omega^2 schedule, splitmix64 seed %d, u_init=0"),
    "c4": ("quadrotor_batch", 8192, 50,
           "C4-shape quadrotor: nx=12, nu=4, T=%d, batch=%d per GPU, fp64 (BASELINE names fp32), hover-perturbed x0 "
           "splitmix64 seed %d, u_init=hover"),
    "c5": ("manipulator_batch", 8192, 30,
           "C5-shape manipulator: nx=14, nu=7, T=%d, batch=%d per GPU (65536 over 8 GPUs), fp64, splitmix64 seed %d"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 (default) is BASELINE.json's metric configuration; c3 / c4 / c5 are the other configs "
                         "(parity-test cases) measured with the same harness")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (0: the workload's own)")
    ap.add_argument("--horizon", type=int, default=0, help="horizon_steps (0: the workload's own)")
    ap.add_argument("--iters-per-solve", type=int, default=8)
    ap.add_argument("--mode", choices=("nominal", "m1", "m2"), default="nominal",
                    help="nominal (default): reference defaults with max_iter = --iters-per-solve.  m1 / m2 are the two "
                         "timing modes of SURVEY.md 8(d): m1 = termination tests disabled (k_rel_norm_thre = 0, "
                         "cost_update_thre = -inf), every instance executes exactly --iters-per-solve iterations unless "
                         "lambda exceeds lambda_max; m2 = solve to convergence with the reference defaults (max_iter 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0,
                    help="sizing target of the CPU baseline sample (the sustained all-core rate is ~3x below the probe: ~20 s)")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


def mode_config(mode: str, iters_per_solve: int) -> dict:
    """Configuration overrides of the three timing modes (same for the GPU solver and the CPU oracle)."""
    if mode == "m1":
        return dict(max_iter=iters_per_solve, k_rel_norm_thre=0.0, cost_update_thre=-1e300)
    if mode == "m2":
        return dict(max_iter=500)
    return dict(max_iter=iters_per_solve)


def cpu_baseline(wl, mode: str, iters_per_solve: int, target_seconds: float):
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload, all host cores."""
    import oracle
    cores = os.cpu_count() or 1
    build_dir = tempfile.mkdtemp(prefix="oracle_native_")
    cfg = oracle.default_config(horizon_steps=wl.T, **mode_config(mode, iters_per_solve))

    def run(nb, threads):
        # the sample is the workload's own instances, repeated cyclically when more than one batch is needed
        idx = np.arange(nb) % wl.B
        r = oracle.solve_batch(wl.model, cfg, wl.x0[idx], wl.u_init[idx], t0=wl.t0[idx], n_threads=threads,
                               want_gains=False, native=True, native_dir=build_dir)
        return r.total_iters, r.seconds

    probe = 16 * cores
    it, sec = run(probe, cores)
    rate = it / max(sec, 1e-9)  # instance-iterations / s
    nb = int(max(probe, rate * target_seconds / max(it / probe, 1.0)))  # iterations per solve as measured by the probe
    it, sec = run(nb, cores)
    nb1 = max(64, int(nb / cores / 4))
    it1, sec1 = run(nb1, 1)
    return {
        "value": (it / sec) / wl.B,  # batch(4096)-iterations / s
        "unit": "DDP iterations/s (batch=%d)" % wl.B,
        "cores": cores,
        "kind": "port",
        "sample": "%d solves (the workload's %d instances, cycled), mode %s, max_iter %d: %d iterations, %d threads, %.1f s; "
                  "1-core leg %d solves, %.1f s; oracle/ built -O3 -march=native"
                  % (nb, wl.B, mode, cfg.max_iter, it, cores, sec, nb1, sec1),
        "instance_iterations_per_s": it / sec,
        "instance_iterations_per_s_1core": it1 / sec1,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch  # device memory for the inputs + torch.distributed (RCCL); imported before the HIP library
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import nmpc_amd
    from nmpc_amd import _capi, workloads

    # per-rank shard: rank r owns instances [r*B, (r+1)*B) of the global splitmix64 stream
    gen, wl_batch, wl_horizon, wl_text = WORKLOADS[args.workload]
    wl = getattr(workloads, gen)(B=args.batch or wl_batch, T=args.horizon or wl_horizon, seed=args.seed + 7919 * rank)
    solver = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B, device=local_rank)
    cfg = solver.config()
    cfg.print_level = 0
    cfg.horizon_steps = wl.T
    for key, val in mode_config(args.mode, args.iters_per_solve).items():
        setattr(cfg, key, val)
    cfg.trace_level = 1

    d_x0 = torch.from_numpy(wl.x0).to(dev)
    d_u0 = torch.from_numpy(wl.u_init).to(dev)
    d_t0 = torch.from_numpy(wl.t0).to(dev)
    n_x = wl.B * (wl.T + 1) * wl.n
    n_u = wl.B * wl.T * max(wl.m, 1)
    d_res = torch.empty(n_x + n_u, dtype=torch.float64, device=dev)  # packed [X | U] send buffer
    d_all = torch.empty(world * (n_x + n_u), dtype=torch.float64, device=dev) if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        solver.solveDevice(d_t0.data_ptr(), d_x0.data_ptr(), d_u0.data_ptr())

    for _ in range(args.warmup):
        step()
    solver.synchronize()
    solver.timingStats(reset=True)

    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        step()
    solver.synchronize()
    t_solve = time.perf_counter()
    # the one collective of the job: gather the final trajectories of every shard
    solver.getDevice(_capi.FIELD_X, d_res.data_ptr(), n_x * 8)
    solver.getDevice(_capi.FIELD_U, d_res.data_ptr() + n_x * 8, n_u * 8)
    solver.synchronize()
    if world > 1:
        dist.all_gather_into_tensor(d_all, d_res)
    barrier()
    t_end = time.perf_counter()

    elapsed = torch.tensor([t_end - t_begin, t_end - t_solve], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed, gather_s = float(elapsed[0]), float(elapsed[1])

    n_solves, total_ms, kernel_ms = solver.timingStats()
    tr = solver.trace()  # (B, max_iter+1, 12) of the last solve
    iters = solver.iters()
    rows = tr[:, 1:, :]
    executed = rows[:, :, 0] > 0
    n_it = int(executed.sum())
    n_bw = float(rows[:, :, _capi.TRACE_COLUMNS.index("n_backward")][executed].sum()) / max(n_it, 1)
    n_fw = float(rows[:, :, _capi.TRACE_COLUMNS.index("n_forward")][executed].sum()) / max(n_it, 1)
    inst_it_per_solve = float(iters.sum())  # this rank's shard
    # whole-job count: every rank solves its own instances (different seeds), so sum the executed iterations over ranks
    job_it = torch.tensor([inst_it_per_solve], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(job_it, op=dist.ReduceOp.SUM)
    job_it_per_solve = float(job_it[0])
    status = solver.status()

    if rank == 0:
        words = workloads.algorithmic_words_per_instance_iteration(wl.n, wl.m, wl.T, n_bw, n_fw)
        fused = workloads.fused_words_per_instance_iteration(wl.n, wl.m, wl.T, n_bw, n_fw)
        k_ms = kernel_ms / max(n_solves, 1)
        bytes_per_launch = words * 8.0 * inst_it_per_solve
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        value = args.steps * (job_it_per_solve / wl.B) / elapsed
        out = {
            "metric": "DDP iterations/s (whole node), batch=%d, T=%d" % (wl.B, wl.T),
            "value": value,
            "unit": "DDP iterations/s (one iteration = procOnce over a batch of %d instances per GPU)" % wl.B,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (wl_text % (wl.T, wl.B, args.seed))
                            + ", default DDPSolver::Configuration with max_iter = iterations_per_step",
                "mode": args.mode,
                "iterations_per_step": args.iters_per_solve if args.mode != "m2" else 500,
                "solves_per_s": world * args.steps * wl.B / elapsed,
                "iteration_histogram": {str(k): int(v) for k, v in zip(*np.unique(iters, return_counts=True))},
                "instance_iterations_per_step": job_it_per_solve,
                "backward_passes_per_iteration": n_bw,
                "forward_passes_per_iteration": n_fw,
                "status_counts": {str(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))},
                "lane_mapping": LANE_MAPPINGS.get(solver.kernelName(), solver.kernelName()),
                "final_gather_ms": 1e3 * gather_s,
            },
            "instance_iterations_per_s": value * wl.B,
            "roofline": {
                "bound": "hbm",
                "kernel": solver.kernelName() + "<%s>" % wl.model,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel_ms_avg": k_ms,
                "launches_timed": int(n_solves),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_instance_iteration": words * 8.0,
                "fused_lower_bound_bytes_per_instance_iteration": fused * 8.0,
            },
        }
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, args.mode, args.iters_per_solve, args.cpu_seconds)
            except Exception as e:  # the GPU number stands on its own; say why the baseline is missing
                out["cpu_baseline"] = {"value": None, "unit": "DDP iterations/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                if (args.workload == "c2" and args.mode == "nominal" and tf.get("batch") == wl.B
                        and tf.get("iterations_per_step") == args.iters_per_solve):
                    out["roofline"]["traffic"] = tf.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = tf.get("source")
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
