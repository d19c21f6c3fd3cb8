#!/usr/bin/env python3
"""Benchmark of the MI355X FMPC path; `python bench.py --workload fmpc ...` lands here, same contract:

    python bench.py --workload fmpc --gpus N --steps K --warmup W

Workload: the reference's cart-pole FMPC problem and solver settings (nmpc_fmpc/tests/src/TestFmpcCartPole.cpp:32-267,303-317:
nx = 4, nu = 1, 4 inequality rows, horizon 2 s / 10 ms = 200 steps, max_iter = 5), batch = 4096 instances per GPU, fp64, every
instance from Variable::reset(0, 0, 0, 1, 1) (:329-330) and its own initial state around the hanging position.  One STEP = one
batched FmpcSolver::solve through the C-ABI (all 5 iterations run: the start is far from the solution); before each step the
resident variable is restored from a device copy of the initial guess (a device-to-device copy outside the solver).  An FMPC
iteration is one FmpcSolver::procOnce (FmpcSolver.hpp:356-491); executed iterations are counted from traceDataList().
value = n_gpus * K * (executed instance-iterations per solve / batch) / t.  Weak scaling, no collective in the data path; the
final variables of every rank are gathered once after the timed job (as in bench.py).

roofline: the dominant kernel is the Riccati kernel (fmpc_riccati_fused_kernel at this batch size: the coefficient records are computed
by its producer waves into LDS, so the contract bytes below — the reference's materialised dataflow — are more than it moves).  Its algorithmic bytes per launch = batch x T x (coefficient record read
by the backward recursion + gain record written by it + A, B, x_bar, K, k read by the forward recursion + dx, du written) x 8;
its average duration comes from HIP events around every launch (config.time_kernels) in a second pass of K steps outside the
timed region (the timed region replays the hipGraph of the solve, where single kernels cannot be bracketed).
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MODEL = "fmpc_cartpole"
BATCH, HORIZON, MAX_ITER = 4096, 200, 5


def workload(B: int, seed: int):
    """Initial states: cart position U[-1, 1], pole angle pi + U[-0.3, 0.3] (hanging, the test's start :327), velocities 0."""
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 4))
    x0[:, 0] = rng.uniform(-1, 1, B)
    x0[:, 1] = np.pi + rng.uniform(-0.3, 0.3, B)
    return x0


def cpu_baseline(x0, T, max_iter, target_seconds, host_cores):
    """The CPU FMPC oracle (kind "port") on a bounded sample of the same workload: thread sweep, best rate."""
    from oracle import fmpc as O
    affinity, quota = host_cores()
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    build_dir = tempfile.mkdtemp(prefix="oracle_fmpc_native_")
    O.lib(native=True, out_dir=build_dir)
    cfg = O.default_config(horizon_steps=T, max_iter=max_iter)
    p = O.default_params(MODEL)
    B = x0.shape[0]

    def run(nb, threads):
        idx = np.arange(nb) % B
        var = O.Variable.reset(MODEL, T, batch=nb)
        t = time.perf_counter()
        r = O.solve_batch(MODEL, cfg, p, 0.0, x0[idx], var, n_threads=threads, native=True, out_dir=build_dir)
        return int(r.iters.sum()), time.perf_counter() - t

    it1, sec1 = run(32, 1)
    rate1 = it1 / max(sec1, 1e-9)
    per_solve = it1 / 32.0
    legs = sorted({1, min(16, usable), min(64, usable), usable})
    budget = target_seconds / len(legs)
    sweep = {}
    for th in legs:
        guess = rate1 * th * (0.5 if th > 1 else 1.0)
        nb = int(max(8 * th, 32, guess * budget / max(per_solve, 1.0)))
        it, sec = run(nb, th)
        sweep[th] = {"instance_iterations_per_s": it / sec, "solves": nb, "seconds": sec}
    best = max(sweep, key=lambda k: sweep[k]["instance_iterations_per_s"])
    return {
        "value": sweep[best]["instance_iterations_per_s"] / B,
        "unit": "FMPC iterations/s (batch=%d)" % B,
        "cores": best,
        "kind": "port",
        "sample": "oracle/fmpc_oracle.hpp (%s) built -O3 -march=native on this host; the workload's %d instances cycled, max_iter "
                  "%d; thread sweep %s, threads pinned, dynamic chunks; best: %d threads, %d solves in %.1f s.  Host: %d CPUs in "
                  "the affinity mask, cgroup quota %s" % (MODEL, B, max_iter, legs, best, sweep[best]["solves"],
                                                         sweep[best]["seconds"], affinity,
                                                         "none" if quota is None else "%.1f CPUs" % quota),
        "instance_iterations_per_s": sweep[best]["instance_iterations_per_s"],
        "thread_sweep": {str(k): round(v["instance_iterations_per_s"], 1) for k, v in sweep.items()},
        "host_cpus_affinity": affinity,
        "host_cpu_quota": quota,
    }


def setup(B: int, T: int, max_iter: int, seed: int, device_index: int):
    """The solver, the step closure (restore the resident initial guess, one batched solve through the C-ABI) and the host x0."""
    import torch
    from nmpc_amd import fmpc as F
    dev = torch.device("cuda", device_index)
    prob = F.FmpcProblemCartPole(0.01)
    x0 = workload(B, seed)
    solver = F.FmpcSolverBatch(prob, B, T, device=device_index)
    solver.config().max_iter = max_iter
    solver._push_config()
    solver._prob_for_bench = prob
    L, h = solver._L, solver._h
    # the initial guess, resident on the device in the boundary layout; restored before every solve
    init = F.Variable.make(prob, T, B)
    init.reset(0.0, 0.0, 0.0, 1.0, 1.0)
    d_init = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in init.arrays()]
    d_eps = torch.full((B,), 1e-4, dtype=torch.float64, device=dev)
    d_x0 = torch.from_numpy(x0).to(dev)
    d_t0 = torch.zeros(B, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    keep = (d_init, d_eps, d_x0, d_t0)

    def step(_keep=keep):
        F.check(L.nmpc_hip_fmpc_set_variable(h, *[a.data_ptr() for a in d_init], d_eps.data_ptr(), 1))
        F.check(L.nmpc_hip_fmpc_solve_device(h, d_t0.data_ptr(), d_x0.data_ptr(), None))

    return solver, step, x0


def riccati_words(n: int, m: int):
    """Algorithmic doubles per (instance, timestep) of the Riccati kernel: (total, coefficient record, gain record, forward reads)."""
    coef = 2 * n * n + 2 * n * m + m * m + 2 * n + m
    gain = m + m * n + n + n * n
    fwd = n * n + n * m + n + m * n + m
    return coef + gain + fwd + (n + m), coef, gain, fwd


def profile_kernels(solver, step, n_prof: int):
    """Second pass with an event pair around every kernel launch (config.time_kernels): per-class ms sums and launch counts."""
    from nmpc_amd import fmpc as F
    solver.config().time_kernels = True
    solver._push_config()
    k_ms = {k: 0.0 for k in F.KERNEL_CLASSES}
    k_n = {k: 0 for k in F.KERNEL_CLASSES}
    for _ in range(n_prof):
        step()
        d = solver.computationDuration()
        for k in F.KERNEL_CLASSES:
            k_ms[k] += d.kernels[k]
            k_n[k] += d.launches[k]
    solver.config().time_kernels = False
    solver._push_config()
    return k_ms, k_n


def traffic_entry(B: int, T: int):
    """(bytes per Riccati launch, source) from profiles/hbm_traffic.json if measured on these device sources, else (None, why)."""
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        from nmpc_amd import build as hip_build
        src_hash = hip_build.source_hash()
        entry = json.load(open(traffic_file)).get("fmpc")
        if entry and entry.get("batch") == B and entry.get("horizon") == T:
            if entry.get("source_hash") != src_hash:
                return None, "profiles/hbm_traffic.json holds a measurement of other device sources (hash %s, now %s): not used" % (
                    entry.get("source_hash"), src_hash)
            return entry.get("hbm_bytes_per_launch"), entry.get("source")
    except Exception:
        pass
    return None, None


def secondary_leg(device_index: int, seed: int, min_seconds: float = 0.5):
    """The FMPC workload as a secondary leg of bench.py's default line: >= min_seconds of timed steps, the Riccati kernel's roofline."""
    from nmpc_amd import fmpc as F
    B, T, max_iter = BATCH, HORIZON, MAX_ITER
    solver, step, _ = setup(B, T, max_iter, seed, device_index)
    prob = solver._prob_for_bench
    L, h = solver._L, solver._h
    for _ in range(3):
        step()
    F.check(L.nmpc_hip_fmpc_synchronize(h))
    t0 = time.perf_counter()
    step()
    F.check(L.nmpc_hip_fmpc_synchronize(h))
    n_steps = int(max(4, np.ceil(min_seconds / max(time.perf_counter() - t0, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    F.check(L.nmpc_hip_fmpc_synchronize(h))
    dt = time.perf_counter() - t0
    iters = solver.iters()
    k_ms, k_n = profile_kernels(solver, step, 4)
    words = riccati_words(prob.state_dim, prob.input_dim)[0]
    bytes_per_launch = float(B) * T * words * 8.0
    ric_ms = k_ms["riccati"] / max(k_n["riccati"], 1)
    ach = bytes_per_launch / (ric_ms * 1e-3) / 1e9
    traffic, src = traffic_entry(B, T)
    out = {"workload": "FMPC cart-pole (TestFmpcCartPole): nx=4, nu=1, 4 inequality rows, T=%d, max_iter=%d, batch=%d, fp64" % (T, max_iter, B),
           "metric": "FMPC iterations/s", "value": n_steps * (float(iters.sum()) / B) / dt, "ms_per_step": 1e3 * dt / n_steps,
           "timed_steps": n_steps, "timed_seconds": dt, "dtype": "f64",
           "kernel": next(k for k in solver.kernelNames() if "riccati" in k), "kernels": solver.kernelNames(),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "kernel_ms_avg": ric_ms, "launches_timed": k_n["riccati"],
                        "algorithmic_bytes_per_launch": bytes_per_launch,
                        "share_of_solve_time": k_ms["riccati"] / max(sum(k_ms.values()), 1e-12)}}
    if traffic:
        out["roofline"]["traffic_source"] = src
        out["roofline"]["hbm_frac_measured"] = traffic / (ric_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    elif src:
        out["roofline"]["traffic_note"] = src
    out["traffic"] = traffic
    return out


def main(args, host_cores):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    import bench
    bench.check_world(args, world, n_dev)  # --gpus = ranks that joined, one device each (unless --share-devices)
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if n_dev >= world else "gloo"
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from nmpc_amd import fmpc as F

    B = args.batch or BATCH
    T = args.horizon or HORIZON
    max_iter = MAX_ITER
    solver, step, x0 = setup(B, T, max_iter, args.seed + 7919 * rank, device_index)
    prob = solver._prob_for_bench
    n, m, g = prob.state_dim, prob.input_dim, prob.ineq_dim
    L, h = solver._L, solver._h

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        step()
    F.check(L.nmpc_hip_fmpc_synchronize(h))
    t_solve = time.perf_counter()
    barrier()
    elapsed_local = time.perf_counter() - t_begin
    iters = solver.iters()
    status = solver.status()
    inst_it = float(iters.sum())

    # one gather of the final variables of every shard (outside the per-step data path)
    n_res = B * ((T + 1) * n + T * m)
    d_res = torch.empty(n_res, dtype=torch.float64, device=dev)
    F.check(L.nmpc_hip_fmpc_get(h, F.FIELD_X, d_res.data_ptr(), B * (T + 1) * n * 8, 1))
    F.check(L.nmpc_hip_fmpc_get(h, F.FIELD_U, d_res.data_ptr() + B * (T + 1) * n * 8, B * T * m * 8, 1))
    t_g = time.perf_counter()
    if world > 1:
        gdev = dev if backend == "nccl" else torch.device("cpu")
        d_all = torch.empty(world * n_res, dtype=torch.float64, device=gdev)
        dist.all_gather_into_tensor(d_all, d_res.to(gdev))
        torch.cuda.synchronize()
    gather_s = time.perf_counter() - t_g

    stats = torch.tensor([elapsed_local, inst_it, t_solve - t_begin], dtype=torch.float64)
    if world > 1:
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats.to(dev) if backend == "nccl" else stats)
        per_rank = [s.cpu() for s in allst]
    else:
        per_rank = [stats]
    elapsed = max(float(s[0]) for s in per_rank)
    job_inst_it = sum(float(s[1]) for s in per_rank)

    # second pass, outside the timed region: the same steps with an event pair around every kernel launch
    n_prof = max(1, min(args.steps, 20))
    k_ms, k_n = profile_kernels(solver, step, n_prof)

    if rank == 0:
        value = args.steps * (job_inst_it / B) / elapsed
        words, coef, gain, fwd = riccati_words(n, m)
        bytes_per_launch = float(B) * T * words * 8.0
        ric_ms = k_ms["riccati"] / max(k_n["riccati"], 1)
        achieved = bytes_per_launch / (ric_ms * 1e-3) / 1e9
        per_iter_ms = {k: (k_ms[k] / max(k_n["riccati"], 1)) for k in F.KERNEL_CLASSES}
        out = {
            "metric": "FMPC iterations/s (whole node), batch=%d, T=%d" % (B, T),
            "value": value,
            "unit": "FMPC iterations/s (one iteration = FmpcSolver::procOnce over a batch of %d instances per GPU)" % B,
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
                "workload": "FMPC cart-pole (the reference's TestFmpcCartPole problem and settings): nx=4, nu=1, 4 inequality rows, "
                            "T=%d, max_iter=%d, batch=%d per GPU, fp64, Variable::reset(0,0,0,1,1), x0 = (U[-1,1], pi+U[-0.3,0.3], 0, 0) "
                            "numpy default_rng seed %d" % (T, max_iter, B, args.seed),
                "iterations_per_step": max_iter,
                "instance_iterations_per_step": inst_it,
                "status_counts": {str(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))},
                "solves_per_s": world * args.steps * B / elapsed,
                "kernels": solver.kernelNames(),
                "kernel_ms_per_iteration": per_iter_ms,
                "lane_mapping": "an iteration of the fused sequence is three launches: Riccati (coefficient records by producer waves), delta, "
                                "tail (step length + update + the next iteration's barrier parameter, KKT-error terms and terminal record: a "
                                "workgroup per 16 instances walks the horizon in slices); "
                                "timestep-parallel kernels: one thread per (instance, timestep), arrays [timestep][element][instance]; "
                                "Riccati recursion: sixteen lanes per instance on v_mfma_f64_4x4x4 (16-instance workgroups, operands "
                                "staged through LDS) up to two workgroups per CU, one lane per instance beyond",
                "final_gather_ms": 1e3 * gather_s,
                "gather_backend": backend,
                "per_rank_solve_ms": [1e3 * float(s[2]) / args.steps for s in per_rank],
            },
            "instance_iterations_per_s": value * B,
            "roofline": {
                "bound": "hbm",
                "kernel": next(k for k in solver.kernelNames() if "riccati" in k),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "accounting": "algorithmic bytes: %d doubles per (instance, timestep) = coefficient record %d (read) + gain record %d "
                              "(written) + forward-recursion reads %d + dx, du %d (written); kernel time: HIP events around each of "
                              "the %d launches of a second pass of %d steps (config.time_kernels), outside the timed region"
                              % (words, coef, gain, fwd, n + m, k_n["riccati"], n_prof),
                "kernel_ms_avg": ric_ms,
                "launches_timed": k_n["riccati"],
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "share_of_solve_time": k_ms["riccati"] / max(sum(k_ms.values()), 1e-12),
            },
        }
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(x0, T, max_iter, args.cpu_seconds, host_cores)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "FMPC iterations/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        traffic, src = traffic_entry(B, T)
        if traffic:
            from nmpc_amd import build as hip_build
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = src
            out["roofline"]["traffic_source_hash"] = hip_build.source_hash()
            out["roofline"]["hbm_frac_measured"] = traffic / (ric_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        elif src:
            out["roofline"]["traffic_note"] = src
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
