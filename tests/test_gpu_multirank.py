"""The N > 1 path on the one-GPU test box: two ranks, both on device 0, each solving its contiguous shard with the HIP kernels;
the gathered records must equal the unsharded HIP solve bit for bit (instances are independent and the kernels deterministic).
With one device RCCL cannot form the group (one device per rank), so the collective runs over gloo here — the sharding
arithmetic, the packing and the HIP path per rank are the same as under nccl on a multi-GPU node.  bench.py is run the same way."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import nmpc_amd
from nmpc_amd import sharding, workloads
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
for gen, kw, B, T, mi in (("cartpole_batch", dict(), 203, 40, 6), ("quadrotor_batch", dict(fp32=True), 75, 20, 3)):
    wl = getattr(workloads, gen)(B=B, T=T, seed=5, **kw)           # uneven shards on purpose
    lo, hi = sharding.shard_range(B, rank, world)
    def solve(x0, u0, t0):
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), x0.shape[0], device=0)
        c = s.config(); c.print_level = 0; c.horizon_steps = T; c.max_iter = mi
        s.solve(t0, x0, u0)
        return sharding.pack_results(s.X(), s.U(), s.cost(), s.status(), s.iters()), s.kernelName()
    rec, kname = solve(wl.x0[lo:hi], wl.u_init[lo:hi], wl.t0[lo:hi])
    allrec = sharding.all_gather_records(torch.from_numpy(rec), B).numpy()
    if rank == 0:
        want, _ = solve(wl.x0, wl.u_init, wl.t0)
        assert allrec.shape == want.shape, (allrec.shape, want.shape)
        assert np.array_equal(allrec, want), gen + ": gathered shards differ from the unsharded HIP solve"
    print("rank", rank, gen, kname, "ok", flush=True)
dist.barrier(); dist.destroy_process_group()
"""


def _torchrun(args, port, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), *args],
                          capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_two_ranks_on_one_gpu_gather_equals_unsharded_hip_solve(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    r = _torchrun([str(script)], 29641)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok") >= 4, r.stdout  # two ranks x two models


def test_bench_under_two_ranks():
    """bench.py launched the way the driver launches it for N > 1 (both ranks share device 0 here): one JSON line from rank 0,
    n_gpus = 2, the whole-job value is the sum of both ranks' work, per-rank solve and gather times are reported."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                   "--no-extra-modes"], 29643)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    cfg = d["config"]
    assert len(cfg["per_rank_solve_ms"]) == 2 and len(cfg["per_rank_gather_ms"]) == 2
    assert 7000 < cfg["instance_iterations_per_step"] / 8  # two shards of 4096 instances, ~7.3 iterations each
    it_per_step = cfg["instance_iterations_per_step"] / 4096
    assert abs(d["value"] - it_per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
