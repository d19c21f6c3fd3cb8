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
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                   "--no-extra-modes", "--min-seconds", "0.5"], 29643)
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


def test_plain_python_bench_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (--gpus was parsed and never read — the line said n_gpus = 1):
    bench.py starts the two ranks itself; rank 0 times the CPU oracle first (the other rank waits at a barrier), so a multi-GPU line has its
    cpu_baseline too.  --share-devices because this box has one GPU; without it a job with fewer devices than ranks is refused, loudly."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-extra-modes", "--cpu-seconds", "0.5",
            "--min-seconds", "0.5"]
    r = subprocess.run(base + ["--share-devices"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["config"]["per_rank_solve_ms"]) == 2
    assert d["config"]["gather_backend"] == "gloo"  # two ranks on one device; "nccl" (= RCCL) when every rank has its own
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and d["value"] > cb["value"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(base, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert r.returncode != 0 and "visible device" in (r.stdout + r.stderr) and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher that started another number of ranks than --gpus says is refused as well
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--share-devices", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-extra-modes", "--min-seconds", "0.2"], 29651)
    assert r.returncode != 0 and "--gpus 4" in (r.stdout + r.stderr)


def test_bench_c5_uneven_strong_split_gathers_the_unsharded_solve():
    """bench.py --workload c5 with ONE batch of 8200 + 1 instances cut over two ranks (4101 / 4100: uneven shards, padded to one
    fixed-size all-gather): the gathered records — X | U | cost | status | iters of every shard — equal the unsharded solve of
    the same batch bit for bit (both sides on the fp64 tile kernel), and the line says so."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--workload", "c5", "--gpus", "2", "--share-devices", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-extra-modes", "--min-seconds", "0.2", "--global-batch", "8201", "--verify-gather"], 29645)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "batch=8201" in d["metric"]
    cfg = d["config"]
    assert cfg["shard_sizes"] == [4101, 4100] and cfg["gather_verified"] is True
    assert d["roofline"]["kernel"] == "ddp_solve_tile64_kernel<manipulator>"
    it_per_step = cfg["instance_iterations_per_step"] / 8201
    assert abs(d["value"] - it_per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_bench_c4_fp32_two_ranks_at_global_batch_8192_equal_the_unsharded_solve():
    """fp32 at the reference's default threshold: 8192 instances run on the fp32 tile kernel, a lone 4096-instance handle would take the
    tile kernel's float instantiation — families that differ in the last bits.  The shards set the whole batch's size as their dispatch
    batch (bench.py --global-batch; nmpc_hip_ddp_set_dispatch_batch) and return the unsharded solve bit for bit."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--workload", "c4", "--gpus", "2", "--share-devices", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-extra-modes", "--min-seconds", "0.2", "--global-batch", "8192", "--cost-update-thre", "1e-7", "--verify-gather"], 29649)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["shard_sizes"] == [4096, 4096] and d["config"]["gather_verified"] is True and d["dtype"] == "f32"
    assert d["roofline"]["kernel"] == "ddp_solve_tile32_kernel<quadrotor_f32>"


def test_fmpc_bench_under_two_ranks():
    """bench.py --workload fmpc under two ranks (both on device 0): FMPC shards like DDP — independent instances, no collective in
    the data path; the job's value counts both ranks' iterations."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--workload", "fmpc", "--gpus", "2", "--share-devices", "--steps", "4", "--warmup", "1",
                   "--no-cpu-baseline", "--batch", "1024"], 29647)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64" and "FMPC" in d["metric"]
    assert len(d["config"]["per_rank_solve_ms"]) == 2
    # 5 iterations per instance per solve on each of the two ranks
    assert abs(d["value"] - 2 * 5 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "fmpc_riccati_fused_kernel" and 0 < d["roofline"]["frac"] < 1.5


def test_cpp_sharded_helper(tmp_path):
    """include/nmpc_amd/DDPSolverSharded.hpp (one host process, one handle per shard, one gather): two and three shards on the
    box's single device with the peer-copy gather, one shard through the RCCL all-gather (ncclCommInitAll needs distinct
    devices); gathered results must equal the unsharded solve bit for bit — fp64 quad kernel and fp32 tile kernel."""
    from nmpc_amd import build as hip_build

    exe = str(tmp_path / "sharded")
    libdir = os.path.dirname(hip_build.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-DNMPC_AMD_WITH_RCCL", f"-I{ROOT}/include", "-I/opt/rocm/include",
           os.path.join(ROOT, "examples", "sharded_solve.cpp"), f"-L{libdir}", "-lnmpc_hip_ddp", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (bit-for-bit needs the same lane mapping on both sides: 4000 / 3 and 4000 both run the quad kernel, 8400 / 2 and 8400 both
    # the two-wave kernel; across kernel families results agree to 1e-13 with identical decisions, test_gpu_parity.py)
    # Since round 5 every shard takes the family the WHOLE batch would get (nmpc_hip_ddp_set_dispatch_batch): 8000 / 2 runs the two-wave
    # kernel like the unsharded 8000, the fp32 quadrotor at 8192 / 2 the fp32 tile kernel the unsharded 8192 runs on.
    for args in (["cartpole", "203", "40", "2"], ["cartpole", "4000", "30", "3"], ["cartpole", "8400", "30", "2"], ["cartpole", "8000", "30", "2"],
                 ["quadrotor_f32", "75", "20", "2"], ["quadrotor_f32", "8192", "20", "2"], ["cartpole", "96", "30", "1", "rccl"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (args, r.stdout[-2000:], r.stderr[-2000:])
