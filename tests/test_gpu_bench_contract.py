"""bench.py keeps its contract: one JSON line on stdout with the driver's keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--cpu-seconds", "0.5", "--min-seconds", "0.5", *extra], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = run_bench()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "batch=4096" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 2.0
    assert rf["kernel"].startswith("ddp_solve_quad_kernel")  # 4096 instances: one quad workgroup per CU
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert d["value"] > cb["value"]  # both are batch-iterations / s of the same workload
    # value = executed iterations: consistent with the step time
    it_per_step = d["config"]["instance_iterations_per_step"] / 4096
    assert abs(d["value"] - it_per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # SURVEY 8(d)'s two timing modes ride in the default line
    cfg = d["config"]
    assert cfg["m1_value"] > 0 and cfg["m2_value"] > 0
    # ... each with the CPU oracle's rate in the same mode on the same 4096 inputs beside it ("same M1/M2 modes")
    for m_name in ("m1", "m2"):
        assert 0 < cfg[m_name]["cpu_value"] < cfg[m_name]["value"] and cfg[m_name]["cpu_cores"] == cb["cores"], m_name
    assert abs(cfg["m2"]["cpu_mean_iterations"] - cfg["m2"]["mean_iterations"]) < 0.05 * cfg["m2"]["mean_iterations"]
    # consecutive batches on eight streams under the ragged-convergence schedule: the tails overlap and converged instances free their slots
    assert cfg["m2_overlapped_value"] > 2.5 * cfg["m2_value"] and cfg["m2_overlapped_value"] > 1.15 * cfg["m2_overlapped"]["whole_solve_launches_value"]
    assert cfg["m2_overlapped"]["launches_per_solve"] == 5 and cfg["m2_overlapped"]["handles"] == 8
    # M2 as a stream through one handle: slots refilled at round boundaries — no batch boundary, so well above the pool of batches
    assert cfg["m2_stream_value"] > 0.75 * cfg["m2_overlapped_value"] and cfg["m2_stream_sustained_value"] > 1.5 * cfg["m2_overlapped_value"]
    assert cfg["m2_stream"]["instances"] == 32768 and cfg["m2_stream"]["sustained"]["instances"] == 262144
    assert cfg["m2_stream"]["status_counts"].get("1", 0) >= 0.99 * 32768
    assert cfg["m1"]["status_counts"].get("1", 0) == 0 and cfg["m1"]["max_iterations"] <= 50
    assert cfg["m2"]["status_counts"].get("1", 0) >= 0.99 * 4096
    assert cfg["per_rank_solve_ms"] and abs(cfg["per_rank_solve_ms"][0] - d["ms_per_step"]) < 0.5 * d["ms_per_step"]
    assert cb["cores"] <= cb["host_cpus_affinity"] and "1" in cb["thread_sweep"]
    # the timed job lasts about --min-seconds whatever --steps says: blocks of --steps steps, their number fixed beforehand
    assert cfg["timed_seconds"] >= 0.4 and cfg["timed_steps_total"] == cfg["timed_blocks"] * 4 and cfg["timed_blocks"] >= 2
    assert cfg["first_block_ms_per_step"] > 0
    # per-block spread of the step time (one number hid a +-10 % box-to-box spread)
    assert 0 < cfg["block_ms_per_step_min"] <= cfg["block_ms_per_step_median"] <= cfg["block_ms_per_step_max"]
    assert cfg["value_at_fastest_block"] >= cfg["value_at_median_block"] > 0
    assert cfg["block_ms_per_step_min"] <= d["ms_per_step"] * 1.001
    # the oracle's first 256 instances against the GPU's solve, in the line itself
    assert cb["gpu_decisions_checked"] == 256 and cb["gpu_decisions_agree_frac"] >= 0.97 and cfg["oracle_agree_frac"] == cb["gpu_decisions_agree_frac"]
    assert "status" in cfg["gathered_record"]
    # both timing modes carry their own roofline (measured pass counts, kernel time, contract fraction)
    for key in ("roofline_m1", "roofline_m2"):
        r = d[key]
        assert r["bound"] == "hbm" and r["kernel_ms_avg"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert r["backward_passes_per_iteration"] >= 1.0 and r["forward_passes_per_iteration"] > 0
    # measured HBM traffic is only taken from profiles/hbm_traffic.json if it was measured on these device sources
    from nmpc_amd import build as hip_build
    if rf["traffic"] is not None:
        assert rf["traffic_source_hash"] == hip_build.source_hash()
    # the other workloads ride along as secondary legs: >= 0.5 s each, own kernel and roofline
    sec = d["secondary"]
    kernels = {"c3": "ddp_solve_quad_kernel<bipedal>", "c4": "ddp_solve_tile64_kernel<quadrotor_f32>", "c4f64": "ddp_solve_tile64_kernel<quadrotor>",
               "c5": "ddp_solve_tile64_kernel<manipulator>", "fmpc": "fmpc_riccati_fused_kernel", "centroidal": "ddp_solve_tile64_kernel<centroidal>"}
    for name, kernel in kernels.items():
        leg = sec[name]
        assert leg.get("error") is None, (name, leg)
        for key in ("value", "ms_per_step", "kernel", "roofline", "traffic", "timed_seconds", "workload"):
            assert key in leg, (name, key)
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["timed_seconds"] >= 0.5 and leg["kernel"] == kernel, (name, leg["kernel"])
        r = leg["roofline"]
        assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.01 < r["frac"] < 2.0 and r["kernel_ms_avg"] > 0
        assert leg["traffic"] == r["traffic"]
    assert sec["seconds_total"] < 60.0
    # c3 is 64 workgroups on 256 CUs — a latency chain: eight batches in flight (DDPSolverPool) is how a caller fills the chip
    assert sec["c3"]["pooled"]["value"] > 1.5 * sec["c3"]["value"] and sec["c3"]["pooled"]["handles"] == 8


def test_c4_runs_in_fp32_on_a_tile_kernel():
    d = run_bench("--workload", "c4", "--cpu-seconds", "0.5")
    assert d["dtype"] == "f32" and "batch=8192" in d["metric"] and "cost_update_thre=1e-3" in d["metric"]
    # the kept fraction of the fp32 parity (decisions equal to the float oracle's) rides in the line
    print("c4 decisions equal to the fp32 oracle's:", d["cpu_baseline"]["gpu_decisions_agree_frac"])
    assert d["cpu_baseline"]["gpu_decisions_checked"] == 256 and d["cpu_baseline"]["gpu_decisions_agree_frac"] >= 0.5
    # (eight iterations of a full chip: the tile kernel's float instantiation, ModelOpsTile32::useTile64Float)
    assert d["roofline"]["kernel"] == "ddp_solve_tile64_kernel<quadrotor_f32>" and "float instantiation" in d["config"]["lane_mapping"]
    # the headline is the threshold an fp32 cost can resolve; the reference's default rides along as the secondary number
    assert d["config"]["cost_update_thre"] == 1e-3 and "cost_update_thre = 0.001" in d["config"]["workload"]
    assert d["config"]["default_threshold_value"] > 0 and d["config"]["default_threshold"]["roofline"]["frac"] > 0
    assert d["config"]["fp32_tolerance_m2"]["status_counts"].get("1", 0) >= 0.98 * 8192
    d64 = run_bench("--workload", "c4f64", "--cpu-seconds", "0.5")
    assert d64["dtype"] == "f64" and d64["roofline"]["kernel"].startswith("ddp_solve_tile64_kernel")
    assert d64["cpu_baseline"]["value"] > 0 and d64["value"] > d64["cpu_baseline"]["value"]
    assert d["value"] > 0 and d64["value"] > 0
    # fp32 has to earn its precision: like for like (the reference's default threshold on both sides)
    assert d["config"]["default_threshold_value"] > 1.1 * d64["value"]


def test_c5_runs_on_the_fp64_tile_kernel_with_a_cpu_baseline():
    d = run_bench("--workload", "c5", "--cpu-seconds", "0.5")
    assert d["dtype"] == "f64" and "batch=8192" in d["metric"] and "T=30" in d["metric"]
    assert d["roofline"]["kernel"] == "ddp_solve_tile64_kernel<manipulator>"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["value"] > d["cpu_baseline"]["value"]
    assert d["config"]["timed_seconds"] >= 0.4


def test_other_workload_and_modes_run():
    d = run_bench("--workload", "c3", "--cpu-seconds", "0.5")
    assert "batch=1024" in d["metric"] and d["cpu_baseline"]["value"] > 0
    # 1024 instances occupy 64 of 256 CUs: the pooled rate (four batches in flight) rides along
    assert d["config"]["pooled_value"] > 1.5 * d["value"] and d["config"]["pooled"]["handles"] == 4
    d = run_bench("--mode", "m1", "--no-cpu-baseline")
    assert d["config"]["mode"] == "m1" and d["config"]["status_counts"].get("1", 0) == 0  # nobody may terminate early


def test_fmpc_workload_keeps_the_contract():
    d = run_bench("--workload", "fmpc")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert "FMPC iterations/s" in d["metric"] and "batch=4096" in d["metric"] and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["config"]["status_counts"] == {"5": 4096} and d["config"]["iterations_per_step"] == 5
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "fmpc_riccati_fused_kernel" and rf["launches_timed"] == 4 * 5
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.5
    assert abs(d["value"] - 5 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and d["value"] > cb["value"]
