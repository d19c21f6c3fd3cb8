"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/nmpc_hip_ddp.h declares,
host-side logic of the Python mirror, workload generators, sharding + the gloo world-size-2 gather path."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import nmpc_amd
import oracle
from nmpc_amd import _capi, sharding, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------
# C-ABI surface
# ---------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nmpc_hip_ddp.h")).read()
    declared = set(re.findall(r"\b(nmpc_hip_ddp_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found in the header"
    L = _capi.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in nmpc_hip_ddp.h but not exported"
    assert declared == set(_capi.EXPORTS), (declared ^ set(_capi.EXPORTS))
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.lib_path()], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), name


def test_model_registry_and_blob_sizes():
    L = _capi.load()
    names = []
    for i in range(L.nmpc_hip_ddp_model_count()):
        p = C.c_char_p()
        assert L.nmpc_hip_ddp_model_name(i, C.byref(p)) == 0
        names.append(p.value.decode())
    assert set(names) >= {"cartpole", "bipedal", "vertical", "centroidal", "quadrotor", "manipulator", "planar_vtol"}
    expect = {"cartpole": (4, 1, False), "bipedal": (2, 1, False), "vertical": (2, 2, True),
              "centroidal": (9, 16, True), "quadrotor": (12, 4, False), "manipulator": (14, 7, False), "planar_vtol": (6, 2, False)}
    for name, (n, m, dyn) in expect.items():
        prob = nmpc_amd.make_problem(name)
        gn, gm, gd, nbytes = prob.dims()
        assert (gn, gm, gd) == (n, m, dyn)
        assert nbytes == C.sizeof(prob.blob)  # ctypes mirror == sizeof(C++ problem object)
        on, om, _ = oracle.model_dims(name)
        assert (on, om) == (n, m)


def test_default_config_matches_reference_defaults():
    c = nmpc_amd.Configuration()
    o = oracle.default_config()
    for k in ("max_iter", "horizon_steps", "reg_type", "initial_lambda", "initial_dlambda", "lambda_factor",
              "lambda_min", "lambda_max", "k_rel_norm_thre", "lambda_thre", "cost_update_ratio_thre",
              "cost_update_thre"):
        assert getattr(c, k) == getattr(o, k), k
    assert c.with_input_constraint is False and c.max_iter == 500 and c.horizon_steps == 100
    np.testing.assert_array_equal(c.alpha_list, np.array([o.alpha_list[i] for i in range(o.n_alpha)]))
    assert (c.qp_max_iter, c.qp_grad_thre, c.qp_rel_improve_thre, c.qp_step_factor, c.qp_min_step,
            c.qp_armijo_param) == (500, 1e-8, 1e-8, 0.6, 1e-22, 0.1)  # BoxQP.h:33-55


def test_default_model_params_match_oracle_defaults():
    for name in ("cartpole", "bipedal", "vertical", "centroidal", "quadrotor", "manipulator", "planar_vtol"):
        prob = nmpc_amd.make_problem(name)
        blob = np.frombuffer(bytes(prob.blob), dtype=np.float64)
        want = oracle.default_params(name)
        np.testing.assert_array_equal(blob, want[: blob.size], err_msg=name)


def test_misuse_is_reported_not_crashed():
    L = _capi.load()
    h = C.c_void_p()
    assert L.nmpc_hip_ddp_create(b"no_such_model", 10, 4, 0, C.byref(h)) == _capi.ERR_UNKNOWN_MODEL
    assert b"unknown model" in L.nmpc_hip_ddp_last_error()
    assert L.nmpc_hip_ddp_create(b"cartpole", 0, 4, 0, C.byref(h)) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_create(b"cartpole", 10, 4, 0, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_default_config(None) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_destroy(None) == 0
    buf = (C.c_double * 3)()
    assert L.nmpc_hip_ddp_model_default_params(b"cartpole", buf, 24) == _capi.ERR_INVALID_ARGUMENT
    # receding-horizon entry points
    opt = _capi.MpcOptions()
    assert L.nmpc_hip_ddp_mpc_default_options(None) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_mpc_default_options(C.byref(opt)) == 0
    assert (opt.n_ticks, opt.shift_warm_start, opt.max_iter_after_first, opt.sim_substeps, opt.sim_dt, opt.clamp_u0) \
        == (1, 1, 0, 0, 0.0, 1)
    assert L.nmpc_hip_ddp_mpc_run(None, None, None, None, C.byref(opt), None, None, None, None, None, None, None,
                                  None) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_kernel_name(None, None) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_set_model_params_batch(None, None, 0) == _capi.ERR_INVALID_ARGUMENT
    assert L.nmpc_hip_ddp_set_input_limits_batch(None, None, None) == _capi.ERR_INVALID_ARGUMENT


def test_no_cpu_fallback_without_a_gpu():
    from conftest import HAVE_GPU
    if HAVE_GPU:
        pytest.skip("a GPU is present")
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(), 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.solve(0.0, np.zeros((4, 4)), np.zeros((4, 100, 1)))


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under nmpc_amd/ or include/ may reference it."""
    for base in (os.path.join(ROOT, "nmpc_amd"), os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not re.search(r"^\s*(import oracle|from oracle)", txt, re.M), f
                    assert "ddp_oracle.hpp" not in txt and 'include "oracle' not in txt and "oracle/" not in txt \
                        or f in ("workloads.py",), f


# ---------------------------------------------------------------------------------------------------
# host-side logic of the mirror
# ---------------------------------------------------------------------------------------------------
def test_initial_u_list_validation_matches_reference_exceptions():
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(), 2)
    s.config().horizon_steps = 5
    s._h_T = 5  # validation happens before any device work
    with pytest.raises(ValueError, match="length should be 5 but 4"):  # std::invalid_argument, DDPSolver.hpp:41-45
        s._pack_u(np.zeros((2, 4, 1)), np.zeros(2))
    with pytest.raises(ValueError):
        s._pack_u([[np.zeros(1)] * 5], np.zeros(2))  # wrong batch
    with pytest.raises(RuntimeError, match="dimension should be 1 but 2"):  # std::runtime_error, :46-58
        s._pack_u([[np.zeros(2)] * 5, [np.zeros(1)] * 5], np.zeros(2))
    u = s._pack_u([[np.full(1, 3.0)] * 5, [np.full(1, 4.0)] * 5], np.zeros(2))
    assert u.shape == (2, 5, 1) and u[1, 2, 0] == 4.0


def test_configuration_roundtrip():
    c = nmpc_amd.Configuration()
    c.with_input_constraint = True
    c.max_iter = 7
    c.alpha_list = np.array([1.0, 0.5])
    cc = c.to_c()
    assert cc.with_input_constraint == 1 and cc.max_iter == 7 and cc.n_alpha == 2 and cc.alpha_list[1] == 0.5
    c.alpha_list = np.ones(40)
    with pytest.raises(ValueError):
        c.to_c()


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
def test_splitmix64_reference_values():
    # published splitmix64 outputs for seed 1234567: 6457827717110365317, 3203168211198807973, ...
    seed = 1234567
    z1 = 6457827717110365317
    z2 = 3203168211198807973
    u = workloads.splitmix64_uniform(seed, 2)
    assert u[0] == (z1 >> 11) / 2.0 ** 53 and u[1] == (z2 >> 11) / 2.0 ** 53
    a = workloads.cartpole_batch(B=16, seed=1234)
    b = workloads.cartpole_batch(B=32, seed=1234)
    np.testing.assert_array_equal(a.x0, b.x0[:16])  # instance-major draw order: prefixes agree
    assert np.all(np.abs(a.x0[:, 1]) <= np.pi) and np.all(np.abs(a.x0[:, [0, 2, 3]]) <= 1)


def test_algorithmic_bytes_formula():
    # SURVEY.md §8 d: C2 = 11 853 words, C3 = 14 419, C4 = 49 599, C5 = 48 163
    f = workloads.algorithmic_words_per_instance_iteration
    assert f(4, 1, 100) == 11853
    assert f(2, 1, 300) == 14419
    assert f(12, 4, 50) == 49599
    assert f(14, 7, 30) == 48163
    assert workloads.fused_words_per_instance_iteration(4, 1, 100) == 100 * 26 + 4 + 9


# ---------------------------------------------------------------------------------------------------
# sharding + the one collective (gloo, world_size 2)
# ---------------------------------------------------------------------------------------------------
def test_shard_ranges_cover_the_batch():
    for B in (1, 7, 64, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_range(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    B, T, n, mm = 5, 7, 4, 2
    X, U, cost = rng.normal(size=(B, T + 1, n)), rng.normal(size=(B, T, mm)), rng.normal(size=(B, T + 1))
    st, it = rng.integers(-1, 2, B).astype(np.int32), rng.integers(0, 50, B).astype(np.int32)
    rec = sharding.pack_results(X, U, cost, st, it)
    assert rec.shape == (B, sharding.record_width(T, n, mm))
    X2, U2, c2, s2, i2 = sharding.unpack_results(rec, T, n, mm)
    for a, b in ((X, X2), (U, U2), (cost, c2), (st, s2), (it, i2)):
        np.testing.assert_array_equal(a, b)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import oracle
from nmpc_amd import sharding, workloads
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B, T = 11, 20                                  # uneven shards on purpose
wl = workloads.cartpole_batch(B=B, T=T, seed=5)
lo, hi = sharding.shard_range(B, rank, world)
cfg = oracle.default_config(horizon_steps=T, max_iter=4)
# the per-shard solve is done by the CPU oracle here (this test covers sharding + the collective, not the kernel)
r = oracle.solve_batch(wl.model, cfg, wl.x0[lo:hi], wl.u_init[lo:hi], t0=wl.t0[lo:hi])
rec = torch.from_numpy(sharding.pack_results(r.X, r.U, r.cost, r.status, r.iters))
allrec = sharding.all_gather_records(rec, B).numpy()
full = oracle.solve_batch(wl.model, cfg, wl.x0, wl.u_init, t0=wl.t0)
want = sharding.pack_results(full.X, full.U, full.cost, full.status, full.iters)
assert allrec.shape == want.shape, (allrec.shape, want.shape)
assert np.array_equal(allrec, want), "gathered shards differ from the unsharded solve"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_gather_equals_unsharded_solve(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29632", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") >= 2  # both ranks passed their in-process asserts (stdout may interleave)


def test_sincos_fast_accuracy(tmp_path):
    """nmpc_amd::sincosFast (the ~32-instruction sin/cos the cart-pole functor inlines on the GPU) against
    long-double references on the host: <= 1.6 ulp for |x| <= 1e3, <= 2.6 ulp up to 1e8, NaN outside 2^27."""
    exe = str(tmp_path / "test_sincos")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                        os.path.join(ROOT, "tests", "cpp", "test_sincos.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SINCOS_OK" in r.stdout, r.stdout


def test_eigen_style_port_matches_the_shipped_model(tmp_path):
    """the reference's problem classes are written in Eigen block / initialiser syntax.  linalg.hpp offers
    that subset (segment / head / tail / block / middleRows / col / diagonal views, `<<` , asDiagonal, cross, products ...).
    tests/cpp/JetGyrostatEigenStyle.hpp is a problem class of this repository's own written that way (dynamic input dimension,
    matrices with a run-time number of columns); tests/cpp/JetGyrostatPlain.hpp states the same arithmetic entry by entry on
    scalars.  The two must agree bit for bit on the host, and compile for gfx950."""
    src = os.path.join(ROOT, "tests", "cpp", "test_eigen_style_port.cpp")
    exe = str(tmp_path / "port")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", f"-I{ROOT}/include", f"-I{ROOT}/tests/cpp", src, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "EIGEN_STYLE_PORT_OK" in r.stdout, r.stdout
    from nmpc_amd import build as hip_build
    r = subprocess.run([hip_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DDEVICE_COMPILE_CHECK", f"-I{ROOT}/include",
                        f"-I{ROOT}/tests/cpp", "-x", "hip", "-c", src, "-o", str(tmp_path / "port_dev.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_constant_limits_replace_a_limits_function_or_table():
    """setInputLimits after setInputLimitsFunc / setInputLimitsHorizon: the constant pair is what the next solve uses (the
    C++ mirror, DDPSolverBatch::setInputLimits, resets both as well); a table survives a change of horizon_steps as "to be
    pushed again" and is re-validated against the new horizon."""
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(), 2)
    s.setInputLimitsFunc(lambda t: (np.array([-1.0 - t]), np.array([1.0 + t])))
    assert s._limits_func is not None
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    assert s._limits_func is None and s._limits[0][0] == -15.0 and s._limits[1][0] == 15.0
    T = int(s.config().horizon_steps)
    s.setInputLimitsHorizon(np.full((T, 1), -2.0), np.full((T, 1), 2.0))
    assert s._limits_horizon is not None and s._limits_horizon_dirty
    s._limits_horizon_dirty = False  # (as if pushed to a handle)
    s.setInputLimits(np.array([-3.0]), np.array([3.0]))
    assert s._limits_horizon is None and s._limits_horizon_dirty  # the device table has to be removed on the next push


def test_centroidal_ridge_constants_match_libm():
    """The friction-pyramid ridge directions the centroidal problem keeps as constants are what the reference's expression
    (TestDDPCentroidalMotion.cpp:206-237: 0.5 cos / 0.5 sin of 2 pi ri / 4, normalised) evaluates to with libm."""
    import math
    text = open(os.path.join(ROOT, "include", "nmpc_amd", "models", "CentroidalMotion.hpp")).read()
    block = text[text.index("kRidgeDir[4][3]"):]
    block = block[:block.index("};")]
    vals = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", block)]
    assert len(vals) == 12
    for ri in range(4):
        theta = 2 * math.pi * (ri / 4)
        rx, ry = 0.5 * math.cos(theta), 0.5 * math.sin(theta)
        ln = math.sqrt((rx * rx + ry * ry) + 1.0 * 1.0)
        assert vals[3 * ri:3 * ri + 3] == [rx / ln, ry / ln, 1.0 / ln]
