"""CPU-side checks of the FMPC path (no GPU): the library exports every symbol include/nmpc_hip_fmpc.h declares, the registered
problem types and their parameter images agree with the oracle's, defaults are the reference's, and the product path fails
loudly without a device (there is no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from nmpc_amd import _capi
from nmpc_amd import fmpc as F
from oracle import fmpc as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_fmpc_symbol():
    hdr = open(os.path.join(ROOT, "include", "nmpc_hip_fmpc.h")).read()
    declared = set(re.findall(r"\b(nmpc_hip_fmpc_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found in the header"
    L = F.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in nmpc_hip_fmpc.h but not exported"
    assert declared == set(F.EXPORTS), (declared ^ set(F.EXPORTS))
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.lib_path()], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), name


def test_field_and_kernel_class_numbers_match_the_header():
    """The Python mirror's FIELD_* / KERNEL_CLASSES are the enum values of include/nmpc_hip_fmpc.h, in order."""
    hdr = open(os.path.join(ROOT, "include", "nmpc_hip_fmpc.h")).read()
    fields = dict((k, int(v)) for k, v in re.findall(r"NMPC_HIP_FMPC_FIELD_(\w+)\s*=\s*(\d+)", hdr))
    assert len(fields) == 20 and sorted(fields.values()) == list(range(20))
    for name, value in fields.items():
        py = {"LAMBDA": "FIELD_LAMBDA", "GAIN_k": "FIELD_GAIN_k"}.get(name, "FIELD_" + name)
        assert getattr(F, py) == value, name
    classes = dict((k.lower(), int(v)) for k, v in re.findall(r"NMPC_HIP_FMPC_KERNEL_(\w+)\s*=\s*(\d+)", hdr))
    assert [k for k, _ in sorted(classes.items(), key=lambda kv: kv[1])] == list(F.KERNEL_CLASSES)
    assert int(re.search(r"NMPC_HIP_FMPC_NKERNELS\s*=\s*(\d+)", hdr).group(1)) == len(F.KERNEL_CLASSES)


def test_problem_types_match_the_oracle_models():
    assert set(F.model_names()) >= {"fmpc_oscillator", "fmpc_cartpole", "fmpc_pointmass"}
    for model, cls in (("fmpc_oscillator", F.FmpcProblemOscillator), ("fmpc_cartpole", F.FmpcProblemCartPole),
                       ("fmpc_pointmass", F.FmpcProblemPointMass)):
        n, m, g, pb = F.model_info(model)
        on, om, og, pd = O.model_info(model)
        assert (n, m, g) == (on, om, og) and pb == 8 * pd
        prob = cls()
        # same memory image as the oracle's struct of doubles: one parameter blob drives both sides of the parity tests
        assert np.array_equal(prob.p, O.default_params(model))
        assert prob.stateDim() == n and prob.inputDim() == m and prob.ineqDim() == g and prob.dt() == prob.p[0]
    assert F.FmpcProblemCartPole(0.02, ref_pos=1.5).p[13] == 1.5


def test_default_config_is_the_reference_default():
    """FmpcSolver::Configuration (FmpcSolver.h:57-89)."""
    c = F.Configuration()
    assert (c.print_level, c.horizon_steps, c.max_iter, c.kkt_error_thre) == (1, 100, 10, 1e-4)
    assert c.check_nan and not c.init_complementary_variable and c.update_barrier_eps and not c.break_if_llt_fails
    assert not c.enable_line_search and not c.merit_const_scale_from_lagrange_multipliers
    o = O.default_config()
    for k in F.Configuration._FIELDS:
        if k not in ("use_graph", "time_kernels"):
            assert getattr(c, k) == getattr(o, k), k
    assert F.Status.Succeeded == 1 and F.Status.MaxIterationReached == 5 and F.Status.IterationContinued == 6


def test_variable_reset_and_shape_checks():
    prob = F.FmpcProblemCartPole()
    v = F.Variable.make(prob, 7, 3)
    v.reset(0.1, 0.2, 0.3, 1.0, 2.0)
    assert v.x_list.shape == (3, 8, 4) and v.u_list.shape == (3, 7, 1) and v.s_list.shape == (3, 7, 4)
    assert (v.x_list == 0.1).all() and (v.nu_list == 2.0).all() and v.horizon_steps == 7


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine without a GPU")
def test_no_device_fails_loudly():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.FmpcSolverBatch(F.FmpcProblemOscillator(), 4, 10)
    with pytest.raises(ValueError, match="unknown FMPC problem type"):
        F.model_info("no_such_model")


def test_cpp_mirror_and_example_compile_on_the_host():
    """include/nmpc_amd/FmpcSolverBatch.hpp is plain C++17 over the C-ABI: g++ alone compiles a caller."""
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT}/include", "-fsyntax-only",
                        os.path.join(ROOT, "examples", "fmpc_oscillator_mpc.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", f"-I{ROOT}/include", "-fsyntax-only",
                        os.path.join(ROOT, "examples", "fmpc_c_api.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]  # include/nmpc_hip_fmpc.h is a C header
