"""Static guards on the shipped gfx950 code objects (no GPU needed: llvm-objdump on libnmpc_hip_ddp.so).

Round 6 found a performance bug that no parity test can see: in SOME builds of the tile kernel the compiler put
`s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every `flat_load` of the line search's ring prefetch — nineteen sequential round trips to
L2 per trip, forward passes 1.6 x slower — because the prefetch pointers had decayed to generic ones.  They are
address_space(1) now; these tests keep it that way and pin a few other properties the measurements rest on."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from nmpc_amd import build as hip_build

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels():
    """name -> list of instruction mnemonics, for every kernel of the library's gfx950 code objects."""
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not in this image")
    lib = hip_build.build()
    tmp = tempfile.mkdtemp(prefix="isa_guard_")
    try:
        out = {}
        obj_dir = hip_build.OBJ_DIR
        objs = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o")] if os.path.isdir(obj_dir) else []
        assert objs, "the objects of the library are built in-tree (nmpc_amd/lib/obj)"
        for o in objs:
            work = os.path.join(tmp, os.path.basename(o)[:-2])
            os.makedirs(work)
            local = os.path.join(work, "x.o")
            shutil.copy(o, local)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=work, check=True, capture_output=True)
            for co in [f for f in os.listdir(work) if "gfx950" in f]:
                text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(work, co)], check=True, capture_output=True, text=True).stdout
                name = None
                for line in text.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                    if m:
                        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                        name = re.sub(r"\(.*", "", name).replace("nmpc_amd::hip::", "").replace("nmpc_amd::", "").replace("void ", "")
                        out[name] = []
                    elif name is not None:
                        parts = line.split()
                        if len(parts) >= 2 and not parts[0].endswith(":"):
                            out[name].append(parts[0] + " " + parts[1] if parts[0] == "s_waitcnt" else parts[0])
        assert lib
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def count(ins, prefix):
    return sum(1 for i in ins if i.startswith(prefix))


def test_tile_kernels_with_a_shared_problem_object_issue_no_flat_loads(kernels):
    tile = {k: v for k, v in kernels.items() if k.startswith("ddp_solve_tile64_kernel<") and k.rstrip(">").endswith("false")}  # kOwnProblem = false
    assert len(tile) >= 8, sorted(kernels)[:20]
    for name, ins in tile.items():
        constrained = name.rstrip(">").endswith("true, false")
        # (box-constrained instantiations read the limits through one of three pointers — shared, per instance, per timestep: two flat loads
        # outside the trips of the passes)
        assert count(ins, "flat_load") <= (2 if constrained else 0), (name, count(ins, "flat_load"))
        assert count(ins, "global_load") > 20, name  # (the ring prefetch, the linearisation's points)


def test_the_headline_kernel_has_no_scratch_and_runs_on_the_4x4x4_matrix_cores(kernels):
    name = next(k for k in kernels if k.startswith("ddp_solve_quad_kernel<DDPProblemCartPoleT<double>, false, false, true, false>"))
    ins = kernels[name]
    assert count(ins, "scratch_") == 0 and count(ins, "flat_load") == 0
    assert count(ins, "v_mfma_f64_4x4x4") >= 100


def test_the_batched_natural_layout_gains_are_what_the_centroidal_kernel_runs(kernels):
    name = next(k for k in kernels if k.startswith("ddp_solve_tile64_kernel<DDPProblemCentroidalMotion, false, false>"))
    ins = kernels[name]
    # a column per lane: the pivot and the pivot column by row broadcasts of the fp64 pipeline's own DPP forms (16 + 120 and 120)
    assert count(ins, "v_mov_b64_dpp") == 136 and count(ins, "v_fmac_f64_dpp") == 120
    assert count(ins, "v_mfma_f64_16x16x4") >= 90  # three unrolled slots' Q terms and value updates


def test_resumable_instantiations_exist_for_the_quad_and_two_wave_kernels(kernels):
    # <Problem, constrained, own problem, fan-out, resumable> / <Problem, constrained, own problem, resumable>: what the ragged schedule and the
    # streamed solves launch
    assert any(k.startswith("ddp_solve_quad_kernel<DDPProblemCartPoleT<double>, false, false, true, true>") for k in kernels)
    assert any(k.startswith("ddp_solve_quad_kernel<DDPProblemCartPoleT<double>, true, false, true, true>") for k in kernels)
    assert any(k.startswith("ddp_solve_tpi2w_kernel<DDPProblemCartPoleT<double>, false, false, true>") for k in kernels)


def test_fmpc_kernels_keep_requests_in_flight(kernels):
    """fmpc_riccati_fused_kernel's forward sweep requests its operands two chunks ahead: that only pays while the wait in front of a
    commit is COUNTED (s_waitcnt vmcnt(N), N > 0) — with a store of the same wavefront in flight, or with the requests behind per-lane
    branches, the compiler waits for everything (vmcnt(0)) and the sweep is a round trip to HBM per chunk again (79 instead of 62 us
    of the launch).  fmpc_tail_kernel: a spilled register is reloaded behind s_waitcnt vmcnt(0), which waits for the next timestep's
    requests — no scratch."""
    fused = next(k for k in kernels if k.startswith("fmpc_riccati_fused_kernel<FmpcProblemCartPole>"))
    tail = next(k for k in kernels if k.startswith("fmpc_tail_kernel<FmpcProblemCartPole>"))
    assert count(kernels[fused], "scratch_") == 0 and count(kernels[tail], "scratch_") == 0
    counted = [i for i in kernels[fused] if i.startswith("s_waitcnt vmcnt(") and int(i.split("(")[1].split(")")[0]) >= 8]
    assert len(counted) >= 4, counted
    assert count(kernels[tail], "global_load") >= 60  # (a timestep's 34 values requested ahead, the neighbour's, the candidates)
