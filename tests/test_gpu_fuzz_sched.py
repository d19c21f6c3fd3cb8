"""Wave-timing fuzz: the library rebuilt with -DNMPC_AMD_FUZZ_SCHED, where every workgroup barrier of every kernel
family is wrapped in per-wave pseudo-random sleeps (include/nmpc_amd/hip/fuzz_sched.hpp), must return the product build's bits.

The kernels exchange data between waves through LDS and HBM behind hand-placed barriers; a missing one shows only when a wave runs far
enough ahead, which an idle chip never makes happen (one such race shipped for several commits in round 4 and was found by luck).
Under the fuzz build a wave is regularly a whole phase behind its neighbours: profiles/r05_fuzz_reopened_race.txt is the log of this
very comparison failing within one run when that race's barrier is taken out again — and profiles/r05_fuzz_fan_adopt_race.txt is the
race it then FOUND in the quad kernel's fan-out line search (a pass without a barrier of its own: the master could post the next
command before a late wave had read this one; since fixed), once the soak ran the long solves that use that search.  scripts/fuzz_soak.py is the worker (one process
per library, NMPC_HIP_DDP_LIB selects it); the whole GPU suite also runs against the fuzz library (scripts/fuzz_suite.sh)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def soak(lib, reps):
    env = dict(os.environ)
    env.pop("NMPC_HIP_DDP_LIB", None)
    if lib:
        env["NMPC_HIP_DDP_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_soak.py"), "--reps", str(reps)], capture_output=True, text=True,
                       timeout=1800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def fuzz_seed() -> int:
    """One of the three prebuilt seeds (nmpc_amd/lib/fuzz{1,2,3}/, __graft_entry__.build()), rotating with the hour of the run so that
    repeated runs of the suite — the driver's at round end, the builder's during the round — do not all sleep the same waves at the same
    barriers (only seed 1 ever ran there).  NMPC_FUZZ_SEED pins it to reproduce a failure; the seed is in every message."""
    import time
    env = os.environ.get("NMPC_FUZZ_SEED")
    return int(env) if env else 1 + int(time.time() // 3600) % 3


def test_fuzzed_wave_timing_changes_no_bit_in_any_kernel_family():
    from nmpc_amd import build as hip_build
    seed = fuzz_seed()
    print("wave-timing fuzz seed", seed)
    fuzz_lib = hip_build.build_fuzz(seed)  # (in-tree, nmpc_amd/lib/fuzz<seed>/: shipped with the tree; rebuilt here only if stale)
    want = soak(None, 1)
    got = soak(fuzz_lib, 3)
    assert set(want) == set(got) and len(want) >= 32
    # same workload, three line searches / two schedules: one digest (in both builds, by the comparison below)
    for same in (("quad c2 fan-out forced", "quad c2 fan-out no scratch", "quad c2 sequential forced"), ("quad c2 to convergence", "quad c2 ragged schedule")):
        assert len({want[c]["digests"][0] for c in same}) == 1, same
    families = {v["kernel"] for v in want.values()}
    for k in ("ddp_solve_quad_kernel", "ddp_solve_tpi2w_kernel", "ddp_solve_tpi_kernel", "ddp_solve_wpi_kernel", "ddp_solve_tile64_kernel",
              "ddp_solve_tile32_kernel"):
        assert k in families, (k, families)
    bad = {}
    for case, ref in want.items():
        assert got[case]["kernel"] == ref["kernel"], (case, "seed", seed)
        d = set(got[case]["digests"]) | set(ref["digests"])
        if len(d) != 1:
            bad[case] = (ref["digests"], got[case]["digests"])
    assert not bad, ("fuzz seed %d" % seed, bad)
