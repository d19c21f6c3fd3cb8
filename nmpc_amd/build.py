"""Build libnmpc_hip_ddp.so (gfx950 code objects + C-ABI) in-tree with hipcc.

`python -m nmpc_amd.build` or `nmpc_amd.build.build()`; hipcc cross-compiles without a GPU.  The library is
written to nmpc_amd/lib/ so that it travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmpc_amd", "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(ROOT, "nmpc_amd", "lib")
LIB_PATH = os.path.join(LIB_DIR, "libnmpc_hip_ddp.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
SOURCES = ("capi.hip", "builtin_models.hip", "model_centroidal.hip", "model_quadrotor.hip", "model_manipulator.hip",
           "model_quadrotor_f32.hip", "model_cartpole_f32.hip", "model_manipulator_f32.hip", "model_planar_vtol.hip", "fmpc_capi.hip", "fmpc_models.hip")
ARCH = "gfx950"
# per-source flags.  builtin_models.hip holds the quad kernel (ddp_kernels_quad.hpp): its fp64 matrix-core results are
# consumed by VALU / DPP instructions right away, so they have to live in ordinary VGPRs — by default a kernel that may
# use 512 registers gets them in accumulation registers plus ~30 v_accvgpr moves per timestep (-5 % on the headline).
EXTRA_FLAGS = {"builtin_models.hip": ["-mllvm", "--amdgpu-mfma-vgpr-form"],
               "fmpc_models.hip": ["-mllvm", "--amdgpu-mfma-vgpr-form"]}  # fmpc_riccati_quad_kernel: same reason


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


HOST_ONLY_HEADERS = ("DDPSolverBatch.hpp", "DDPSolverSharded.hpp", "FmpcSolverBatch.hpp")  # mirrors over the C-ABI: no translation unit of the library includes them


def _headers():
    out = []
    for base, _, files in os.walk(INCLUDE):
        out += [os.path.join(base, f) for f in files if f not in HOST_ONLY_HEADERS]
    return out


def _obj_deps(obj: str, src: str):
    """Files the object was compiled from: the compiler's own dependency file (-MMD) when there is one from the last build,
    every header of the tree otherwise."""
    dep = obj[:-2] + ".d"
    if os.path.exists(dep) and os.path.exists(obj):
        # written on the machine that compiled the object; the tree may live elsewhere now (the GPU box): re-root the paths
        tokens = open(dep).read().split(":", 1)[-1].split()
        files = []
        for f in tokens:
            for marker in ("/include/nmpc", "/nmpc_amd/csrc/"):
                k = f.find(marker)
                if k >= 0:
                    files.append(ROOT + f[k:])
                    break
        if files:
            return [f for f in files if os.path.exists(f)] + [src]
    return [src] + _headers()


def _stale_objects():
    out = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _obj_deps(obj, src)):
            out.append(s)
    return out


def source_hash() -> str:
    """Hash of everything the device code is compiled from (translation units + headers, by relative path and content,
    comments and white space excluded).
    profiles/hbm_traffic.json keys its measured HBM traffic on it: bench.py refuses an entry measured on other sources."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + _headers()
    for f in sorted(files, key=lambda f: os.path.relpath(f, ROOT)):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(b"\0")
        h.update(_code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


_COMMENT = None


def _code_only(text: str) -> str:
    """The text without comments and with runs of white space collapsed: editing a comment does not invalidate the measured
    traffic (string and character literals are kept as they are)."""
    global _COMMENT
    import re
    if _COMMENT is None:
        _COMMENT = re.compile(r'''("(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')|(/\*.*?\*/|//[^\n]*)''', re.S)
    stripped = _COMMENT.sub(lambda m: m.group(1) if m.group(1) is not None else " ", text)
    return " ".join(stripped.split())


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if os.path.isdir(OBJ_DIR) and os.listdir(OBJ_DIR):
        return bool(_stale_objects()) or any(
            os.path.getmtime(os.path.join(OBJ_DIR, f)) > os.path.getmtime(LIB_PATH) for f in os.listdir(OBJ_DIR) if f.endswith(".o"))
    t = os.path.getmtime(LIB_PATH)  # a shipped library without its objects (the GPU box): sources and headers decide
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}"]
    flags += os.environ.get("NMPC_AMD_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DNMPC_AMD_PROFILE_2W
    objs = []
    procs = []
    stale = set(SOURCES) if force else set(_stale_objects())
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if s not in stale:
            continue
        cmd = [cc] + flags + EXTRA_FLAGS.get(s, []) + ["-MMD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB_PATH


def variant_path(name: str) -> str:
    return os.path.join(LIB_DIR, name, "libnmpc_hip_ddp.so")


def build_variant(name: str, extra_flags, force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """A second library with extra compiler flags on EVERY translation unit, written to nmpc_amd/lib/<name>/ (objects beside it):
    test builds such as the wave-timing fuzz (`fuzz_path()`); load it with NMPC_HIP_DDP_LIB=<path>.  Rebuilt when a source or header is
    newer than the library."""
    out_dir = os.path.join(LIB_DIR, name)
    lib = variant_path(name)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + _headers()
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    os.makedirs(os.path.join(out_dir, "obj"), exist_ok=True)
    cc = hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}"] + list(extra_flags)
    objs, pending = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(out_dir, "obj", s.replace(".hip", ".o"))
        objs.append(obj)
        pending.append([cc] + flags + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])
    jobs = jobs or (os.cpu_count() or 4)
    running = []
    while pending or running:
        while pending and len(running) < jobs:
            cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            running.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        cmd, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return lib


def fuzz_path(seed: int = 1) -> str:
    return variant_path("fuzz%d" % seed)


def build_fuzz(seed: int = 1, force: bool = False, verbose: bool = False) -> str:
    """The wave-timing fuzz build (include/nmpc_amd/hip/fuzz_sched.hpp): every barrier of every kernel family is preceded and
    followed by a per-wave pseudo-random sleep.  Results must be bit-identical to the product build's (tests/test_gpu_fuzz_sched.py)."""
    return build_variant("fuzz%d" % seed, ["-DNMPC_AMD_FUZZ_SCHED=%d" % seed] + os.environ.get("NMPC_AMD_FUZZ_EXTRA_FLAGS", "").split(),
                         force=force, verbose=verbose)


if __name__ == "__main__":
    if "--fuzz" in sys.argv:
        k = sys.argv.index("--fuzz")
        seeds = [int(a) for a in sys.argv[k + 1:] if a.isdigit()] or [1]
        for sd in seeds:
            print(build_fuzz(sd, force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
