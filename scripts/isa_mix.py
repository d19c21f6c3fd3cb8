"""Instruction mix of a kernel's hot region, from the built objects (no GPU needed): the region between the two s_barriers that
enclose the most matrix-core instructions — for the fp64 tile kernel that is the matrix waves' backward step (one trip of the slot
loop, all reg_type variants), for others the recursion's chunk loop.
    python scripts/isa_mix.py <object under nmpc_amd/lib/obj> <mangled-name substring> [<substring> ...]
    e.g.  python scripts/isa_mix.py model_manipulator.o tile64_kernelINS_21DDPProblemManipulatorELb0ELb0"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, os.path.basename(obj))
    subprocess.run(["cp", obj, local], check=True)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_mfma"):
        return "matrix core (v_mfma)"
    if re.match(r"v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|cmp\w*)_f64|v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|pk_\w+)_f32|v_ldexp|v_frexp|v_rndne|v_cvt", op):
        return "VALU floating point"
    if op.startswith("v_cndmask"):
        return "VALU select (v_cndmask)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "VALU move" + (" (DPP: cross-lane)" if False else "")
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "VALU lane <-> scalar"
    if op.startswith("v_"):
        return "VALU integer / address / compare"
    if op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "LDS crossbar (ds_bpermute)"
    if op.startswith("ds_"):
        return "LDS read / write"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global memory"
    if op.startswith("scratch_"):
        return "scratch (spill)"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait / nop / barrier"
    return "scalar"


def main():
    obj = sys.argv[1] if os.path.isabs(sys.argv[1]) else os.path.join(ROOT, "nmpc_amd", "lib", "obj", sys.argv[1])
    text = disassemble(obj).splitlines()
    starts = [(k, l) for k, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l)]
    for pat in sys.argv[2:]:
        hit = [(k, l) for k, l in starts if pat in l]
        if not hit:
            print(f"{pat}: no such kernel in {os.path.basename(obj)}")
            continue
        k0 = hit[0][0]
        k1 = min([k for k, _ in starts if k > k0] + [len(text)])
        body = [l.split()[0] for l in text[k0 + 1:k1] if l.strip() and not l.lstrip().startswith("//") and l.startswith("\t")]
        barriers = [-1] + [k for k, op in enumerate(body) if op == "s_barrier"] + [len(body)]
        best = max(zip(barriers[:-1], barriers[1:]), key=lambda ab: sum(op.startswith("v_mfma") for op in body[ab[0] + 1:ab[1]]))
        region = body[best[0] + 1:best[1]]
        dpp = sum(1 for l in text[k0 + 1:k1] if "row_newbcast" in l)
        mix = collections.Counter(classify(op) for op in region)
        print(f"{pat}: kernel of {len(body)} instructions ({sum(op.startswith('scratch_') for op in body)} scratch, "
              f"{sum(op.startswith('v_mfma') for op in body)} v_mfma, {dpp} DPP row broadcasts); region between barriers "
              f"{best[0] + 1} .. {best[1]}: {len(region)} instructions")
        for name, n in mix.most_common():
            print(f"    {name:38s} {n:5d}")
        valu = sum(n for name, n in mix.items() if name.startswith("VALU"))
        print(f"    VALU : v_mfma in the region = {valu} : {mix['matrix core (v_mfma)']}"
              f" = {valu / max(mix['matrix core (v_mfma)'], 1):.1f} (static count: every reg_type variant's products are in the region)")


if __name__ == "__main__":
    main()
