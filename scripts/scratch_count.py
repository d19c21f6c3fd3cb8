"""scratch_* (spill) and v_mfma instruction counts per kernel of the built objects (no GPU needed).
    python scripts/scratch_count.py [object substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "nmpc_amd", "lib", "obj")
for o in sorted(os.listdir(OBJ)):
    if not o.endswith(".o") or (len(sys.argv) > 1 and sys.argv[1] not in o):
        continue
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, o)
    subprocess.run(["cp", os.path.join(OBJ, o), local], check=True)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f]
    if not co:
        continue
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    name, cnt = None, {}
    for l in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            name = m.group(1)
            cnt[name] = [0, 0, 0]
            continue
        if name is None:
            continue
        parts = l.split()
        if len(parts) < 2:
            continue
        op = parts[0]
        cnt[name][2] += 1
        if op.startswith("scratch_"):
            cnt[name][0] += 1
        elif op.startswith("v_mfma"):
            cnt[name][1] += 1
    for k, (s, m, n) in cnt.items():
        if n > 200:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            d = re.sub(r"\(.*", "", d).replace("nmpc_amd::hip::", "").replace("nmpc_amd::", "").replace("void ", "")
            print(f"{o:26s} scratch {s:6d}  mfma {m:5d}  instr {n:7d}  {d[:110]}")
