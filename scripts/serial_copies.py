"""Loops of a gfx950 disassembly that copy (or reduce) one element per round trip: a backward branch whose body is short, has global
loads and as many `s_waitcnt vmcnt(0)` — load, wait, use, branch.  (Round 6: the trajectory copy of the quad kernel's fan-out adopt and
the horizon reductions of fmpc_tail_kernel were such loops; the compiler does not move a load of the next trip over a store or a
compare of this one.)    python scripts/serial_copies.py <objdump -d output> [max body length]"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().splitlines()
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80
addr, func = {}, {}
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m:
        cur = m.group(1)
    m = re.search(r"//\s*([0-9A-F]{12}):", l)
    if m:
        addr[int(m.group(1), 16)] = i
        func[i] = cur
seen = {}
for a, i in addr.items():
    m = re.match(r"\s*s_cbranch_\w+\s+(\d+)", lines[i])
    if not m:
        continue
    off = int(m.group(1))
    if off < 32768:
        continue
    tgt = a + 4 + 4 * (off - 65536)
    if tgt not in addr:
        continue
    j = addr[tgt]
    body = lines[j:i + 1]
    if len(body) > limit:
        continue
    gl = sum("global_load" in x or "flat_load" in x for x in body)
    gs = sum("global_store" in x for x in body)
    vm0 = sum("vmcnt(0)" in x for x in body)
    if gl >= 1 and vm0 >= gl:
        seen.setdefault(func[i], []).append((len(body), gl, gs, vm0))
for f, loops in sorted(seen.items(), key=lambda kv: -len(kv[1])):
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("nmpc_amd::hip::", "").replace("nmpc_amd::", "").replace("void ", "")
    print(f"{len(loops):3d} loops  {name[:150]}: " + ", ".join(f"{n} instr {gl}ld/{gs}st" for n, gl, gs, _ in loops[:8]))
