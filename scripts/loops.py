"""Barrier loops of a disassembled gfx950 kernel (llvm-objdump -d ... > k.s) with their counts of scratch accesses, global loads / stores,
s_waitcnt vmcnt(0) and LDS instructions: where a loop that should keep loads in flight waits for each of them instead.
    python scripts/loops.py k.s"""
import re,sys
lines=open(sys.argv[1]).read().splitlines()
# map address->line idx
addr={}
ins=[]
for i,l in enumerate(lines):
    m=re.search(r'//\s*([0-9A-F]{12}):',l)
    if m:
        a=int(m.group(1),16); addr[a]=i; ins.append((i,a,l))
for i,a,l in ins:
    m=re.match(r'\s*s_cbranch_\w+\s+(\d+)',l)
    if m:
        off=int(m.group(1))
        if off>=32768:
            off-=65536
            tgt=a+4+4*off
            if tgt in addr:
                j=addr[tgt]
                body=lines[j:i+1]
                nb=sum('s_barrier' in x for x in body)
                if nb:
                    sc=sum('scratch_' in x for x in body); gl=sum('global_load' in x for x in body); gs=sum('global_store' in x for x in body)
                    vm0=sum('vmcnt(0)' in x for x in body); ds=sum('ds_write' in x or 'ds_read' in x for x in body)
                    print(f"loop lines {j}-{i} ({i-j} instr): barriers {nb} scratch {sc} global_load {gl} global_store {gs} vmcnt(0) {vm0} ds {ds}")
