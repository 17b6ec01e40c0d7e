#!/usr/bin/env python3
"""Lint for register loads hidden from hipcc (inline-asm `buffer_load_dword vN, ... offen` without `lds`; conv_wino4.hip, SPEC 2):
hipcc does not know the destination is still in flight, so it may copy, spill or reuse vN before the data lands.  For every such
load in a kernel's ISA the first later instruction that names vN (program order, wrapping around the enclosing item loop) must
come behind an `s_waitcnt vmcnt` -- ours, since hipcc emits none for them.  usage: lint_asm_loads.py FILE.s KERNEL_SYMBOL;
exit code 1 on a violation (tests/test_host_cpu.py compiles the kernel and runs this)."""
import re,sys
fn,kern=sys.argv[1],sys.argv[2]
L=open(fn).read().split('\n')
st=next(i for i,l in enumerate(L) if l.startswith(kern+':'))
en=next(i for i in range(st,len(L)) if L[i].startswith('.Lfunc_end'))
seg=L[st:en]
def refs(line, n):
    # does the line reference vgpr n (token vN or range v[a:b])?
    for m in re.finditer(r'\bv(\d+)\b', line):
        if int(m.group(1))==n: return True
    for m in re.finditer(r'v\[(\d+):(\d+)\]', line):
        if int(m.group(1))<=n<=int(m.group(2)): return True
    return False
bad=0;nloads=0
for i,l in enumerate(seg):
    t=l.strip()
    m=re.match(r'buffer_load_dword v(\d+), v\d+, s\[\d+:\d+\], s\d+ offen$', t)
    if not m: continue
    nloads+=1
    n=int(m.group(1))
    waited=False
    hdr=max([k for k in range(i) if 'Loop Header: Depth=1' in seg[k]] or [0])
    order=list(range(i+1,len(seg)))+list(range(hdr,i))
    for j in order:
        u=seg[j].strip()
        if u.startswith('s_waitcnt') and 'vmcnt' in u: waited=True
        if u.startswith(';') or u.startswith('.'): continue
        if refs(u,n):
            if re.match(r'buffer_load_dword v%d,'%n,u): break  # reloaded (loop)
            if not waited:
                bad+=1; print('VIOLATION: load at',i,'->',t,'| first ref at',j,':',u)
            break
print(kern[-40:], 'asm register loads:',nloads,'violations:',bad)
sys.exit(1 if bad else 0)
