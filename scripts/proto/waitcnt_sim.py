"""Path-sensitive check of the vector-memory wait counts of ONE kernel in a hipcc -S listing: every path through the kernel's branches
(s_cbranch_* taken / not taken, loops until a (pc, outstanding loads) state repeats) is walked with the in-order vmcnt queue, and any
instruction that reads or writes a VGPR a load may still be writing is reported.  r05: run on conv3x3_select4_kernel built with and
without SLP packing -- no violation in either (profiles/r05/x_select_strips_packed_fma.txt): the packed build's intermittent low halves
are not a missing s_waitcnt vmcnt.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only conv_select.hip -o /tmp/sel.s && python scripts/proto/waitcnt_sim.py /tmp/sel.s conv3x3_select4_kernel"""
import re, sys
src, kname = sys.argv[1], sys.argv[2]
s = open(src).read()
m = re.search(r'^(_Z\w*' + kname + r'\w*):', s, re.M)
st = m.end(); en = s.index('.Lfunc_end', st)
prog = []; labels = {}
for l in s[st:en].split('\n'):
    t = l.split(';')[0].strip()
    if not t: continue
    if t.endswith(':'):
        labels[t[:-1]] = len(prog); continue
    if t.startswith('.'): continue
    prog.append(t)
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok):
        out.add(int(a))
    return out
def parse(t):
    op, _, rest = t.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    return op, ops
viol = {}
seen = set()
stack = [(0, ())]
nstates = 0
while stack:
    pc, pend = stack.pop()
    while True:
        if pc >= len(prog): break
        key = (pc, pend)
        if key in seen: break
        seen.add(key); nstates += 1
        if nstates > 3000000: print('state limit'); sys.exit(1)
        t = prog[pc]; op, ops = parse(t)
        if op == 's_endpgm': break
        if op == 's_waitcnt':
            mm = re.search(r'vmcnt\((\d+)\)', t)
            if mm:
                n = int(mm.group(1))
                pend = pend[len(pend) - n:] if n < len(pend) else pend
                if n == 0: pend = ()
            pc += 1; continue
        if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
            d = frozenset(regs(ops[0])); srcs = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            busy = set().union(*[p[0] for p in pend]) if pend else set()
            if (srcs | d) & busy: viol.setdefault(pc, set()).add(tuple(sorted((srcs | d) & busy)))
            pend = pend + ((d, pc),); pc += 1; continue
        if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic')):
            srcs = set().union(*[regs(o) for o in ops])
            busy = set().union(*[p[0] for p in pend]) if pend else set()
            if srcs & busy: viol.setdefault(pc, set()).add(tuple(sorted(srcs & busy)))
            pend = pend + ((frozenset(), pc),); pc += 1; continue
        if op in ('s_branch',):
            pc = labels[ops[0]]; continue
        if op.startswith('s_cbranch'):
            stack.append((labels[ops[0]], pend)); pc += 1; continue
        used = set().union(*[regs(o) for o in ops]) if ops else set()
        busy = set().union(*[p[0] for p in pend]) if pend else set()
        if used & busy: viol.setdefault(pc, set()).add(tuple(sorted(used & busy)))
        pc += 1
print('instructions', len(prog), 'states', nstates, 'violations', len(viol))
for pc in sorted(viol)[:40]:
    print(pc, prog[pc], sorted(viol[pc])[:3])
