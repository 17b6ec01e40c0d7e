#!/usr/bin/env python3
"""Lint for the counted epilogue stores of conv_wino4_kernel (non-RAG instantiations).  The item-crossing pipeline orders its LDS-DMA
data by COUNTING: the first chunk behind an epilogue waits with s_waitcnt vmcnt(NL + 16) -- correct only while every wave issues the same
16 output stores (+ 4 statistics stores) per item on EVERY path through the epilogue.  hipcc lays the residual / aux branches out of
line, so textual order says nothing: this walks the control-flow graph of each `conv_wino4_kernel<.., .., false>` symbol (basic blocks
split at labels and branches; back edges found by depth-first search) and, for every loop whose body holds a store -- the item loop of
the heavy and of the light wave-class body --, requires that the minimum and the maximum number of buffer stores over all paths of one
iteration agree: 16 x dwordx4 + 4 x dwordx2.  A store sunk into a conditional block, merged or duplicated by the compiler, or an edit
that puts a branch around one, changes one of the two and fails here instead of producing silently wrong tiles on the GPU.
usage: lint_asm_stores.py FILE.s ; exit code 1 on a violation (tests/test_host_cpu.py compiles conv_wino4.hip and runs this)."""
import re
import sys

sys.setrecursionlimit(100000)
L = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(L) if re.match(r'_ZN\S*conv_wino4_kernelILi\dELi\dELb0E\S*:', l)]
bad = 0
for st in starts:
    en = next(i for i in range(st, len(L)) if L[i].startswith('.Lfunc_end'))
    name = "conv_wino4_kernel<%s, %s, false>" % re.search(r'ILi(\d)ELi(\d)ELb0E', L[st]).groups()
    # ---- basic blocks: a new block at every label and behind every branch ----
    blocks, cur, label_of = [], {"ins": [], "label": None}, {}
    for i in range(st + 1, en):
        t = L[i].strip()
        m = re.match(r'(\.LBB\d+_\d+):', t)
        if m:
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"ins": [], "label": m.group(1)}
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        cur["ins"].append(t)
        if re.match(r's_c?branch', t) or t.startswith('s_endpgm'):
            blocks.append(cur)
            cur = {"ins": [], "label": None}
    if cur["ins"]:
        blocks.append(cur)
    for k, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = k
    succ, x4, x2 = [], [], []
    for k, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else ""
        s = []
        m = re.match(r's_(c?)branch\w*\s+(\.LBB\d+_\d+)', last)
        if m:
            s.append(label_of[m.group(2)])
            if m.group(1) == 'c' and k + 1 < len(blocks):
                s.append(k + 1)
        elif not last.startswith('s_endpgm') and k + 1 < len(blocks):
            s.append(k + 1)
        succ.append(s)
        x4.append(sum(1 for t in b["ins"] if t.startswith('buffer_store_dwordx4')))
        x2.append(sum(1 for t in b["ins"] if t.startswith('buffer_store_dwordx2')))
    # ---- back edges (iterative DFS) ----
    color, back, stack = [0] * len(blocks), [], [(0, 0)]
    color[0] = 1
    while stack:
        n, j = stack.pop()
        if j < len(succ[n]):
            stack.append((n, j + 1))
            m_ = succ[n][j]
            if color[m_] == 1:
                back.append((n, m_))
            elif color[m_] == 0:
                color[m_] = 1
                stack.append((m_, 0))
        else:
            color[n] = 2
    backset = set(back)
    # ---- per loop with stores: min / max stores over one iteration (paths header -> latch in the graph without back edges) ----
    found = 0
    for latch, head in back:
        memo = {}

        def walk(n):
            """(min, max) of (x4, x2) store counts from block n to the latch, None if the latch is not reachable"""
            if n in memo:
                return memo[n]
            memo[n] = None  # (cycle guard; cannot happen without back edges)
            here = (x4[n], x2[n])
            if n == latch:
                res = (here, here)
            else:
                outs = [walk(m_) for m_ in succ[n] if (n, m_) not in backset]
                outs = [o for o in outs if o is not None]
                if not outs:
                    res = None
                else:
                    lo = min(o[0] for o in outs)
                    hi = max(o[1] for o in outs)
                    lo4 = min(o[0][0] for o in outs), min(o[0][1] for o in outs)
                    hi4 = max(o[1][0] for o in outs), max(o[1][1] for o in outs)
                    res = ((here[0] + lo4[0], here[1] + lo4[1]), (here[0] + hi4[0], here[1] + hi4[1]))
            memo[n] = res
            return res
        r = walk(head)
        if r is None or r[1] == (0, 0):
            continue
        found += 1
        if r[0] != r[1] or r[0] != (16, 4):
            print("VIOLATION: %s: a loop's iteration issues between %s and %s (dwordx4, dwordx2) stores, expected exactly (16, 4) on every path" % (name, r[0], r[1]))
            bad += 1
    if found != 2:
        print("VIOLATION: %s: %d store-carrying loops found, expected the item loops of the two wave-class bodies" % (name, found))
        bad += 1
    print("%s: %d blocks, item loops with stores: %d, every path (16, 4)" % (name, len(blocks), found))
print("symbols: %d violations: %d" % (len(starts), bad))
sys.exit(1 if bad or not starts else 0)
