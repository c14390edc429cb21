#!/usr/bin/env python3
"""Audit of hv_attention_w4's generated code (hipcc -save-temps .s): resource summary, compiler v_accvgpr_* outside inline asm,
scratch / s_load / v_readlane inside the steady-state loop, and the instruction mix per MFMA gap of the loop's first body.
usage: viz_w4_loop.py <file.s> [--gaps]"""
import re, sys
S = sys.argv[1]
L = open(S).read().split('\n')
for l in L:
    if re.search(r'\.(vgpr_count|agpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):', l): print(l.strip())
inasm = False; bad = []
for i, l in enumerate(L):
    if '#ASMSTART' in l: inasm = True; continue
    if '#ASMEND' in l: inasm = False; continue
    if not inasm and 'v_accvgpr' in l: bad.append(i)
print("compiler v_accvgpr outside asm:", len(bad), bad[:8])
# steady-state loop = the innermost loop containing the most MFMAs
starts = [i for i, l in enumerate(L) if 'Loop Header' in l]
best = None
for st in starts:
    # loop ends at the s_cbranch back to its label
    lab = None
    for k in range(st, max(st - 3, 0), -1):
        m = re.match(r'(\.LBB\d+_\d+):', L[k])
        if m: lab = m.group(1); break
    if not lab: continue
    end = None
    for k in range(st, len(L)):
        if re.search(r's_cbranch_\w+\s+' + re.escape(lab) + r'\b', L[k]) or re.search(r's_branch\s+' + re.escape(lab) + r'\b', L[k]): end = k
    if end is None: continue
    n = sum('v_mfma' in l for l in L[st:end])
    if best is None or n > best[2]: best = (st, end, n, lab)
st, end, n, lab = best
body = L[st:end + 1]
print(f"steady-state loop {lab}: lines {st}-{end}, {n} MFMAs, {len([l for l in body if l.startswith(chr(9)) and not l.strip().startswith(';')])} instructions")
for pat in ('scratch_', 's_load', 'v_readlane', 'v_writelane', 's_waitcnt', 'v_mov_b32', 'v_accvgpr', 's_nop', 'buffer_load', 'ds_read', 'v_exp', 'v_cvt_pk', 'v_add_f32', 'v_max3', 'v_xor', 'v_xad', 's_barrier'):
    print(f"  {pat:14s} {sum(pat in l for l in body)}")
if '--gaps' in sys.argv:
    gap, cur = 0, []
    for l in body:
        t = l.strip()
        if not t or t.startswith(';;') or t.startswith('.'): continue
        if 'sched_barrier' in t:
            ops = [re.split(r'\s+', x)[0] for x in cur]
            print(f"gap {gap:3d}: " + ' '.join(ops))
            gap += 1; cur = []
            continue
        if t.startswith(';'): continue
        cur.append(t)
