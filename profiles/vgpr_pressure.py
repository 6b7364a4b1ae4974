#!/usr/bin/env python3
"""Rough VGPR liveness over the ISA text of one kernel (hipcc -S --cuda-device-only): backward dataflow over the basic
blocks, first operand = destination for everything but stores / compares-to-SGPR; prints the blocks with the highest
number of live VGPRs and, for the peak, which source lines (";" comments are absent in -S output, so: instruction text)."""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
s = open(path).read()
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
lines = s[i:j].split('\n')[1:]
blocks = []      # (label, [instr])
cur = ('entry', [])
for ln in lines:
    t = ln.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    elif t and not t.startswith(('.', ';', '#')):
        cur[1].append(t.split(';')[0].strip())
blocks.append(cur)
idx = {b[0]: k for k, b in enumerate(blocks)}
def regs(op):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', op):
        if m.group(5) is not None: out.append(int(m.group(5)) + (1000 if m.group(4) == 'a' else 0))
        else: out.extend(range(int(m.group(2)) + (1000 if m.group(1) == 'a' else 0), int(m.group(3)) + 1 + (1000 if m.group(1) == 'a' else 0)))
    return out
NODST = ('global_store', 'scratch_store', 'ds_write', 'buffer_store', 's_', 'v_cmp', 'v_cmpx', 'ds_bpermute_nothing', 'global_load_lds', 'v_writelane_nothing')
def du(ins):
    parts = ins.split(None, 1)
    op = parts[0]; args = parts[1] if len(parts) > 1 else ''
    ops = [a.strip() for a in re.split(r',(?![^\[]*\])', args)]
    d, u = [], []
    if op.startswith(NODST):
        for a in ops: u += regs(a)
    else:
        if ops: d = regs(ops[0])
        for a in ops[1:]: u += regs(a)
        if op.startswith(('v_mac', 'v_fmac', 'v_writelane', 'v_mad_mix')) or 'accvgpr' in op and False:
            u += d
        if op.startswith('v_mad_u64_u32') or op.startswith('v_add_co') or op.startswith('v_addc') or op.startswith('v_sub_co') or op.startswith('v_subb'):
            pass
    return set(d), set(u)
succ = []
for k, (lab, ins) in enumerate(blocks):
    sc = set()
    fall = True
    for t in ins:
        m = re.match(r'^(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', t)
        if m:
            if m.group(2) in idx: sc.add(idx[m.group(2)])
            if m.group(1) == 's_branch': fall = False
        if t.startswith('s_endpgm'): fall = False
    if fall and k + 1 < len(blocks): sc.add(k + 1)
    succ.append(sc)
gen, kill = [], []
for lab, ins in blocks:
    g, kl = set(), set()
    for t in ins:
        d, u = du(t)
        g |= (u - kl); kl |= d
    gen.append(g); kill.append(kl)
live_in = [set() for _ in blocks]; live_out = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for k in range(len(blocks) - 1, -1, -1):
        lo = set()
        for s_ in succ[k]: lo |= live_in[s_]
        li = gen[k] | (lo - kill[k])
        if lo != live_out[k] or li != live_in[k]:
            live_out[k], live_in[k] = lo, li; changed = True
peak = []
for k, (lab, ins) in enumerate(blocks):
    live = set(live_out[k]); mx = len(live); at = len(ins)
    for n in range(len(ins) - 1, -1, -1):
        d, u = du(ins[n])
        live -= d; live |= u
        if len(live) > mx: mx, at = len(live), n
    peak.append((mx, k, at))
for mx, k, at in sorted(peak, reverse=True)[:topn]:
    lab, ins = blocks[k]
    print('%4d live  %-14s %5d instrs  peak at #%d: %s' % (mx, lab, len(ins), at, ins[at] if at < len(ins) else '(end)'))
if len(sys.argv) > 4:
    # live-through analysis for the loop whose header label is argv[4]: registers live into the header that no block
    # of the loop (blocks whose comment says Header=<that>) references; and where they are next used
    hdr = sys.argv[4]
    raw = s[i:j].split('\n')
    inloop = set()
    for ln in raw:
        m = re.match(r'^(\.LBB\d+_\d+):.*(Header=%s |Inner Loop Header|Loop Header)' % hdr.replace('.L', ''), ln.strip())
        if m and (('Header=' + hdr.replace('.L', '')) in ln or ln.strip().startswith(hdr + ':')):
            inloop.add(idx[m.group(1)])
    refd = set()
    for k in inloop:
        refd |= gen[k] | kill[k]
        for t in blocks[k][1]:
            d, u = du(t); refd |= d | u
    lt = sorted(live_in[idx[hdr]] - refd)
    print('loop %s: %d blocks, live-in %d, live-through (unreferenced) %d: %s' % (hdr, len(inloop), len(live_in[idx[hdr]]), len(lt), lt))
    first = {}
    for k in range(len(blocks)):
        if k in inloop: continue
        for n, t in enumerate(blocks[k][1]):
            d, u = du(t)
            for r in u:
                if r in lt and r not in first and k > max(inloop): first[r] = (blocks[k][0], t)
    by = collections.defaultdict(list)
    for r, (lab, t) in first.items(): by[lab].append((r, t))
    for lab in by: print(lab, len(by[lab]), by[lab][:6])
