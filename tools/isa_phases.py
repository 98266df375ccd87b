"""exploratory: per-phase instruction mix of k_blind_rotate's main loop.
usage: hipcc ... -DTXD_PHASE_MARK -S --cuda-device-only -o x.s csrc/tx_api.hip ; python tools/isa_phases.py x.s"""
import re, sys, collections
L = [l.strip() for l in open(sys.argv[1])]
st = [i for i, l in enumerate(L) if re.match(r'^_ZN3txd14k_blind_rotate.*:', l)][0]
en = [i for i, l in enumerate(L) if i > st and 's_endpgm' in l][0]
body = L[st:en]
lab = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r'^s_cbranch\S*\s+(\.LBB\d+_\d+)', l)
    if m and lab.get(m.group(1), 1 << 30) < i: loops.append((lab[m.group(1)], i))
s, e = max(loops, key=lambda x: x[1] - x[0])
loop = body[s:e]
marks = [(i, l) for i, l in enumerate(loop) if 'TXPHASE' in l]
cuts = [0] + [i for i, _ in marks] + [len(loop)]
names = ['rotate+decompose'] + [m.split('TXPHASE')[1].strip() for _, m in marks]
def cls(op):
    if op.startswith('v_accvgpr'): return 'agpr_mov'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_'): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op == 's_nop': return 'nop'
    return 'salu'
for k in range(len(cuts) - 1):
    c = collections.Counter()
    for l in loop[cuts[k]:cuts[k + 1]]:
        if not l or l[0] in ';.': continue
        c[cls(l.split()[0])] += 1
    print(names[k] if k < len(names) else '?', dict(c))
