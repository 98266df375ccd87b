"""exploratory: instruction mix of the largest loop of a kernel in a hipcc -S dump.
usage: python tools/isa_loop.py x.s k_extract_trace"""
import re, sys, collections
L = [l.strip() for l in open(sys.argv[1])]
name = sys.argv[2]
st = [i for i, l in enumerate(L) if re.match(r'^_ZN3txd\d+' + name + r'E.*:', l)][0]
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
ops = collections.Counter(l.split()[0] for l in body[s:e] if l and l[0] not in ';.')
print('loop', s, e, 'instructions', sum(ops.values()), 'valu', sum(v for k, v in ops.items() if k.startswith('v_')))
print(ops.most_common(60))
print(collections.Counter(l for l in body[s:e] if l.startswith('s_waitcnt') and 'vmcnt' in l))
print('scratch in loop:', sum(1 for l in body[s:e] if l.startswith('scratch_')))
