"""Where a kernel's register pressure sits: for every basic block of one kernel in a `hipcc -S --cuda-device-only`
listing, the highest VGPR index referenced and the instruction mix.   python tools/isa_pressure.py file.s <mangled-substring>"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2]
m = [m for m in re.finditer(r'^(\S*%s\S*):.*$' % re.escape(key), txt, re.M) if not m.group(1).startswith('.')][-1]
start = m.end()
end = txt.index('.Lfunc_end', start)
body = txt[start:end].split('\n')
blocks, cur = [], ['entry', []]
for l in body:
    if re.match(r'^\.LBB\S+:', l):
        blocks.append(cur)
        cur = [l.split(':')[0] + ' ' + (l.split(';')[1].strip() if ';' in l else ''), []]
    elif l.startswith('\t') and not l.startswith('\t.') and not l.strip().startswith(';'):
        cur[1].append(l.strip())
blocks.append(cur)
print(m.group(1))
for name, ins in blocks:
    if not ins:
        continue
    mx = -1
    for l in ins:
        regs = [int(x) for x in re.findall(r'\bv(\d+)\b', l)] + [int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', l)]
        mx = max([mx] + regs)
    kinds = {}
    for l in ins:
        op = l.split()[0]
        k = ('ds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else
             'salu' if op.startswith('s_') else 'trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op) else
             'dpp' if 'dpp' in l else 'lane' if 'lane' in op else 'valu')
        kinds[k] = kinds.get(k, 0) + 1
    print('%-70s n=%4d maxv=%3d %s' % (name[:70], len(ins), mx, kinds))
