"""Count the instructions of the kernels whose mangled name contains the given substrings in a hipcc -S dump:
   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o x.s file.hip && python scripts/isa_count.py x.s k_legodo"""
import sys
from collections import Counter

path, pats = sys.argv[1], sys.argv[2:]
cur, body = None, {}
for line in open(path):
    t = line.strip()
    if line.startswith('_Z') and ':' in t:
        cur = t.split(':')[0]
        body[cur] = []
    elif cur and t.startswith('s_endpgm'):
        body[cur].append('s_endpgm')
        cur = None
    elif cur and t and not t.startswith(('.', ';', '/')) and not t.endswith(':'):
        body[cur].append(t.split()[0])
for name, ins in body.items():
    if pats and not any(p in name for p in pats):
        continue
    c = Counter()
    for i in ins:
        if i.startswith('v_') and 'f64' in i: c['v_f64'] += 1
        elif i.startswith('v_'): c['v_other'] += 1
        elif i.startswith('s_'): c['scalar'] += 1
        elif i.startswith(('global', 'buffer', 'flat', 'scratch')): c['vmem'] += 1
        elif i.startswith('ds_'): c['lds'] += 1
        else: c['other'] += 1
    print(name[:90], len(ins), dict(c))
