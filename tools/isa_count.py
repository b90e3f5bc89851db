"""Instruction mix of selected kernels from a hipcc -S dump: tools/isa_count.py file.s substr [substr...]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
subs = sys.argv[2:]
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for k, (i, name) in enumerate(starts):
    if subs and not all(s in name for s in subs):
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    c = collections.Counter()
    tot = 0
    for l in lines[i:end]:
        if not l.startswith('\t'):
            continue
        op = l.strip().split()[0] if l.strip() else ''
        if not op or op.startswith(('.', ';')):
            continue
        tot += 1
        if op.startswith('v_'):
            if 'f64' in op: c['v_f64'] += 1
            elif op.startswith('v_cndmask'): c['v_cndmask'] += 1
            elif op.startswith('v_cmp'): c['v_cmp(other)'] += 1
            elif 'f32' in op: c['v_f32'] += 1
            else: c['v_int/mov'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): c['vmem'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        else: c['other'] += 1
    print(name[:70], 'total', tot, dict(c))
