"""Instruction mix of the MFMA loop of every kernel in a hipcc -S listing.  usage: isa_mix.py file.s [kernel-substring]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l) for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l)]
ends = [i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')]
for (i, l), e in zip(starts, ends):
    if want not in l: continue
    body = lines[i:e]
    labels = {m.group(1): j for j, x in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', x))}
    best = None
    for j, x in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', x)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            a = labels[m.group(1)]
            cnt = sum(1 for y in body[a:j] if 'v_mfma' in y)
            if cnt and (best is None or cnt > best[2]): best = (a, j, cnt)
    if not best: continue
    c = collections.Counter()
    for y in body[best[0]:best[1]]:
        y = y.strip()
        if not y or y[0] in '.;/': continue
        c[y.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    trans = sum(v for k, v in c.items() if k in ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32'))
    print(l.split(':')[0][:70], '| loop lines', best[1] - best[0], '| mfma', best[2], '| VALU', valu, '(trans', trans, ') | ds', sum(v for k, v in c.items() if k.startswith('ds_')),
          '| vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))), '| salu', sum(v for k, v in c.items() if k.startswith('s_')))
    print('   ', ', '.join(f'{k} {v}' for k, v in c.most_common(40)))
