#!/usr/bin/env python3
"""Static instruction mix per '; MARK_x' region of one kernel in a hipcc -S listing.
    python scripts/asm_phase_stats.py file.s kernel_substring
"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + key + r'\w*:', l)][0]
end = [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][-1] + start
end = min(end, [i for i, l in enumerate(lines[start:]) if '.amdhsa_kernel' in l][0] + start)
cur = 'PRE'; stats = {}; order = []
for l in lines[start:end]:
    m = re.search(r'; (MARK_\w+)', l)
    if m: cur = m.group(1)
    if cur not in stats:
        stats[cur] = dict(total=0, valu=0, readlane=0, salu=0, ds=0, sld=0, sst=0, gld=0, gst=0, wait=0); order.append(cur)
    t = l.strip().split(' ')[0] if l.strip() else ''
    if not t or t[0] in ';.' or t.endswith(':'): continue
    st = stats[cur]; st['total'] += 1
    if t.startswith('scratch_load'): st['sld'] += 1
    elif t.startswith('scratch_store'): st['sst'] += 1
    elif t.startswith(('global_load', 'buffer_load', 'flat_load')): st['gld'] += 1
    elif t.startswith(('global_store', 'buffer_store', 'flat_store')): st['gst'] += 1
    elif t.startswith('ds_'): st['ds'] += 1
    elif t.startswith('v_readlane'): st['readlane'] += 1; st['valu'] += 1
    elif t.startswith('v_'): st['valu'] += 1
    elif t.startswith('s_waitcnt'): st['wait'] += 1
    elif t.startswith('s_'): st['salu'] += 1
for k in order: print(k, stats[k])
