#!/usr/bin/env python
"""Instruction mix of the largest loop (the time loop) of one kernel in a hipcc device assembly listing:
    python tools/loop_mix.py x.s <mangled-name-prefix>"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith(sys.argv[2]) and ':' in l)
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
lines = [l for l in text[start:end] if not l.strip().startswith((';', '.')) or re.match(r'^\.LBB', l)]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
best = None
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
span, a, b = best
body = [l.split()[0] for l in lines[a:b] if l.startswith('\t')]
c = Counter()
for ins in body:
    key = ('mfma' if ins.startswith('v_mfma') else 'valu' if ins.startswith('v_') else 'lds' if ins.startswith('ds_') else
           'vmem' if ins.startswith(('global_', 'buffer_', 'scratch_')) else 'waitcnt' if ins.startswith('s_waitcnt') else
           'nop' if ins.startswith('s_nop') else 'salu' if ins.startswith('s_') else 'other')
    c[key] += 1
print('loop of', span, 'lines:', dict(c))
print(Counter(i for i in body if i.startswith('v_') and not i.startswith('v_mfma')).most_common(16))
