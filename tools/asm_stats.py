#!/usr/bin/env python
"""Static per-kernel statistics from a hipcc device assembly listing
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S x.hip -o x.s):
instruction totals, non-MFMA VALU, MFMA, and the metadata's VGPR / spill / LDS figures."""
import re
import sys


def stats(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'^(_ZN3sbr\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
        ins = [l.split()[0] for l in m.group(2).split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        out[m.group(1)] = dict(total=len(ins), valu=sum(1 for i in ins if i.startswith('v_') and not i.startswith('v_mfma')),
                               mfma=sum(1 for i in ins if i.startswith('v_mfma')), lds=sum(1 for i in ins if i.startswith('ds_')),
                               vmem=sum(1 for i in ins if i.startswith(('global_', 'buffer_', 'flat_'))))
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)(?=\n  - \.|\namdhsa\.target)', s, re.S):
        body = m.group(2)
        d = out.setdefault(m.group(1), {})
        for key in ('vgpr_count', 'vgpr_spill_count', 'agpr_count', 'group_segment_fixed_size'):
            g = re.search(r'\.' + key + r':\s+(\d+)', body)
            if g:
                d[key] = int(g.group(1))
    return out


if __name__ == "__main__":
    st = stats(sys.argv[1])
    pats = sys.argv[2:]
    for k, v in st.items():
        if not pats or any(p in k for p in pats):
            print(k[8:72], v)
