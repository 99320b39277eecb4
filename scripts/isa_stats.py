"""Instruction statistics of one kernel in a hipcc -S dump: python scripts/isa_stats.py file.s substring-of-the-mangled-name"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    if pat not in m.group(1):
        continue
    lines = m.group(2).split('\n')
    c = collections.Counter()
    for l in lines:
        t = l.strip().split(' ')[0]
        if t.startswith(('v_mfma', 'scratch_', 'ds_read', 'ds_write', 'v_exp', 'v_cvt_pk', 's_barrier', 'global_load_lds', 'v_accvgpr', 'buffer_', 's_cbranch')):
            c[t] += 1
    print(m.group(1), len(lines), 'lines')
    print('  ', dict(c))
    for i, l in enumerate(lines):
        if 'scratch_' in l:
            print('   scratch at', i, l.strip())
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    print('   mfma lines', mf[:6], '...', mf[-6:])
