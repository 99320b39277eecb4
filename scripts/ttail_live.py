"""approximate VGPR liveness just after MFMA number N of k_ttail<BF16> (reads /tmp/ttail.s): python scripts/ttail_live.py N"""
import re, collections, sys
s = open('/tmp/ttail.s').read()
m = re.search(r'^(_ZN\w*k_ttailIN2dn4BF16ELi0ELi0\w*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
L = m.group(2).split('\n')
mf = [i for i, l in enumerate(L) if 'v_mfma' in l]
N = int(sys.argv[1])
a = mf[N - 1]
def regs(t):
    r = set()
    for x in re.findall(r'\bv\[(\d+):(\d+)\]', t): r.update(range(int(x[0]), int(x[1]) + 1))
    for x in re.findall(r'\bv(\d+)\b', t): r.add(int(x))
    return r
written = set(); first = collections.OrderedDict()
for l in L[a + 1:]:
    t = l.strip()
    if not t or t.startswith((';', '.')): continue
    ops = t.split(None, 1)
    if len(ops) < 2: continue
    args = ops[1].split(',')
    isst = ops[0].startswith(('scratch_store', 'ds_write', 'global_store', 's_', 'v_cmp', 'v_writelane', 'global_load_lds'))
    dst = regs(args[0]) if not isst else set()
    src = set()
    for a_ in (args[1:] if dst else args): src |= regs(a_)
    for r in src - written:
        if r not in first: first[r] = ops[0]
    written |= dst
print(len(first), collections.Counter(first.values()).most_common(12))
