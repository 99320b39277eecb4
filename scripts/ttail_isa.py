"""instruction mix of k_ttail's feed-forward loop and of the whole kernel, bucketed by MFMA count (reads /tmp/ttail.s)"""
import re, collections, sys
s = open('/tmp/ttail.s').read()
m = re.search(r'^(_ZN\w*k_ttailIN2dn4BF16\w*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
L = m.group(2).split('\n')
hdr = [i for i, l in enumerate(L) if 'Loop Header' in l]
for st in hdr:
    lab = None
    for j in range(st, -1, -1):
        if L[j].startswith('.LBB'): lab = L[j].split(':')[0]; break
    end = None
    for j in range(st, len(L)):
        if re.search(r's_c?branch\w*\s+' + re.escape(lab) + r'\b', L[j]): end = j
    if end is None: continue
    body = [l.strip() for l in L[st:end + 1] if l.strip() and not l.strip().startswith((';', '.'))]
    c = collections.Counter(i.split(' ')[0] for i in body)
    if c.get('v_mfma_f32_32x32x16_bf16', 0) > 10:
        print(lab, len(body), {k: v for k, v in c.most_common(30)})
ins = [l.strip() for l in L if l.strip() and not l.strip().startswith((';', '.'))]
n = 0; buckets = collections.defaultdict(collections.Counter)
for i in ins:
    op = i.split(' ')[0]
    if 'mfma' in op: n += 1
    b = n // 100
    k = 'scr' if op.startswith('scratch') else 'acc' if op.startswith('v_accvgpr') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'ds' if op.startswith('ds_') else 'other'
    buckets[b][k] += 1; buckets[b]['all'] += 1
for b in sorted(buckets): print(b * 100, dict(buckets[b]))
