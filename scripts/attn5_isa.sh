#!/bin/bash
# hot-loop instruction mix of k_attn5 (bf16, prescaled Q): bash scripts/attn5_isa.sh [extra hipcc flags]
cd "$(dirname "$0")/../gaussctrl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-function -Wno-unused-variable -ffp-contract=fast -munsafe-fp-atomics -fno-honor-nans "$@" -S --cuda-device-only dn_attn5.hip -o /tmp/dn_attn5.s 2>&1 | grep -v "hip-link" | head
python - <<'P'
import re, collections
s=open('/tmp/dn_attn5.s').read()
m=re.search(r'^(_ZN\w*k_attn5IN2dn4BF16ELb1\w*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S|re.M)
L=m.group(2).split('\n')
vg=re.search(r'\.vgpr_count:\s+(\d+)', s); 
print('vgpr', re.findall(r'k_attn5IN2dn4BF16ELb1\w*\.num_vgpr, (\d+)', s)[:1], 'scratch', re.findall(r'k_attn5IN2dn4BF16ELb1\w*\.private_seg_size, (\d+)', s)[:1])
idx=[i for i,l in enumerate(L) if 'Inner Loop Header' in l]
for st in idx:
    lab=None
    for j in range(st,-1,-1):
        if L[j].startswith('.LBB'): lab=L[j].split(':')[0]; break
    end=None
    for j in range(st,len(L)):
        if ('s_branch '+lab) in L[j] or (('s_cbranch' in L[j]) and L[j].strip().endswith(lab)): end=j
    if end is None: continue
    body=L[st:end+1]
    c=collections.Counter(l.strip().split(' ')[0] for l in body if l.strip() and not l.strip().startswith((';','.')))
    if c.get('v_mfma_f32_32x32x16_bf16',0)>=14:
        print('loop',lab,'instrs',sum(c.values()), {k:v for k,v in sorted(c.items(), key=lambda kv:-kv[1]) if v>=3})
        print('   scratch ops in loop:', sum(v for k,v in c.items() if k.startswith('scratch')))
P
