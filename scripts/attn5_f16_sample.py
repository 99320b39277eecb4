"""k_attn5 in f16: does the exponent offset of a K/V set (round 4: row maximum over 64 SAMPLED keys, one per 8 x 8 block of a 64-wide token map; before:
over the set's first 32 keys) keep P = exp2(s - offset) inside f16's range?  A row that overflows sends its 256-query workgroup to the
online-softmax body (about 3x the time), so time per launch against bf16 (which never overflows) shows the fallback rate.
Two kinds of logits at the production launch (L = 4096 = a 64 x 64 map, 8 heads x 40, B = 6, own frame + 4 reference sets):
  iid     q, k ~ qscale N(0, 1): no spatial structure, the sample can only be as good as any 64 keys
  local   q_p = k_p = sqrt(G) u(p), u a smooth unit-vector field (correlation length ~4 tokens): a query's logits peak at G binades around its own
          position in EVERY frame and sit near 0 +- G / sqrt(40) elsewhere -- the spatial locality trained self / cross-view attention has
python scripts/attn5_f16_sample.py        (GC_HIP_LIB=<older build> for the before / after)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussctrl_amd.sd import ops
DEV = "cuda:0"
f, L, heads, D = 3, 4096, 8, 40
B, C = 2 * f, heads * D


def ref_attn(q, k, v):
    qh = q.float().view(q.shape[0], L, heads, D).transpose(1, 2); kh = k.float().view(k.shape[0], L, heads, D).transpose(1, 2)
    vh = v.float().view(v.shape[0], L, heads, D).transpose(1, 2)
    p = (qh @ kh.transpose(-1, -2) * 0.6931471805599453).softmax(-1)          # logits are in binades (q_prescaled)
    return (p @ vh).transpose(1, 2).reshape(q.shape[0], L, C)


def fields(n, gen, G):
    """n frames of [L, C]: per head a smooth unit-vector field times sqrt(G)"""
    z = torch.randn(n * heads, D, 16, 16, device=DEV, generator=gen)
    u = torch.nn.functional.interpolate(z, size=(64, 64), mode="bilinear", align_corners=False)
    u = u / u.norm(dim=1, keepdim=True)
    return (u * G ** 0.5).view(n, heads, D, L).permute(0, 3, 1, 2).reshape(n, L, C)


def case(kind, dt, param):
    g = torch.Generator(device=DEV).manual_seed(0)
    if kind == "iid":
        q = (torch.randn(B, L, C, device=DEV, generator=g) * param).to(dt); k = (torch.randn(B, L, C, device=DEV, generator=g) * param).to(dt)
        kr = torch.randn(8, L, C, device=DEV, generator=g).to(dt)
    else:
        base = fields(1, g, param)                                   # one scene: every frame sees (nearly) the same field
        mk = lambda n: (base + 0.05 * param ** 0.5 * torch.randn(n, L, C, device=DEV, generator=g) / D ** 0.5).to(dt).contiguous()
        q, k, kr = mk(B), mk(B), mk(8)
    v = torch.randn(B, L, C, device=DEV, generator=g).to(dt); vr = torch.randn(8, L, C, device=DEV, generator=g).to(dt)
    vt, vtr = v.transpose(1, 2).contiguous(), vr.transpose(1, 2).contiguous()
    sets = [(-1, 0.6)] + [(r, 0.1) for r in range(4)]
    call = lambda: ops.attention(q, k, vt, heads, sets, f, Lk=L, kref=kr, vtref=vtr, ref_fph=4, q_prescaled=True)
    out = call().float()
    ref = 0.6 * ref_attn(q, k, v)
    for r in range(4):
        idx = torch.arange(B, device=DEV) // f * 4 + r
        ref = ref + 0.1 * ref_attn(q, kr[idx], vr[idx])
    err = float((out - ref).norm() / ref.norm())
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / 20, err, bool(torch.isfinite(out).all())


print(f"library: {os.environ.get('GC_HIP_LIB', 'shipped build')}")
print("logits      parameter        f16 us/launch   rel L2 vs fp32     bf16 us/launch   rel L2")
for kind, params in (("iid", (0.05, 0.25, 0.5, 0.7)), ("local", (12.0, 20.0, 30.0, 45.0))):
    for p in params:
        t16, e16, ok16 = case(kind, torch.float16, p)
        tb, eb, okb = case(kind, torch.bfloat16, p)
        name = f"qscale {p}" if kind == "iid" else f"peak {p:g} binades"
        print(f"{kind:10s}  {name:16s} {t16:10.0f}      {e16:.2e}{'' if ok16 else ' NOT FINITE'}        {tb:10.0f}      {eb:.2e}{'' if okb else ' NOT FINITE'}", flush=True)
