"""Per-kernel microbenchmarks at the real SD1.5 shapes of BASELINE configs[1] (chunk_size=3 -> CFG batch 6).
usage: python scripts/bench_kernels.py [conv|linear|attn|norm|all] [bf16|f16]"""
import sys, torch
sys.path.insert(0, '.')
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight, geglu_permute
ops.configure(ops.options_from_env())       # GC_GEMM_MT / GC_GEMM_PW / ... experiment switches
DEV = 'cuda:0'
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
if __name__ != '__main__': what = 'none'
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == 'f16') else torch.bfloat16
B = 6

def timeit(fn, n=20, warm=3):
    """GPU time per call: the n calls are captured into one HIP graph (no host launch gaps)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2): fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us

def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(dt)

tot = {}
if what in ('conv', 'all'):
    print(f"--- conv3x3 ({dt}) B={B}")
    for (H, Cin, Cout, stride, ups, cnt) in [(64, 320, 320, 1, False, 16), (64, 640, 320, 1, False, 2), (64, 960, 320, 1, False, 1), (64, 320, 320, 2, False, 1),
                                        (32, 320, 640, 1, False, 1), (32, 640, 640, 1, False, 12), (32, 1280, 640, 1, False, 2), (32, 1920, 640, 1, False, 1), (32, 960, 640, 1, False, 1),
                                        (32, 640, 640, 2, False, 1), (32, 640, 640, 1, True, 1),
                                        (16, 640, 1280, 1, False, 1), (16, 1280, 1280, 1, False, 12), (16, 2560, 1280, 1, False, 5), (16, 1920, 1280, 1, False, 1), (16, 1280, 1280, 1, True, 1),
                                        (8, 1280, 1280, 1, False, 9), (8, 2560, 1280, 1, False, 3), (8, 1280, 1280, 1, True, 1),
                                        (64, 8, 320, 1, False, 1)]:
        x = rnd(B, H, H, Cin); w = conv3x3_weight(rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), dt); b = torch.randn(Cout, device=DEV)
        us = timeit(lambda: ops.conv3x3(x, w, b, stride=stride, upsample=ups))
        Ho = (2 * H if ups else H) // stride
        fl = 2.0 * B * Ho * Ho * Cout * 9 * Cin
        tot['conv'] = tot.get('conv', 0) + us * cnt
        print(f"  {H:3d}x{H:<3d} {Cin:5d}->{Cout:5d} s{stride} ups{int(ups)} : {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s   (x{cnt} per UNet fwd)")
if what in ('linear', 'all'):
    print(f"--- linear ({dt}) B={B}")
    for (L, K, N, geglu, cnt) in [(4096, 320, 320, False, 30), (4096, 320, 2560, True, 5), (4096, 1280, 320, False, 5), (4096, 320, 960, False, 0),
                             (1024, 640, 640, False, 30), (1024, 640, 5120, True, 5), (1024, 2560, 640, False, 5),
                             (256, 1280, 1280, False, 30), (256, 1280, 10240, True, 5), (256, 5120, 1280, False, 5), (64, 1280, 1280, False, 6), (64, 5120, 1280, False, 1), (64, 1280, 10240, True, 1), (1, 1280, 1280, False, 22)]:
        x = rnd(B, L, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV)
        if geglu: w, b = geglu_permute(w, b)
        us = timeit(lambda: ops.linear(x, w, b, geglu=geglu))
        fl = 2.0 * B * L * N * K
        tot['linear'] = tot.get('linear', 0) + us * cnt
        print(f"  M={B * L:6d} K={K:5d} N={N:5d} geglu{int(geglu)} : {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s   (x{cnt})")
if what in ('attn', 'all'):
    print(f"--- attention ({dt}) B={B}")
    for (L, C, nsets, Lk, cnt) in [(4096, 320, 5, 4096, 5), (4096, 320, 4, 4096, 2), (1024, 640, 5, 1024, 5), (256, 1280, 5, 256, 5), (64, 1280, 5, 64, 1), (4096, 320, 1, 77, 5), (4096, 320, 1, 4096, 0)]:
        heads = 8
        q = rnd(B, L, C, scale=0.4); Bk = B if Lk == L else 2
        k = rnd(Bk, Lk, C); Lp = (Lk + 7) // 8 * 8
        vt = torch.zeros(Bk, C, Lp, dtype=dt, device=DEV); vt[:, :, :Lk] = rnd(Bk, C, Lk)
        kr = rnd(8, Lk, C); vtr = rnd(8, C, Lp)
        if Lk != L: sets = [(-2, 1.0)]
        elif nsets == 1: sets = [(-1, 1.0)]
        else: sets = ([(-1, 0.6)] if nsets == 5 else []) + [(r, 0.1) for r in range(4)]
        fn = (lambda: ops.attention(q, k, vt, heads, sets, B // 2, Lk=Lk, kref=kr, vtref=vtr, ref_fph=4, q_prescaled=True)) if (Lk == L and nsets > 1) else (lambda: ops.attention(q, k, vt, heads, sets, B // 2, Lk=Lk, q_prescaled=True))
        us = timeit(fn)
        fl = 4.0 * B * L * Lk * C * len(sets)
        tot['attn'] = tot.get('attn', 0) + us * cnt
        print(f"  L={L:5d} Lk={Lk:5d} C={C:5d} sets={len(sets)} : {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s   (x{cnt})")
if what in ('norm', 'all'):
    print(f"--- norms ({dt}) B={B}")
    for (HW, C, cnt) in [(4096, 320, 20), (4096, 640, 2), (4096, 960, 1), (1024, 640, 18), (1024, 1280, 2), (1024, 1920, 1), (256, 1280, 18), (256, 2560, 5), (64, 1280, 12), (64, 2560, 3)]:
        x = rnd(B, HW, C); g = torch.randn(C, device=DEV); bb = torch.randn(C, device=DEV)
        us = timeit(lambda: ops.groupnorm(x, g, bb, 32, 1e-5, True))
        tot['gn'] = tot.get('gn', 0) + us * cnt
        print(f"  groupnorm HW={HW:5d} C={C:5d}: {us:8.1f} us  {B * HW * C * 2 * 3 / us / 1e6:6.2f} TB/s (2R+1W)  (x{cnt})")
    for (M, C) in [(B * 4096, 320), (B * 1024, 640), (B * 256, 1280)]:
        x = rnd(M, C); g = torch.randn(C, device=DEV); bb = torch.randn(C, device=DEV)
        us = timeit(lambda: ops.layernorm(x, g, bb))
        print(f"  layernorm M={M:6d} C={C:5d}: {us:8.1f} us  {M * C * 2 * 2 / us / 1e6:6.2f} TB/s")
if what == "scan":
    for K, N in ((320, 320), (1280, 1280), (320, 1280), (1280, 320)):
        for M in (128, 1536, 6144, 24576, 98304):
            x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV)
            us = timeit(lambda: ops.linear(x, w, b))
            print(f"  M={M:6d} K={K:5d} N={N:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
print("weighted per-UNet-forward totals (us):", {k: round(v) for k, v in tot.items()})
