"""Do two part-filled kernels on two HIP streams overlap on the GPU?  Both variants are captured into one HIP graph (no host cost):
20 + 20 GEMMs of 120 tiles (M=1536 K=1280 N=1280, one 8-wave workgroup per CU) on one stream vs forked onto two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussctrl_amd.sd import ops
dt = torch.bfloat16; dev = "cuda:0"
M, K, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1536, 1280, 1280)))
xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(2)]
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(2)]
for i in range(2): ops.linear(xs[i], ws[i])
torch.cuda.synchronize()

def capture(two, n=20):
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); side = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            if two:
                side.wait_stream(st)
                with torch.cuda.stream(side):
                    for _ in range(n): ops.linear(xs[1], ws[1])
                for _ in range(n): ops.linear(xs[0], ws[0])
                st.wait_stream(side)
            else:
                for _ in range(n): ops.linear(xs[0], ws[0]); ops.linear(xs[1], ws[1])
    return g

def timeg(g):
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / 20

g1, g2 = capture(False), capture(True)
for _ in range(2):
    print(f"M={M} K={K} N={N}: one stream {timeg(g1):7.1f} us per pair, two streams {timeg(g2):7.1f} us per pair")
