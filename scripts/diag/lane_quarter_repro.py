"""Stand-alone reproducer (torch + one small HIP library, nothing else of this repository):
    python scripts/diag/lane_quarter_repro.py load SECONDS           torch.matmul (hipBLASLt) in a loop -- the disturbing process
    python scripts/diag/lane_quarter_repro.py victim MODE SECONDS    k_echo<MODE> (lane_quarter_repro.hip) on static inputs, every call compared with the first
Run the load in one process and the victim in another on the same GPU."""
import collections, ctypes as C, os, sys, time
import torch
dev = "cuda:0"
if sys.argv[1] == "load":
    secs = float(sys.argv[2])
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1536, 1280, device=dev, generator=g).half(); w = torch.randn(1280, 1280, device=dev, generator=g).half()
    n, t0 = 0, time.time()
    while time.time() - t0 < secs:
        for _ in range(20):
            x @ w
        torch.cuda.synchronize(); n += 20
    print(f"load: {n} matmuls", flush=True)
else:
    mode, secs = int(sys.argv[2]), float(sys.argv[3])
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lane_quarter_repro_scalar.so" if os.environ.get("LQ_LIB") == "scalar" else "lane_quarter_repro.so"))   # _scalar: the same file built with -fno-slp-vectorize -fno-vectorize
    N = 20000
    g = torch.Generator(device=dev); g.manual_seed(0)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    means, ls, quats, opl, dc, rest = mk(N, 3), mk(N, 3), mk(N, 4), mk(N), mk(N, 3), mk(N, 45)
    p = lambda t: C.c_void_p(t.data_ptr())

    def call():
        out = torch.empty(N, 16, device=dev)
        rc = lib.diag_echo(mode, C.c_int64(N), p(means), p(ls), p(quats), p(opl), p(dc), p(rest), p(out),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        return out

    ref = call()
    torch.cuda.synchronize()
    if len(sys.argv) > 4:                       # the reference is taken BEFORE the disturbing process starts: wait for it
        time.sleep(float(sys.argv[4]))
    want = torch.cat([means, ls, quats, opl[:, None], dc], 1)
    if mode < 5:
        assert torch.equal(ref[:, :14], want), "the first call is not a copy of its inputs"
    else:
        assert torch.equal(ref[:, 6:10], quats)
    COLS = ["means"] * 3 + ["scales"] * 3 + ["quats"] * 4 + ["opacity"] + ["dc"] * 3 + ["rest (LDS) / fma x"] + ["math / fma y"]
    quarters = collections.Counter(); cols = collections.Counter(); bad = total = 0
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(10):
            out = call()
            total += 1
            d = out != ref
            if bool(d.any()):
                bad += 1
                for j in d.any(1).nonzero().flatten().tolist():
                    quarters[j % 64 // 16] += 1
                for c in sorted({COLS[k] for k in d.any(0).nonzero().flatten().tolist()}):
                    cols[c] += 1
    print(f"victim mode {mode}: {bad} deviating calls of {total}; rows by quarter of the wavefront {dict(sorted(quarters.items()))}; by input {dict(cols)}", flush=True)
