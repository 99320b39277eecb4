"""Finds the operand layout of the MX-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) on the GPU box: packs random small-integer
A[16,128], B[128,16] (exact in e4m3) and E8M0 block scales under candidate layouts and compares with the host product.
usage: python scripts/mx_probe.py"""
import ctypes as C
import itertools
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "scripts", "ubench", "libmx_probe.so"))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
A = torch.randint(-3, 4, (16, 128), generator=g).float()
B = torch.randint(-3, 4, (128, 16), generator=g).float()
ea = torch.randint(-2, 3, (16, 4), generator=g)            # block exponents per (row, 32-block of k)
eb = torch.randint(-2, 3, (4, 16), generator=g)
A8 = A.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
B8 = B.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
ref = ((A * (2.0 ** ea.float()).repeat_interleave(32, 1)) @ (B * (2.0 ** eb.float()).repeat_interleave(32, 0))).numpy()
ref_noscale = (A @ B).numpy()


def run(Ab, Bb, SA, SB, opa=0, opb=0):
    t = lambda x, dt: torch.tensor(np.ascontiguousarray(x).view(dt)).to(dev)
    a, b = t(Ab, np.int32), t(Bb, np.int32)
    sa, sb = torch.tensor(SA.astype(np.int32)).to(dev), torch.tensor(SB.astype(np.int32)).to(dev)
    d = torch.zeros(64, 4, device=dev)
    rc = lib.mx_probe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sa.data_ptr()), C.c_void_p(sb.data_ptr()),
                      C.c_void_p(d.data_ptr()), opa, opb, None)
    assert rc == 0
    torch.cuda.synchronize()
    return d.cpu().numpy()


def unpack_d(d):                      # C/D layout of the 16x16 shapes: col = lane & 15, row = 4 * (lane >> 4) + reg
    out = np.zeros((16, 16), np.float32)
    for lane in range(64):
        for r in range(4):
            out[4 * (lane >> 4) + r, lane & 15] = d[lane, r]
    return out


def pack(M8, kmap):                   # M8[i, k] (A: rows; for B pass B8.T so that "row" = column j): lane holds 32 bytes
    out = np.zeros((64, 32), np.uint8)
    for lane in range(64):
        for j in range(32):
            out[lane, j] = M8[lane & 15, kmap(lane, j)]
    return out


cands = {
    "k = 32*(lane>>4) + byte": lambda l, j: 32 * (l >> 4) + j,
    "k = 64*(byte>>4) + 16*(lane>>4) + (byte&15)": lambda l, j: 64 * (j >> 4) + 16 * (l >> 4) + (j & 15),
    "k = 4*(8*(byte>>2)... interleave 4B": lambda l, j: 16 * (j >> 2) + 4 * (l >> 4) + (j & 3),
    "k = 8-byte interleave": lambda l, j: 32 * (j >> 3) + 8 * (l >> 4) + (j & 7),
}
one = np.full(64, 127, np.int64)
print("== layout (scales = 1.0)")
for name, km in cands.items():
    d = unpack_d(run(pack(A8, km), pack(B8.T, km), one, one))
    print(f"  {name:50s} max |D - A@B| = {np.abs(d - ref_noscale).max():.3f}")
best = None
for name, km in cands.items():
    d = unpack_d(run(pack(A8, km), pack(B8.T, km), one, one))
    if np.abs(d - ref_noscale).max() == 0:
        best = (name, km)
if best is None:
    raise SystemExit("no candidate layout matches")
print("layout:", best[0])
km = best[1]
# scale semantics: which k-block does lane l's scale byte apply to?  candidates: block = lane>>4 (own data block); byte select via opsel
def scales(E, which):                 # E [16 rows, 4 blocks] exponents -> per-lane int32 with the E8M0 byte replicated / placed
    out = np.zeros(64, np.int64)
    for lane in range(64):
        row = lane & 15
        if which == "own":
            v = 127 + int(E[row, lane >> 4]); out[lane] = v | (v << 8) | (v << 16) | (v << 24)
        elif which == "bytes":            # all four block scales of the row in the four bytes
            out[lane] = sum((127 + int(E[row, kb])) << (8 * kb) for kb in range(4))
    return out
print("== scales")
for which in ("own", "bytes"):
    for opa, opb in ((0, 0), (1, 0), (2, 3), (3, 1)):
        d = unpack_d(run(pack(A8, km), pack(B8.T, km), scales(ea.numpy(), which), scales(eb.numpy().T, which), opa, opb))
        print(f"  scale vreg = {which:5s} opsel = ({opa},{opb}): max |D - ref| = {np.abs(d - ref).max():.3f}   (ref max {np.abs(ref).max():.1f})")
