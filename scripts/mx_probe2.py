"""Which operand bytes does lane l's scale register apply to in v_mfma_scale_f32_16x16x128_f8f6f4?  A = ones; B one-hot at
(lane lb, byte jb); scale_a = 2.0 in ONE lane l0 (1.0 elsewhere).  Two operand positions meet in the dot product iff they have the same
(lane >> 4, byte index), so D[row l0 & 15][col lb & 15] == 2 tells that bytes (lb >> 4, jb) of A-lane l0's row are scaled by lane l0.
usage: python scripts/mx_probe2.py"""
import ctypes as C
import os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "scripts", "ubench", "libmx_probe.so"))
dev = "cuda:0"
ONE = 0x38


def run(Ab, Bb, SA, SB, opa=0, opb=0):
    t = lambda x: torch.tensor(np.ascontiguousarray(x).view(np.int32)).to(dev)
    a, b = t(Ab), t(Bb)
    sa, sb = torch.tensor(SA.astype(np.int32)).to(dev), torch.tensor(SB.astype(np.int32)).to(dev)
    d = torch.zeros(64, 4, device=dev)
    assert lib.mx_probe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sa.data_ptr()), C.c_void_p(sb.data_ptr()),
                        C.c_void_p(d.data_ptr()), opa, opb, None) == 0
    torch.cuda.synchronize()
    d = d.cpu().numpy()
    out = np.zeros((16, 16), np.float32)
    for lane in range(64):
        for r in range(4):
            out[4 * (lane >> 4) + r, lane & 15] = d[lane, r]
    return out


ones = np.full((64, 32), ONE, np.uint8)
s1 = np.full(64, 127, np.int64)
for opa, byte in ((0, 0), (1, 1), (2, 2), (3, 3)):
    print(f"== scale in byte {byte} of the scale VGPR, opsel_a = {opa}")
    for l0 in (0, 5, 16, 37, 48, 63):
        sa = s1.copy()
        sa[l0] = (sa[l0] & ~(0xFF << (8 * byte))) | (128 << (8 * byte))      # 2^1 in the selected byte of lane l0; the other bytes 2^0
        for b_ in range(4):
            if b_ != byte:
                sa[l0] = (sa[l0] & ~(0xFF << (8 * b_))) | (127 << (8 * b_))
        sa_all = np.array([v if i == l0 else (127 | 127 << 8 | 127 << 16 | 127 << 24) for i, v in enumerate(sa)], np.int64)
        hit = np.zeros((4, 32), np.int32)
        for kb in range(4):
            for jb in range(32):
                Bb = np.zeros((64, 32), np.uint8)
                Bb[16 * kb + 3, jb] = ONE                      # column 3, k position (kb, jb)
                d = run(ones, Bb, sa_all, np.full(64, 127 | 127 << 8 | 127 << 16 | 127 << 24, np.int64), opa, 0)
                hit[kb, jb] = int(round(float(d[l0 & 15, 3])))
        rows = ["".join(str(v) for v in hit[kb]) for kb in range(4)]
        print(f"   scale lane {l0:2d} (row {l0 & 15}, lane>>4 = {l0 >> 4}): D values per (lane>>4 of data, byte 0..31):")
        for kb in range(4):
            print(f"        data lane>>4 = {kb}: {rows[kb]}")
