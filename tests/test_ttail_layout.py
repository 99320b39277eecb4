"""Host side of the fused transformer tail (gaussctrl_amd/sd/weights.py::mfma_blocks / lane_order / tail_streams, csrc/dn_ttail.hip):
the operand-stream layout is checked on the CPU by emulating v_mfma_f32_32x32x16's lane <-> element mapping -- the accumulator registers
of one GEMM, taken 8 at a time, must BE the B operand of the next GEMM when its weight tiles use the PERM16 k-order."""
import numpy as np
import torch

from gaussctrl_amd.sd import weights


def mfma32(a_blk, b_frag):
    """one 32x32x16 MFMA: a_blk, b_frag [64 lanes, 8] -> accumulator registers [64 lanes, 16]"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        hg, i = l >> 5, l & 31
        A[i, 8 * hg:8 * hg + 8] = a_blk[l]
        B[8 * hg:8 * hg + 8, i] = b_frag[l]
    D = A @ B
    out = np.zeros((64, 16))
    for l in range(64):
        hg, j = l >> 5, l & 31
        for r in range(16):
            out[l, r] = D[8 * (r >> 2) + 4 * hg + (r & 3), j]
    return out


def gemm(blocks, frags):
    """blocks [KS, NB, 64, 8], frags [KS][64, 8] -> accumulators [NB][64, 16]"""
    KS, NB = blocks.shape[:2]
    acc = [np.zeros((64, 16)) for _ in range(NB)]
    for ks in range(KS):
        for nb in range(NB):
            acc[nb] += mfma32(blocks[ks, nb], frags[ks])
    return acc


def rows_to_frags(x):
    """x [32 rows, K] -> lane-order B fragments, one per k-step (what the kernel's load_rows builds from 8-byte loads)"""
    K = x.shape[1]
    fr = []
    for ks in range(K // 16):
        f = np.zeros((64, 8))
        for l in range(64):
            hg, m = l >> 5, l & 31
            for t in range(8):
                f[l, t] = x[m, 16 * ks + weights.PERM16[8 * hg + t]]
        fr.append(f)
    return fr


def test_accumulator_registers_are_the_next_b_operand():
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, (32, 32)).astype(np.float64)
    w1 = rng.integers(-3, 4, (64, 32)).astype(np.float64)
    w2 = rng.integers(-3, 4, (32, 64)).astype(np.float64)
    b1 = weights.mfma_blocks(torch.from_numpy(w1)).numpy()
    b2 = weights.mfma_blocks(torch.from_numpy(w2)).numpy()
    acc = gemm(b1, rows_to_frags(x))                       # y1[n][m] in accumulator layout
    frags = []
    for nb in range(2):
        for j in range(2):
            frags.append(acc[nb][:, 8 * j:8 * j + 8])     # registers 8 j .. 8 j + 7 of block nb = k-step 2 nb + j
    acc2 = gemm(b2, frags)
    ref = (x @ w1.T) @ w2.T                                # [m, n]
    for l in range(64):
        hg, m = l >> 5, l & 31
        for r in range(16):
            assert acc2[0][l, r] == ref[m, 8 * (r >> 2) + 4 * hg + (r & 3)]


def test_lane_order_matches_accumulator_rows():
    v = torch.arange(64.)
    lo = weights.lane_order(v).numpy()
    for nb in range(2):
        for hg in range(2):
            for r in range(16):
                assert lo[32 * nb + 16 * hg + r] == 32 * nb + 8 * (r >> 2) + 4 * hg + (r & 3)


def test_tail_stream_sizes_and_geglu_pairing():
    torch.manual_seed(0)
    C, FFN = 320, 1280
    p = "tb"; t = p + ".transformer_blocks.0"
    sd = {p + ".proj_out.weight": torch.randn(C, C, 1, 1), p + ".proj_out.bias": torch.randn(C),
          t + ".ff.net.0.proj.weight": torch.randn(2 * FFN, C), t + ".ff.net.0.proj.bias": torch.arange(2. * FFN),
          t + ".ff.net.2.weight": torch.randn(C, FFN), t + ".ff.net.2.bias": torch.randn(C)}
    for n in ("norm2", "norm3"):
        sd[t + f".{n}.weight"] = torch.randn(C); sd[t + f".{n}.bias"] = torch.randn(C)
    for a_ in ("attn1", "attn2"):
        sd[t + f".{a_}.to_out.0.weight"] = torch.randn(C, C); sd[t + f".{a_}.to_out.0.bias"] = torch.randn(C)
    sd[t + ".attn2.to_q.weight"] = torch.randn(C, C)
    out = {}
    weights.tail_streams(out, lambda n: sd[n].float().reshape(sd[n].shape[0], -1).squeeze(-1) if sd[n].dim() == 4 else sd[n].float(), p, 8, torch.float32)
    assert out[p + ".tail.a"].numel() == 400 * 512 and out[p + ".tail.b"].numel() == 2800 * 512      # 1 KB blocks of 512 2-byte elements
    prm = out[p + ".tail.params"]
    assert prm.numel() == 2560 + 2560 + 2 * weights.GELU_N
    x = torch.tensor([-3.0, -0.5, 0.0, 0.75, 2.5])          # table entries at x * 128 + 1024
    tab = prm[5120:].reshape(-1, 2)
    assert torch.allclose(tab[(x * 128 + 1024).long(), 0], torch.nn.functional.gelu(x), atol=1e-6)
    # GEGLU bias, lane order of up-block 0: registers 0..3 / 8..11 hold hidden channels, 4..7 / 12..15 their gates (hidden + 1280)
    bup = prm[2560:2560 + 32].reshape(2, 16)
    for hg in range(2):
        for pr in range(2):
            for c in range(4):
                assert bup[hg, 8 * pr + c] == 8 * pr + 4 * hg + c
                assert bup[hg, 8 * pr + 4 + c] == FFN + 8 * pr + 4 * hg + c
    # text stream: ones row (channel 40 of every head) only under valid keys
    k = torch.randn(2, 77, C); vt = torch.randn(2, C, 80)
    kv = weights.tail_text_stream(k, vt, 77, 8)
    assert kv.shape == (2, 8 * 21 * 512)
