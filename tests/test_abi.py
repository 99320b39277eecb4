"""CPU: the C-ABI library loads and exports every symbol include/gaussctrl_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names += re.findall(r"\b(gc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "gaussctrl_amd", "libgaussctrl_hip.so")):
        ge.build()
    return os.path.join(ROOT, "gaussctrl_amd", "libgaussctrl_hip.so")


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(built)
    decl = _declared()
    assert len(decl) >= 15
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing


def test_loader_symbol_list_matches_header(built):
    from gaussctrl_amd import _lib
    assert sorted(_lib.SYMBOLS) == _declared()
    l = _lib.lib()
    assert l.gc_abi_version() >= 1
    assert l.gc_raster_scan_workspace_bytes(ctypes.c_int64(5000)) >= 3 * 4


def test_documented_entry_point_count_is_current():
    """DESIGN.md / INTEGRATION.md state how many entry points the header declares: the number must be the header's (it went stale twice)"""
    n = len(_declared())
    for doc in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        found = [int(m) for m in re.findall(r"(\d+) (?:`extern \"C\"` )?entry points", text)]
        assert found, doc
        assert all(f == n for f in found), (doc, found, n)


def test_product_path_refuses_cpu_tensors(built):
    import torch
    from gaussctrl_amd import gsplat_ops as ops
    from gaussctrl_amd._lib import GaussCtrlHipError
    with pytest.raises(GaussCtrlHipError):
        ops.spherical_harmonics(3, torch.zeros(4, 3), torch.zeros(4, 16, 3))


def test_no_oracle_import_in_product():
    """the product package must never import oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "gaussctrl_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, fn)


def test_library_has_no_packed_fp32_arithmetic(built):
    """DESIGN.md 7.0 (round 4): v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 return wrong results in lanes 48..63 on the MI355X boxes of this
    pool whenever a wavefront of ANOTHER process issues MFMAs on the same SIMD (profiles/r04_packed_fp32_fault.txt; found through the
    two-ranks-on-one-GPU test).  The library is built with the SLP and loop vectorizers off, which is where every one of them came from;
    this holds the disassembly of the shipped code objects at zero."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("packed_fp32_audit", os.path.join(ROOT, "scripts", "packed_fp32_audit.py"))
    audit = importlib.util.module_from_spec(spec); spec.loader.exec_module(audit)
    found, functions = audit.audit(built)
    assert functions > 300, functions            # every translation unit's code object was found
    assert not found, found


def test_descriptor_structs_match_the_header(tmp_path):
    """the ctypes mirrors of gc_gemm_desc / gc_attn_desc (gaussctrl_amd/sd/ops.py) have the size and the last-field offset the C compiler
    gives the header's structs: a field added on one side only would shift every later argument silently."""
    import subprocess
    from gaussctrl_amd.sd import ops
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gaussctrl_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(gc_gemm_desc), offsetof(gc_gemm_desc, softmax_keys), sizeof(gc_attn_desc),'
                   ' offsetof(gc_attn_desc, workspace_bytes)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(ops.GemmDesc), ops.GemmDesc.softmax_keys.offset, ctypes.sizeof(ops.AttnDesc), ops.AttnDesc.workspace_bytes.offset]


def test_fp8_linear_copies_of_transformer_weights():
    """weights.add_fp8_linears (host side, CPU): e4m3 copies exist for the C % 128 == 0 blocks only, dequantise to within one e4m3 step of
    the prepared 2-byte weights (row scale = power of two, row maximum in the top binade), and keep the GEGLU row permutation."""
    import torch
    from gaussctrl_amd.sd.weights import add_fp8_linears
    g = torch.Generator().manual_seed(0)
    out = {}
    for name, C in (("a.transformer_blocks.0.", 640), ("b.transformer_blocks.0.", 320)):
        for lin, (n, k) in (("attn1.to_qkv", (3 * C, C)), ("attn2.to_q", (C, C)), ("ff.net.0.proj", (8 * C, C)), ("ff.net.2", (C, 4 * C))):
            out[name + lin + ".weight"] = (torch.randn(n, k, generator=g) * k ** -0.5 * torch.exp2(torch.randint(-6, 6, (n, 1), generator=g).float())).to(torch.bfloat16)
    add_fp8_linears(out, 5)
    assert out["_fp8_linears"] == 5
    assert not any(k.startswith("b.") and k.endswith(".w8") for k in out)
    for lin in ("attn1.to_qkv", "attn2.to_q", "ff.net.0.proj", "ff.net.2"):
        w = out["a.transformer_blocks.0." + lin + ".weight"].double()
        q, sc = out["a.transformer_blocks.0." + lin + ".w8"], out["a.transformer_blocks.0." + lin + ".w8_scale"]
        assert q.dtype == torch.uint8 and q.shape == w.shape and sc.shape == (w.shape[0],)
        deq = q.view(torch.float8_e4m3fn).double() * torch.exp2(sc.double() - 127)[:, None]
        amax = w.abs().amax(1, keepdim=True)
        assert bool(((deq - w).abs() <= 0.0626 * w.abs() + amax * 2.0 ** -9).all())          # half an e4m3 step (normal), subnormal floor
        stored_max = q.view(torch.float8_e4m3fn).float().abs().amax(1)
        assert bool((stored_max >= 224).all()) and bool((stored_max <= 448).all())


def test_fp8_gemm_planning_queries(built):
    """host-side planning of the fp8 GEMM (no GPU needed: gc_dn_gemm_workspace_bytes / gc_dn_gemm_chan_parts_layout are pure functions of
    the descriptor): the 16 x 16-map convolutions at the benchmark's batch are k-sliced and leave their GroupNorm partials through the
    reduce kernel (32-row slabs, 64-column blocks); full grids are not sliced and leave them through k_gemm8q's own epilogue (row-tile
    slabs); GEGLU / e4m3-output / statistics problems are never sliced; with plan_rows (batch-invariant mode) the slice count depends on the
    rows ONE frame contributes, not on the batch."""
    from gaussctrl_amd.sd.ops import GemmDesc
    lib = ctypes.CDLL(built)
    lib.gc_dn_gemm_workspace_bytes.restype = ctypes.c_size_t

    def conv(B, hw, cin, cout, **kw):
        d = GemmDesc()
        d.dtype = 0; d.mode = 1; d.M, d.N, d.K = B * hw * hw, cout, 9 * cin
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.stride, d.pad_lo = B, hw, hw, cin, hw, hw, 1, 1
        d.rows_per_batch = hw * hw; d.fp8 = 1; d.a_scale = 127; d.out = 1; d.ldc = cout; d.lda = cin
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    def splits(d):
        return lib.gc_dn_gemm_workspace_bytes(ctypes.byref(d)) // (4 * d.M * d.N)

    def layout(d, with_ws=True):
        if with_ws:
            d.workspace = 1; d.workspace_bytes = lib.gc_dn_gemm_workspace_bytes(ctypes.byref(d))
        d.gn_groups = 32
        rows, ns, ct = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
        assert lib.gc_dn_gemm_chan_parts_layout(ctypes.byref(d), ctypes.byref(rows), ctypes.byref(ns), ctypes.byref(ct)) == 0
        return rows.value, ns.value, ct.value

    assert splits(conv(6, 16, 1280, 1280)) == 2                 # 120 tiles of 90 k-steps -> 240 workgroups
    assert splits(conv(3, 16, 1280, 1280)) == 5                 # 60 tiles -> 300 workgroups of 18 k-steps
    assert splits(conv(6, 16, 2560, 1280)) == 2
    assert splits(conv(14, 16, 1280, 1280)) == 0                # 280 tiles: a full grid
    assert splits(conv(6, 32, 640, 640)) == 0 and splits(conv(6, 64, 384, 320)) == 0
    assert splits(conv(6, 16, 1280, 1280, geglu=1)) == 0 and splits(conv(6, 16, 1280, 1280, out_fp8=127)) == 0
    assert splits(conv(6, 16, 1280, 1280, out_group_stats=1)) == 0
    assert splits(conv(6, 16, 1280, 1280, kernel_variant=2)) == 0          # a forced tile height does not slice
    assert layout(conv(6, 16, 1280, 1280)) == (32, 8, 64)       # reduce kernel: 32-row slabs, 64-column blocks
    assert layout(conv(6, 16, 1280, 1280), with_ws=False) == (128, 2, 128)   # no workspace -> unsliced, k_gemm8q's own partial epilogue
    assert layout(conv(14, 16, 1280, 1280)) == (192, 3, 128)    # MT 3 tiles of 192 rows straddle the 256-row batches
    assert layout(conv(6, 64, 384, 320)) == (128, 32, 160)      # N = 320 (Cin 320 padded to 384): 160-column tiles
    assert layout(conv(6, 8, 1280, 1280))[0] == 0               # 8 x 8 maps: fewer than 256 rows per batch -> no partials
    # batch-invariant planning: the same slices whatever shares the batch
    a, b = conv(6, 16, 1280, 1280, plan_rows=256), conv(14, 16, 1280, 1280, plan_rows=256)
    assert splits(a) == splits(b) == 10


def test_gemm_planning_matches_the_documented_design(built):
    """the 2-byte GEMM's host-side planning at the benchmark's shapes (CFG batch 6), as DESIGN.md 3.2 / 7.0 states it: 64 x 64 maps at C = 320 on
    192-row x 160-column tiles; 32 x 32 maps on 128 x 128 tiles; the long-K part-filled grids in k-slices whose reduce kernel leaves the GroupNorm
    partials in 32-row slabs; the 8 x 8 maps in 15 slices ("128 x 128 x 15 slices") without partials (fewer than 256 rows per batch); GEGLU
    never sliced.  Pure functions of the descriptor: no GPU needed."""
    from gaussctrl_amd.sd.ops import GemmDesc
    lib = ctypes.CDLL(built)
    lib.gc_dn_gemm_workspace_bytes.restype = ctypes.c_size_t

    def conv(B, hw, cin, cout):
        d = GemmDesc()
        d.dtype = 0; d.mode = 1; d.M, d.N, d.K = B * hw * hw, cout, 9 * cin
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.stride, d.pad_lo = B, hw, hw, cin, hw, hw, 1, 1
        d.rows_per_batch = hw * hw; d.out = 1; d.ldc = cout; d.lda = cin; d.zeros = 1
        return d

    def lin(M, N, K, rpb=0, geglu=0):
        d = GemmDesc()
        d.dtype = 0; d.mode = 0; d.M, d.N, d.K = M, N, K
        d.lda = K; d.out = 1; d.ldc = N; d.zeros = 1; d.rows_per_batch = rpb; d.geglu = geglu
        return d

    def plan(d):
        ws = lib.gc_dn_gemm_workspace_bytes(ctypes.byref(d))
        d.workspace = 1; d.workspace_bytes = ws; d.gn_groups = 32
        rows, ns, ct = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
        assert lib.gc_dn_gemm_chan_parts_layout(ctypes.byref(d), ctypes.byref(rows), ctypes.byref(ns), ctypes.byref(ct)) == 0
        return ws // (4 * d.M * d.N), (rows.value, ns.value, ct.value)

    assert plan(conv(6, 64, 320, 320)) == (0, (192, 23, 160))
    assert plan(conv(6, 32, 640, 640)) == (0, (128, 8, 128))
    assert plan(conv(6, 16, 1280, 1280)) == (4, (32, 8, 64))
    assert plan(conv(6, 16, 2560, 1280)) == (4, (32, 8, 64))
    assert plan(conv(6, 8, 1280, 1280)) == (15, (0, 0, 0))
    assert plan(lin(6144, 640, 640, 1024)) == (0, (128, 8, 128))              # proj_out of a 32 x 32 block
    assert plan(lin(1536, 1280, 5120, 256)) == (4, (32, 8, 64))               # FF down projection, 16 x 16 block
    assert plan(lin(384, 1280, 5120, 64))[0] == 10                            # ... 8 x 8 block
    assert plan(lin(6144, 5120, 640, geglu=1)) == (0, (0, 0, 0))
