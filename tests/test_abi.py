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
                   'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(gc_gemm_desc), offsetof(gc_gemm_desc, out_fp8), sizeof(gc_attn_desc),'
                   ' offsetof(gc_attn_desc, workspace_bytes)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(ops.GemmDesc), ops.GemmDesc.out_fp8.offset, ctypes.sizeof(ops.AttnDesc), ops.AttnDesc.workspace_bytes.offset]


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
