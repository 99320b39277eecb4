"""Packed-fp32 audit of the built library: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 per kernel of libgaussctrl_hip.so.
DESIGN.md 7.0 (round 4): on the MI355X boxes of this pool the upper 16 lanes of these instructions return wrong results when a wavefront of
ANOTHER process issues MFMAs on the same SIMD -- the library is built without them (-fno-slp-vectorize + scalar source where the back end still
packed), and tests/test_build.py holds the count at zero.
    python scripts/packed_fp32_audit.py [path/to/lib.so]        prints the kernels that contain such instructions; exit code 1 if any"""
import os, re, struct, subprocess, sys, tempfile

BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PAT = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")


def code_objects(lib):
    """gfx950 code objects embedded in the shared library (one clang offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        subprocess.check_call([f"{BIN}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", lib, os.path.join(td, "unused")])
        data = open(fb, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                yield data[pos + off:pos + off + size]
        pos += len(MAGIC)


def audit(lib):
    """{kernel symbol: count of packed fp32 arithmetic instructions} for the kernels that have any; and the number of kernels seen."""
    found, kernels = {}, 0
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            dis = subprocess.run([f"{BIN}/llvm-objdump", "-d", f.name], capture_output=True, text=True, check=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                name = m.group(1); kernels += 1
            elif name and PAT.search(line):
                found[name] = found.get(name, 0) + 1
    return found, kernels


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussctrl_amd", "libgaussctrl_hip.so")
    found, kernels = audit(lib)
    for k, v in sorted(found.items(), key=lambda kv: -kv[1]):
        d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{v:6d}  {d[:140]}")
    print(f"{sum(found.values())} packed fp32 arithmetic instructions in {len(found)} of {kernels} functions of {lib}")
    sys.exit(1 if found else 0)
