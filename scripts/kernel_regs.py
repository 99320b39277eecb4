"""Register / scratch usage of every kernel of one translation unit: python scripts/kernel_regs.py gaussctrl_amd/csrc/dn_gemm_cs.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/" + os.path.basename(src) + ".s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-ffp-contract=fast", "-munsafe-fp-atomics",
                       "-fno-honor-nans", "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S):
    ag, name, priv, vg, sp = m.groups()
    if flt in name:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110]
        print(f"vgpr {vg:>3s} agpr {ag:>3s} scratch {priv:>4s} spills {sp:>3s}  {name}")
