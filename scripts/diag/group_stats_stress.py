"""Stress of the round-2 group-statistics epilogue (gc_gemm_desc.out_group_stats) on the shape that failed ONCE in the forced GC_GEMM_MT=2 child of the round-6
final-evidence run: conv3x3 B = 14, 8 x 8, 1280 -> 1280 (test_conv_output_group_statistics[14-8-8-1280-1280-1-dt1]).  python scripts/diag/group_stats_stress.py <iters> [mt]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussctrl_amd.sd import ops
from gaussctrl_amd.sd.weights import conv3x3_weight

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda:0"
bad = 0
for dt in (torch.float16, torch.bfloat16):
  for (B, H, Cin, Cout) in ((14, 8, 1280, 1280), (3, 16, 1280, 1280), (2, 32, 640, 640)):
      g = torch.Generator(device=dev).manual_seed(1)
      x = torch.randn(B, H, H, Cin, device=dev, generator=g).to(dt)
      w = conv3x3_weight((torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * (9 * Cin) ** -0.5).to(dt), dt)
      b = torch.randn(Cout, device=dev, generator=g); rv = torch.randn(B, Cout, device=dev, generator=g)
      ops.KERNEL_VARIANT["gemm"] = mt
      ref_out = None
      for it in range(iters):
          gs = torch.zeros(B, 32, 2, device=dev)
          out = ops.conv3x3(x, w, b, rowvec=rv, group_stats=gs)
          if ref_out is None:
              ref_out = out.clone()
              o = out.double().reshape(B, H * H, 32, Cout // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
              ref = torch.stack([o.sum(-1), (o * o).sum(-1)], -1)
          elif not torch.equal(out, ref_out):
              bad += 1; print(dt, it, "OUTPUT differs: max abs", float((out.float() - ref_out.float()).abs().max()))
          err = (gs.double() - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())
          if float(err.max()) >= 2e-5:
              bad += 1
              idx = (err == err.max()).nonzero()[0].tolist()
              print(dt, it, "STATS off: rel", float(err.max()), "at (batch, group, which)", idx, "got", float(gs[tuple(idx)]), "want", float(ref[tuple(idx)]))
      ops.KERNEL_VARIANT["gemm"] = 0
print("bad", bad, "of", 2 * iters)
