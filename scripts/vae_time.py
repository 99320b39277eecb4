import sys, time, torch
sys.path.insert(0, '/root/repo')
from oracle import sd15_torch as sd
from gaussctrl_amd.sd.pipeline import to_nhwc8
from gaussctrl_amd.sd.vae import VAEDecoder, prepare_vae_weights
from gaussctrl_amd.sd import ops
DEV='cuda:0'; dt=torch.bfloat16
vw = sd.make_vae_decoder_weights(sd.VAE_SD, 300)
dec = VAEDecoder(prepare_vae_weights(vw, dt, DEV))
lat = torch.randn(3, 4, 64, 64)
x = to_nhwc8(lat.to(DEV), dt)
for _ in range(2): dec.decode(x, postprocess=True)
torch.cuda.synchronize()
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): dec.decode(x, postprocess=True)
e.record(); torch.cuda.synchronize()
print("VAE decode of 3 views: %.2f ms" % (s.elapsed_time(e)/3))
# per-op timing
rec=[]
_conv=ops.conv3x3; _lin=ops.linear
def conv(xx,w,*a,**k):
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record(); o=_conv(xx,w,*a,**k); e.record(); rec.append(("conv %s->%d %s"%(tuple(xx.shape),w.shape[0],k.get('upsample',a)), 2.0*o.numel()//o.shape[-1]*w.shape[0]*w.shape[1], s,e)); return o
ops.conv3x3=conv
import gaussctrl_amd.sd.vae as V
dec.decode(x, postprocess=True); torch.cuda.synchronize()
tot=0
for n,fl,s,e in rec:
    t=s.elapsed_time(e)*1e3; tot+=t
    print("%-60s %8.1f us %7.1f TF/s"%(n[:60],t,fl/t/1e6))
print("conv total %.2f ms"%(tot/1e3))
