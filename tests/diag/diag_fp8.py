import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.diaglib as D
if len(sys.argv) > 1: D.use(sys.argv[1])
from unet_bssfp_amd import functional as Fn, ops
from unet_bssfp_amd.nn import Conv3d
DEV="cuda:0"
def q8(t):
    amax=float(t.abs().max()); s=224.0/amax
    return (t*s).to(torch.float8_e4m3fn).float()/s
g=torch.Generator().manual_seed(3); torch.manual_seed(1)
layer=Conv3d(32,32,3,1,1); layer.fp8=True
sp=(32,64,128)
x=(torch.rand(1,32,*sp,generator=g)-0.3).to(torch.bfloat16).float()
z_ref=F.conv3d(q8(x), q8(layer.weight.detach()), layer.bias.detach(),1,1)
layer=layer.to(DEV)
a=ops.new_act(1,*sp,32,torch.bfloat16,DEV); ops.pack_ncdhw(x.to(DEV).contiguous(), a, 0, 32)
with torch.no_grad():
    z,_=Fn.ConvFn.apply(a,None,layer.weight,layer.bias,layer.spec,True,False,0,True)
zz=ops.unpack_ncdhw(z,32,0).cpu()
bad=~torch.isclose(zz,z_ref,rtol=1e-2,atol=1e-3)
print("bad", int(bad.sum()), "nan", int(torch.isnan(zz).sum()))
idx=bad.nonzero()
for dim,name in ((1,'c'),(2,'d'),(3,'h'),(4,'w')):
    vals,cnt=torch.unique(idx[:,dim],return_counts=True)
    print(name, list(zip(vals.tolist(),cnt.tolist()))[:40])
print(zz[bad][:10], z_ref[bad][:10])
