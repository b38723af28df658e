import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boostmvsnerfs_amd import ops
from boostmvsnerfs_amd.synthetic import make_batch
dev="cuda"
H,W=256,320
b=make_batch(H,W,device=dev)
torch.manual_seed(0)
C,D,h,w,Hs,Ws=16,8,128,160,128,160
feats=torch.randn(1,3,C,Hs,Ws,device=dev)
P=ops.proj_mats(b["src_exts"],b["src_ixts"],b["tar_ext"],b["tar_ixt"],0.5,0.5)
dv=(3.0+torch.rand(1,1,h,w,device=dev)+torch.linspace(-1,1,D,device=dev).view(1,-1,1,1)).contiguous()
os.environ["BMV_SWEEP_WIN_FLAGS"]="34"
o=ops.sweep_variance(feats,P,dv,algo=40)
torch.cuda.synchronize()
o=o[0].cpu()
for (d,y,x) in [(0,0,0),(0,0,31),(0,7,0),(3,8,32),(7,100,100)]:
    print((d,y,x), [round(float(o[c,d,y,x]),4) for c in range(16)], "1/dv", float(1/dv[0,d,y,x]))
p=P[0,0].cpu()
for (d,y,x) in [(0,0,0),(0,7,31),(7,100,100)]:
    i=1/float(dv[0,d,y,x]); px=p[0,0]*x+p[0,1]*y+p[0,2]+p[0,3]*i; py=p[1,0]*x+p[1,1]*y+p[1,2]+p[1,3]*i; pz=p[2,0]*x+p[2,1]*y+p[2,2]+p[2,3]*i
    print("expect view0 tap", (d,y,x), float(px/pz), float(py/pz))
