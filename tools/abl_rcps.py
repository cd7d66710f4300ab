import sys; sys.path.insert(0,'.')
import torch
from im2im_uq_amd import hip_ops
from im2im_uq_amd._lib import lib
dev='cuda:0'; M=432; hw=320
g=torch.Generator(device=dev).manual_seed(0)
pred=torch.rand(M,1,hw,hw,device=dev,generator=g)
out3=torch.stack([pred-0.05*torch.rand_like(pred),pred,pred+0.05*torch.rand_like(pred)],1).contiguous()
lab=pred+0.05*torch.randn(pred.shape,device=dev,generator=g)
lam=torch.linspace(0,6,1000); lam=(lam-(lam[1]-lam[0])).to(dev)
import os
table=torch.empty((M,1000),device=dev)
x=torch.empty(M*hw*hw*4, device=dev); 
for rnd in range(2):
  for bpc,mode in ((8,0),):
    hist=torch.empty((lib.im2im_rcps_workspace_bytes(M,hw*hw,1000)//4,),dtype=torch.int32,device=dev)
    for _ in range(5): hip_ops.rcps_loss_table_raw(out3,lab,M,hw*hw,lam,hist,table)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): hip_ops.rcps_loss_table_raw(out3,lab,M,hw*hw,lam,hist,table)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print(f"bpc {bpc:3d} mode {mode} ms {ms:.4f}  GB/s {M*16*hw*hw/ms/1e6:.0f}")
# reference: plain copy-style read of the same bytes with torch (sum) for the achievable rate on this box
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
for _ in range(3): out3.sum()
e0.record()
for _ in range(20): out3.sum()
e1.record(); torch.cuda.synchronize()
print("torch sum of out3 GB/s", out3.numel()*4/(e0.elapsed_time(e1)/20)/1e6)
