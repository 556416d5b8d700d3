import sys, os, math, ctypes as C
R=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from lookaheaddecoding_amd import ops, cabi
from lookaheaddecoding_amd.cabi import AttnArgs, ptr, call
torch.manual_seed(0)
H=Hkv=1; dh=64; T=32; P=0; S_max=128
q=(torch.arange(T*dh).view(T,dh).float()/64).bfloat16().cuda()   # value encodes (row, col)
k=torch.randn(Hkv,S_max,dh).bfloat16().cuda(); vt=torch.randn(Hkv,dh,S_max).bfloat16().cuda()
out=torch.empty(T,H*dh,dtype=torch.bfloat16,device="cuda")
part_ml=torch.zeros(4096,dtype=torch.float32,device="cuda")
part_o=torch.zeros(T*H*dh,dtype=torch.bfloat16,device="cuda")
a=AttnArgs(ptr(q),ptr(k),ptr(vt),ptr(out),ptr(part_o),ptr(part_ml),None,q.stride(0),out.stride(0),H,Hkv,dh,S_max,0,1,1/8.0,ops.StepMask(T=T,P=P,is_prefill=True).c_struct())
call("lade_attn_fwd", C.byref(a)); torch.cuda.synchronize()
d=part_ml.view(torch.int32).cpu()
n=d[0].item(); print("mismatches", n)
qh=q.cpu().view(torch.int16).int() & 0xffff
for i in range(min(n,24)):
    row,chunk,exp,got=d[4+i*4:8+i*4].tolist()
    gw = got & 0xffff
    loc=(qh==gw).nonzero()
    print("row",row,"chunk",chunk,"expected word0 %08x got %08x"%(exp&0xffffffff,got&0xffffffff), "got value found at (row,col):", loc[:3].tolist())
sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, lade_oracle as O
vis=np.tril(np.ones((T,T),dtype=bool))
ref=O.attention_dense(q.cpu().float().view(T,H,dh).transpose(0,1), k.cpu().float()[:,:T], vt.cpu().float().transpose(1,2)[:,:T], vis).transpose(0,1).reshape(T,H*dh)
for rep in range(3):
    call("lade_attn_fwd", C.byref(a)); torch.cuda.synchronize()
    err=(out.float().cpu()-ref).abs()
    print("rep",rep,"max err",err.max().item(),"bad rows",(err.amax(1)>0.05).nonzero().flatten().tolist())
