import torch, sys
sys.path.insert(0,'.')
from yelprecommendation_amd import engine
dev=torch.device('cuda:0')
nu,ni,d=31668,38048,int(sys.argv[4]) if len(sys.argv) > 4 else 64
g=torch.Generator(device=dev).manual_seed(0)
U=torch.randn(nu,d,device=dev,generator=g)*0.1; I=torch.randn(ni,d,device=dev,generator=g)*0.1
users=torch.arange(nu,device=dev)
cnt=torch.randint(10,60,(nu,),device=dev,generator=g)
ptr=torch.zeros(nu+1,dtype=torch.int64,device=dev); ptr[1:]=torch.cumsum(cnt,0)
idx=torch.randint(0,ni,(int(ptr[-1]),),device=dev,generator=g)
sidx=engine.sort_mask_rows(ptr,idx)
# argv[1]: "f32" / "bf16x3" (default); argv[2]: "hint" = every call after the first takes the previous lists as hints;
# argv[3]: "two_roles" / "four_waves" = that form of the sweep (default: the library's rule); argv[4]: D (default 64)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
hinted = len(sys.argv) > 2 and sys.argv[2] == "hint"
form = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] in ("two_roles", "four_waves") else None
top = None
for _ in range(4):
    top = engine.mf_eval_topk(U,I,users,ptr,sidx,10,precision=prec,hint=top if hinted else None,form=form)
torch.cuda.synchronize()
