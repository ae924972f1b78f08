import torch, time, numpy as np, sys
sys.path.insert(0,'.')
from yelprecommendation_amd import engine
dev=torch.device('cuda:0')
U_, I_, D = 31668, 38048, 64
U=torch.randn(U_,D,device=dev)*0.1; I=torch.randn(I_,D,device=dev)*0.1
gU=torch.zeros_like(U); gI=torch.zeros_like(I)
mU=torch.zeros_like(U); vU=torch.zeros_like(U); mI=torch.zeros_like(I); vI=torch.zeros_like(I)
part=torch.zeros(engine.LOSS_PARTIALS,device=dev)
def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3  # us
for B in [4096, 65536, 262144, 1048576, 4194304]:
    u=torch.randint(0,U_,(B,),device=dev); p=torch.randint(0,I_,(B,),device=dev); n=torch.randint(0,I_,(B,),device=dev)
    t_fb=timeit(lambda: engine.bpr_mf_fwd_bwd(U,I,u,p,n,gU,gI,part))
    t_f=timeit(lambda: engine.bpr_mf_fwd_bwd(U,I,u,p,n,None,None,part))
    t_aU=timeit(lambda: engine.adam_dense(U,gU,mU,vU,1,1e-4,zero_grad=True))
    t_aI=timeit(lambda: engine.adam_dense(I,gI,mI,vI,1,1e-4,zero_grad=True))
    def step():
        engine.bpr_mf_fwd_bwd(U,I,u,p,n,gU,gI,part)
        engine.loss_finalize(part,1.0/B)
        engine.adam_dense(U,gU,mU,vU,1,1e-4,zero_grad=True)
        engine.adam_dense(I,gI,mI,vI,1,1e-4,zero_grad=True)
    t_s=timeit(step)
    print(f"B={B:8d} fwd_bwd {t_fb:9.1f}us ({B*1560/t_fb/1e6:7.2f} TB/s alg, {B/t_fb:8.1f} Mtrip/s) fwd-only {t_f:8.1f}us ({B*792/t_f/1e6:6.2f} TB/s) adamU {t_aU:6.1f} adamI {t_aI:6.1f} step {t_s:9.1f}us -> {B/t_s:8.1f} Mtrip/s")
