import torch, time, sys
sys.path.insert(0,'.')
from yelprecommendation_amd import engine
dev=torch.device('cuda:0')
nu,ni,d=31668,38048,64
g=torch.Generator(device=dev).manual_seed(0)
U=torch.randn(nu,d,device=dev,generator=g)*0.1; I=torch.randn(ni,d,device=dev,generator=g)*0.1
users=torch.arange(nu,device=dev)
cnt=torch.randint(10,60,(nu,),device=dev,generator=g)
ptr=torch.zeros(nu+1,dtype=torch.int64,device=dev); ptr[1:]=torch.cumsum(cnt,0)
idx=torch.randint(0,ni,(int(ptr[-1]),),device=dev,generator=g)
sidx=engine.sort_mask_rows(ptr,idx)
for fused in (False, True):
    for _ in range(2):
        torch.cuda.synchronize(); t=time.time()
        out=engine.mf_recommend(U,I,users,ptr,idx,10,fused=fused)
        torch.cuda.synchronize(); print('fused' if fused else 'gemm+topk','recommend all users', round((time.time()-t)*1e3,2),'ms')
torch.cuda.synchronize(); t=time.time()
for _ in range(5): o2=engine.mf_eval_topk(U,I,users,ptr,sidx,10)
torch.cuda.synchronize(); print('fused kernel only', round((time.time()-t)/5*1e3,2),'ms  -> ', round(2*nu*ni*d/((time.time()-t)/5)/1e12,1),'TFLOP/s f32')
a=engine.mf_recommend(U,I,users,ptr,idx,10,fused=False); print('agree rows', (a==o2).all(1).float().mean().item())
def tk(name, f, n=5):
    f(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); print(name, round((time.time()-t)/n*1e3,3),'ms')
tk('k=10 masks   ', lambda: engine.mf_eval_topk(U,I,users,ptr,sidx,10))
tk('k=10 no masks', lambda: engine.mf_eval_topk(U,I,users,None,None,10))
tk('k=4  masks   ', lambda: engine.mf_eval_topk(U,I,users,ptr,sidx,4))
tk('k=16 masks   ', lambda: engine.mf_eval_topk(U,I,users,ptr,sidx,16))
tk('k=10 unsliced', lambda: engine.mf_eval_topk(U,I,users,ptr,sidx,10,sliced=False))
tk('scores gemm only (unfused, 4.8 GB out)', lambda: engine.mf_scores_gemm(U,I,users[:8192]), 3)
