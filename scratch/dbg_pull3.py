import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import bpr_mf as obpr
from yelprecommendation_amd.bpr_step import BPRMFStep
dev=torch.device('cuda:0')
def run(zero_ws, zero_alt, fill):
    junk=torch.full((64<<20,), fill, dtype=torch.float32, device=dev); del junk
    rs=np.random.RandomState(705)
    nu,ni,d,B=211,307,64,257
    U=(rs.standard_normal((nu,d))*0.2).astype(np.float32); I=(rs.standard_normal((ni,d))*0.2).astype(np.float32)
    ref=obpr.MFState(U,I,'adam',lr=5e-3)
    st=BPRMFStep(torch.from_numpy(U).to(dev),torch.from_numpy(I).to(dev),lr=5e-3,impl='pull')
    st._workspace(B)
    if zero_ws: st._ws.zero_()
    if zero_alt: st._U_alt.zero_()
    out=[]
    for k in range(2):
        u=rs.randint(0,nu,B).astype(np.int64); p=rs.randint(0,ni,B).astype(np.int64); n=rs.randint(0,ni,B).astype(np.int64)
        ref.train_step(u,p,n)
        st.step(*(torch.from_numpy(a).to(dev) for a in (u,p,n)))
        dU=np.abs(st.U.cpu().numpy()-ref.U).max(1); dI=np.abs(st.I.cpu().numpy()-ref.I).max(1)
        out.append(((dU>1e-5).sum(), (dI>1e-5).sum(), np.nonzero(dU>1e-5)[0][:6].tolist(), np.nonzero(dI>1e-5)[0][:6].tolist()))
    return out
for fill in (float('nan'), 1e30, 7.0):
    for zw,za in ((0,0),(1,0),(0,1),(1,1)):
        print('fill',fill,'zero_ws',zw,'zero_alt',za, run(zw,za,fill))
