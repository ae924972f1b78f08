import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import bpr_mf as obpr
from yelprecommendation_amd.bpr_step import BPRMFStep
dev=torch.device('cuda:0')
rs=np.random.RandomState(7*64+257)
nu,ni,d,B=211,307,64,257
U=(rs.standard_normal((nu,d))*0.2).astype(np.float32); I=(rs.standard_normal((ni,d))*0.2).astype(np.float32)
ref=obpr.MFState(U,I,'adam',lr=5e-3)
st=BPRMFStep(torch.from_numpy(U).to(dev),torch.from_numpy(I).to(dev),lr=5e-3,impl='pull')
for k in range(4):
    b = B if k!=2 else B//3
    u=rs.randint(0,nu,b).astype(np.int64); p=rs.randint(0,ni,b).astype(np.int64); n=rs.randint(0,ni,b).astype(np.int64)
    l=ref.train_step(u,p,n)
    st.step(*(torch.from_numpy(a).to(dev) for a in (u,p,n)))
    torch.cuda.synchronize()
    dU=np.abs(st.U.cpu().numpy()-ref.U).max(1); dI=np.abs(st.I.cpu().numpy()-ref.I).max(1)
    print(k,'b',b,'loss',l,st.loss.item(),'bad users',(dU>1e-5).sum(),'bad items',(dI>1e-5).sum(), np.nonzero(dU>1e-5)[0][:10], np.nonzero(dI>1e-5)[0][:10])
