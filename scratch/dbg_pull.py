import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import bpr_mf as obpr
from yelprecommendation_amd.bpr_step import BPRMFStep
dev=torch.device('cuda:0')
rs=np.random.RandomState(0)
nu,ni,d,B=211,307,64,257
U=(rs.standard_normal((nu,d))*0.2).astype(np.float32); I=(rs.standard_normal((ni,d))*0.2).astype(np.float32)
u=rs.randint(0,nu,B).astype(np.int64); p=rs.randint(0,ni,B).astype(np.int64); n=rs.randint(0,ni,B).astype(np.int64)
ref=obpr.MFState(U,I,'adam',lr=5e-3)
st=BPRMFStep(torch.from_numpy(U).to(dev),torch.from_numpy(I).to(dev),lr=5e-3,impl='pull')
for k in range(2):
    l=ref.train_step(u,p,n)
    st.step(*(torch.from_numpy(a).to(dev) for a in (u,p,n)))
    print('loss',l,st.epoch_loss())
print('mU diff', np.abs(st.mU.cpu().numpy()-ref.opt.m[0]).max(), 'vU', np.abs(st.vU.cpu().numpy()-ref.opt.v[0]).max(),'mI', np.abs(st.mI.cpu().numpy()-ref.opt.m[1]).max())
dU=np.abs(st.U.cpu().numpy()-ref.U).max(1); dI=np.abs(st.I.cpu().numpy()-ref.I).max(1)
print('bad user rows', np.nonzero(dU>1e-5)[0][:40], (dU>1e-5).sum())
print('bad item rows', np.nonzero(dI>1e-5)[0][:40], (dI>1e-5).sum())
cntI=np.bincount(np.concatenate([p,n]),minlength=ni); cntU=np.bincount(u,minlength=nu)
bi=np.nonzero(dI>1e-5)[0]
print('counts of bad item rows', cntI[bi][:40]); print('counts bad users', cntU[np.nonzero(dU>1e-5)[0]][:40])
# check gradient implied: (p_new - p_old) sign etc
r=bi[0] if len(bi) else 0
print('row',r,'gpu',st.I[r,:8].cpu().numpy(),'ref',ref.I[r,:8],'old',I[r,:8])
