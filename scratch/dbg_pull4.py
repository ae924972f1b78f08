import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import bpr_mf as obpr
from yelprecommendation_amd.bpr_step import BPRMFStep
dev=torch.device('cuda:0')
np.set_printoptions(linewidth=200, precision=3)
found=0
for trial in range(40):
    rs=np.random.RandomState(705+trial)
    nu,ni,d,B=211,307,64,257
    U=(rs.standard_normal((nu,d))*0.2).astype(np.float32); I=(rs.standard_normal((ni,d))*0.2).astype(np.float32)
    st=BPRMFStep(torch.from_numpy(U).to(dev),torch.from_numpy(I).to(dev),lr=5e-3,impl='pull')
    u=rs.randint(0,nu,B).astype(np.int64); p=rs.randint(0,ni,B).astype(np.int64); n=rs.randint(0,ni,B).astype(np.int64)
    loss,gU,gI=obpr.loss_and_grads(U,I,u,p,n)
    st.step(*(torch.from_numpy(a).to(dev) for a in (u,p,n)))
    # recover grads from m after one step: m = 0.1*g
    gI_gpu=st.mI.cpu().numpy()/np.float32(0.1); gU_gpu=st.mU.cpu().numpy()/np.float32(0.1)
    dI=np.abs(gI_gpu-gI).max(1); dU=np.abs(gU_gpu-gU).max(1)
    bad=np.nonzero(dI>1e-6)[0]; badu=np.nonzero(dU>1e-6)[0]
    if len(bad) or len(badu):
        found+=1
        print('trial',trial,'bad items',bad,'bad users',badu)
        for r in bad[:2]:
            occ=[(b,'p') for b in np.nonzero(p==r)[0]]+[(b,'n') for b in np.nonzero(n==r)[0]]
            print(' item',r,'occurrences',occ)
            print('  diff',(gI_gpu[r]-gI[r])[:16])
            print('  ref ',gI[r][:16])
            # does the diff equal a multiple of some user row?
            for b,kind in occ:
                ur=U[u[b]]
                ratio=(gI_gpu[r]-gI[r])/ur
                print('   vs user',u[b],'ratio spread',ratio.min(),ratio.max())
        if found>=3: break
print('found',found)
