import sys, time
sys.path.insert(0,'.')
from yelprecommendation_amd.train import main
t=time.time()
m=main(["model_name=MF","synthetic=1000x1000x30","epochs=4","batch_size=256","lr=0.005","embed_size=32","model_dir=/tmp/yr_models","fast_loader=true"])
print('MF fast loader', m, time.time()-t)
t=time.time()
m=main(["model_name=MF","synthetic=1000x1000x30","epochs=2","batch_size=256","lr=0.005","embed_size=32","model_dir=/tmp/yr_models"])
print('MF DataLoader', m, time.time()-t)
t=time.time()
m=main(["model_name=NGCF","synthetic=300x200x12","epochs=2","batch_size=64","lr=0.01","embed_size=32","model_dir=/tmp/yr_models2"])
print('NGCF', m, time.time()-t)
