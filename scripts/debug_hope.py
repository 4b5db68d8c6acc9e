import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from conftest import load_karate, load_sbm1024, golden_path
from gem_amd.embedding.hope import HOPE
from gem_amd.graph import edge_arrays
from oracle import hope_oracle
for name,G,d in (('karate',load_karate(),4),('sbm',load_sbm1024(),32)):
    n,src,dst,w,order=edge_arrays(G)
    A=hope_oracle.adjacency(n,src,dst,w,order)
    Xo,so=hope_oracle.hope_dense(A,0.01,d)
    for kw in (dict(), dict(krylov_steps=1, max_restarts=60), dict(oversample=4)):
        m=HOPE(d=d,beta=0.01,**kw)
        Y=m.learn_embedding(graph=G)
        print(name,kw,'sigma dev',m._sigma[-4:],'oracle',so[-4:],'stats',m._stats)
        k=d//2
        R=Y[:,:k]@Y[:,k:].T; Ro=Xo[:,:k]@Xo[:,k:].T
        print('   recon rel err',np.linalg.norm(R-Ro)/np.linalg.norm(Ro))
        HOPE.hyper_params.pop('krylov_steps',None); HOPE.hyper_params.pop('max_restarts',None); HOPE.hyper_params.pop('oversample',None)
