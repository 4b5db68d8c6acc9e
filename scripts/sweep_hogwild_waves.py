import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import ctypes as C, numpy as np
from gem_amd import _hip
from gem_amd.graph import edge_arrays, sbm_graph
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from conftest import load_sbm1024
from test_n2v_gpu import Dev
def run(G, n, src, dst, w, d, waves, seed=1):
    dev = Dev(n, src, dst, w)
    dev.walks(1.0,1.0,10,80,seed,11); dev.unigram()
    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, waves))
    t=time.time(); P,N = dev.sgns(d,10,1,seed,11); el=time.time()-t
    dev.close()
    m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    return gr.evaluateStaticGraphReconstruction(G, m, P.astype(np.float64), None)[0], el
G=load_sbm1024(); n,src,dst,w,_=edge_arrays(G)
for waves in (1,2,4,8,16,32,64,256,1024,8192):
    print('sbm1024 d16 waves',waves, run(G,n,src,dst,w,16,waves), flush=True)
g=sbm_graph(8192, 8192*20, 8, seed=5); n,src,dst,w,_=edge_arrays(g)
for waves in (8,64,256,1024,8192):
    print('sbm8192 d32 waves',waves, run(g,n,src,dst,w,32,waves), flush=True)
