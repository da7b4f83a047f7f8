import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from lv_slam_amd import ndt, synth
from oracle import oracle_py as O
from conftest import se3_err
for off in ([0,0,0],[500,-300,40],[5000,-3000,40]):
    base,_,_ = synth.make_pair(50,128,n_beams=32); base=base.numpy()
    off=np.float32(off)
    tgt=(base+off).astype(np.float32); src=(base[::3]+off+np.float32([0.3,-0.2,0.05])).astype(np.float32)
    kw=dict(trans_epsilon=0.01,max_iterations=64)
    eng=ndt.Engine(ndt.default_params(**kw)); grid=O.Grid(tgt,O.default_params(**kw))
    eng.set_target(tgt); eng.set_source(src)
    G=np.eye(4,dtype=np.float32)
    r,ro=eng.align(G),O.align(grid,src,G)
    c=np.append(tgt[np.isfinite(tgt).all(1)].mean(0).astype(np.float64),1.0)
    d=(r["final"].astype(np.float64)-ro["final"].astype(np.float64))@c
    print(off, r["iterations"],ro["iterations"], se3_err(ro["final"],r["final"]), "displacement at data centroid %.3e m"%np.linalg.norm(d[:3]), "score", r["score"], ro["score"])
