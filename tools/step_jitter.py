import sys, os, time, gc
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lv_slam_amd import ndt, synth
B, NAZ = 271, 1024
dev = torch.device("cuda:0"); N = NAZ*64
T = torch.empty(B,3,N,device=dev); S = torch.empty(B,3,N,device=dev)
for b in range(B):
    t,s,_ = synth.make_pair(b,NAZ,device=dev); T[b]=t.T; S[b]=s.T
eng = ndt.Engine(ndt.default_params(trans_epsilon=0.01,max_iterations=64),device=0)
eng.batch_bind_device(T.data_ptr(),[N]*B,N,S.data_ptr(),[N]*B,N)
G = synth.default_guess(); g = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1,16),(B,16)),dtype=np.float32); r=(ndt.Result*B)()
for prof in (False, True):
    eng.profile_enable(prof); eng.profile_reset()
    for _ in range(5): eng.batch_build_targets(); eng.batch_align_raw(g,r)
    gc.collect(); gc.disable()
    ts=[]
    for _ in range(400):
        t0=time.perf_counter(); eng.batch_build_targets(); t1=time.perf_counter(); eng.batch_align_raw(g,r); t2=time.perf_counter()
        ts.append((t1-t0,t2-t1))
    gc.enable()
    a=np.array(ts)*1e3; tot=a.sum(1)
    print(f"prof={prof}: median {np.median(tot):.3f} ms, mean {tot.mean():.3f}, p99 {np.percentile(tot,99):.3f}, max {tot.max():.3f}; steps > 1.2x median: {(tot>1.2*np.median(tot)).sum()} of {len(tot)}")
    big=np.argsort(tot)[-5:]
    print("   worst steps (build ms, align ms):", [(round(a[i,0],2),round(a[i,1],2)) for i in big])
eng.close()
