import sys, numpy as np, faulthandler, ctypes as C, threading, time
sys.path.insert(0, '.')
from lv_slam_amd import ndt
rng = np.random.default_rng(0)
tgt = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
src = rng.uniform(-4, 4, (100, 3)).astype(np.float32)
e = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64))
e.set_target(tgt); e.set_source(src)
print(e.derivatives(np.zeros(6))[0], flush=True)
out = (C.c_int * 17)()
e.lib.mi355ndt_debug_ctl.argtypes = [C.c_void_p, C.c_void_p]
print(e.lib.mi355ndt_debug_ctl(e.h, out), list(out), flush=True)
