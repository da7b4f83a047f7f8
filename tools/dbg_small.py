import sys, numpy as np, faulthandler
sys.path.insert(0, '.')
faulthandler.dump_traceback_later(25, exit=True)
from lv_slam_amd import ndt
rng = np.random.default_rng(0)
tgt = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
src = rng.uniform(-4, 4, (int(sys.argv[1]) if len(sys.argv) > 1 else 100, 3)).astype(np.float32)
e = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64))
print("engine ok", flush=True)
e.set_target(tgt); print("target ok", e.get_grid(), flush=True)
e.set_source(src); print("source ok", flush=True)
print(e.derivatives(np.zeros(6))[0], flush=True)
print(e.align(np.eye(4, dtype=np.float32))["iterations"], flush=True)
