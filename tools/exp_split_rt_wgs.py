"""Measurement helper (round 2): the run-time split kernels at device-filling batches with 1..4 workgroups per CU (MH_SPLIT_RT_WGS) against the one-wave kernels (FORCE=0)."""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream); return t.elapsed_ms() / iters * 1e3
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
hum = rt.nextHumanoid(np.random.default_rng(43))
for name, s, dt, B in (("tree128 f32", tree, torch.float32, 131072), ("humanoid f64", hum, torch.float64, 262144), ("humanoid f64", hum, torch.float64, 32768)):
    os.environ["MH_SPLIT_RT"] = os.environ.get("FORCE", "1")
    hm = HipModel(s.toModelDesc())
    q, qd, qdd, tau = (torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt) for x in rt.nextState(np.random.default_rng(2342), s, 8192))
    q, qd, qdd, tau = (x.repeat(B // 8192, 1).t().contiguous() for x in (q, qd, qdd, tau))
    from mecano_amd import _lib
    L = _lib.LAYOUT_SOA
    print(f"WGS={os.environ.get('MH_SPLIT_RT_WGS','1')} force={os.environ['MH_SPLIT_RT']} {name} B={B}: RNEA {timeit(lambda: hm.rnea(q, qd, qdd, g, layout=L)):8.1f} us  ABA {timeit(lambda: hm.aba(q, qd, tau, g, layout=L)):8.1f} us", flush=True)
