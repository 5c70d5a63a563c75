"""Exploration helper: per-kernel time of the 7-joint arm (small code) vs the humanoid (large code)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.multibody import MultiBodySystem
from mecano_amd.engine import HipModel, HipTimer

def timeit(fn, stream, iters=30, warm=5):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream)
    return t.elapsed_ms() / iters * 1e3

rng = np.random.default_rng(0)
stream = torch.cuda.current_stream().cuda_stream
g = (0, 0, -9.81)
for name, sys_ in (("arm7", MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7)[0].getPredecessor())), ("humanoid", rt.nextHumanoid(rng))):
    hm = HipModel(sys_.toModelDesc())
    for B in (64, 4096):
        q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(rng, sys_, B))
        r = timeit(lambda: hm.rnea(q, qd, qdd, g), stream); a = timeit(lambda: hm.aba(q, qd, tau, g), stream)
        n = hm.n_joints
        print(f"{name} {hm.kernel_variant} B={B}: rnea {r:.1f} us ({r/n:.2f}/body)  aba {a:.1f} us ({a/n:.2f}/body)")
