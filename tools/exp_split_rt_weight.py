"""Measurement helper (round 2): planner trunk weights of the run-time tree split (MH_SPLIT_RT_TRUNK_WEIGHT), B = 4096, models without a code object."""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream); return t.elapsed_ms() / iters * 1e3
systems = {"humanoid": rt.nextHumanoid(np.random.default_rng(43)), "tree30": rt.referenceBenchmarkSystems()["tree30"], "floating_tree30": rt.referenceBenchmarkSystems()["floating_tree30"],
           "tree128": MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())}
for w in (1, 2, 3, 4, 6):
    os.environ["MH_SPLIT_RT_TRUNK_WEIGHT"] = str(w)
    for name, s in systems.items():
        hm = HipModel(s.toModelDesc())
        B = 4096
        q, qd, qdd, tau = (torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64) for x in rt.nextState(np.random.default_rng(2342), s, B))
        print(f"weight {w}/2 {name:16s} RNEA {timeit(lambda: hm.rnea(q, qd, qdd, g)):7.1f} us  ABA {timeit(lambda: hm.aba(q, qd, tau, g)):7.1f} us  {hm.kernel_variant[40:]}", flush=True)
