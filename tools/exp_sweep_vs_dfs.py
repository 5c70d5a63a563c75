"""Measurement helper: one-wave run-time-topology kernels, sweep (MH_DFS=0) against depth-first (default), run-time split off; fp64 and fp32."""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
os.environ["MH_SPLIT_RT"] = "0"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream); return t.elapsed_ms() / iters * 1e3
systems = {"humanoid": rt.nextHumanoid(np.random.default_rng(43)), "chain30": rt.referenceBenchmarkSystems()["chain30"], "tree30": rt.referenceBenchmarkSystems()["tree30"],
           "tree128": MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())}
for name, s in systems.items():
    d = s.toModelDesc()
    os.environ["MH_DFS"] = "0"; sw = HipModel(d)
    os.environ.pop("MH_DFS"); df = HipModel(d)
    os.environ["MH_DFS_ABA64"] = "1"; dfa = HipModel(d); os.environ.pop("MH_DFS_ABA64")
    for dt, dn in ((torch.float64, "f64"), (torch.float32, "f32")):
        for B in (4096, 32768, 131072 if name == "tree128" else 262144):
            q, qd, qdd, tau = (torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt) for x in rt.nextState(np.random.default_rng(2342), s, min(B, 32768)))
            if B > 32768:
                q, qd, qdd, tau = (x.repeat(B // 32768, 1) for x in (q, qd, qdd, tau))
            print(f"{name:9s} {dn} B={B:6d}  RNEA sweep {timeit(lambda: sw.rnea(q, qd, qdd, g)):8.1f} / dfs {timeit(lambda: df.rnea(q, qd, qdd, g)):8.1f} us   "
                  f"ABA sweep {timeit(lambda: sw.aba(q, qd, tau, g)):8.1f} / dfs {timeit(lambda: dfa.aba(q, qd, tau, g)):8.1f} us", flush=True)
