"""Config 5's pair call (mh_rnea_aba_f32, AoS) and its two halves at batch sizes around the per-GPU shard: where are the cliffs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
hm = HipModel(tree.toModelDesc())
base = 8192
f32 = torch.float32
st0 = rt.nextState(np.random.default_rng(1), tree, base)
g = (0.0, 0.0, -9.81)
stream = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=5):
    for _ in range(2): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream)
    return t.elapsed_ms() / iters
for B in [int(a) for a in sys.argv[1:]]:
    reps = (B + base - 1) // base
    q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=f32).repeat(reps, 1)[:B].contiguous() for x in st0)
    o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
    pair = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
    T = lambda x: x.t().contiguous()
    qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
    tp = timeit(pair)
    tr, ta = timeit(lambda: hm.rnea(q, qd, qdd, g)), timeit(lambda: hm.aba(q, qd, tau, g))
    trs, tas = timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=_lib.LAYOUT_SOA)), timeit(lambda: hm.aba(qs, qds, taus, g, layout=_lib.LAYOUT_SOA))
    print(f"B={B:8d}: pair {tp:7.3f} ms = {B / tp / 1e3:6.1f} M/s | AoS rnea {tr:6.3f} aba {ta:6.3f} | SoA rnea {trs:6.3f} ({B / trs / 1e3:5.0f} M/s) aba {tas:6.3f} ({B / tas / 1e3:5.0f} M/s)", flush=True)
