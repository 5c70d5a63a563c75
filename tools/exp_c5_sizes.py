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
SPREAD = {}
def timeit(fn, iters=9, tag=None):
    """median of per-call event times (round 4 averaged five calls behind two warm-up calls: one 40 ms hiccup -- a first-use allocation
    that the warm-up did not reach -- read as "8.3 ms per call" at B = 49 152)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t = HipTimer(); t.start(stream); fn(); t.stop(stream)
        ts.append(t.elapsed_ms())
    ts.sort()
    if tag: SPREAD[tag] = (ts[0], ts[-1])
    return ts[len(ts) // 2]
for B in [int(a) for a in sys.argv[1:]]:
    reps = (B + base - 1) // base
    q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=f32).repeat(reps, 1)[:B].contiguous() for x in st0)
    o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
    pair = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
    T = lambda x: x.t().contiguous()
    qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
    tp = timeit(pair)
    tr, ta = timeit(lambda: hm.rnea(q, qd, qdd, g), tag='rnea'), timeit(lambda: hm.aba(q, qd, tau, g), tag='aba')
    trs, tas = timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=_lib.LAYOUT_SOA)), timeit(lambda: hm.aba(qs, qds, taus, g, layout=_lib.LAYOUT_SOA))
    print(f"B={B:8d}: pair {tp:7.3f} ms = {B / tp / 1e3:6.1f} M/s | AoS rnea {tr:6.3f} aba {ta:6.3f} | SoA rnea {trs:6.3f} ({B / trs / 1e3:5.0f} M/s) aba {tas:6.3f} ({B / tas / 1e3:5.0f} M/s) | AoS rnea min..max {SPREAD['rnea'][0]:.3f}..{SPREAD['rnea'][1]:.3f} aba {SPREAD['aba'][0]:.3f}..{SPREAD['aba'][1]:.3f}", flush=True)
