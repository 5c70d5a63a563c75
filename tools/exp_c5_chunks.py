"""Config 5 (128-body tree, fp32, AoS, RNEA + ABA of 1 M configurations): one mh_rnea_aba_f32 call against the same batch cut into chunks
that alternate between S streams, each stream with a context of its own -- do the transposed scratch copies of one chunk (bound by memory)
overlap with the walks of another (bound by issue and latency)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
hm = HipModel(tree.toModelDesc())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
base = 8192
f32 = torch.float32
q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=f32).repeat(B // base, 1).contiguous() for x in rt.nextState(np.random.default_rng(1), tree, base))
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
g = (0.0, 0.0, -9.81)
main = torch.cuda.current_stream()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
one = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
t = timed(one)
ref1, ref2 = o1.clone(), o2.clone()
print(f"one call: {t * 1e3:.2f} ms  {B / t / 1e6:.1f} M configs/s", flush=True)
for S in (1, 2, 3):
    for C in (4, 8, 16, 32):
        streams = [torch.cuda.Stream() for _ in range(S)]
        views = [hm.context() for _ in range(S)]
        Bc = B // C
        calls = []
        for k in range(C):
            sl = slice(k * Bc, (k + 1) * Bc)
            with torch.cuda.stream(streams[k % S]):
                calls.append((streams[k % S], views[k % S].bind_rnea_aba(q[sl], qd[sl], qdd[sl], tau[sl], o1[sl], o2[sl], g)))
        def run():
            for st in streams:
                st.wait_stream(main)
            for st, fn in calls:
                with torch.cuda.stream(st):
                    fn()
            for st in streams:
                main.wait_stream(st)
        o1.zero_(), o2.zero_()
        t = timed(run)
        ok = torch.equal(o1, ref1) and torch.equal(o2, ref2)
        print(f"{C:3d} chunks on {S} stream(s): {t * 1e3:.2f} ms  {B / t / 1e6:.1f} M configs/s  same results: {ok}", flush=True)
        for v in views:
            v.close()
