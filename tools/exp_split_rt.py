"""Measurement helper: the run-time tree split (mh_split_kernels.h, MH_SPLIT_RT=1) against the one-wave run-time-topology kernels
(MH_SPLIT_RT=0) over the batch size, fp64 and fp32; models without a code object.  python tools/exp_split_rt.py"""
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
    for _ in range(warm):
        fn()
    t = HipTimer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    return t.elapsed_ms() / iters * 1e3


systems = {"humanoid": rt.nextHumanoid(np.random.default_rng(43)), "tree30": rt.referenceBenchmarkSystems()["tree30"],
           "tree128": MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())}
for name, s in systems.items():
    d = s.toModelDesc()
    os.environ["MH_SPLIT_RT"] = "1"
    on = HipModel(d)
    os.environ["MH_SPLIT_RT"] = "0"
    off = HipModel(d)
    for dt, dn in ((torch.float64, "f64"), (torch.float32, "f32")):
        for B in (256, 4096, 8192, 16384, 24576, 32768, 65536):
            q, qd, qdd, tau = (torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt) for x in rt.nextState(np.random.default_rng(2342), s, B))
            r1, r0 = timeit(lambda: on.rnea(q, qd, qdd, g)), timeit(lambda: off.rnea(q, qd, qdd, g))
            a1, a0 = timeit(lambda: on.aba(q, qd, tau, g)), timeit(lambda: off.aba(q, qd, tau, g))
            print(f"{name:9s} {dn} B={B:6d}  RNEA split {r1:8.1f} us / one wave {r0:8.1f} us   ABA split {a1:8.1f} us / one wave {a0:8.1f} us", flush=True)
