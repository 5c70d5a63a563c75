"""Measurement helper: fp64 RNEA / ABA on the run-time-topology kernels (no code object) for the humanoid and the reference's 30-joint
benchmark shapes at three batch sizes.  python tools/exp_generic_rnea.py"""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    t = HipTimer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    return t.elapsed_ms() / iters * 1e3


systems = {"humanoid": rt.nextHumanoid(np.random.default_rng(43))}
systems.update(rt.referenceBenchmarkSystems())
for name, s in systems.items():
    hm = HipModel(s.toModelDesc())
    for B in (4096, 32768, 262144):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), s, B))
        print(f"{name:18s} B={B:7d}  RNEA {timeit(lambda: hm.rnea(q, qd, qdd, g)):8.1f} us   ABA {timeit(lambda: hm.aba(q, qd, tau, g)):8.1f} us", flush=True)
