"""Measurement helper: mh_rnea_aba_f64 on the run-time-topology kernels (MH_DISABLE_SPEC=1), the two launches side by side against one after the
other (MH_DISABLE_PAIR=1), humanoid and the reference's benchmark shapes.  python tools/exp_pair_generic.py"""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


def timeit(fn, iters=50, warm=10):
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
    d = s.toModelDesc()
    os.environ.pop("MH_DISABLE_PAIR", None)
    side = HipModel(d)
    os.environ["MH_DISABLE_PAIR"] = "1"
    seq = HipModel(d)
    for B in (1024, 4096, 16384, 32768, 65536):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), s, B))
        a, b = side.rnea_aba(q, qd, qdd, tau, g), seq.rnea_aba(q, qd, qdd, tau, g)
        same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
        print(f"{name:18s} B={B:6d}  side by side {timeit(lambda: side.rnea_aba(q, qd, qdd, tau, g)):8.1f} us   one after the other {timeit(lambda: seq.rnea_aba(q, qd, qdd, tau, g)):8.1f} us   identical results: {same}", flush=True)
