"""Measurement helper: CRBA on the run-time-topology kernel (no code object) against the specialised one."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
stream = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream); return t.elapsed_ms() / iters * 1e3
s = rt.nextHumanoid(np.random.default_rng(43)); d = s.toModelDesc()
spec = HipModel(d)
os.environ["MH_DISABLE_SPEC"] = "1"; gen = HipModel(d)
for B in (4096, 32768, 262144):
    q = torch.tensor(np.ascontiguousarray(rt.nextState(np.random.default_rng(1), s, min(B, 32768))[0]), device="cuda")
    if B > 32768: q = q.repeat(B // 32768, 1)
    from mecano_amd import _lib
    qs = q.t().contiguous()
    print(f"humanoid CRBA B={B:6d}: specialised {timeit(lambda: spec.crba(q)):8.1f} us   generic AoS {timeit(lambda: gen.crba(q)):8.1f} us   generic SoA {timeit(lambda: gen.crba(qs, layout=_lib.LAYOUT_SOA)):8.1f} us", flush=True)
