"""A/B helper: AoS RNEA / ABA time at B = 4096 and 262144 (run with and without MH_SPEC_DIR=exp_build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
s = rt.nextHumanoid(np.random.default_rng(43)); hm = HipModel(s.toModelDesc())
st = torch.cuda.current_stream().cuda_stream
out = []
for B in (4096, 262144):
    q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), s, B))
    for name, fn in (("rnea", lambda: hm.rnea(q, qd, qdd)), ("aba", lambda: hm.aba(q, qd, tau))):
        for _ in range(10): fn()
        t = HipTimer(); t.start(st)
        for _ in range(40): fn()
        t.stop(st); out.append("%s@%d %.1f us" % (name, B, t.elapsed_ms() / 40 * 1e3))
print(os.environ.get("MH_SPEC_DIR", "shipped"), " | ".join(out))
