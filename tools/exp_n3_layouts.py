import sys, numpy as np, torch
sys.path.insert(0, ".")
from mecano_amd import random_tools as rt, _lib
from mecano_amd.engine import HipModel, HipTimer
s = rt.nextHumanoid(np.random.default_rng(43)); hm = HipModel(s.toModelDesc())
st = torch.cuda.current_stream().cuda_stream
for B in (4096, 32768, 262144):
    q, qd, _, _ = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), s, B))
    qs, qds = q.t().contiguous(), qd.t().contiguous()
    for name, fn in (("coriolis AoS", lambda: hm.crba_coriolis(q, qd)), ("coriolis SoA", lambda: hm.crba_coriolis(qs, qds, _lib.LAYOUT_SOA)),
                     ("centroidal AoS", lambda: hm.centroidal(q, qd, None, True)), ("centroidal SoA", lambda: hm.centroidal(qs, qds, None, True, _lib.LAYOUT_SOA))):
        for _ in range(3): fn()
        t = HipTimer(); t.start(st)
        for _ in range(5): fn()
        t.stop(st); print(B, name, "%.1f us" % (t.elapsed_ms() / 5 * 1e3), flush=True)
