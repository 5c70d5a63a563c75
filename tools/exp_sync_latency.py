"""What the closing synchronize of a short timed region costs: 20 back-to-back launches of the bound pair call, host clock from before
the first launch to after the wait.  python tools/exp_sync_latency.py [mode]   mode: sync (torch.cuda.synchronize), spin (hipEventQuery loop
through torch events), both; run also with HSA_ENABLE_INTERRUPT=0 in the environment."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
B, K = 4096, 20
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
f = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, (0, 0, -9.81))
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20): f()
torch.cuda.synchronize()
def region(mode):
    ev = torch.cuda.Event()
    t = HipTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.start(stream)
    for _ in range(K): f()
    t.stop(stream)
    if mode == "spin":
        ev.record()
        while not ev.query():
            pass
    if mode == "event":  # wait on the stop event the timer has recorded anyway, then the (now idle) device synchronize
        t.elapsed_ms()
    if mode == "nosync_after_event":
        t.elapsed_ms()
    else:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt * 1e6, t.elapsed_ms() * 1e3
for mode in ("sync", "event", "nosync_after_event", "sync"):
    rs = [region(mode) for _ in range(15)]
    print("HSA_ENABLE_INTERRUPT=%s %s: host %.1f us (min %.1f) for %d steps, events %.1f us -> %.2f / %.2f us per step" % (
        os.environ.get("HSA_ENABLE_INTERRUPT"), mode, np.median([r[0] for r in rs]), min(r[0] for r in rs), K, np.median([r[1] for r in rs]),
        np.median([r[0] for r in rs]) / K, np.median([r[1] for r in rs]) / K), flush=True)
