"""What a timed region of bench.py costs besides its K steps (headline workload, B = 4096): regions of K = 5, 20, 80, 200 steps with and
without the HIP event pair around them; a + b K fitted through the medians."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
B = 4096
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(7), sys_, B))
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
fn = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, (0.0, 0.0, -9.81))
stream = torch.cuda.current_stream().cuda_stream
for _ in range(50): fn()
def region(K, events):
    t = HipTimer()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    if events: t.start(stream)
    t0 = time.perf_counter()
    for _ in range(K): fn()
    if events: t.stop(stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
for events in (True, False):
    med = {}
    for K in (5, 20, 80, 200):
        med[K] = float(np.median([region(K, events) for _ in range(41)]))
    b = (med[200] - med[20]) / 180
    a = med[20] - 20 * b
    print(f"events={events}: " + "  ".join(f"K={K}: {v:.1f} us ({v / K:.2f}/step)" for K, v in med.items()) + f"   fit: {a:.1f} + {b:.2f} K", flush=True)
# host-only cost of K launches (no wait)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per launch {((t1 - t0) / 200) * 1e6:.2f} us")
# an idle device: synchronize alone
ts = []
for _ in range(50):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(f"synchronize on an idle device {np.median(ts):.2f} us")
# one launch + synchronize
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(f"one launch + synchronize {np.median(ts):.2f} us")
