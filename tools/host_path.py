"""PCIe-inclusive rate of the host-pointer entry points (mh_rnea_f64_host + mh_aba_f64_host), for DESIGN.md."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); hm = HipModel(sys_.toModelDesc())
for B in (4096, 262144):
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(1), sys_, B)
    g = (0, 0, -9.81)
    for _ in range(3):
        hm.rnea(q, qd, qdd, g); hm.aba(q, qd, tau, g)
    n = 20 if B < 100000 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        hm.rnea(q, qd, qdd, g); hm.aba(q, qd, tau, g)
    dt = (time.perf_counter() - t0) / n
    print(f"host-pointer path B={B}: {dt*1e3:.3f} ms per RNEA+ABA step, {B/dt/1e6:.2f} M configs/s (PCIe copies and synchronisation included)")
