"""PCIe-inclusive rate of the host-pointer entry points (what a Java shim calls), for DESIGN.md: pageable numpy arrays against pinned ones
(mh_host_alloc), two calls (mh_rnea_f64_host + mh_aba_f64_host) against the pair call (mh_rnea_aba_f64_host)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, pinned_empty
sys_ = rt.nextHumanoid(np.random.default_rng(43)); hm = HipModel(sys_.toModelDesc())
g = (0, 0, -9.81)
for B in (4096, 32768, 262144):
    state = rt.nextState(np.random.default_rng(1), sys_, B)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            arrs = []
            for x in state:
                a = pinned_empty(x.shape); a[...] = x; arrs.append(a)
            out = (pinned_empty(state[1].shape), pinned_empty(state[1].shape))
        else:
            arrs, out = list(state), (np.empty_like(state[1]), np.empty_like(state[1]))
        q, qd, qdd, tau = arrs
        n = 20 if B < 100000 else 5
        for label, fn in (("rnea_host + aba_host", lambda: (hm.rnea(q, qd, qdd, g), hm.aba(q, qd, tau, g))),
                          ("rnea_aba_host (pair)", lambda: hm.rnea_aba(q, qd, qdd, tau, g, out=out))):
            for _ in range(3):
                fn()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            dt = (time.perf_counter() - t0) / n
            print(f"host-pointer path B={B:7d} {kind:9s} {label:22s}: {dt*1e3:8.3f} ms per RNEA+ABA step, {B/dt/1e6:7.2f} M pairs/s", flush=True)
