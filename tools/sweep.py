"""Exploration helper (not part of the product): per-kernel time over batch sizes and layouts on one GPU."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt, _lib
from mecano_amd.engine import HipModel, HipTimer

def timeit(fn, stream, iters=20, warm=3):
    for _ in range(warm): fn()
    t = HipTimer(); t.start(stream)
    for _ in range(iters): fn()
    t.stop(stream)
    return t.elapsed_ms() / iters

def main():
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    stream = torch.cuda.current_stream().cuda_stream
    g = (0, 0, -9.81)
    for B in [int(x) for x in (sys.argv[1:] or [4096, 32768, 262144])]:
        q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
        T = lambda x: x.t().contiguous()
        qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
        r = {}
        r["rnea_aos"] = timeit(lambda: hm.rnea(q, qd, qdd, g), stream)
        r["rnea_soa"] = timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=_lib.LAYOUT_SOA), stream)
        r["aba_aos"] = timeit(lambda: hm.aba(q, qd, tau, g), stream)
        r["aba_soa"] = timeit(lambda: hm.aba(qs, qds, taus, g, layout=_lib.LAYOUT_SOA), stream)
        r["crba_aos"] = timeit(lambda: hm.crba(q), stream, iters=5)
        print(B, {k: round(v * 1e3, 1) for k, v in r.items()}, "us | Mcfg/s:", {k: round(B / v / 1e3, 1) for k, v in r.items()}, flush=True)

main()
