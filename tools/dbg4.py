import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); d = sys_.toModelDesc()
hm, om = HipModel(d), OracleModel(d)
for B in (4096, 16384, 16448, 32768, 65536, 200000):
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(5), sys_, B)
    g = (0, 0, -9.81)
    dv = lambda x: torch.tensor(x, device="cuda")
    idx = np.arange(0, B, 61)
    t = hm.rnea(dv(q), dv(qd), dv(qdd), g).cpu().numpy()[idx]; tr = om.rnea(q[idx], qd[idx], qdd[idx], g)
    a = hm.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy()[idx]; ar = om.aba(q[idx], qd[idx], tau[idx], g)
    et = np.abs(t - tr).max(axis=1); ea = np.abs(a - ar).max(axis=1)
    print(B, "rnea err", et.max(), "bad", idx[np.nonzero(et > 1e-8)[0]][:8], "| aba err", ea.max(), "bad", idx[np.nonzero(ea > 1e-8)[0]][:8], (ea > 1e-8).sum(), "of", len(idx))
