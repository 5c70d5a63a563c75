import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); d = sys_.toModelDesc()
hm, om = HipModel(d), OracleModel(d)
print("variant", hm.kernel_variant)
for B in (64, 100, 4096):
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, B)
    g = (0, 0, -9.81)
    dv = lambda x: torch.tensor(x, device="cuda")
    t = hm.rnea(dv(q), dv(qd), dv(qdd), g).cpu().numpy(); tr = om.rnea(q, qd, qdd, g)
    a = hm.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy(); ar = om.aba(q, qd, tau, g)
    et = np.abs(t - tr).max(axis=1); ea = np.abs(a - ar).max(axis=1)
    print(B, "rnea err", et.max(), "bad rows", np.nonzero(et > 1e-8)[0][:10], "aba err", ea.max(), "bad rows", np.nonzero(ea > 1e-8)[0][:10], "bad cols", np.nonzero(np.abs(a-ar).max(axis=0) > 1e-8)[0])
