import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); d = sys_.toModelDesc()
os.environ["MH_SPEC_ST"] = "0"
os.environ["MH_SPEC_IO"] = "0"; h0 = HipModel(d)
os.environ["MH_SPEC_IO"] = "1"; h1 = HipModel(d)
B = 64
q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, B)
dv = lambda x: torch.tensor(x, device="cuda")
g = (0, 0, -9.81)
a0 = h0.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy()
a1 = h1.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy()
a2 = h1.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy()
print("repeatable:", np.array_equal(a1, a2, equal_nan=True))
bad = np.abs(a0 - a1) > 1e-8
print("bad per lane:", bad.any(axis=1).astype(int))
print("bad per col :", bad.any(axis=0).astype(int))
