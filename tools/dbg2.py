import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); d = sys_.toModelDesc()
os.environ["MH_SPEC_IO"] = "0"
os.environ["MH_SPEC_ST"] = "1"; h0 = HipModel(d)
os.environ["MH_SPEC_ST"] = "0"; h1 = HipModel(d)
B = 2
q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, B)
dv = lambda x: torch.tensor(x, device="cuda")
g = (0, 0, -9.81)
np.set_printoptions(precision=4, linewidth=200)
for name, args in (("tau=0,qd=0", (q, 0*qd, 0*tau)), ("qd=0", (q, 0*qd, tau)), ("tau=0", (q, qd, 0*tau)), ("full", (q, qd, tau))):
    a0 = h0.aba(*[dv(x) for x in args], g).cpu().numpy(); a1 = h1.aba(*[dv(x) for x in args], g).cpu().numpy()
    print(name, "maxdiff", np.abs(a0-a1).max()); print((a0-a1)[0])
