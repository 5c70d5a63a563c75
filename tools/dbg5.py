import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.multibody import MultiBodySystem
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel
rng = np.random.default_rng(0)
sys_ = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7)[0].getPredecessor()); d = sys_.toModelDesc()
om = OracleModel(d)
B = 128
q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, B)
dv = lambda x: torch.tensor(x, device="cuda")
g = (0, 0, -9.81)
ar = om.aba(q, qd, tau, g)
for io in "01":
    for st in "01":
        os.environ["MH_SPEC_IO"] = io; os.environ["MH_SPEC_ST"] = st
        h = HipModel(d)
        a = h.aba(dv(q), dv(qd), dv(tau), g).cpu().numpy()
        print("arm7", h.kernel_variant, "IO", io, "ST", st, "aba err", np.abs(a - ar).max())
