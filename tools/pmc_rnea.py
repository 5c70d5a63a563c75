import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
sys_ = rt.nextHumanoid(np.random.default_rng(43)); hm = HipModel(sys_.toModelDesc())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
g = (0, 0, -9.81)
for _ in range(5):
    hm.rnea(q, qd, qdd, g); hm.aba(q, qd, tau, g)
torch.cuda.synchronize()
