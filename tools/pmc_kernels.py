"""PMC workload: the humanoid's RNEA, ABA, CRBA, fused and integrate calls, 10 launches each at B = argv[1] (default 4096)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), sys_, B))
g = (0.0, 0.0, -9.81)
for _ in range(10):
    hm.rnea(q, qd, qdd, g)
for _ in range(10):
    hm.aba(q, qd, tau, g)
for _ in range(10):
    hm.rnea_aba(q, qd, qdd, tau, g)
for _ in range(10):
    hm.crba(q)
torch.cuda.synchronize()
