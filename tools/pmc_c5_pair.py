"""PMC workload for config 5: mh_rnea_aba_f32 on the random 128-body tree at B = argv[1] (default 131072, the per-GPU shard), AoS then SoA,
five calls each (MH_DFS_PAIR in the environment picks the fused walk or the two launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
hm = HipModel(tree.toModelDesc())
base = 8192
st = rt.nextState(np.random.default_rng(1), tree, base)
q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=torch.float32).repeat((B + base - 1) // base, 1)[:B].contiguous() for x in st)
g = (0.0, 0.0, -9.81)
which = sys.argv[2] if len(sys.argv) > 2 else "aos"
if which == "soa":
    q, qd, qdd, tau = (x.t().contiguous() for x in (q, qd, qdd, tau))
for _ in range(5):
    hm.rnea_aba(q, qd, qdd, tau, g, layout=_lib.LAYOUT_SOA if which == "soa" else _lib.LAYOUT_AOS)
torch.cuda.synchronize()
print("nq", hm.nq, "nv", hm.nv, "B", B, which)
