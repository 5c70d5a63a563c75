"""PMC workload for the run-time-topology kernels: MH_DISABLE_SPEC=1 humanoid at B = 4096 (argv[1] = hum) or the 128-body tree, fp32, SoA,
B = 131072 (argv[1] = c5); 5 RNEA then 5 ABA launches."""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem
which = sys.argv[1] if len(sys.argv) > 1 else "hum"
g = (0.0, 0.0, -9.81)
if which in ("hum", "reg"):
    sys_ = rt.nextHumanoid(np.random.default_rng(43)); B = 4096; dt = torch.float64; layout = _lib.LAYOUT_AOS
else:
    sys_ = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
    B = 131072; dt = torch.float32; layout = _lib.LAYOUT_AOS if which == "c5aos" else _lib.LAYOUT_SOA
hm = HipModel(sys_.toModelDesc())
st = rt.nextState(np.random.default_rng(2342), sys_, min(B, 8192))
q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=dt).repeat((B + len(x) - 1) // len(x), 1)[:B].contiguous() for x in st)
if layout == _lib.LAYOUT_SOA:
    q, qd, qdd, tau = (x.t().contiguous() for x in (q, qd, qdd, tau))
if which == "reg":  # joint torque regressor of the humanoid, B = 4096 (run as: tools/pmc_dfs.sh reg <tag>)
    for _ in range(5):
        hm.regressor(q, qd, qdd, g)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(5):
    hm.rnea(q, qd, qdd, g, layout=layout)
for _ in range(5):
    hm.aba(q, qd, tau, g, layout=layout)
torch.cuda.synchronize()
