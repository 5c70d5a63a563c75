"""Config 5 at B = argv[1]: RNEA / ABA / pair, SoA, for the MH_WAVES_PER_CU in the environment (median of 7 calls, ms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
hm = HipModel(tree.toModelDesc())
st0 = rt.nextState(np.random.default_rng(1), tree, 8192)
q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=torch.float32).repeat((B + 8191) // 8192, 1)[:B].t().contiguous() for x in st0)
stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
S = _lib.LAYOUT_SOA
def med(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t = HipTimer(); t.start(stream); fn(); t.stop(stream); ts.append(t.elapsed_ms())
    return sorted(ts)[3]
print("MH_WAVES_PER_CU", os.environ.get("MH_WAVES_PER_CU", "default"), "B", B, "SoA: rnea %.3f ms, aba %.3f ms, pair %.3f ms" % (
    med(lambda: hm.rnea(q, qd, qdd, g, layout=S)), med(lambda: hm.aba(q, qd, tau, g, layout=S)), med(lambda: hm.rnea_aba(q, qd, qdd, tau, g, layout=S))), flush=True)
