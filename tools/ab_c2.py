"""C2 (7-DoF arm, RNEA fp64) at B = 1024 and 262144 from the code object in MH_SPEC_DIR: device time per call (graph replay of 20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bench_configs import timeit
from mecano_amd import build as b
from mecano_amd.engine import HipModel
from mecano_amd import random_tools as rt
from mecano_amd.multibody import MultiBodySystem
from oracle.cpu_oracle import OracleModel
rng = np.random.default_rng(0)
rt.nextHumanoid(rng)
sys_ = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7)[0].getPredecessor())
d = sys_.toModelDesc()
hm = HipModel(d)
stream = torch.cuda.current_stream().cuda_stream
for B in (1024, 262144):
    q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2), sys_, B))
    t = timeit(lambda: hm.rnea(q, qd, qdd, (0, 0, -9.81)), stream)
    ta = timeit(lambda: hm.aba(q, qd, tau, (0, 0, -9.81)), stream)
    err = np.abs(hm.rnea(q[:64], qd[:64], qdd[:64], (0, 0, -9.81)).cpu().numpy() - OracleModel(d).rnea(q[:64].cpu().numpy(), qd[:64].cpu().numpy(), qdd[:64].cpu().numpy(), (0, 0, -9.81))).max()
    print(os.environ.get("MH_SPEC_DIR", "shipped")[-12:], hm.kernel_variant[:22], f"B={B}: RNEA {t * 1e6:.2f} us, ABA {ta * 1e6:.2f} us, RNEA err {err:.1e}", flush=True)
