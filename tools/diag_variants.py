"""Diagnostic (GPU box): ABA / RNEA error of every forced memory plan of the specialised code object against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(77 + B)
sys_ = rt.nextHumanoid(rng)
d = sys_.toModelDesc()
om = OracleModel(d)
q, qd, qdd, tau = rt.nextState(rng, sys_, B)
g = (0.3, -0.2, -9.81)
t_ref, a_ref = om.rnea(q, qd, qdd, g), om.aba(q, qd, tau, g)
a_ref0 = om.aba(q, 0 * qd, tau, g)
dev = lambda x: torch.tensor(x, device="cuda")
off = {"MH_SPEC_SPLIT": "0"}
for env in ({"MH_DISABLE_SPEC": "1"}, {"MH_SPEC_IO": "0", "MH_SPEC_ST": "0", **off}, {"MH_SPEC_IO": "0", "MH_SPEC_ST": "1", **off},
            {"MH_SPEC_IO": "1", "MH_SPEC_ST": "0", **off}, {"MH_SPEC_IO": "1", "MH_SPEC_ST": "1", **off}, {"MH_SPEC_SPLIT": "1"}, {}):
    for k in ("MH_DISABLE_SPEC", "MH_SPEC_IO", "MH_SPEC_ST", "MH_SPEC_SPLIT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    hm = HipModel(d)
    e_r = np.abs(hm.rnea(dev(q), dev(qd), dev(qdd), g).cpu().numpy() - t_ref).max()
    e_a = np.abs(hm.aba(dev(q), dev(qd), dev(tau), g).cpu().numpy() - a_ref).max()
    e_a0 = np.abs(hm.aba(dev(q), dev(0 * qd), dev(tau), g).cpu().numpy() - a_ref0).max()
    print(env, "rnea %.2e aba %.2e aba(qd=0) %.2e" % (e_r, e_a, e_a0), flush=True)
