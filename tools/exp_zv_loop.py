"""Steady-state loop of the bound mh_rnea_aba_f64 call (for rocprofv3 --kernel-trace --stats): python tools/exp_zv_loop.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
f = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, (0, 0, -9.81))
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20): f()
torch.cuda.synchronize()
for rep in range(3):
    t = HipTimer(); t.start(stream)
    for _ in range(K): f()
    t.stop(stream); torch.cuda.synchronize()
    print("B", B, "variant", hm.kernel_variant, "MH_ZV", os.environ.get("MH_ZV"), ": %.2f us per step" % (t.elapsed_ms() / K * 1e3), flush=True)
