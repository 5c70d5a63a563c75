import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from mecano_amd import _lib, build as b, random_tools as rt
from mecano_amd.engine import HipModel
desc = b.registered_models()["humanoid30"]
hum = rt.nextHumanoid(np.random.default_rng(0))
dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)
def model(**env):
    keys = ("MH_SPEC_SPLIT", "MH_SPEC_IO", "MH_SPEC_ST", "MH_DISABLE_SPEC", "MH_DISABLE_FUSED")
    for k in keys:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    m = HipModel(desc)
    for k in keys:
        os.environ.pop(k, None)
    return m
ref = model(MH_DISABLE_SPEC=1)
bad = model(MH_SPEC_SPLIT=0, MH_SPEC_IO=int(os.environ.get("DIAG_IO", "1")), MH_SPEC_ST=int(os.environ.get("DIAG_ST", "0")), MH_DISABLE_FUSED=1)
g = (0.0, 0.0, -9.81)
for B in (64, 4096):
    q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(B), hum, B))
    e = (bad.aba(q, qd, tau, g) - ref.aba(q, qd, tau, g)).abs().max().item()
    e0 = (bad.aba(q, 0 * qd, tau, g) - ref.aba(q, 0 * qd, tau, g)).abs().max().item()
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} [{bad.kernel_variant[:12]}] B={B}: max err {e:.3e} (qd=0: {e0:.3e}) {'WRONG' if not (max(e, e0) <= 1e-8) else ('exact-zero: generic fallback?' if e == 0 else 'OK')}", flush=True)
