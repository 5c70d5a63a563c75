"""Forward dynamics and the pair call at device-filling batches (tree-split kernels): MH_SPEC_DIR=... python tools/exp_aba_big.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
stream = torch.cuda.current_stream().cuda_stream
g = (0, 0, -9.81)
out = []
for B in [int(a) for a in sys.argv[1:]] or [32768, 262144]:
    q, qd, qdd, tau = (torch.tensor(x, device="cuda").repeat((B + 16383) // 16384, 1)[:B].contiguous() for x in rt.nextState(np.random.default_rng(1), sys_, min(B, 16384)))
    res = {}
    for name, fn in (("rnea", lambda: hm.rnea(q, qd, qdd, g)), ("aba", lambda: hm.aba(q, qd, tau, g))):
        for _ in range(3): fn()
        best = 1e9
        for rep in range(3):
            t = HipTimer(); t.start(stream)
            for _ in range(20): fn()
            t.stop(stream)
            best = min(best, t.elapsed_ms() / 20 * 1e3)
        res[name] = best
    out.append(f"B={B}: rnea {res['rnea']:.1f} us, aba {res['aba']:.1f} us ({B / res['aba'] / 1e3:.3f} G/s)")
print(os.environ.get("MH_SPEC_DIR", "shipped"), hm.kernel_variant, " | ".join(out), flush=True)
