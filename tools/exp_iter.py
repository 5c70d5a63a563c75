"""Cold vs warm iterations: with MH_FAKE_CU_COUNT=1 a grid is 1-2 workgroups that loop over the batch; time per loop iteration for
1 iteration (cold caches) against 32 (warm I$ / K$)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import build as b, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

def timeit(fn, n=50):
    t = HipTimer()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t.start(); fn(); t.stop(); ts.append(t.elapsed_ms())
    return float(np.median(ts)) * 1e3

os.environ["MH_FAKE_CU_COUNT"] = "1"
for name, desc in b.registered_models().items():
    hm = HipModel(desc)
    rng = np.random.default_rng(1)
    for algo in ("rnea", "aba"):
        for split in ("0", "1"):
            os.environ["MH_SPEC_SPLIT"] = split
            hm = HipModel(desc)
            for iters in (1, 2, 4, 32):
                groups = 2 if split == "1" else 1
                B = 64 * groups * iters
                q = torch.randn(B, desc.nq, device="cuda", dtype=torch.float64)
                if desc.joint_type[0] == 2: q[:, :4] /= q[:, :4].norm(dim=1, keepdim=True)
                qd = torch.randn(B, desc.nv, device="cuda", dtype=torch.float64); x = torch.randn_like(qd)
                fn = (lambda: hm.rnea(q, qd, x)) if algo == "rnea" else (lambda: hm.aba(q, qd, x))
                us = timeit(fn)
                print(f"{name:10s} {algo} split={split} variant={hm.kernel_variant} iters={iters:3d} B={B:5d}  {us:8.1f} us  -> {us/iters:7.2f} us/iter", flush=True)
