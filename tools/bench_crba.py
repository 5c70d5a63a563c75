"""CRBA of the 30-DoF humanoid: whole-wave packed kernel (MH_SPEC_SPLIT=0) against the tree-split kernel (MH_SPEC_SPLIT=1), per batch size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import build as b, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from oracle.cpu_oracle import OracleModel

sys_ = rt.nextHumanoid(np.random.default_rng(43)); desc = sys_.toModelDesc()
om = OracleModel(desc)
t = HipTimer()
for B in (64, 1024, 4096, 16384, 65536, 262144):
    q = rt.nextState(np.random.default_rng(1), sys_, B)[0]
    tq = torch.tensor(q, device="cuda")
    ref = om.crba(q[:256])
    line = f"B={B:7d}"
    for split in (sys.argv[1:] or ("0", "1")):
        os.environ["MH_SPEC_SPLIT"] = split
        hm = HipModel(desc)
        H = hm.crba(tq)
        err = float(np.abs(H[:256].cpu().numpy() - ref).max())
        for _ in range(5): hm.crba(tq)
        torch.cuda.synchronize()
        n = 30
        t.start()
        for _ in range(n): hm.crba(tq)
        t.stop()
        us = t.elapsed_ms() * 1e3 / n
        line += f" | split={split}: {us:8.1f} us {B/us:7.1f} M/s {B*7448/us/1e3:7.1f} GB/s err {err:.1e}"
    print(line, flush=True)
