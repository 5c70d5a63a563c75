"""Thin CRBA workgroups after the image shrank with them: python tools/exp_crba_lpg.py  (GPU box; MH_CRBA_LPG / MH_RNEA_CRBA_LPG per run)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(rt.humanoid30Desc())
stream = torch.cuda.current_stream().cuda_stream
for B in (2048, 4096, 8192, 16384, 32768):
    q, qd, qdd, _ = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), sys_, B))
    for name, fn in (("crba", lambda: hm.crba(q)), ("rnea_crba", lambda: hm.rnea_crba(q, qd, qdd))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t = HipTimer(); t.start(stream)
        for _ in range(100):
            fn()
        t.stop(stream); torch.cuda.synchronize()
        print(f"{name:10s} B={B:6d} {t.elapsed_ms()*10:8.2f} us  [CRBA_LPG={os.environ.get('MH_CRBA_LPG','-')} RNEA_CRBA_LPG={os.environ.get('MH_RNEA_CRBA_LPG','-')}]", flush=True)
