"""Generic (run-time-topology) mass matrix / Coriolis / centroidal kernels with the bodies' columns spread over several waves per group of
64 configurations: MH_DISABLE_SPEC=1 python tools/exp_column_parts.py  (GPU box; MH_REGRESSOR_PARTS=1 reproduces the old form;
MH_SPLIT_RT=0 keeps the CRBA off the run-time tree split)"""
import os, sys
os.environ.setdefault("MH_DISABLE_SPEC", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(rt.humanoid30Desc())
stream = torch.cuda.current_stream().cuda_stream
tag = f"[parts={os.environ.get('MH_REGRESSOR_PARTS', 'auto')} split_rt={os.environ.get('MH_SPLIT_RT', 'auto')}]"
for B in (1024, 4096, 16384, 65536):
    q, qd, _, _ = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), sys_, B))
    for name, fn in (("crba", lambda: hm.crba(q)), ("crba+coriolis", lambda: hm.crba_coriolis(q, qd)), ("centroidal A, b", lambda: hm.centroidal(q, qd))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = HipTimer(); t.start(stream)
        for _ in range(30):
            fn()
        t.stop(stream); torch.cuda.synchronize()
        print(f"{name:16s} B={B:6d} {t.elapsed_ms() / 30 * 1e3:9.1f} us  {tag}  {hm.kernel_variant[:40]}", flush=True)
