"""Rates of mh_regressor_f64 (joint torque regressor) on the benchmark humanoid: python tools/exp_regressor.py  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

sys_ = rt.nextHumanoid(np.random.default_rng(43))
d = rt.humanoid30Desc()
hm = HipModel(d)
stream = torch.cuda.current_stream().cuda_stream
for B in (4096, 32768, 131072):
    q, qd, qdd, _ = rt.nextState(np.random.default_rng(2342), sys_, min(B, 16384))
    reps = (B + len(q) - 1) // len(q)
    tq, tqd, tqdd = (torch.tensor(x, device="cuda").repeat(reps, 1)[:B].contiguous() for x in (q, qd, qdd))
    from mecano_amd import _lib
    sq, sqd, sqdd = (x.t().contiguous() for x in (tq, tqd, tqdd))
    for _ in range(3):
        Y = hm.regressor(sq, sqd, sqdd, layout=_lib.LAYOUT_SOA)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t = HipTimer(); t.start(stream); Y = hm.regressor(sq, sqd, sqdd, layout=_lib.LAYOUT_SOA); t.stop(stream); torch.cuda.synchronize(); ts.append(t.elapsed_ms())
    ms = float(np.median(ts))
    bytes_cfg = 8 * (d.nq + 3 * d.nv) + 8 * d.nv * 10 * d.n_joints
    print(f"regressor B={B:7d} SoA state and Y       {ms*1e3:9.1f} us  {B/ms/1e3:8.2f} M configs/s  {bytes_cfg*B/ms/1e6:8.1f} GB/s ({bytes_cfg*B/ms/1e6/8000*100:5.1f} % of 8 TB/s)  [incl. torch.empty + memset]", flush=True)
    del Y
    for fm in (False, True):
        for _ in range(3):
            Y = hm.regressor(tq, tqd, tqdd, first_moment_columns=fm)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t = HipTimer(); t.start(stream); Y = hm.regressor(tq, tqd, tqdd, first_moment_columns=fm); t.stop(stream); torch.cuda.synchronize(); ts.append(t.elapsed_ms())
        ms = float(np.median(ts))
        bytes_cfg = 8 * (d.nq + 3 * d.nv) + 8 * d.nv * 10 * d.n_joints
        print(f"regressor B={B:7d} first_moments={int(fm)}  {ms*1e3:9.1f} us  {B/ms/1e3:8.2f} M configs/s  {bytes_cfg*B/ms/1e6:8.1f} GB/s ({bytes_cfg*B/ms/1e6/8000*100:5.1f} % of 8 TB/s)  [incl. torch.empty + memset]", flush=True)
        del Y
