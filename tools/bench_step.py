"""Device-resident simulation step on the 30-DoF humanoid: mh_aba_integrate_f64 (one launch) against mh_aba_f64 + mh_integrate_f64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

sys_ = rt.nextHumanoid(np.random.default_rng(43)); desc = sys_.toModelDesc()
hm = HipModel(desc); t = HipTimer(); g = (0.0, 0.0, -9.81)
for B in (4096, 32768, 262144):
    q, qd, _, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
    tau = tau * 0
    import ctypes
    from mecano_amd import _lib
    lib = _lib.load()
    gv = (ctypes.c_double * 3)(*g)
    opts = hm._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream().cuda_stream)
    acc = torch.empty_like(qd)
    a_step = (hm._h, B, ctypes.c_double(1e-4), q.data_ptr(), qd.data_ptr(), tau.data_ptr(), gv, None, ctypes.byref(opts), acc.data_ptr(), q.data_ptr(), qd.data_ptr())
    a_aba = (hm._h, B, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), gv, None, ctypes.byref(opts), acc.data_ptr())
    a_int = (hm._h, B, ctypes.c_double(1e-4), q.data_ptr(), qd.data_ptr(), acc.data_ptr(), ctypes.byref(opts), q.data_ptr(), qd.data_ptr(), None)
    def fused():
        lib.mh_aba_integrate_f64(*a_step)
    def two():
        lib.mh_aba_f64(*a_aba)
        lib.mh_integrate_f64(*a_int)
    line = f"B={B:7d}"
    for name, fn in (("one launch", fused), ("aba + integrate", two)):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        n = 200
        t.start()
        for _ in range(n): fn()
        t.stop()
        us = t.elapsed_ms() * 1e3 / n
        line += f" | {name}: {us:7.1f} us/step {B/us:7.1f} M configs/s"
    print(line, flush=True)
