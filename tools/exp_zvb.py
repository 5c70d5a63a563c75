"""Experiment: forward dynamics of device-filling batches as two launches (mh_zv_kernels.h, spec_zvb_*) against the oracle and against the
one-job tree-split kernel.  MH_SPEC_DIR=exp_build python tools/exp_zvb.py [B ...]   (child processes toggle MH_ZVB / MH_ZVB_WHICH)"""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    from oracle.cpu_oracle import OracleModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    desc = sys_.toModelDesc()
    hm = HipModel(desc)
    om = OracleModel(desc)
    tag = f"MH_ZVB={os.environ.get('MH_ZVB')} WHICH={os.environ.get('MH_ZVB_WHICH', '3')}"
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.3, -0.2, -9.81)
    for B in [int(a) for a in sys.argv[2:]] or [32768]:
        q, qd, qdd, tau = rt.nextState(np.random.default_rng(B), sys_, min(B, 32768 + 77))
        rep = (B + q.shape[0] - 1) // q.shape[0]
        dq, dqd, dtau = (torch.tensor(x, device="cuda").repeat(rep, 1)[:B].contiguous() for x in (q, qd, tau))
        n = min(B, 200)
        a = hm.aba(dq, dqd, dtau, g)
        torch.cuda.synchronize()
        ah = a.cpu().numpy()
        head = np.abs(ah[:n] - om.aba(q[:n], qd[:n], tau[:n], g)).max()
        tq, tqd, ttau = (x[-n:].cpu().numpy() for x in (dq, dqd, dtau))
        tail = np.abs(ah[-n:] - om.aba(tq, tqd, ttau, g)).max()
        nan = int(np.isnan(ah).sum())
        fn = lambda: hm.aba(dq, dqd, dtau, g)
        for _ in range(5): fn()
        best = 1e9
        for r in range(3):
            t = HipTimer(); t.start(stream)
            for _ in range(20): fn()
            t.stop(stream)
            best = min(best, t.elapsed_ms() / 20 * 1e3)
        print(f"{tag} B={B}: err head {head:.2e} tail {tail:.2e} nan {nan};  {best:.1f} us  {B / best / 1e3:.3f} G/s", flush=True)
else:
    for env in ({"MH_ZVB": "0"}, {"MH_ZVB": "2"}, {"MH_ZVB": "2", "MH_ZVB_WHICH": "1"}, {"MH_ZVB": "2", "MH_ZVB_WHICH": "2"}):
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, **env))
