"""Experiment: the bias-split forward dynamics (mh_zv_kernels.h) against the oracle and against the tree-split kernels it replaces.
MH_SPEC_DIR=exp_build python tools/exp_zv.py [B ...]   (child processes toggle MH_ZV)"""
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
    print("MH_ZV =", os.environ.get("MH_ZV"), "variant:", hm.kernel_variant, flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.3, -0.2, -9.81)
    for B in [int(a) for a in sys.argv[2:]] or [4096]:
        q, qd, qdd, tau = rt.nextState(np.random.default_rng(B), sys_, B)
        dq, dqd, dqdd, dtau = (torch.tensor(x, device="cuda") for x in (q, qd, qdd, tau))
        n = min(B, 256)
        ref_a, ref_r = om.aba(q[:n], qd[:n], tau[:n], g), om.rnea(q[:n], qd[:n], qdd[:n], g)
        a = hm.aba(dq, dqd, dtau, g)
        t2, a2 = hm.rnea_aba(dq, dqd, dqdd, dtau, g)
        torch.cuda.synchronize()
        ea = np.abs(a.cpu().numpy()[:n] - ref_a).max(); ea2 = np.abs(a2.cpu().numpy()[:n] - ref_a).max(); er = np.abs(t2.cpu().numpy()[:n] - ref_r).max()
        tail_a = np.abs(a.cpu().numpy()[-n:] - om.aba(q[-n:], qd[-n:], tau[-n:], g)).max()
        same = bool((a == a2).all())
        times = []
        o1, o2 = torch.empty_like(dqd), torch.empty_like(dqd)
        bound = hm.bind_rnea_aba(dq, dqd, dqdd, dtau, o1, o2, g)
        for fn in (lambda: hm.aba(dq, dqd, dtau, g), bound):
            for _ in range(5): fn()
            best = 1e9
            for rep in range(3):
                t = HipTimer(); t.start(stream)
                for _ in range(50): fn()
                t.stop(stream)
                best = min(best, t.elapsed_ms() / 50 * 1e3)
            times.append(best)
        print(f"B={B}: aba err {ea:.2e} (tail {tail_a:.2e}), pair: qdd err {ea2:.2e} tau err {er:.2e}, aba == pair's: {same};  aba (python path) {times[0]:.2f} us, pair (bound call) {times[1]:.2f} us", flush=True)
else:
    for zv in ("1", "0"):
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, MH_ZV=zv, MH_SPEC_SELFCHECK_VERBOSE="1"))
