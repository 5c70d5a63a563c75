"""Are the device entry points capturable in a HIP graph?  python tools/exp_graph.py  (GPU box)
Captures mh_rnea_aba_f64 (one fused launch with the code object; side by side on two streams without it), mh_rnea_crba_f64 and one
simulation step into torch.cuda.CUDAGraph objects on a side stream, replays them and compares with the eager results; times 200 replays
of a 10-step graph against 2000 eager steps."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel

def run(tag):
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(rt.humanoid30Desc())
    B = 4096
    hm.reserve(B)
    q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), sys_, B))
    g = (0.0, 0.0, -9.81)
    t_ref, a_ref = hm.rnea_aba(q, qd, qdd, tau, g)
    t2_ref, H_ref = hm.rnea_crba(q, qd, qdd, g)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    t_out, a_out = torch.empty_like(qd), torch.empty_like(qd)
    t2_out, H_out = torch.empty_like(qd), torch.empty((B, hm.nv, hm.nv), dtype=torch.float64, device="cuda")
    with torch.cuda.stream(s):
        pair = hm.bind_rnea_aba(q, qd, qdd, tau, t_out, a_out, g)
        rc = hm.bind_rnea_crba(q, qd, qdd, t2_out, H_out, g)
        pair(); rc()
        torch.cuda.synchronize()
        t_out.zero_(); a_out.zero_(); t2_out.zero_(); H_out.zero_()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            pair(); rc()
        g10 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g10, stream=s):
            for _ in range(10):
                pair()
    g1.replay()
    torch.cuda.synchronize()
    ok = torch.equal(t_out, t_ref) and torch.equal(a_out, a_ref) and torch.equal(t2_out, t2_ref) and torch.equal(H_out, H_ref)
    print(f"[{tag}] {hm.kernel_variant[:60]}: graph replay equals eager: {ok}", flush=True)
    for _ in range(20):
        g10.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g10.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 2000
    with torch.cuda.stream(s):
        for _ in range(50):
            pair()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            pair()
        torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 2000
    print(f"[{tag}] mh_rnea_aba_f64, B = {B}: eager {te*1e6:.2f} us / step, 10-step graph {tg*1e6:.2f} us / step", flush=True)
    return ok

if __name__ == "__main__":
    ok = run("env MH_DISABLE_SPEC=" + os.environ.get("MH_DISABLE_SPEC", "0"))
    sys.exit(0 if ok else 1)
