"""Config 5, AoS RNEA (LDS windows): time against the LDS budget of the depth stack (MH_DFS_BUDGET, slots per lane) at batch sizes between
the occupancy steps -- why was 98 304 slower than 131 072?"""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    from mecano_amd.multibody import MultiBodySystem
    tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
    hm = HipModel(tree.toModelDesc())
    st0 = rt.nextState(np.random.default_rng(1), tree, 8192)
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for B in (49152, 65536, 98304, 131072):
        q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=torch.float32).repeat((B + 8191) // 8192, 1)[:B].contiguous() for x in st0)
        for _ in range(3): hm.rnea(q, qd, qdd, (0, 0, -9.81))
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t = HipTimer(); t.start(stream); hm.rnea(q, qd, qdd, (0, 0, -9.81)); t.stop(stream); ts.append(t.elapsed_ms())
        out[B] = sorted(ts)[3]
    print(json.dumps(out))
    sys.exit(0)
for budget in (None, 8, 16, 24, 32, 48, 64):
    env = dict(os.environ)
    if budget is not None:
        env["MH_DFS_BUDGET"] = str(budget)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print("MH_DFS_BUDGET", budget, line[-1] if line else r.stderr[-300:], flush=True)
