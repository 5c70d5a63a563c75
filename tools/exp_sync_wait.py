"""How long does a region of K = 20 headline steps take between two synchronisations under different host wait policies?
python tools/exp_sync_wait.py [spin|yield|blocking|auto]   (hipSetDeviceFlags before anything touches the device)"""
import ctypes, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
hip = ctypes.CDLL("libamdhip64.so")
FLAGS = {"auto": 0, "spin": 1, "yield": 2, "blocking": 4}
rc = hip.hipSetDeviceFlags(ctypes.c_uint(FLAGS[mode]))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
B = 4096
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
f = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, (0, 0, -9.81))
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    f()
torch.cuda.synchronize()
for K in (20, 200):
    ts = []
    for r in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    ts.sort()
    print(f"mode {mode} (hipSetDeviceFlags rc {rc}) ROC_ACTIVE_WAIT_TIMEOUT={os.environ.get('ROC_ACTIVE_WAIT_TIMEOUT')} K={K}: median {ts[12]:.3f} us per step, min {ts[0]:.3f}, max {ts[-1]:.3f}", flush=True)
