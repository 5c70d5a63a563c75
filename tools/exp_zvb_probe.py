"""Phase stamps (100 MHz real-time counter) of the two-launch forward dynamics at a device-filling batch; code object built with -DMH_ZV_PROBE:
EXP_DIR=exp_probe python tools/isa.py --so -DMH_ZV_PROBE ; MH_SPEC_DIR=exp_probe MH_ZVB=2 python tools/exp_zvb_probe.py [B]"""
import ctypes, os, sys, glob
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
so = ctypes.CDLL(glob.glob(os.path.join(os.environ["MH_SPEC_DIR"], "libmecano_hip_topo_b5c1*.so"))[0])
q, qd, qdd, tau = rt.nextState(np.random.default_rng(1), sys_, min(B, 32768))
rep = (B + q.shape[0] - 1) // q.shape[0]
q, qd, tau = (torch.tensor(x, device="cuda").repeat(rep, 1)[:B].contiguous() for x in (q, qd, tau))
g = (0, 0, -9.81)
for _ in range(4):
    hm.aba(q, qd, tau, g)
    torch.cuda.synchronize()
b = np.zeros(4096 * 3 * 4 * 16, dtype=np.uint64)
assert so.mh_spec_zv_probe_read(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(b.nbytes)) == 0
G = min((B + 63) // 64, 4096)
st = b.reshape(4096, 3, 4, 16)[:G].astype(np.int64)
print("B", B, "variant", hm.kernel_variant, "groups", G, "(durations in us, median over groups [10 %, 90 %])")
def d(job, w, a, z):
    x = (st[:, job, w, z] - st[:, job, w, a]) / 100.0
    return "%5.2f [%5.2f %5.2f]" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90))
for w in range(4):
    print(f"bias    wave {w}: stage {d(0, w, 0, 1)}  limbs {d(0, w, 1, 2)}  wait {d(0, w, 2, 3)}  trunk {d(0, w, 3, 4)}  copy {d(0, w, 4, 5)}  bar {d(0, w, 5, 6)}  | group {d(0, w, 0, 6)}")
for w in range(4):
    print(f"inertia wave {w}: limbs_in {d(1, w, 0, 2)}  wait {d(1, w, 2, 3)}  root {d(1, w, 3, 4)}  wait {d(1, w, 4, 5)}  stage {d(1, w, 5, 6)}  early {d(1, w, 6, 7)}  late+sub {d(1, w, 7, 8)}  root {d(1, w, 8, 9)}  out {d(1, w, 9, 10)}  copy {d(1, w, 10, 11)}  bar {d(1, w, 11, 12)}  | group {d(1, w, 0, 12)}")
for job, last in ((0, 6), (1, 12)):
    t0 = st[:, job, :, 0].min()
    print("job", job, "span %.1f us; groups per workgroup %.1f" % ((st[:, job, :, last].max() - t0) / 100.0, G / 512.0))
