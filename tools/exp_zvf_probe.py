"""Phase stamps (100 MHz real-time counter) of the fused forward dynamics at a device-filling batch; code object built with -DMH_ZV_PROBE:
EXP_DIR=exp_probe python tools/isa.py --so -DMH_ZV_PROBE ; MH_SPEC_DIR=$PWD/exp_probe MH_ZVF=2 MH_ZV=0 python tools/exp_zvf_probe.py [B]"""
import ctypes, os, sys, glob
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
PAIR = len(sys.argv) > 2 and sys.argv[2] == "pair"  # the pair call (one launch: the M(q) qdd phase behind the outward sweep)
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
so = ctypes.CDLL(glob.glob(os.path.join(os.environ["MH_SPEC_DIR"], "libmecano_hip_topo_b5c1*.so"))[0])
q, qd, qdd, tau = rt.nextState(np.random.default_rng(1), sys_, min(B, 32768))
rep = (B + q.shape[0] - 1) // q.shape[0]
q, qd, qdd, tau = (torch.tensor(x, device="cuda").repeat(rep, 1)[:B].contiguous() for x in (q, qd, qdd, tau))
g = (0, 0, -9.81)
for _ in range(4):
    hm.rnea_aba(q, qd, qdd, tau, g) if PAIR else hm.aba(q, qd, tau, g)
    torch.cuda.synchronize()
b = np.zeros(4096 * 3 * 4 * 16, dtype=np.uint64)
assert so.mh_spec_zv_probe_read(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(b.nbytes)) == 0
G = min((B + 63) // 64, 4096)
st = b.reshape(4096, 3, 4, 16)[:G].astype(np.int64)
print("B", B, "variant", hm.kernel_variant[:40], "groups", G, "(durations in us, median over groups [10 %, 90 %])")
def d(w, a, z, ja=2, jz=2):
    x = (st[:, jz, w, z] - st[:, ja, w, a]) / 100.0
    return "%5.2f [%5.2f %5.2f]" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90))
for w in range(4):
    print(f"wave {w}: stage {d(w, 0, 1)}  rnea limbs {d(w, 1, 2)}  wait {d(w, 2, 3)}  rnea trunk {d(w, 3, 4)}  bar {d(w, 4, 4)}  limbs_in {d(w, 4, 5)}  wait {d(w, 5, 6)}  "
          f"root {d(w, 6, 7)}  bar {d(w, 7, 8)}  early {d(w, 8, 7, 2, 1)}  late+sub {d(w, 7, 8, 1, 1)}  root {d(w, 8, 9, 1, 1)}  out {d(w, 9, 10, 1, 2)}  copy {d(w, 10, 11)}  " + (f"M qdd limbs {d(w, 11, 13)}  trunk {d(w, 13, 14)}  copy {d(w, 14, 12)}" if PAIR else f"bar {d(w, 11, 12)}") + f"  | group {d(w, 0, 12)}")
t0 = st[:, 2, :, 0].min()
print("span %.1f us; groups per workgroup %.1f" % ((st[:, 2, :, 12].max() - t0) / 100.0, G / 512.0))
