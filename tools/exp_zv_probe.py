"""Phase stamps (100 MHz real-time counter, 10 ns ticks) of the bias-split forward dynamics; needs the code object built with -DMH_ZV_PROBE:
python tools/isa.py --so -DMH_ZV_PROBE ; MH_SPEC_DIR=build/exp python tools/exp_zv_probe.py [B] [jobs: aba|pair]"""
import ctypes, os, sys, glob
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
what = sys.argv[2] if len(sys.argv) > 2 else "pair"
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc())
so = ctypes.CDLL(glob.glob(os.path.join(os.environ["MH_SPEC_DIR"], "libmecano_hip_topo_b5c1*.so"))[0])
q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
g = (0, 0, -9.81)
o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
f = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
def read():
    b = np.zeros(4096 * 3 * 4 * 16, dtype=np.uint64)
    assert so.mh_spec_zv_probe_read(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(b.nbytes)) == 0
    return b
for _ in range(6):
    hm.rnea_aba(q, qd, qdd, tau, g) if what == "pair" else hm.aba(q, qd, tau, g)
    torch.cuda.synchronize()
if what == "pair":  # back-to-back launches: where does launch n + 1 start relative to the end of launch n?
    for n in (9, 10):
        for _ in range(n): f()
        torch.cuda.synchronize()
        b = read().reshape(4096, 3, 4, 16)[:min((B + 63) // 64, 4096)].astype(np.int64)
        print(f"after {n} back-to-back launches: last launch first kernel entry {b[:, :, :, 15].min()}, last stamp {max(b[:, :2, :, :12].max(), b[:, 2, :, 1].max())} (10 ns ticks)")
buf = read()
G = min((B + 63) // 64, 4096)
st = buf.reshape(4096, 3, 4, 16)[:G].astype(np.int64)
t0 = st[:, :, :, 15].min()  # first instruction of the launch
names = {0: ["entry", "staged", "limbs", "barrier", "trunk", "copied", "fenced", "flag", "q_staged", "pairs"],
         1: ["entry", "staged(q)", "limbs_in", "barrier2", "root", "flag_seen", "tau_staged", "early_fold", "late_fold+sub", "root_fold", "out", "copied"]}
print("B", B, what, "variant", hm.kernel_variant, "(times in us relative to the first entry; median over groups)")
for job in (0, 1):
    for w in range(4):
        print(f"job {job} wave {w}: " + "  ".join(f"{n}={np.median(st[:, job, w, k] - t0) / 100:6.2f}" for k, n in enumerate(names[job])))
print("wait for the flag (flag_seen - root), wave 0: median %.2f us, max %.2f us" % (np.median(st[:, 1, 0, 5] - st[:, 1, 0, 4]) / 100, (st[:, 1, 0, 5] - st[:, 1, 0, 4]).max() / 100))
print("flag stored -> flag seen (same group): median %.2f us" % (np.median(st[:, 1, 0, 5] - st[:, 0, 0, 7]) / 100))
print("entry skew: bias job %.2f..%.2f us, inertia job %.2f..%.2f us" % ((st[:, 0, 0, 0] - t0).min() / 100, (st[:, 0, 0, 0] - t0).max() / 100, (st[:, 1, 0, 0] - t0).min() / 100, (st[:, 1, 0, 0] - t0).max() / 100))
if (st[:, 1, :, 12] > 0).all():
    for w in range(4):
        print(f"inward limb phase, wave {w}: first pass (cold instruction cache) {np.median(st[:, 1, w, 12] - st[:, 1, w, 1]) / 100:.2f} us, second pass (warm) {np.median(st[:, 1, w, 2] - st[:, 1, w, 13]) / 100:.2f} us")
print("after the prologue (before the scalar-cache warm-up), median: bias %.2f, inertia %.2f us" % (np.median(st[:, 0, :, 14] - t0) / 100, np.median(st[:, 1, :, 14] - t0) / 100))
print("kernel entry (first instruction) relative to t0: bias %.2f..%.2f, inertia %.2f..%.2f us" % ((st[:, 0, :, 15].min() - t0) / 100, (st[:, 0, :, 15].max() - t0) / 100, (st[:, 1, :, 15].min() - t0) / 100, (st[:, 1, :, 15].max() - t0) / 100))
if what == "pair":
    print("inverse dynamics job: kernel entry %.2f..%.2f us, end %.2f..%.2f us (median %.2f)" % ((st[:, 2, :, 15].min() - t0) / 100, (st[:, 2, :, 15].max() - t0) / 100, (st[:, 2, :, 1].min() - t0) / 100, (st[:, 2, :, 1].max() - t0) / 100, np.median(st[:, 2, :, 1] - t0) / 100))
print("span: %.2f us" % ((st[:, :2, :, :12].max() - t0) / 100))
# ---- per-body stamps of the inertia job's inward walk (ZV_STAMP_BODY): 0 children done, 1 constants in + inertia summed, 2 division and
#      downdate done, 3 handed up; us relative to the first entry, median over the groups
if hasattr(so, "mh_spec_zv_probe_body_read"):
    bb = np.zeros(4096 * 32 * 4, dtype=np.uint64)
    assert so.mh_spec_zv_probe_body_read(bb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bb.nbytes)) == 0
    bb = bb.reshape(4096, 32, 4)[:G].astype(np.int64)
    print("inertia job, inward walk, per body (us after the first entry): children_done  summed  downdated  handed_up | step = handed_up - children_done")
    for j in range(hm.n_joints):
        m = np.median(bb[:, j, :] - t0, axis=0) / 100
        if (bb[:, j, 3] > 0).all():
            print(f"  body {j:2d}: {m[0]:6.2f} {m[1]:6.2f} {m[2]:6.2f} {m[3]:6.2f} | {m[3] - m[0]:5.2f}   (sum {m[1]-m[0]:.2f}, divide+downdate {m[2]-m[1]:.2f}, congruence+translate {m[3]-m[2]:.2f})")
