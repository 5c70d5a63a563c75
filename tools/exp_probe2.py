"""Phase stamps per slice ORDER: with MH_FAKE_CU_COUNT=16 every tree-split workgroup loops over two slices of a 4096 batch; prints the
median stamps of the first and of the second slice of the workgroups separately (cold vs warm: kernel start, caches, TLB)."""
import ctypes, os, sys
os.environ.setdefault("MH_FAKE_CU_COUNT", "16")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, build as b
from mecano_amd.engine import HipModel
algo = sys.argv[1] if len(sys.argv) > 1 else "aba"
B = 4096
desc = b.registered_models()["humanoid30"]
hm = HipModel(desc)
lib = _lib.load()
q = torch.randn(B, desc.nq, device="cuda", dtype=torch.float64); q[:, :4] /= q[:, :4].norm(dim=1, keepdim=True)
qd = torch.randn(B, desc.nv, device="cuda", dtype=torch.float64); x = torch.randn_like(qd)
nslice = (B + 63) // 64
out = torch.zeros(B * desc.nv + nslice * 32, device="cuda", dtype=torch.float64)
g = (ctypes.c_double * 3)(0, 0, -9.81)
opts = hm._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream().cuda_stream)
fn = lib.mh_rnea_f64 if algo == "rnea" else lib.mh_aba_f64
for rep in range(6):
    _lib.check(fn(hm._h, B, q.data_ptr(), qd.data_ptr(), x.data_ptr(), g, None, ctypes.byref(opts), out.data_ptr()))
    torch.cuda.synchronize()
st = out[B * desc.nv:].cpu().numpy().view(np.uint64).reshape(nslice, 4, 8).astype(np.int64)
ngroups = 32
names = ["entry", "staged", "limbs", "barrier", "trunk_in", "trunk_out", "barrier2", "copied"]
for order, sl in (("first slice ", slice(0, ngroups)), ("second slice", slice(ngroups, 2 * ngroups))):
    d = st[sl] - st[sl][:, :, :1]
    print(order, algo, "wave 0: " + "  ".join(f"{n}={np.median(d[:, 0, k]):7.0f}" for k, n in enumerate(names) if k != 4 or algo == "aba"))
gap = st[ngroups:2 * ngroups, 0, 0] - st[:ngroups, 0, 7]
print("gap between the end of the first and the entry of the second slice:", int(np.median(gap)))
