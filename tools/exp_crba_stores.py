"""mh_crba_f64 on the humanoid: time per call at several batch sizes (code object picked through MH_SPEC_DIR: write-out experiments)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
sys_ = rt.nextHumanoid(np.random.default_rng(43))
hm = HipModel(sys_.toModelDesc()); lib = _lib.load()
print(hm.kernel_variant, flush=True)
stream = torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]]:
    q = torch.tensor(rt.nextState(np.random.default_rng(B), sys_, B)[0], device="cuda")
    out = torch.empty(B, hm.nv, hm.nv, device="cuda", dtype=torch.float64)
    opts = hm._options(_lib.LAYOUT_AOS, True, True, stream)
    fn = lambda: _lib.check(lib.mh_crba_f64(hm._h, B, q.data_ptr(), ctypes.byref(opts), out.data_ptr()))
    for _ in range(10): fn()
    best = 1e9
    for r in range(5):
        t = HipTimer(); t.start(stream)
        for _ in range(30): fn()
        t.stop(stream)
        best = min(best, t.elapsed_ms() / 30 * 1e3)
    print(f"CRBA B={B}: {best:.1f} us  {B / best / 1e3:.3f} G/s  {B * 7448 / best / 1e3:.0f} GB/s  checksum {float(out.abs().sum()):.9e}", flush=True)
