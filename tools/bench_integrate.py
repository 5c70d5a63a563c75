"""Streaming rate of mh_integrate_f64 (HBM-bound: reads q, qd, qdd, writes q', qd') on the 30-DoF humanoid."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, build as b
from mecano_amd.engine import HipModel, HipTimer

desc = b.registered_models()["humanoid30"]
hm = HipModel(desc)
t = HipTimer()
for B in (4096, 262144, 1048576, 4194304):
    for layout, name in ((_lib.LAYOUT_AOS, "AoS"), (_lib.LAYOUT_SOA, "SoA")):
        shp = (lambda n: (B, n)) if layout == _lib.LAYOUT_AOS else (lambda n: (n, B))
        q = torch.randn(shp(desc.nq), device="cuda", dtype=torch.float64); qd = torch.randn(shp(desc.nv), device="cuda", dtype=torch.float64); qdd = torch.randn_like(qd)
        qo, vo = torch.empty_like(q), torch.empty_like(qd)
        for _ in range(5): hm.integrate(1e-3, q, qd, qdd, layout, out=(qo, vo))
        torch.cuda.synchronize()
        n = 50
        t.start()
        for _ in range(n): hm.integrate(1e-3, q, qd, qdd, layout, out=(qo, vo))
        t.stop()
        us = t.elapsed_ms() * 1e3 / n
        byts = B * (desc.nq + 2 * desc.nv + desc.nq + desc.nv) * 8
        print(f"B={B:8d} {name}: {us:9.1f} us/launch  {B/us:8.1f} M configs/s  {byts/us/1e3:8.1f} GB/s algorithmic ({byts/us/1e3/8000*100:5.1f}% of 8 TB/s)", flush=True)
