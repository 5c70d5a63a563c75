"""Diagnosis: kernel variant (and the self-check's verdict) of the registered tree shapes with identity and permuted index maps, and of the
torso of tests/test_gpu_parity.py::test_specialised_coriolis_kernel_variants."""
import os, sys, zlib, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
rng = np.random.default_rng(zlib.crc32(("speccor" + "torso").encode()))
sys_ = rt.nextFixedBaseTorso(rng)
d = sys_.toModelDesc()
for B in (1, 100, 4096):
    rt.nextState(rng, sys_, B)
Rf = [np.linalg.qr(rng.normal(size=(3, 3)))[0] for _ in range(0)]
for rep in range(4):
    perm_v, perm_q = rng.permutation(d.nv).astype(np.int32), rng.permutation(d.nq).astype(np.int32)
    d2 = sys_.toModelDesc()
    d2.dof_indices = perm_v[np.asarray(d.dof_indices)]
    d2.cfg_indices = perm_q[np.asarray(d.cfg_indices)]
    print("permuted torso", rep, ":", HipModel(d2).kernel_variant[:400], flush=True)
