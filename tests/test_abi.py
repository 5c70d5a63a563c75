"""The C-ABI shared library: builds, loads, exports every symbol include/mecano_hip.h declares, and validates models.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from mecano_amd import _lib
from mecano_amd import random_tools as rt
from mecano_amd.multibody import MultiBodySystem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mecano_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), name
    assert hip_lib.mh_abi_version() == 5


def _desc_struct(desc, keep):
    d = _lib.MhModelDesc()
    d.n_joints, d.nq, d.nv = desc.n_joints, desc.nq, desc.nv
    for k in ("parent", "joint_type", "dof_indices", "cfg_indices"):
        a = np.ascontiguousarray(getattr(desc, k), dtype=np.int32)
        keep.append(a)
        setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
    for k in ("axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com"):
        a = np.ascontiguousarray(getattr(desc, k), dtype=np.float64)
        keep.append(a)
        setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
    return d


def _create(hip_lib, desc):
    keep = []
    d = _desc_struct(desc, keep)
    h = ctypes.c_void_p()
    st = hip_lib.mh_model_create(ctypes.byref(d), ctypes.byref(h))
    if st == 0:
        hip_lib.mh_model_destroy(h)
    return st, hip_lib.mh_last_error().decode()


def _chain_desc(n=5, seed=0):
    rng = np.random.default_rng(seed)
    joints = rt.nextJointChain(rng, n, ("revolute", "prismatic"))
    return MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor()).toModelDesc()


def test_model_validation_error_codes(hip_lib):
    """Errors come back as status codes + mh_last_error(), never as exceptions across the ABI."""
    good = _chain_desc()
    st, msg = _create(hip_lib, good)
    assert st in (0, 7), msg  # MH_OK on a GPU box, MH_ERR_NO_DEVICE here: validation itself passed

    bad = _chain_desc()
    bad.joint_type = bad.joint_type.copy()
    bad.joint_type[2] = 9
    assert _create(hip_lib, bad)[0] == 3  # MH_ERR_UNSUPPORTED_JOINT

    bad = _chain_desc()
    bad.parent = bad.parent.copy()
    bad.parent[1], bad.parent[2] = 2, 1  # 1 <-> 2 cycle
    assert _create(hip_lib, bad)[0] == 4  # MH_ERR_LOOP_CLOSURE

    bad = _chain_desc()
    bad.parent = bad.parent.copy()
    bad.parent[3] = 17
    assert _create(hip_lib, bad)[0] == 5  # MH_ERR_BAD_TOPOLOGY

    bad = _chain_desc()
    bad.axis = bad.axis.copy()
    bad.axis[3:6] *= 1.5
    assert _create(hip_lib, bad)[0] == 6  # MH_ERR_BAD_AXIS

    bad = _chain_desc()
    bad.dof_indices = bad.dof_indices.copy()
    bad.dof_indices[1] = bad.dof_indices[0]
    st, msg = _create(hip_lib, bad)
    assert st == 5 and "dof_indices" in msg

    h = ctypes.c_void_p()
    assert hip_lib.mh_model_create(None, ctypes.byref(h)) == 1  # MH_ERR_INVALID_ARGUMENT


def test_device_count_call_is_safe_without_gpu(hip_lib):
    n = ctypes.c_int32(-1)
    assert hip_lib.mh_device_count(ctypes.byref(n)) == 0
    assert n.value >= 0


def test_options_default(hip_lib):
    o = _lib.MhOptions()
    hip_lib.mh_options_default(ctypes.byref(o))
    assert (o.consider_coriolis, o.consider_accelerations, o.layout, o.stream) == (1, 1, 0, None)


def test_no_cpu_fallback_without_device(hip_lib):
    """The product path must fail loudly when no GPU is present: model creation reports MH_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mecano_amd.engine import HipModel
    with pytest.raises(_lib.MecanoHipError) as e:
        HipModel(_chain_desc())
    assert e.value.status == 7


def test_header_is_plain_c_and_a_c_host_links(hip_lib, tmp_path):
    """The boundary is a C ABI: include/mecano_hip.h must compile as C99 (no C++-isms, no torch types), and a C host that calls the
    device-free entry points must link against the shared library and run (model validation is host-only; without a HIP device
    mh_model_create reports MH_ERR_NO_DEVICE, on a GPU box MH_OK)."""
    import shutil
    import subprocess
    from mecano_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "mecano_hip.h"
int main(void)
{
   /* a two-joint arm: revolute about z, then prismatic along x */
   int32_t parent[2] = {-1, 0}, type[2] = {MH_JOINT_REVOLUTE, MH_JOINT_PRISMATIC}, dof[2] = {0, 1}, cfg[2] = {0, 1};
   double axis[6] = {0, 0, 1, 1, 0, 0}, X[24] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0.5, 0, 0};
   double J[18] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1}, mass[2] = {1, 2}, com[6] = {0};
   mh_model_desc d;
   memset(&d, 0, sizeof d);
   d.n_joints = 2, d.nq = 2, d.nv = 2;
   d.parent = parent, d.joint_type = type, d.axis = axis, d.X_before = X, d.X_com = X, d.inertia_J = J, d.inertia_mass = mass;
   d.inertia_com = com, d.dof_indices = dof, d.cfg_indices = cfg;
   char key[17];
   int32_t ep[2], et[2], count = -1;
   if (mh_abi_version() != MH_ABI_VERSION) return 1;
   if (mh_topology_key(&d, key, ep, et) != MH_OK || strlen(key) != 16 || ep[0] != -1 || ep[1] != 0) return 2;
   parent[1] = 1; /* its own parent */
   if (mh_topology_key(&d, key, ep, et) != MH_ERR_BAD_TOPOLOGY || strlen(mh_last_error()) == 0) return 3;
   parent[1] = 0;
   if (mh_device_count(&count) != MH_OK || count < 0) return 4;
   mh_model_t m = NULL;
   mh_status st = mh_model_create(&d, &m);
   if (count == 0 && st != MH_ERR_NO_DEVICE) return 5;
   if (count > 0 && (st != MH_OK || mh_model_nv(m) != 2)) return 6;
   if (m) mh_model_destroy(m);
   printf("ok %s %d\n", key, (int)count);
   return 0;
}
''')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "host"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lmecano_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("ok ")


def test_code_objects_can_be_built_through_the_c_abi(hip_lib, tmp_path, monkeypatch):
    """mh_build_code_object: a host without Python gets the topology-specialised code object of its robot from the library itself
    (hipcc cross-compiles here without a GPU); deep chains and planar / spherical joints are refused with a reason."""
    import ctypes
    import shutil
    import numpy as np
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.multibody import MultiBodySystem
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    monkeypatch.setenv("MH_HIPCC_FLAGS", "-DMH_SPEC_MINIMAL")  # seconds instead of minutes: the entry point is under test, not the kernels

    def c_desc(desc):
        keep, d = [], _lib.MhModelDesc()
        d.n_joints, d.nq, d.nv = int(desc.n_joints), int(desc.nq), int(desc.nv)
        for k, dt in (("parent", np.int32), ("joint_type", np.int32), ("dof_indices", np.int32), ("cfg_indices", np.int32), ("axis", np.float64),
                      ("X_before", np.float64), ("X_com", np.float64), ("inertia_J", np.float64), ("inertia_mass", np.float64), ("inertia_com", np.float64)):
            a = np.ascontiguousarray(getattr(desc, k), dtype=dt)
            keep.append(a)
            setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
        return d, keep

    rng = np.random.default_rng(3)
    arm = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 3, ("revolute", "prismatic"))[0].getPredecessor()).toModelDesc()
    d, keep = c_desc(arm)
    path = ctypes.create_string_buffer(1024)
    st = hip_lib.mh_build_code_object(ctypes.byref(d), str(tmp_path).encode(), path, 1024)
    assert st == 0, hip_lib.mh_last_error()
    built = path.value.decode()
    assert os.path.exists(built) and os.path.basename(built).startswith("libmecano_hip_topo_")
    obj = ctypes.CDLL(built)
    obj.mh_spec_abi.restype = ctypes.c_uint64
    assert obj.mh_spec_abi() == hip_lib.mh_spec_abi_stamp() and obj.mh_spec_n() == 3
    deep, keep2 = c_desc(rt.referenceBenchmarkSystems()["chain30"].toModelDesc())
    assert hip_lib.mh_build_code_object(ctypes.byref(deep), str(tmp_path).encode(), path, 1024) == 5 and b"joints deep" in hip_lib.mh_last_error()
    sph, keep3 = c_desc(MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(rng, 4, ("spherical",))[0].getPredecessor()).toModelDesc())
    assert hip_lib.mh_build_code_object(ctypes.byref(sph), str(tmp_path).encode(), path, 1024) == 3


def test_code_object_build_runs_hipcc_without_a_shell(hip_lib, tmp_path, monkeypatch):
    """mh_build_code_object assembled a shell command from its out_dir argument and MH_HIPCC_FLAGS; `$(...)`, back-ticks and
    quotes in a directory name reached /bin/sh.  It now starts hipcc with an argument vector (posix_spawn): a directory whose NAME is shell
    syntax is just a directory, the fast build lands in it under a name of its own (.min.so), and nothing else is executed.  (Cross-compiles
    here without a GPU; the three-finger hand builds in a few seconds.)"""
    import ctypes
    import shutil
    import numpy as np
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.multibody import MultiBodySystem, RigidBody
    rng = np.random.default_rng(515)
    root = RigidBody("root")
    for k in range(3):
        rt.nextJointChain(rng, 2, ("revolute",), rootBody=root, prefix=f"finger{k}_")
    desc = MultiBodySystem.toMultiBodySystemInput(root).toModelDesc()
    keep, d = [], _lib.MhModelDesc()
    d.n_joints, d.nq, d.nv = int(desc.n_joints), int(desc.nq), int(desc.nv)
    for k, dt in (("parent", np.int32), ("joint_type", np.int32), ("dof_indices", np.int32), ("cfg_indices", np.int32), ("axis", np.float64),
                  ("X_before", np.float64), ("X_com", np.float64), ("inertia_J", np.float64), ("inertia_mass", np.float64), ("inertia_com", np.float64)):
        a = np.ascontiguousarray(getattr(desc, k), dtype=dt)
        keep.append(a)
        setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
    marker = tmp_path / "INJECTED"
    evil = tmp_path / "out $(touch INJECTED) `touch INJECTED` \"; touch INJECTED; \""  # cwd = tmp_path below: the marker would land there
    evil.mkdir()
    monkeypatch.setenv("MH_BUILD_FAST", "1")
    monkeypatch.setenv("MH_HIPCC_FLAGS", f"-DMH_UNUSED_A=1   -DMH_UNUSED_B=2")
    monkeypatch.chdir(tmp_path)
    path = ctypes.create_string_buffer(2048)
    rc = hip_lib.mh_build_code_object(ctypes.byref(d), str(evil).encode(), path, 2048)
    assert rc == 0, hip_lib.mh_last_error()
    assert not marker.exists(), "the out_dir argument was interpreted by a shell"
    built = path.value.decode()
    assert built.endswith(".min.so") and os.path.dirname(built) == str(evil) and os.path.getsize(built) > 0
    assert sorted(os.listdir(evil)) == [os.path.basename(built)]  # no temporaries left behind
    # a compiler that fails: the exit status is decoded, the temporary is removed
    monkeypatch.setenv("MH_HIPCC_FLAGS", "--no-such-compiler-flag")  # the compiler refuses to start the build
    rc = hip_lib.mh_build_code_object(ctypes.byref(d), str(evil).encode(), path, 2048)
    assert rc != 0 and b"exited with status" in hip_lib.mh_last_error()
    assert sorted(os.listdir(evil)) == [os.path.basename(built)]
