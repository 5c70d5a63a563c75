"""Builds the HIP shared libraries in-tree (gfx950 only):

* mecano_amd/libmecano_hip.so                  the C-ABI + the generic (run-time topology) kernels
* mecano_amd/libmecano_hip_topo_<key>.so       one topology-specialised code object per registered kinematic-tree shape;
                                               libmecano_hip.so picks it up at mh_model_create when the key matches.

`python -m mecano_amd.build` builds everything; `build_spec(desc)` builds the code object of any other model
(hipcc is needed at that moment; it takes about half a minute per topology).
"""
from __future__ import annotations

import ctypes
import functools
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libmecano_hip.so")
CSRC = os.path.join(HERE, "csrc")
SOURCES = [os.path.join(CSRC, "mh_api.hip"), os.path.join(CSRC, "mh_comm.hip")]
HEADERS = [os.path.join(CSRC, "mh_kernels.h"), os.path.join(CSRC, "mh_device.h"), os.path.join(ROOT, "include", "mecano_hip.h")]
LIB_HEADERS = HEADERS + [os.path.join(CSRC, "mh_dfs_kernels.h"), os.path.join(CSRC, "mh_split_kernels.h")]  # the library's own kernels
SPEC_SOURCE = os.path.join(CSRC, "mh_spec.hip")
SPEC_HEADERS = HEADERS + [os.path.join(CSRC, "mh_spec_kernels.h"), os.path.join(CSRC, "mh_zv_kernels.h")]
# what a code object is hashed over, in this order (mh_api.hip: kSpecHashFiles / kSpecCodegenFlags hold the same lists: the library's
# own builder, mh_build_code_object, computes the same number; tests/test_build_provenance.py compares the two implementations)
SPEC_HASH_FILES = [os.path.join(CSRC, n) for n in ("mh_spec.hip", "mh_spec_kernels.h", "mh_zv_kernels.h", "mh_kernels.h", "mh_device.h")]
# -fno-signed-zeros -ffinite-math-only: lets the compiler drop the multiplications by the structural zeros of the canonical
# joint frames (S = e_z); no reassociation is enabled, products and sums keep their written order.
# -fno-slp-vectorize: hipcc 7.2 packs adjacent fp32 operations into v_pk_* instructions; in crba_kernel<float> that came with a wrong
# component select for the first row of a multi-DoF joint's diagonal block (tools/diag_f32_crba2.py, DESIGN.md open issues), and packed
# fp32 VALU is no faster on gfx950 anyway.  fp64 code is unaffected (there are no packed fp64 instructions).
CODEGEN_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-signed-zeros", "-ffinite-math-only", "-fno-slp-vectorize"]
FLAGS = CODEGEN_FLAGS + ["-fPIC", "-shared"]
# the code objects only.  -disable-machine-licm: the kernels that loop over groups of 64 configurations (persistent workgroups at device-
# filling batches) are a few thousand instructions of straight-line code per turn; the machine-level loop-invariant code motion lifts the
# literal constants of that body (sincos coefficients, 1.0, ...) out of the loop into ~24 VGPRs that then live across every phase.  Without
# it: fused forward dynamics 256 registers + 32 bytes of scratch -> 240 and none, tree-split RNEA 220 -> 190, tree-split CRBA 125 -> 98.
SPEC_CODEGEN_FLAGS = CODEGEN_FLAGS + ["-mllvm", "-disable-machine-licm"]
SPEC_FLAGS = SPEC_CODEGEN_FLAGS + ["-fPIC", "-shared"]


# deepest root-to-leaf path (in joints) a topology-specialised code object is built for: the humanoid is 9 deep; a 30-joint chain needs
# the whole register file plus ~600 spilled VGPRs for forward dynamics (mh_spec.hip: kWholeTreeMaxBodies)
MAX_SPEC_DEPTH = 16


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


# ---- build provenance: binaries are tied to their sources by CONTENT, not by file times (a reverted experiment plus a
#      `touch` shipped a stale kernel with every test green against its own self-check).  FNV-1a 64 over, per file, name 0 contents 0,
#      then the code-generation flags; every binary carries the number in the string "MH_BUILD_ID=...;" (found here by reading the file,
#      no dlopen) and the library refuses a code object whose number is not the one it was built beside (mh_api.hip: try_load_spec).
def _fnv1a(data: bytes, h: int = 0xCBF29CE484222325) -> int:
    for byte in data:
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _hash_files(paths, flags, h: int = 0xCBF29CE484222325) -> int:
    stamp = tuple((p, os.stat(p).st_mtime_ns, os.stat(p).st_size) for p in paths)  # memo key only: the hash is over contents
    return _hash_files_memo(stamp, tuple(flags), h)


@functools.lru_cache(maxsize=64)
def _hash_files_memo(stamp, flags, h):
    for p, _, _ in stamp:
        h = _fnv1a(os.path.basename(p).encode() + b"\0", h)
        with open(p, "rb") as f:
            h = _fnv1a(f.read(), h)
        h = _fnv1a(b"\0", h)
    return _fnv1a(" ".join(flags).encode(), h)


def spec_sources_hash(csrc: str = CSRC) -> str:
    """What a code object must carry to be loaded: kernel sources + code-generation flags (the topology is in its name and its tables)."""
    return "h%016x" % _hash_files([os.path.join(csrc, os.path.basename(p)) for p in SPEC_HASH_FILES], SPEC_CODEGEN_FLAGS)


def lib_hash() -> str:
    """The library's own sources, headers and flags, and the code-object hash it is going to expect."""
    return "h%016x" % _fnv1a(spec_sources_hash().encode(), _hash_files(SOURCES + LIB_HEADERS, CODEGEN_FLAGS))


def extra_hash(extra=()) -> str:
    return "h%016x" % _fnv1a(" ".join(extra).encode()) if extra else "none"


def build_id_of(path: str):
    """The "MH_BUILD_ID=...;" string a binary carries (None: it has none)."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(rb"MH_BUILD_ID=([ -~]*?;)(?=[^ -~]|$)", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def lib_build_id() -> str:
    return f"{lib_hash()};spec={spec_sources_hash()};"


def spec_build_id(parents, kinds, extra=()) -> str:
    return (f"{spec_sources_hash()};N={len(parents)};P={','.join(str(int(x)) for x in parents)};T={','.join(str(int(x)) for x in kinds)};"
            f"X={extra_hash(extra)};")


def spec_defines(parents, kinds, extra=()):
    """The -D flags of a code object: the tree, the hash of its sources (checked by the library at load) and of any extra flags."""
    return [f"-DMH_TOPO_N={len(parents)}", "-DMH_TOPO_PARENTS=" + ",".join(str(int(x)) for x in parents),
            "-DMH_TOPO_TYPES=" + ",".join(str(int(x)) for x in kinds), "-DMH_SPEC_SOURCES_HASH=" + spec_sources_hash(),
            "-DMH_BUILD_EXTRA=" + extra_hash(extra)] + list(extra)


def needs_build() -> bool:
    return build_id_of(LIB) != lib_build_id()


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    tmp = f"{LIB}.tmp{os.getpid()}"
    cmd = [hipcc()] + FLAGS + [f"-DMH_BUILD_HASH={lib_hash()}", f"-DMH_SPEC_SOURCES_HASH={spec_sources_hash()}",
                               "-I" + os.path.join(ROOT, "include"), "-o", tmp] + SOURCES + ["-ldl"]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)  # atomic: a concurrent load never sees a half-written library
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def topology_of(desc):
    """(key, parents, kinds) of a ModelDesc in the engine's parents-first order, from the library's own host-side planner."""
    from . import _lib
    build_lib()
    lib = _lib.load()
    keep = []
    d = _lib.MhModelDesc()
    d.n_joints, d.nq, d.nv = int(desc.n_joints), int(desc.nq), int(desc.nv)
    for k, dt in (("parent", np.int32), ("joint_type", np.int32), ("dof_indices", np.int32), ("cfg_indices", np.int32), ("axis", np.float64),
                  ("X_before", np.float64), ("X_com", np.float64), ("inertia_J", np.float64), ("inertia_mass", np.float64),
                  ("inertia_com", np.float64)):
        a = np.ascontiguousarray(getattr(desc, k), dtype=dt)
        keep.append(a)
        setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
    key = ctypes.create_string_buffer(17)
    parents = np.zeros(desc.n_joints, dtype=np.int32)
    kinds = np.zeros(desc.n_joints, dtype=np.int32)
    _lib.check(lib.mh_topology_key(ctypes.byref(d), key, parents.ctypes.data, kinds.ctypes.data))
    return key.value.decode(), parents, kinds


def spec_path(key: str) -> str:
    return os.path.join(HERE, f"libmecano_hip_topo_{key}.so")


def build_spec(desc, force: bool = False, verbose: bool = False) -> str:
    """Builds the topology-specialised code object of a model (any ModelDesc).  Returns its path."""
    key, parents, kinds = topology_of(desc)
    if any(int(k) > 3 for k in kinds):
        raise ValueError("specialised code objects cover revolute, prismatic, 6-DoF and fixed joints; planar / spherical joints run on the generic kernels")
    depth = [0] * len(parents)
    for j, pj in enumerate(parents):
        depth[j] = 1 + (depth[int(pj)] if int(pj) >= 0 else 0)
    if max(depth) > MAX_SPEC_DEPTH:
        raise ValueError(f"the tree is {max(depth)} joints deep: a compile-time walk keeps the state of every body on a root-to-leaf path in "
                         f"registers, which stops paying (512 registers plus hundreds of spills, minutes of compile time) beyond "
                         f"{MAX_SPEC_DEPTH}; such models run on the run-time-topology kernels")
    out = spec_path(key)
    if not force and build_id_of(out) == spec_build_id(parents, kinds):
        return out
    defs = spec_defines(parents, kinds)
    extra = ["-Rpass-analysis=kernel-resource-usage"] if verbose else []
    # two translation units (mh_spec.hip, MH_SPEC_PART = 1 | 2), compiled side by side and linked into one code object: hipcc generates the
    # device code of a unit kernel by kernel on one core, and the humanoid's single unit took five minutes.  Objects and the linked file
    # carry this process' id until the finished object is moved into place (two ranks or two test workers may build the same tree at once).
    from concurrent.futures import ThreadPoolExecutor
    compile_flags = [f for f in SPEC_FLAGS if f != "-shared"]
    objs = [f"{out[:-3]}.part{part}.{os.getpid()}.o" for part in (1, 2)]
    tmp = f"{out}.tmp{os.getpid()}"

    def one(part):
        subprocess.check_call([hipcc()] + compile_flags + defs + extra + [f"-DMH_SPEC_PART={part}", "-c", "-o", objs[part - 1], SPEC_SOURCE])

    try:
        with ThreadPoolExecutor(max_workers=2) as pool:
            list(pool.map(one, (1, 2)))
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
        os.replace(tmp, out)
    finally:
        for o in objs + [tmp]:
            if os.path.exists(o):
                os.remove(o)
    return out


def verify(paths=None):
    """Every shipped binary must carry the build id of the CURRENT sources: [(path, found, expected)] of those that do not."""
    bad = []
    if build_id_of(LIB) != lib_build_id():
        bad.append((LIB, build_id_of(LIB), lib_build_id()))
    for name, desc in registered_models().items():
        key, parents, kinds = topology_of(desc)
        want = spec_build_id(parents, kinds)
        if build_id_of(spec_path(key)) != want:
            bad.append((spec_path(key), build_id_of(spec_path(key)), want))
    return bad


def registered_models():
    """Topologies that get a specialised code object at build time: the shapes BASELINE.json's configs name."""
    from . import random_tools as rt
    from .multibody import MultiBodySystem
    rng = np.random.default_rng(0)
    humanoid = rt.nextHumanoid(rng).toModelDesc()                                                     # configs[2], configs[3], the metric
    arm7 = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7)[0].getPredecessor()).toModelDesc()  # configs[0], configs[1]
    # two more tree shapes, so that every branch of the tree-split planner has a code object the GPU tests run (seconds to compile):
    quadruped = rt.nextQuadruped(rng).toModelDesc()       # limbs on the root only: plain split, no staged trunk
    torso = rt.nextFixedBaseTorso(rng).toModelDesc()      # revolute root, sub-trunk, mixed joints, a one-body late limb
    centaur = rt.nextCentaur(rng).toModelDesc()           # two sub-trunks folded by two waves, a one-body late limb on the root
    return {"humanoid30": humanoid, "arm7": arm7, "quadruped18": quadruped, "torso13": torso, "centaur20": centaur}


def build_all(force: bool = False, verbose: bool = False, jobs: int = 4):
    """The library first (the topology keys come from its host-side planner), then the code objects of the registered shapes side by
    side: they are independent hipcc runs, the humanoid's alone takes minutes."""
    from concurrent.futures import ThreadPoolExecutor
    out = [build_lib(force, verbose)]
    from . import _lib
    _lib.load()  # once, on this thread, before the workers ask the library for topology keys
    models = list(registered_models().values())
    with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(models)))) as pool:
        out += list(pool.map(lambda desc: build_spec(desc, force, verbose), models))
    return out


if __name__ == "__main__":
    import sys
    for p in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv):
        print(p)
