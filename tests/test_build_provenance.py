"""Build provenance: binaries are tied to their sources by content, not by file times.  Every binary carries
"MH_BUILD_ID=<hash>;..." -- FNV-1a 64 over its sources, headers and code-generation flags (+ the tree and extra flags for a code object);
mecano_amd/build.py rebuilds whatever does not carry the current hash, libmecano_hip.so refuses a code object whose source hash is not the
one it was built beside (tests/test_gpu_code_objects.py::test_code_object_from_other_sources_is_refused_by_its_source_hash).  No GPU."""
import ctypes
import os
import shutil

import pytest

from mecano_amd import build as b


def test_python_and_library_compute_the_same_source_hash(hip_lib):
    """Two implementations (build.py and mh_api.hip's, which mh_build_code_object uses on a box without Python) of one number."""
    out = ctypes.create_string_buffer(18)
    assert hip_lib.mh_spec_sources_hash_of(b.CSRC.encode(), out) == 0
    assert out.value.decode() == b.spec_sources_hash()
    # ... and the loaded library was built beside these very sources (conftest's hip_lib rebuilt it if not)
    assert hip_lib.mh_spec_sources_hash().decode() == b.spec_sources_hash()
    assert hip_lib.mh_build_hash().decode() == b.lib_hash()
    assert b.build_id_of(b.LIB) == b.lib_build_id()


def test_an_edited_header_makes_every_binary_stale(hip_lib, tmp_path):
    """A copy of csrc/ with one character of a kernel header changed: another hash from both implementations, so the shipped code objects
    would be rebuilt (build_spec compares build ids) and refused at load by a library built beside the edited sources.  A `touch` changes
    nothing."""
    csrc = tmp_path / "csrc"
    shutil.copytree(b.CSRC, csrc)
    assert b.spec_sources_hash(str(csrc)) == b.spec_sources_hash()
    os.utime(csrc / "mh_zv_kernels.h", (1, 1))                      # older than everything: irrelevant
    assert b.spec_sources_hash(str(csrc)) == b.spec_sources_hash()
    with open(csrc / "mh_zv_kernels.h", "a") as f:
        f.write("\n// edited\n")
    edited = b.spec_sources_hash(str(csrc))
    assert edited != b.spec_sources_hash()
    out = ctypes.create_string_buffer(18)
    assert hip_lib.mh_spec_sources_hash_of(str(csrc).encode(), out) == 0 and out.value.decode() == edited
    # what build_spec's staleness test sees for a shipped object against the edited sources: stale
    desc = b.registered_models()["arm7"]
    key, parents, kinds = b.topology_of(desc)
    have = b.build_id_of(b.spec_path(key))
    if have is None:
        pytest.skip("code objects not built")
    assert have == b.spec_build_id(parents, kinds)
    assert have.split(";")[0] != edited
    # a missing source is an error, not a hash
    os.remove(csrc / "mh_device.h")
    assert hip_lib.mh_spec_sources_hash_of(str(csrc).encode(), out) != 0


def test_every_shipped_binary_is_built_from_the_current_sources():
    """What __graft_entry__.build() asserts after building: the library and the five registered code objects carry the current build ids
    (tree, source hash, no extra flags)."""
    if not all(os.path.exists(b.spec_path(b.topology_of(d)[0])) for d in b.registered_models().values()):
        pytest.skip("code objects not built")
    assert b.verify() == []


def test_build_id_distinguishes_extra_flags():
    desc = b.registered_models()["arm7"]
    _, parents, kinds = b.topology_of(desc)
    plain, probe = b.spec_build_id(parents, kinds), b.spec_build_id(parents, kinds, ("-DMH_ZV_PROBE",))
    assert plain != probe and plain.endswith("X=none;") and "-DMH_BUILD_EXTRA=none" in b.spec_defines(parents, kinds)
