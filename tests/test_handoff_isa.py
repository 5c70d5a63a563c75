"""The hand-off of the bias-split forward dynamics (mh_zv_kernels.h) is correct by an ISA-level convention -- relaxed agent-scope atomics
that hipcc lowers to sc1 accesses, an inline `s_waitcnt vmcnt(0)`, workgroup barriers -- not by anything the language promises.  A compiler
update that re-scopes one access or moves the flag store would break it silently.  So every registered code object is
disassembled and its spec_zv_kernel instruction streams are checked against the protocol (tools/isa_handoff.py); and one deliberately
broken build (the flag stored BEFORE the rows are drained) proves that the check sees such a break.  No GPU needed: hipcc cross-compiles and
llvm-objdump reads the code objects the build left in the tree.  Round 4: with identity index maps the hand-off runs in two stages (the limbs'
columns under one flag, the trunk's under another; three waves counted in through LDS in front of the first) -- the checker recognises
which form a stream is and holds it to that form's rules; the broken build breaks the two-stage form (flag B in front of wave 0's drain)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_handoff  # noqa: E402

from mecano_amd import build as mbuild  # noqa: E402


def _registered():
    return mbuild.registered_models()


@pytest.fixture(scope="module")
def code_objects():
    """{name: path} of the registered code objects (built by __graft_entry__.build(); built here when they are missing or stale)."""
    out = {}
    for name, desc in _registered().items():
        out[name] = mbuild.build_spec(desc)
    return out


def test_every_registered_code_object_keeps_the_hand_off_protocol(code_objects):
    checked = 0
    for name, path in code_objects.items():
        found = isa_handoff.zv_kernels(isa_handoff.disassemble(path))
        if name == "arm7":
            assert not found, "a chain has no tree-split form, hence no bias-split kernel"
            continue
        assert found, f"{name}: no spec_zv_kernel in {os.path.basename(path)}"
        for kernel, instrs in found.items():
            bad = isa_handoff.check_handoff(instrs)
            assert not bad, f"{name} {kernel[:60]}: " + "; ".join(bad)
            checked += 1
    assert checked >= 4  # humanoid, quadruped, torso, centaur; identity and general index maps each


def test_the_check_catches_a_flag_stored_before_the_rows_are_drained(tmp_path):
    """-DMH_ZV_TEST_FLAG_BEFORE_DRAIN moves the flag store in front of `s_waitcnt vmcnt(0)` + barrier (mh_zv_kernels.h): compiled to ISA text
    only (the smallest staged topology, the minimal kernel set), never linked.  The same source without the macro must pass."""
    desc = _registered()["torso13"]
    key, parents, kinds = mbuild.topology_of(desc)
    defs = mbuild.spec_defines(parents, kinds, ("-DMH_SPEC_MINIMAL",))
    flags = [f for f in mbuild.SPEC_FLAGS if f not in ("-shared", "-fPIC")]
    verdicts = {}
    for tag, extra in (("as shipped", []), ("flag before drain", ["-DMH_ZV_TEST_FLAG_BEFORE_DRAIN"])):
        out = tmp_path / (tag.replace(" ", "_") + ".s")
        subprocess.check_call([mbuild.hipcc()] + flags + defs + extra + ["--cuda-device-only", "-S", "-o", str(out), mbuild.SPEC_SOURCE],
                              stderr=subprocess.DEVNULL)
        found = isa_handoff.zv_kernels(out.read_text())
        assert found, tag
        verdicts[tag] = [b for instrs in found.values() for b in isa_handoff.check_handoff(instrs)]
    assert verdicts["as shipped"] == []
    assert any("flag can overtake its" in b for b in verdicts["flag before drain"]), verdicts["flag before drain"]
