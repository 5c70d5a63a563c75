"""Consumer of the golden vectors java/us/ihmc/mecano/hip/tools/MecanoGoldenVectorHarness.java writes from the REAL Mecano calculators.

No JVM exists in this repository's image, so no such file is committed: `tests/golden/mecano_*.json` is what a maintainer with a JDK and the
Mecano / Euclid / EJML jars produces (one command, see the harness's header) and drops next to the committed inputs
(`tests/golden/states_humanoid30.json`, `mecano_amd/models/humanoid30.json`).  From that moment these tests compare

* the CPU oracle (oracle/mecano_oracle.c) with Mecano's own tau, qdd, H and C to 1e-10 -- which is what turns "parity unpinned" into a pin, and
* (-m gpu) the HIP path with the same numbers through the C-ABI.

Until then the Mecano comparisons are SKIPPED, loudly; what does run on every CPU test pass is the plumbing: the committed state file is what
its generator produces (inputs only, no results), and the consumer itself is exercised on a file of the harness's schema written by the
oracle into a temporary directory (labelled as such -- it proves the reader and the comparison code, not parity).
"""
import glob
import json
import os

import numpy as np
import pytest

from mecano_amd import random_tools as rt
from oracle.cpu_oracle import OracleModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MECANO_FILES = sorted(glob.glob(os.path.join(GOLDEN, "mecano_*.json")))
TOL = 1.0e-10


def load_cases(path):
    doc = json.load(open(path))
    cases = doc["cases"]
    arr = lambda key: np.array([c[key] for c in cases], dtype=np.float64)
    nv = int(doc["nv"])
    out = {"gravity": tuple(doc["gravity"]), "q": arr("q"), "qd": arr("qd"), "qdd": arr("qdd"), "tau_in": arr("tau_in"), "tau": arr("tau"),
           "qdd_out": arr("qdd_out"), "H": arr("H").reshape(len(cases), nv, nv)}
    if "C" in cases[0]:
        out["C"] = arr("C").reshape(len(cases), nv, nv)
    return doc, out


def compare(results, want, label):
    """results: callables of this repository's implementation; want: the file's numbers.  Absolute 1e-10 on tau and qdd (north_star), relative
    to max(1, |ref|) on the matrices."""
    g = want["gravity"]
    tau = results["rnea"](want["q"], want["qd"], want["qdd"], g)
    assert np.abs(tau - want["tau"]).max() <= TOL, (label, "tau", np.abs(tau - want["tau"]).max())
    qdd = results["aba"](want["q"], want["qd"], want["tau_in"], g)
    assert np.abs(qdd - want["qdd_out"]).max() <= TOL, (label, "qdd", np.abs(qdd - want["qdd_out"]).max())
    H = results["crba"](want["q"])
    assert np.abs(H - want["H"]).max() <= TOL * max(1.0, np.abs(want["H"]).max()), (label, "H")
    if "C" in want and "coriolis" in results:
        C = results["coriolis"](want["q"], want["qd"])
        assert np.abs(C - want["C"]).max() <= TOL * max(1.0, np.abs(want["C"]).max()), (label, "C")


def oracle_results(desc):
    om = OracleModel(desc)
    return {"rnea": om.rnea, "aba": om.aba, "crba": om.crba, "coriolis": lambda q, qd: om.crba_coriolis(q, qd)[1]}


def test_committed_state_file_is_what_its_generator_writes():
    doc = json.load(open(os.path.join(GOLDEN, "states_humanoid30.json")))
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, 16)
    d = rt.humanoid30Desc()
    assert (doc["nq"], doc["nv"]) == (d.nq, d.nv) and doc["gravity"] == [0.0, 0.0, -9.81]
    for key, ref in (("q", q), ("qd", qd), ("qdd", qdd), ("tau", tau)):
        assert np.array_equal(np.array(doc[key]), ref), key
    assert not any(k in doc for k in ("tau_out", "qdd_out", "H", "cases")), "the state file holds inputs only"


def test_consumer_on_a_file_of_the_harness_schema(tmp_path):
    """The reader and the comparison, on a file this test writes itself with the ORACLE's numbers (so it can only prove the plumbing)."""
    states = json.load(open(os.path.join(GOLDEN, "states_humanoid30.json")))
    d = rt.humanoid30Desc()
    om = OracleModel(d)
    q, qd, qdd, tau = (np.array(states[k]) for k in ("q", "qd", "qdd", "tau"))
    g = tuple(states["gravity"])
    t, a, (H, C) = om.rnea(q, qd, qdd, g), om.aba(q, qd, tau, g), om.crba_coriolis(q, qd)
    fmt = lambda v: [float("%.17g" % x) for x in np.asarray(v).reshape(-1)]
    doc = {"generator": "oracle/mecano_oracle.c (self-generated: NOT reference output)", "reference_output": False, "model": "humanoid30.json",
           "n_joints": d.n_joints, "nq": d.nq, "nv": d.nv, "gravity": list(g),
           "cases": [{"q": fmt(q[s]), "qd": fmt(qd[s]), "qdd": fmt(qdd[s]), "tau_in": fmt(tau[s]), "tau": fmt(t[s]), "qdd_out": fmt(a[s]),
                      "H": fmt(H[s]), "C": fmt(C[s])} for s in range(len(q))]}
    path = tmp_path / "selfgenerated_humanoid30.json"
    path.write_text(json.dumps(doc))
    meta, want = load_cases(str(path))
    assert meta["reference_output"] is False and want["H"].shape == (16, d.nv, d.nv)
    compare(oracle_results(d), want, "oracle vs its own file")
    want["tau"][3, 7] += 1e-9  # and the comparison does notice a difference of 1e-9
    with pytest.raises(AssertionError):
        compare(oracle_results(d), want, "perturbed")


@pytest.mark.skipif(not MECANO_FILES, reason="no tests/golden/mecano_*.json: golden vectors of the real Mecano calculators need a JVM "
                                             "(java/us/ihmc/mecano/hip/tools/MecanoGoldenVectorHarness.java); parity against the Java reference stays unpinned")
@pytest.mark.parametrize("path", MECANO_FILES or [None])
def test_oracle_matches_mecano(path):
    meta, want = load_cases(path)
    assert meta.get("reference_output") is True, "only files written by the Java harness pin parity"
    assert meta["model"] == "humanoid30.json"
    compare(oracle_results(rt.humanoid30Desc()), want, "oracle vs Mecano")


@pytest.mark.gpu
@pytest.mark.skipif(not MECANO_FILES, reason="no tests/golden/mecano_*.json (needs a JVM once; see MecanoGoldenVectorHarness.java)")
@pytest.mark.parametrize("path", MECANO_FILES or [None])
def test_hip_path_matches_mecano(path):
    import torch
    from mecano_amd.engine import HipModel
    meta, want = load_cases(path)
    assert meta.get("reference_output") is True
    hm = HipModel(rt.humanoid30Desc())
    dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda")
    results = {"rnea": lambda q, qd, qdd, g: hm.rnea(dev(q), dev(qd), dev(qdd), g).cpu().numpy(),
               "aba": lambda q, qd, tau, g: hm.aba(dev(q), dev(qd), dev(tau), g).cpu().numpy(),
               "crba": lambda q: hm.crba(dev(q)).cpu().numpy(),
               "coriolis": lambda q, qd: hm.crba_coriolis(dev(q), dev(qd))[1].cpu().numpy()}
    compare(results, want, "HIP vs Mecano")
