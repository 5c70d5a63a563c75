"""Consumer of the golden vectors java/us/ihmc/mecano/hip/tools/MecanoGoldenVectorHarness.java writes from the REAL Mecano calculators.

No JVM exists in this repository's image, so no such file is committed: `tests/golden/mecano_*.json` is what a maintainer with a JDK and the
Mecano / Euclid / EJML jars produces (one command, see the harness's header) and drops next to the committed inputs
(`tests/golden/states_<name>.json`, `mecano_amd/models/<name>.json` for the three mechanisms of BASELINE.json's configurations: the 30-DoF humanoid, the
7-DoF arm, the 128-body tree -- ONE harness run, `--all <repository root>`, writes all three).  From that moment these tests compare

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
MECHANISMS = ("humanoid30", "arm7", "tree128")  # BASELINE.json configs 3 / 4 + the metric; configs 1 / 2; config 5
TOL = 1.0e-10


def load_cases(path):
    doc = json.load(open(path))
    cases = doc["cases"]
    arr = lambda key, rows=None: np.array([c[key] for c in (cases if rows is None else cases[:rows])], dtype=np.float64)
    nv = int(doc["nv"])
    out = {"gravity": tuple(doc["gravity"]), "q": arr("q"), "qd": arr("qd"), "qdd": arr("qdd"), "tau_in": arr("tau_in"), "tau": arr("tau"),
           "qdd_out": arr("qdd_out")}
    n_mat = sum(1 for c in cases if "H" in c)  # the matrices come with the leading `with_matrices` states only (323 x 323 on the tree)
    assert all("H" in c for c in cases[:n_mat])
    if n_mat:
        out["H"] = arr("H", n_mat).reshape(n_mat, nv, nv)
        if "C" in cases[0]:
            out["C"] = arr("C", n_mat).reshape(n_mat, nv, nv)
    return doc, out


def compare(results, want, label, tol=TOL):
    """results: callables of this repository's implementation; want: the file's numbers.  1e-10 (north_star) relative to max(1, |ref|) per
    output -- absolute on the humanoid and the arm, whose outputs are O(1..100)."""
    g = want["gravity"]
    rel = lambda ref: tol * max(1.0, float(np.abs(ref).max()))
    tau = results["rnea"](want["q"], want["qd"], want["qdd"], g)
    assert np.abs(tau - want["tau"]).max() <= rel(want["tau"]), (label, "tau", np.abs(tau - want["tau"]).max())
    qdd = results["aba"](want["q"], want["qd"], want["tau_in"], g)
    assert np.abs(qdd - want["qdd_out"]).max() <= rel(want["qdd_out"]), (label, "qdd", np.abs(qdd - want["qdd_out"]).max())
    if "H" in want:
        m = len(want["H"])
        H = results["crba"](want["q"][:m])
        assert np.abs(H - want["H"]).max() <= rel(want["H"]), (label, "H")
        if "C" in want and "coriolis" in results:
            C = results["coriolis"](want["q"][:m], want["qd"][:m])
            assert np.abs(C - want["C"]).max() <= rel(want["C"]), (label, "C")


def oracle_results(desc):
    om = OracleModel(desc)
    return {"rnea": om.rnea, "aba": om.aba, "crba": om.crba, "coriolis": lambda q, qd: om.crba_coriolis(q, qd)[1]}


@pytest.mark.parametrize("name", MECHANISMS)
def test_committed_model_and_state_files_are_what_their_generators_write(name):
    """mecano_amd/models/<name>.json and tests/golden/states_<name>.json (inputs only) against tests/golden/make_*_fixtures.py's recipes."""
    sys_ = rt.committedBenchmarkSystems()[name]
    want, have = sys_.toModelDesc(), rt.modelDescFromJson(name)
    assert (have.n_joints, have.nq, have.nv) == (want.n_joints, want.nq, want.nv)
    for f in ("parent", "joint_type", "dof_indices", "cfg_indices", "axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com"):
        assert np.array_equal(np.asarray(getattr(have, f)).reshape(-1), np.asarray(getattr(want, f)).reshape(-1)), (name, f)
    doc = json.load(open(os.path.join(GOLDEN, f"states_{name}.json")))
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(2342), sys_, len(doc["q"]))
    assert (doc["nq"], doc["nv"], doc["model"]) == (want.nq, want.nv, name + ".json") and doc["gravity"] == [0.0, 0.0, -9.81]
    for key, ref in (("q", q), ("qd", qd), ("qdd", qdd), ("tau", tau)):
        assert np.array_equal(np.array(doc[key]), ref), (name, key)
    assert not any(k in doc for k in ("tau_out", "qdd_out", "H", "cases")), "the state files hold inputs only"


def test_the_harness_processes_every_committed_mechanism_in_one_run():
    """java/.../MecanoGoldenVectorHarness.java --all <root>: its list of mechanisms is this file's, and the paths it derives exist."""
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "java", "us", "ihmc", "mecano", "hip", "tools", "MecanoGoldenVectorHarness.java")).read()
    import re
    listed = re.search(r"MECHANISMS = \{([^}]*)\}", src).group(1)
    assert tuple(x.strip().strip('"') for x in listed.split(",")) == MECHANISMS
    root = os.path.dirname(os.path.dirname(GOLDEN))
    for name in MECHANISMS:
        assert os.path.exists(os.path.join(root, "mecano_amd", "models", name + ".json")) and os.path.exists(os.path.join(GOLDEN, f"states_{name}.json"))
    assert '"with_matrices"' in src and "--all" in src


@pytest.mark.parametrize("name", MECHANISMS)
def test_consumer_on_a_file_of_the_harness_schema(tmp_path, name):
    """The reader and the comparison, on a file this test writes itself with the ORACLE's numbers (so it can only prove the plumbing)."""
    states = json.load(open(os.path.join(GOLDEN, f"states_{name}.json")))
    d = rt.modelDescFromJson(name)
    om = OracleModel(d)
    q, qd, qdd, tau = (np.array(states[k]) for k in ("q", "qd", "qdd", "tau"))
    g = tuple(states["gravity"])
    n_mat = int(states.get("with_matrices", len(q)))
    t, a, (H, C) = om.rnea(q, qd, qdd, g), om.aba(q, qd, tau, g), om.crba_coriolis(q[:n_mat], qd[:n_mat])
    fmt = lambda v: [float("%.17g" % x) for x in np.asarray(v).reshape(-1)]
    cases = []
    for s in range(len(q)):
        case = {"q": fmt(q[s]), "qd": fmt(qd[s]), "qdd": fmt(qdd[s]), "tau_in": fmt(tau[s]), "tau": fmt(t[s]), "qdd_out": fmt(a[s])}
        if s < n_mat:
            case["H"], case["C"] = fmt(H[s]), fmt(C[s])
        cases.append(case)
    doc = {"generator": "oracle/mecano_oracle.c (self-generated: NOT reference output)", "reference_output": False, "model": name + ".json",
           "n_joints": d.n_joints, "nq": d.nq, "nv": d.nv, "gravity": list(g), "cases": cases}
    path = tmp_path / f"selfgenerated_{name}.json"
    path.write_text(json.dumps(doc))
    meta, want = load_cases(str(path))
    assert meta["reference_output"] is False and want["H"].shape == (n_mat, d.nv, d.nv)
    compare(oracle_results(d), want, "oracle vs its own file")
    want["tau"][1, min(5, d.nv - 1)] += 1e-9 * max(1.0, float(np.abs(want["tau"]).max()))  # and the comparison does notice a difference of 1e-9
    with pytest.raises(AssertionError):
        compare(oracle_results(d), want, "perturbed")


@pytest.mark.skipif(not MECANO_FILES, reason="no tests/golden/mecano_*.json: golden vectors of the real Mecano calculators need a JVM "
                                             "(java/us/ihmc/mecano/hip/tools/MecanoGoldenVectorHarness.java --all); parity against the Java reference stays unpinned")
@pytest.mark.parametrize("path", MECANO_FILES or [None])
def test_oracle_matches_mecano(path):
    meta, want = load_cases(path)
    assert meta.get("reference_output") is True, "only files written by the Java harness pin parity"
    name = meta["model"][:-len(".json")]
    assert name in MECHANISMS
    compare(oracle_results(rt.modelDescFromJson(name)), want, f"oracle vs Mecano, {name}")


@pytest.mark.gpu
@pytest.mark.skipif(not MECANO_FILES, reason="no tests/golden/mecano_*.json (needs a JVM once; see MecanoGoldenVectorHarness.java --all)")
@pytest.mark.parametrize("path", MECANO_FILES or [None])
def test_hip_path_matches_mecano(path):
    import torch
    from mecano_amd.engine import HipModel
    meta, want = load_cases(path)
    assert meta.get("reference_output") is True
    name = meta["model"][:-len(".json")]
    hm = HipModel(rt.modelDescFromJson(name))
    dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda")
    results = {"rnea": lambda q, qd, qdd, g: hm.rnea(dev(q), dev(qd), dev(qdd), g).cpu().numpy(),
               "aba": lambda q, qd, tau, g: hm.aba(dev(q), dev(qd), dev(tau), g).cpu().numpy(),
               "crba": lambda q: hm.crba(dev(q)).cpu().numpy(),
               "coriolis": lambda q, qd: hm.crba_coriolis(dev(q), dev(qd))[1].cpu().numpy()}
    compare(results, want, f"HIP vs Mecano, {name}")
