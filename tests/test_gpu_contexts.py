"""SURVEY.md section 8(b), "Threading": the model handle is read-only and shareable.  Everything a compute call writes besides its outputs
(workspace, scratch matrices, staging buffers, the hand-off flags and the error word of the bias-split forward dynamics) belongs to a
context (mh_context_create); the reference keeps the same state inside the calculator object, which is why Mecano needs one calculator per
thread (InverseDynamicsCalculator.java:706-707).  Here: two host threads, each with two contexts on two streams of its own, all on ONE
handle and all at once, bit for bit equal to the same calls made one after the other."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, x):
    return torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


# batch sizes that take three formulations of forward dynamics: bias split (hand-off flags, scratch rows), one job, fused device-filling
BATCHES = (4096, 12000, 40000)


def _inputs(torch, sys_, seed):
    from mecano_amd import random_tools as rt
    out = {}
    for B in BATCHES:
        q, qd, qdd, tau = rt.nextState(np.random.default_rng(seed + B), sys_, B)
        out[B] = tuple(dev(torch, x) for x in (q, qd, qdd, tau))
    return out


def _evaluate(torch, hm, inputs, g, stream, rounds):
    """rounds x (RNEA, ABA, pair, CRBA on the small batch) on `stream`; returns the results of the LAST round (earlier rounds only keep
    the scratch buffers busy while the other contexts use theirs)."""
    res = {}
    with torch.cuda.stream(stream):
        for _ in range(rounds):
            for B, (q, qd, qdd, tau) in inputs.items():
                res[("rnea", B)] = hm.rnea(q, qd, qdd, g)
                res[("aba", B)] = hm.aba(q, qd, tau, g)
                t2, a2 = hm.rnea_aba(q, qd, qdd, tau, g)
                res[("pair_tau", B)], res[("pair_qdd", B)] = t2, a2
            q = inputs[BATCHES[0]][0]
            res[("crba", BATCHES[0])] = hm.crba(q)
        hm.check(stream.cuda_stream)  # mh_model_check: synchronises the stream, reports asynchronous failures of this context
    return res


def test_two_threads_two_streams_each_on_one_shared_handle(torch_cuda):
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    desc = sys_.toModelDesc()
    hm = HipModel(desc)
    assert hm.kernel_variant.startswith("topo:"), hm.kernel_variant
    g = (0.2, -0.1, -9.81)
    jobs = [(_inputs(torch, sys_, 1000 * (k + 1)), hm.context(), torch.cuda.Stream()) for k in range(4)]
    torch.cuda.synchronize()
    # serial: one context after the other, on the default stream, through the model's own default context
    serial = [_evaluate(torch, hm, inputs, g, torch.cuda.current_stream(), 1) for inputs, _, _ in jobs]
    torch.cuda.synchronize()
    # concurrent: thread t drives contexts 2 t and 2 t + 1, each on a stream of its own, several rounds so that the calls really overlap
    results, errors = [None] * 4, []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            for k in (2 * t, 2 * t + 1):
                inputs, view, stream = jobs[k]
                results[k] = _evaluate(torch, view, inputs, g, stream, 6)
        except Exception as e:  # noqa: BLE001 (reported below, on the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(4):
        for key, ref in serial[k].items():
            assert torch.equal(results[k][key], ref), (k, key, (results[k][key] - ref).abs().max().item())
    # ... and the serial results are the oracle's
    om = OracleModel(desc)
    q, qd, qdd, tau = (x[:64].cpu().numpy() for x in jobs[0][0][BATCHES[0]])
    assert np.abs(serial[0][("aba", BATCHES[0])][:64].cpu().numpy() - om.aba(q, qd, tau, g)).max() < 1e-10
    assert np.abs(serial[0][("rnea", BATCHES[0])][:64].cpu().numpy() - om.rnea(q, qd, qdd, g)).max() < 1e-9
    for _, view, _ in jobs:
        view.close()
    hm.close()


def test_context_misuse_is_refused(torch_cuda):
    torch = torch_cuda
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    lib = _lib.load()
    a = HipModel(rt.nextHumanoid(np.random.default_rng(1)).toModelDesc())
    b = HipModel(rt.nextQuadruped(np.random.default_rng(2)).toModelDesc())
    view = a.context()
    q, qd, qdd, tau = (dev(torch, x) for x in rt.nextState(np.random.default_rng(3), rt.nextQuadruped(np.random.default_rng(2)), 8))
    # a context of model a handed to a call on model b
    stolen = b.context()
    lib.mh_context_destroy(stolen._ctx)
    stolen._ctx = view._ctx
    with pytest.raises(_lib.MecanoHipError) as e:
        stolen.rnea(q, qd, qdd, (0, 0, -9.81))
    assert e.value.status == 1 and "another model" in str(e.value)
    stolen._ctx = None
    # mh_model_check with a foreign context
    assert lib.mh_model_check(b._h, view._ctx, None) == 1
    # joint source modes are set on the model, before its contexts are created
    modes = (ctypes.c_int32 * a.n_joints)(*([0] * a.n_joints))
    assert lib.mh_model_set_joint_source_modes(a._h, modes) == 1
    view.close()
    assert lib.mh_model_set_joint_source_modes(a._h, modes) == 0
    assert lib.mh_model_check(a._h, None, None) == 0


def test_a_model_destroyed_before_its_contexts_lives_on_for_them(torch_cuda):
    """The model is reference-counted by its contexts (include/mecano_hip.h, contexts): mh_model_destroy with live contexts gives up the
    caller's reference only, calls through the surviving contexts stay valid, no new context can be made, the last mh_context_destroy
    releases the device records (this used to free the records under the contexts)."""
    torch = torch_cuda
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    q, qd, qdd, tau = (dev(torch, x) for x in rt.nextState(np.random.default_rng(5), sys_, 4096))
    g = (0.0, 0.0, -9.81)
    want_tau, want_qdd = hm.rnea(q, qd, qdd, g).clone(), hm.aba(q, qd, tau, g).clone()
    torch.cuda.synchronize()
    v1, v2 = hm.context(), hm.context()
    handle = hm._h
    hm.close()                          # mh_model_destroy while two contexts are alive
    assert torch.equal(v1.rnea(q, qd, qdd, g), want_tau) and torch.equal(v2.aba(q, qd, tau, g), want_qdd)
    w, qdd2 = v2.aba_joint_wrenches(q, qd, tau, g) if hasattr(v2, "aba_joint_wrenches") else (None, None)
    torch.cuda.synchronize()
    ctx = ctypes.c_void_p()
    assert _lib.load().mh_context_create(handle, ctypes.byref(ctx)) == 1  # MH_ERR_INVALID_ARGUMENT: the model has been destroyed
    v1.close()
    assert torch.equal(v2.rnea(q, qd, qdd, g), want_tau)
    torch.cuda.synchronize()
    v2.close()                          # the last reference: releases the model


def test_joint_wrenches_of_forward_dynamics_use_the_context_of_the_call(torch_cuda):
    """mh_aba_joint_wrenches_f64 keeps the efforts of its Newton-Euler sweep in scratch: two threads with a context each on one handle
    (the scratch used to be the model's whatever opts->context said)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    if not hasattr(hm, "aba_joint_wrenches"):
        pytest.skip("no Python mirror of mh_aba_joint_wrenches_f64")
    g = (0.0, 0.0, -9.81)
    data = {B: tuple(dev(torch, x) for x in rt.nextState(np.random.default_rng(B), sys_, B)) for B in (1000, 9000)}
    want = {B: [t.clone() for t in hm.aba_joint_wrenches(q, qd, tau, g)] for B, (q, qd, qdd, tau) in data.items()}
    torch.cuda.synchronize()
    errors = []

    def worker(view, stream):
        try:
            with torch.cuda.stream(stream):
                for _ in range(20):
                    for B, (q, qd, qdd, tau) in data.items():  # alternating sizes: the scratch is re-allocated in every call but the first
                        got = view.aba_joint_wrenches(q, qd, tau, g)
                        stream.synchronize()
                        for a, b in zip(got, want[B]):
                            if not torch.equal(a, b):
                                errors.append(B)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    views = [hm.context(), hm.context()]
    threads = [threading.Thread(target=worker, args=(v, torch.cuda.Stream())) for v in views]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for v in views:
        v.close()
    assert not errors, errors[:4]
