"""N>1 path on CPU: world_size 2 over gloo.  Model broadcast, contiguous batch sharding, output all-gather.
The per-rank compute is the ORACLE here (tests may use it as the checker; no GPU in this container) -- what is under
test is the sharding / broadcast / gather logic of mecano_amd.distributed, which bench.py and a multi-GPU host use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mecano_amd import distributed as mdist
    from mecano_amd import random_tools as rt
    from oracle.cpu_oracle import OracleModel
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(99)  # same stream on every rank: the full batch is known everywhere for the check
    sys_ = rt.nextHumanoid(rng)
    desc0 = sys_.toModelDesc()
    desc = mdist.broadcast_model_desc(desc0 if rank == 0 else None, src=0)
    for f in ("parent", "joint_type", "axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com", "dof_indices", "cfg_indices"):
        assert np.array_equal(np.asarray(getattr(desc, f)), np.asarray(getattr(desc0, f))), f
    om = OracleModel(desc)
    q, qd, qdd, tau = rt.nextState(rng, sys_, B)
    tq, tqd, tqdd = (torch.from_numpy(x) for x in (q, qd, qdd))

    def fn(q_, qd_, qdd_):
        return torch.from_numpy(om.rnea(q_.numpy(), qd_.numpy(), qdd_.numpy()))

    full = mdist.sharded_compute(fn, B, tq, tqd, tqdd, gather=True)
    ref = om.rnea(q, qd, qdd)
    assert full.shape == (B, desc.nv)
    assert np.array_equal(full.numpy(), ref)
    lo, hi = mdist.shard_range(B, rank, world)
    local = mdist.sharded_compute(fn, B, tq, tqd, tqdd, gather=False)
    assert np.array_equal(local.numpy(), ref[lo:hi])
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").close()


@pytest.mark.parametrize("B", [64, 37])  # equal shards (all_gather_into_tensor) and ragged shards (padded all_gather)
def test_world_size_2_gloo(tmp_path, B):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_shard_range_partitions_the_batch():
    from mecano_amd.distributed import shard_range
    for B in (0, 1, 7, 4096, 262144, 1000003):
        for world in (1, 2, 4, 8):
            edges = [shard_range(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_c_abi_shard_range_is_the_python_one_and_rejects_bad_arguments():
    """mh_shard_range (what a host without torch calls, include/mecano_hip.h) against mecano_amd.distributed.shard_range, including more
    ranks than rows; the communicator entry points refuse NULL handles without touching RCCL or a device."""
    import ctypes
    from mecano_amd import _lib
    from mecano_amd.distributed import shard_range
    from mecano_amd.engine import HipCommunicator
    for B in (0, 1, 3, 7, 4096, 4099, 262144, 1000003):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert HipCommunicator.shard_range(B, r, world) == shard_range(B, r, world)
    lib = _lib.load()
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for B, r, w in ((-1, 0, 1), (8, 2, 2), (8, -1, 2), (8, 0, 0)):
        assert lib.mh_shard_range(B, r, w, ctypes.byref(lo), ctypes.byref(hi)) == 1  # MH_ERR_INVALID_ARGUMENT
        assert b"mh_shard_range" in lib.mh_last_error()
    assert lib.mh_shard_range(8, 0, 2, None, ctypes.byref(hi)) == 1
    out = ctypes.c_void_p()
    assert lib.mh_comm_create(None, 0, 1, ctypes.byref(out)) == 1
    assert lib.mh_comm_unique_id(None) == 1
    assert lib.mh_comm_barrier(None, None) == 1
    assert lib.mh_comm_all_gather_rows(None, None, 8, 8, None, None) == 1
    assert lib.mh_comm_destroy(None) == _lib.MH_OK
    with pytest.raises(ValueError):
        HipCommunicator(b"short", 0, 1)


def test_pack_unpack_roundtrip():
    from mecano_amd import distributed as mdist
    from mecano_amd import random_tools as rt
    d = rt.nextHumanoid(np.random.default_rng(3)).toModelDesc()
    d2 = mdist.unpack_desc(*mdist.pack_desc(d))
    assert d2.n_joints == d.n_joints and d2.nq == d.nq and d2.nv == d.nv
    for f in ("parent", "joint_type", "axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com", "dof_indices", "cfg_indices"):
        assert np.array_equal(np.asarray(getattr(d2, f)), np.asarray(getattr(d, f)))


def test_bench_self_launcher_starts_one_process_per_rank(tmp_path, monkeypatch):
    """bench.self_launch (what `python3 bench.py --gpus N` runs when no launcher set WORLD_SIZE): N child processes with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, rank 0's stdout relayed, first non-zero exit code returned -- exercised with a stand-in script (no GPU here)."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    script = tmp_path / "fake_bench.py"
    script.write_text("import os, sys\n"
                      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                      "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "open(os.path.join(os.path.dirname(__file__), f'rank{r}of{w}'), 'w').close()\n"
                      "print('{\"rank\": %d}' % r)\n"
                      "sys.exit(3 if (r == 1 and 'fail' in sys.argv) else 0)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for argv, want in ((["bench.py", "--gpus", "3"], 0), (["bench.py", "--gpus", "3", "fail"], 3)):
        monkeypatch.setattr(sys, "argv", argv)
        out = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.__file__ = {str(script)!r}; "
                              f"sys.argv = {argv!r}; sys.exit(bench.self_launch(3))"], capture_output=True, text=True, timeout=120)
        assert out.returncode == want, out.stderr
        assert out.stdout.strip() == '{"rank": 0}'  # only rank 0's line is relayed
        assert all((tmp_path / f"rank{r}of3").exists() for r in range(3))
