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


def test_world_size_8_gloo_ragged_config4_total(tmp_path):
    """Eight ranks -- the node size the driver's scaling run uses -- over gloo with a ragged total (8 does not divide 262 144 + 5 scaled down to
    what the oracle evaluates in seconds: 1024 + 5): shard sizes 129, 129, 129, 129, 129, 128, 128, 128; broadcast of the model, sharded
    compute, padded all-gather, every row equal to the unsharded result."""
    port = _free_port()
    mp.spawn(_worker, args=(8, port, 1029, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(8))


def test_gather_plan_is_the_same_on_every_rank_and_tiles_the_output():
    """mh_comm_all_gather_rows issues what mh_comm_gather_plan lists (mecano_amd/csrc/mh_comm.hip).  RCCL cannot put two ranks on one GPU and
    the builder's box has one, so the ragged schedule -- a group of broadcasts, send != receive on the root, in place elsewhere -- had only
    ever run with world = 1.  The plan is a pure function: here it is evaluated for every rank of every world size up to nine and EXECUTED
    on host buffers standing in for the ranks' device memory: all ranks list the same operations in the same order (a grouped collective
    deadlocks otherwise), the received ranges tile the output exactly, and every rank ends with all rows in batch order."""
    import ctypes
    from mecano_amd import _lib
    from mecano_amd.distributed import shard_range
    lib = _lib.load()

    def plan(B, row_bytes, rank, world, force):
        n = ctypes.c_int32()
        steps = (_lib.MhGatherStep * (world + 1))()
        assert lib.mh_comm_gather_plan(B, row_bytes, rank, world, force, steps, world + 1, ctypes.byref(n)) == _lib.MH_OK
        count_only = ctypes.c_int32()
        assert lib.mh_comm_gather_plan(B, row_bytes, rank, world, force, None, 0, ctypes.byref(count_only)) == _lib.MH_OK
        assert count_only.value == n.value <= world
        return [(s.root, s.send_local, s.recv_offset, s.bytes) for s in steps[:n.value]]

    row = 24  # bytes per row
    for world in range(1, 10):
        for B in (0, 1, 5, world, 8 * world, 8 * world + 1, 9 * world - 1, 4099, 262144 + 5):
            for force in (0, 1):
                plans = [plan(B, row, r, world, force) for r in range(world)]
                # same operations, same order, on every rank (send_local differs: it marks the root's own step)
                for r in range(1, world):
                    if plans[0] and plans[0][0][0] == -1:  # the plain all-gather: one step everywhere, same size; the offset is where the rank's own shard lands
                        assert [(a, d) for a, _, _, d in plans[r]] == [(a, d) for a, _, _, d in plans[0]], (world, B, force, r)
                    else:
                        assert [(a, c, d) for a, _, c, d in plans[r]] == [(a, c, d) for a, _, c, d in plans[0]], (world, B, force, r)
                if B == 0:
                    assert plans[0] == []
                    continue
                full = np.random.default_rng(B + world).integers(0, 255, size=B * row, dtype=np.uint8)
                shards = [full[shard_range(B, r, world)[0] * row:shard_range(B, r, world)[1] * row] for r in range(world)]
                outs = [np.zeros(B * row, dtype=np.uint8) for _ in range(world)]
                covered = np.zeros(B * row, dtype=np.int32)
                if len(plans[0]) == 1 and plans[0][0][0] == -1:  # equal shards: the plain all-gather, every rank's shard at rank * bytes
                    assert B % world == 0 and not force
                    for r in range(world):
                        root, send_local, ofs, nbytes = plans[r][0]
                        assert (send_local, ofs, nbytes) == (1, r * (B // world) * row, (B // world) * row)
                        for o in outs:
                            o[ofs:ofs + nbytes] = shards[r]
                        covered[ofs:ofs + nbytes] += 1
                else:
                    for k, (root, _, ofs, nbytes) in enumerate(plans[0]):
                        assert 0 <= root < world and nbytes == len(shards[root]) > 0 and ofs == shard_range(B, root, world)[0] * row
                        for r in range(world):
                            assert plans[r][k][1] == (1 if r == root else 0)  # the root sends its local rows, everybody else receives in place
                            outs[r][ofs:ofs + nbytes] = shards[root]
                        covered[ofs:ofs + nbytes] += 1
                assert (covered == 1).all(), (world, B, force)
                assert all(np.array_equal(o, full) for o in outs)
    n = ctypes.c_int32()
    for bad in ((-1, 8, 0, 1), (8, 8, 2, 2), (8, 8, -1, 2), (8, 8, 0, 0)):
        assert lib.mh_comm_gather_plan(bad[0], bad[1], bad[2], bad[3], 0, None, 0, ctypes.byref(n)) == 1
    assert lib.mh_comm_gather_plan(8, 8, 0, 2, 0, None, 0, None) == 1


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
