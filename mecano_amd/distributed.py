"""Multi-GPU driver: one process per GPU, batch sharded contiguously, no collective on the data path.

Every configuration is independent and the model is read-only, so the only exchanges are
  * one broadcast of the flattened model (a few KB) from the rank that owns the MultiBodySystem, and
  * optionally one all-gather / gather of the outputs ((B/R) * nv * sizeof(T) bytes per rank) over xGMI.
``torch.distributed`` is used as the transport: backend "nccl" is RCCL on ROCm, "gloo" runs the same code on CPU
(tests/test_distributed_cpu.py, world_size 2).  The reference has no counterpart (it is single-threaded Java).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import numpy as np

from .multibody import ModelDesc

_INT_FIELDS = ("parent", "joint_type", "dof_indices", "cfg_indices")
_F64_FIELDS = ("axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com")


def init_from_env(backend: Optional[str] = None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MECANO_DIST_BACKEND=gloo: run the multi-rank path without RCCL (e.g. several ranks sharing one GPU in a test)
            backend = os.environ.get("MECANO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a batch of B configurations owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_desc(desc: ModelDesc):
    ints = np.concatenate([[desc.n_joints, desc.nq, desc.nv] + [len(getattr(desc, f)) for f in _INT_FIELDS]]
                          + [np.asarray(getattr(desc, f), dtype=np.int64) for f in _INT_FIELDS]).astype(np.int64)
    f64 = np.concatenate([np.asarray(getattr(desc, f), dtype=np.float64).reshape(-1) for f in _F64_FIELDS])
    return ints, f64


def unpack_desc(ints: np.ndarray, f64: np.ndarray) -> ModelDesc:
    n, nq, nv = int(ints[0]), int(ints[1]), int(ints[2])
    lens = [int(x) for x in ints[3:3 + len(_INT_FIELDS)]]
    pos = 3 + len(_INT_FIELDS)
    iv = {}
    for f, ln in zip(_INT_FIELDS, lens):
        iv[f] = ints[pos:pos + ln].astype(np.int32)
        pos += ln
    sizes = dict(axis=3 * n, X_before=12 * n, X_com=12 * n, inertia_J=9 * n, inertia_mass=n, inertia_com=3 * n)
    fv, pos = {}, 0
    for f in _F64_FIELDS:
        fv[f] = f64[pos:pos + sizes[f]].copy()
        pos += sizes[f]
    return ModelDesc(n, nq, nv, iv["parent"], iv["joint_type"], fv["axis"], fv["X_before"], fv["X_com"], fv["inertia_J"], fv["inertia_mass"],
                     fv["inertia_com"], iv["dof_indices"], iv["cfg_indices"])


def broadcast_model_desc(desc: Optional[ModelDesc], src: int = 0, device=None) -> ModelDesc:
    """Rank ``src`` passes its ModelDesc, the others pass None; everybody returns the same model (bit-identical arrays)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return desc
    dev = device if device is not None else (f"cuda:{torch.cuda.current_device()}" if dist.get_backend() == "nccl" else "cpu")
    if dist.get_rank() == src:
        ints, f64 = pack_desc(desc)
        header = torch.tensor([len(ints), len(f64)], dtype=torch.int64, device=dev)
    else:
        header = torch.zeros(2, dtype=torch.int64, device=dev)
    dist.broadcast(header, src)
    ni, nf = int(header[0]), int(header[1])
    if dist.get_rank() == src:
        ti = torch.from_numpy(ints).to(dev)
        tf = torch.from_numpy(f64).to(dev)
    else:
        ti = torch.empty(ni, dtype=torch.int64, device=dev)
        tf = torch.empty(nf, dtype=torch.float64, device=dev)
    dist.broadcast(ti, src)
    dist.broadcast(tf, src)
    return unpack_desc(ti.cpu().numpy(), tf.cpu().numpy())


def all_gather_rows(local, B_total: int):
    """Concatenates the ranks' contiguous row slices (shard_range) of a [B, ...] matrix on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(B_total, r, world)[1] - shard_range(B_total, r, world)[0] for r in range(world)]
    if dist.get_backend() == "gloo" and local.is_cuda:
        return all_gather_rows(local.cpu(), B_total).to(local.device)
    if len(set(sizes)) == 1:
        out = torch.empty((B_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def sharded_compute(fn: Callable, B_total: int, *full_inputs, gather: bool = True):
    """Runs ``fn`` on this rank's contiguous slice of every [B, ...] input and (optionally) all-gathers the result."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(B_total, rank, world)
    local = fn(*[None if x is None else x[lo:hi] for x in full_inputs])
    return all_gather_rows(local, B_total) if gather else local
